"""svgpu_ba_partition_keyframe_segments (host only: runs without a GPU): the landmark -> rank map of north_star's BA partition -- keyframes in
segments, a landmark with the segment that owns its keyframes (optimize/global_bundle_adjuster.cc:26-192 is the workload it cuts)."""
import numpy as np
import pytest

from stella_vslam_amd import distributed as D, optimize, synthetic as S


@pytest.fixture(scope="module")
def scene():
    return S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_partition_follows_keyframe_segments(scene, world, monkeypatch):
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    lm_rank, info = optimize.partition_keyframe_segments(scene, world)
    assert info["segmented"] == 1 and info["jobs"] >= world and info["cuts"] >= 1
    assert lm_rank.shape == (len(scene["points"]),) and lm_rank.min() >= 0 and lm_rank.max() == world - 1
    obs_pose, obs_rank = np.asarray(scene["obs_pose"]), lm_rank[np.asarray(scene["obs_point"])]
    # every keyframe that is not a separator keyframe has ALL its observations on one rank: the ranks that hold observations of a
    # keyframe number more than one for at most `separator_keyframes` keyframes
    P = len(scene["pose_cw"])
    seen = np.zeros((P, world), bool)
    seen[obs_pose, obs_rank] = True
    multi = int((seen.sum(1) > 1).sum())
    assert 0 < multi <= info["separator_keyframes"] < info["free_keyframes"] // 4, (multi, info)
    # landmarks on separators are the only ones whose blocks cross ranks: a minority
    assert info["landmarks_on_separators"] < len(lm_rank) // 2 and info["landmarks_on_separators_only"] <= info["landmarks_on_separators"]
    # balanced to the granularity of the jobs
    share = np.bincount(obs_rank, minlength=world) / len(obs_rank)
    assert share.max() < 2.0 / world + 0.05, share
    # what crosses ranks per trial is far smaller than the reduced system
    assert 288 * info["separator_blocks"] + 8 * info["job_exchange_doubles"] < 288 * info["kept_blocks"]
    # deterministic
    again, info2 = optimize.partition_keyframe_segments(scene, world)
    assert np.array_equal(lm_rank, again) and info == info2


def test_shards_cover_the_problem_once(scene, monkeypatch):
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    world = 3
    idx = np.concatenate([D.shard_by_keyframe_segment(scene, r, world)["_obs_index"] for r in range(world)])
    assert len(idx) == len(scene["obs_pose"]) and np.array_equal(np.sort(idx), np.arange(len(idx)))
    for r in range(world):   # by landmark: no landmark on two ranks
        sh = D.shard_by_keyframe_segment(scene, r, world)
        assert np.all(scene["_kfseg"][0][sh["obs_point"]] == r)
        assert sh["_partition"]["segmented"] == 1 and "_kfseg" not in sh


def test_small_or_unsegmented_problems_fall_back_to_modulo(monkeypatch):
    sc = S.ba_scene(num_kf=12, num_lm=500, obs_per_lm=5, num_fixed=2, seed=3)
    lm_rank, info = optimize.partition_keyframe_segments(sc, 4)
    assert info["segmented"] == 0 and np.array_equal(lm_rank, np.arange(len(sc["points"])) % 4)
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "0")   # the segmented plan switched off: the solve would not cut, neither does the partition
    big = S.ba_scene_large(num_kf=200, num_lm=6000, obs_per_lm=4)
    lm_rank, info = optimize.partition_keyframe_segments(big, 2)
    assert info["segmented"] == 0 and np.array_equal(lm_rank, np.arange(len(big["points"])) % 2)
    lm_rank, info = optimize.partition_keyframe_segments(big, 1)
    assert info["segmented"] == 0 and not lm_rank.any()


def test_a_fixed_landmark_seen_from_two_pieces_keeps_modulo_shards(scene, monkeypatch):
    """A fixed landmark couples no keyframes, so it may be observed from two pieces of the cut graph; its observations cannot sit on one rank
    without feeding a foreign piece's pose blocks -- such a problem keeps the l % world shards (the solve then exchanges the whole system)."""
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    sc = dict(scene)
    P, L = len(sc["pose_cw"]), len(sc["points"])
    pf = np.zeros(L, np.uint8)
    pf[0] = 1
    # two keyframes from the INSIDE of two different pieces (every observation of each on one rank, and not the same rank) observe the fixed landmark 0
    lm_rank0, info0 = optimize.partition_keyframe_segments(scene, 4)
    assert info0["segmented"] == 1
    seen = np.zeros((P, 4), bool)
    seen[np.asarray(scene["obs_pose"]), lm_rank0[np.asarray(scene["obs_point"])]] = True
    inside = np.flatnonzero((seen.sum(1) == 1) & (np.asarray(scene["pose_fixed"]) == 0))
    a = int(inside[seen[inside, 0]][0])
    b = int(inside[seen[inside, 1]][0])
    far = np.array([a, b], np.int32)
    keep = np.asarray(sc["obs_point"]) != 0
    sc["obs_pose"] = np.concatenate([np.asarray(sc["obs_pose"])[keep], far]).astype(np.int32)
    sc["obs_point"] = np.concatenate([np.asarray(sc["obs_point"])[keep], np.zeros(2, np.int32)]).astype(np.int32)
    sc["point_fixed"] = pf
    lm_rank, info = optimize.partition_keyframe_segments(sc, 4)
    assert info["segmented"] == 0 and np.array_equal(lm_rank, np.arange(L) % 4)
    sc["point_fixed"] = np.zeros(L, np.uint8)           # the same landmark FREE couples the two keyframes: the graph is no longer a band the planner cuts, or it is cut elsewhere -- either way a valid partition
    lm_rank, info = optimize.partition_keyframe_segments(sc, 4)
    assert lm_rank.min() >= 0 and lm_rank.max() <= 3


def test_partition_rejects_bad_indices(scene):
    sc = dict(scene)
    bad = np.array(sc["obs_pose"], np.int32, copy=True)
    bad[5] = len(sc["pose_cw"])
    sc["obs_pose"] = bad
    with pytest.raises(RuntimeError):
        optimize.partition_keyframe_segments(sc, 2)
