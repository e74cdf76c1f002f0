"""GPU parity of the function-specific matchers (through the C ABI) against the per-method oracles of oracle/match2_oracle.c:
identical match lists -- the matchers are integer / index work, so the bar is bit-exactness.  The only inexact ingredients are
libm calls (logf in landmark::predict_scale_level, acos in check_epipolar_constraint): the tests assert that no input sits within
rounding distance of those decisions."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import match_problems as MP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from stella_vslam_amd import feature
    return feature.Context()


@pytest.fixture(scope="module", params=[False, True], ids=["mono", "stereo"])
def world(request):
    sc = MP.scene(seed=9 if request.param else 7, stereo=request.param)
    return sc, MP.make_cams(sc, "oracle"), MP.make_cams(sc, "svgpu")


def _no_level_ties(max_valid, dist, T):
    q = np.log(max_valid.astype(np.float64) / np.maximum(dist, 1e-9)) / T["log_scale_factor"]
    assert np.abs(q - np.round(q)).min() > 1e-5


def test_current_and_last_frames(ctx, world):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.current_and_last(sc)
    for check_ori in (True, False):
        got, num = match.projection_flat(0.9, check_ori, ctx).match_current_and_last_frames(gcam, **kw)
        ref, rnum = O.match_current_and_last_frames(check_ori, ocam, **kw)
        assert np.array_equal(got, ref) and num == rnum > 300
    # landmarks without observations do not block their keypoint: the overwrite semantics of curr_frm.add_landmark
    taken = ref[ref >= 0]
    kw2 = dict(kw, lm_has_observation=np.zeros_like(kw["lm_has_observation"]))
    got, num = match.projection_flat(0.9, True, ctx).match_current_and_last_frames(gcam, **kw2)
    ref2, _ = O.match_current_and_last_frames(True, ocam, **kw2)
    assert np.array_equal(got, ref2) and len(np.unique(ref2[ref2 >= 0])) < (ref2 >= 0).sum() and len(taken) > 0


def test_frame_and_keyframe_projection(ctx, world):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.frame_and_keyframe(sc)
    R, t = np.asarray(kw["rot_cw"]), np.asarray(kw["trans_cw"])
    _no_level_ties(kw["max_valid_dist"][kw["valid"] > 0], np.linalg.norm(kw["pos_w"][kw["valid"] > 0] - (-R.T @ t), axis=1), sc["tables"])
    for thr in (100, 50):
        got, num = match.projection_flat(0.9, True, ctx).match_frame_and_keyframe(gcam, **dict(kw, hamm_dist_thr=thr))
        ref, rnum = O.match_frame_and_keyframe_projection(True, ocam, **dict(kw, hamm_dist_thr=thr))
        assert np.array_equal(got, ref) and num == rnum > 200


def test_by_sim3_transform(ctx, world):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.by_sim3(sc)
    got, num = match.projection_flat(0.9, True, ctx).match_by_Sim3_transform(gcam, **kw)
    ref, rnum = O.match_by_sim3_transform(ocam, **kw)
    assert np.array_equal(got, ref) and num == rnum > 300


def test_keyframes_mutually(ctx, world):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.mutually(sc)
    g21, g12, gmut, gnum = match.projection_flat(0.9, True, ctx).match_keyframes_mutually(gcam, gcam, **kw)
    r21, r12, rmut, rnum = O.match_keyframes_mutually(ocam, ocam, **kw)
    assert np.array_equal(g21, r21) and np.array_equal(g12, r12) and np.array_equal(gmut, rmut) and gnum == rnum > 200


@pytest.mark.parametrize("reproj", [False, True])
def test_fuse_detect_duplication(ctx, world, reproj):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.fuse(sc, do_reprojection_matching=reproj)
    got, num = match.fuse(0.6, ctx).detect_duplication(gcam, **kw)
    ref, rnum = O.fuse_detect_duplication(ocam, **kw)
    assert np.array_equal(got, ref) and num == rnum > 300


@pytest.mark.parametrize("with_nodes", [False, True], ids=["robust", "bow_tree"])
def test_match_for_triangulation(ctx, world, with_nodes):
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    okw = MP.triangulation(sc, lambda R, t, c: O.reproject_to_bearing(ocam, R, t, c), with_nodes=with_nodes)
    gkw = MP.triangulation(sc, lambda R, t, c: match.reproject_to_bearing(gcam, R, t, c), with_nodes=with_nodes)
    assert np.array_equal(okw["epipole_in_2"], gkw["epipole_in_2"]) and okw["valid_epipole"] == gkw["valid_epipole"]
    for check_ori in (True, False):
        got, num = match.match_for_triangulation(ctx, 0.8, check_ori, **gkw)
        ref, rnum = O.match_for_triangulation(0.8, check_ori, **okw)
        assert np.array_equal(got, ref) and num == rnum > (30 if with_nodes else 100)


@pytest.mark.parametrize("keyframes", [False, True], ids=["frame_and_keyframe", "keyframes"])
def test_bow_match(ctx, world, keyframes):
    from stella_vslam_amd import match
    sc, _, _ = world
    kw = MP.bow(sc, keyframes=keyframes)
    for check_ori in (True, False):
        got, num = match.bow_tree(0.75, check_ori, ctx).match(**kw)
        ref, rnum = O.bow_match(0.75, check_ori, **kw)
        assert np.array_equal(got, ref) and num == rnum > 200


def test_degenerate_inputs(ctx, world):
    """empty sides, nothing valid, nothing visible: every entry point returns all -1 and a zero count"""
    from stella_vslam_amd import match
    sc, ocam, gcam = world
    kw = MP.fuse(sc)
    got, num = match.fuse(0.6, ctx).detect_duplication(gcam, **dict(kw, valid=np.zeros_like(kw["valid"])))
    assert num == 0 and (got == -1).all()
    far = dict(kw, pos_w=kw["pos_w"] + np.array([0, 0, -100.0]))   # behind the camera
    got, num = match.fuse(0.6, ctx).detect_duplication(gcam, **far)
    assert num == 0 and (got == -1).all()
    kb = MP.bow(sc)
    got, num = match.bow_tree(0.75, True, ctx).match(**dict(kb, node2=np.full_like(kb["node2"], -1)))
    assert num == 0 and (got == -1).all()
    e = np.zeros((0, 32), np.uint8)
    got, num = match.bow_tree(0.75, True, ctx).match(desc1=e, angle1=np.zeros(0, np.float32), valid1=np.zeros(0, np.uint8), node1=np.zeros(0, np.int32),
                                                     desc2=kb["desc2"], angle2=kb["angle2"], node2=kb["node2"])
    assert num == 0 and len(got) == 0
