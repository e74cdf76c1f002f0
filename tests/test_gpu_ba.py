"""GPU parity: HIP local BA (through the C ABI) vs the CPU oracle -- optimised SE3 poses within 1e-4 relative
(BASELINE.json north_star tolerance), identical LM schedule and outlier lists."""
import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu
TOL = 1e-4  # relative, on pose entries (north_star)


@pytest.fixture(scope="module")
def ba():
    from stella_vslam_amd import optimize
    return optimize.local_bundle_adjuster()


def _rel(a, b):
    """Largest PER-POINT relative error |p_got - p_ref| / |p_ref| for n x 3 point arrays (a far point must not hide the error of a near one);
    max|delta| / max|ref| for anything else."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.ndim == 2 and a.shape[1] == 3:
        return float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-12)).max())
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _pose_err(got, ref):
    """north_star: "within 1e-4 relative on optimised SE3 poses", judged per pose and per part: the angle of R_ref^T R_got (rad)
    and |t_got - t_ref| / |t_ref| -- NOT max|delta| over a 3x4 block scaled by its largest entry (a 10 m translation would hide
    a 5e-4 error on a rotation entry)."""
    g, r = np.asarray(got).reshape(-1, 3, 4), np.asarray(ref).reshape(-1, 3, 4)
    dR = np.einsum("pji,pjk->pik", r[:, :, :3], g[:, :, :3])
    ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0))
    # arccos loses half the digits near 0: use the skew part there
    skew = 0.5 * np.sqrt((dR[:, 2, 1] - dR[:, 1, 2]) ** 2 + (dR[:, 0, 2] - dR[:, 2, 0]) ** 2 + (dR[:, 1, 0] - dR[:, 0, 1]) ** 2)
    ang = np.where(ang < 1e-3, skew, ang)
    dt = np.linalg.norm(g[:, :, 3] - r[:, :, 3], axis=1) / np.maximum(np.linalg.norm(r[:, :, 3], axis=1), 1e-12)
    return float(ang.max()), float(dt.max())


def _assert_poses(got, ref, tol=TOL):
    ang, dt = _pose_err(got, ref)
    assert ang < tol and dt < tol, (ang, dt)


@pytest.mark.parametrize("kw", [dict(num_kf=6, num_lm=300, obs_per_lm=4, num_fixed=2, seed=3),
                                dict(num_kf=10, num_lm=1500, obs_per_lm=5, num_fixed=3, seed=5),
                                dict(num_kf=12, num_lm=900, obs_per_lm=6, num_fixed=2, seed=6, stereo=True),
                                dict(num_kf=20, num_lm=10000, obs_per_lm=6, num_fixed=4, seed=1234),
                                # the on-chip LL^T at its limits: n = 126 (946 of 1024 tiles, one per thread), n = 132 (two per thread)
                                dict(num_kf=24, num_lm=3000, obs_per_lm=6, num_fixed=3, seed=21),
                                dict(num_kf=24, num_lm=3000, obs_per_lm=6, num_fixed=2, seed=22),
                                # windows beyond one workgroup's LDS with a dense block pattern: the tiled LL^T of ba_dense_tiled.hip (n = 192: four
                                # whole tile columns; n = 228: a part-filled last tile with the right-hand side inside its row range)
                                dict(num_kf=34, num_lm=4000, obs_per_lm=6, num_fixed=2, seed=31),
                                dict(num_kf=40, num_lm=5000, obs_per_lm=6, num_fixed=2, seed=32),
                                # equirectangular cameras (intrinsics rows {0, 0, cols, rows, 0}): equirectangular_reproj_edge.h:64-134
                                dict(num_kf=8, num_lm=1200, obs_per_lm=5, num_fixed=2, seed=8, equirect=True)])
def test_local_ba_matches_oracle(ba, kw):
    sc = S.ba_scene(**kw)
    got = ba.optimize_flat(sc)
    ref = O.local_ba(sc)
    gs, rs = got["stats"], ref["stats"]
    assert gs["iters_stage1"] == rs[2] and gs["iters_stage2"] == rs[3] and gs["stage2_entered"] == rs[4]
    assert gs["num_gated"] == rs[5]
    assert gs["chi2_initial"] == pytest.approx(rs[0], rel=1e-9)
    assert gs["chi2_final"] == pytest.approx(rs[1], rel=1e-6)
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    assert np.array_equal(got["outlier"], ref["outlier"])
    # and the optimisation did its job
    assert gs["chi2_final"] < 0.5 * gs["chi2_initial"]
    assert np.abs(got["pose_cw"] - sc["pose_gt"]).mean() < np.abs(sc["pose_cw"] - sc["pose_gt"]).mean()


def test_tiled_dense_solver_is_what_auto_takes_and_agrees_with_the_others(ba, monkeypatch):
    """A 40-keyframe window (n = 228, every upper block kept): AUTO hands the reduced system to the tiled dense LL^T (ba_dense_tiled.hip: the
    only solver of this size that leaves the envelope plan empty); the envelope factorisation and the one-workgroup dense form must land on
    the same estimate with the same LM schedule."""
    from stella_vslam_amd import optimize
    sc = S.ba_scene(num_kf=40, num_lm=5000, obs_per_lm=6, num_fixed=2, seed=32)
    auto = optimize.local_bundle_adjuster()
    got = auto.optimize_flat(sc)
    assert got["stats"]["cholesky_failures"] == 0 and got["stats"]["pcg_iterations"] == 0
    env = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE).optimize_flat(sc)
    monkeypatch.setenv("SVGPU_BA_NO_DENSE_TILED", "1")
    before = optimize.local_bundle_adjuster().optimize_flat(sc)  # what AUTO took before: the envelope factorisation
    monkeypatch.delenv("SVGPU_BA_NO_DENSE_TILED")
    for other in (env, before):
        for key in ("iters_stage1", "iters_stage2", "num_gated", "lm_trials"):
            assert got["stats"][key] == other["stats"][key], key
        assert got["stats"]["chi2_final"] == pytest.approx(other["stats"]["chi2_final"], rel=1e-9)
        assert np.abs(got["pose_cw"] - other["pose_cw"]).max() < 1e-8 and np.abs(got["points"] - other["points"]).max() < 1e-8
        assert np.array_equal(got["outlier"], other["outlier"])
    again = auto.optimize_flat(sc)
    assert np.array_equal(again["pose_cw"], got["pose_cw"]) and np.array_equal(again["points"], got["points"])  # fixed-order sums


def test_tiled_dense_solver_fails_trials_where_the_envelope_solver_does():
    """A pivot that is not positive fails the damping trial (ctl.solve_failed; g2o's solver returning false): an INDEFINITE problem -- a third of
    the observations with negative information -- makes a dozen factorisations fail.  The tiled dense LL^T (AUTO at n = 228) and the envelope
    factorisation must fail the same trials: same failure count, same LM schedule, same damping at the end, same estimate."""
    from stella_vslam_amd import optimize
    sc = dict(S.ba_scene(num_kf=40, num_lm=5000, obs_per_lm=6, num_fixed=2, seed=32))
    w = sc["obs_inv_sigma_sq"].copy()
    w[np.random.default_rng(3).random(len(w)) < 0.3] *= -3.0
    sc["obs_inv_sigma_sq"] = w
    tiled = optimize.local_bundle_adjuster().optimize_flat(sc)
    env = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE).optimize_flat(sc)
    assert tiled["stats"]["cholesky_failures"] >= 5
    for key in ("iters_stage1", "iters_stage2", "lm_trials", "cholesky_failures", "num_gated"):
        assert tiled["stats"][key] == env["stats"][key], key
    assert tiled["stats"]["lambda_final"] == pytest.approx(env["stats"]["lambda_final"], rel=1e-9)
    assert tiled["stats"]["chi2_final"] == pytest.approx(env["stats"]["chi2_final"], rel=1e-9)
    assert np.abs(tiled["pose_cw"] - env["pose_cw"]).max() < 1e-7 and np.abs(tiled["points"] - env["points"]).max() < 1e-6


def test_local_ba_stop_flag_semantics(ba):
    sc = S.ba_scene(num_kf=6, num_lm=200, obs_per_lm=4, num_fixed=2, seed=3, outlier_frac=0.0, pose_noise=(1e-4, 1e-3),
                    point_noise=1e-4)
    from stella_vslam_amd import optimize
    ba20 = optimize.local_bundle_adjuster(num_first_iter=20, ctx=ba.ctx)  # enough iterations for the gain rule to fire in stage 1
    flag = np.zeros(1, np.uint8)
    got = ba20.optimize_flat(sc, force_stop_flag=flag)
    f2 = np.zeros(1, np.uint8)
    ref = O.local_ba(sc, iters1=20, stop=f2)
    assert flag[0] == f2[0] == 1  # the terminate rule wrote through the caller's flag
    assert ref["stats"][4] == 0 and ref["stats"][2] < 20
    assert got["stats"]["stage2_entered"] == 0 and got["stats"]["iters_stage1"] == ref["stats"][2]
    # without a caller flag g2o installs its own: stage 2's gate runs but its LM loop does not (svgpu.h)
    got2, ref2 = ba20.optimize_flat(sc), O.local_ba(sc, iters1=20)
    assert got2["stats"]["stage2_entered"] == 1 == ref2["stats"][4] and got2["stats"]["iters_stage2"] == 0 == ref2["stats"][3]
    assert np.array_equal(got2["outlier"], ref2["outlier"])
    assert _rel(got["pose_cw"], ref["pose_cw"]) < TOL
    flag[:] = 1
    got = ba.optimize_flat(sc, force_stop_flag=flag)
    assert got["rc"] == 7 and np.array_equal(got["pose_cw"], sc["pose_cw"])  # SVGPU_STOPPED: nothing touched


def test_local_ba_fixed_points_and_no_kernel(ba):
    sc = S.ba_scene(num_kf=6, num_lm=300, obs_per_lm=4, num_fixed=2, seed=9)
    sc["point_fixed"] = (np.arange(len(sc["points"])) % 7 == 0).astype(np.uint8)  # marker corners kept fixed
    sc["obs_huber"][::5] = 0.0                                                     # ... observed without a kernel
    got = ba.optimize_flat(sc)
    ref = O.local_ba(sc)
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    fixed = sc["point_fixed"] == 1
    assert np.array_equal(got["points"][fixed], sc["points"][fixed])
    assert np.array_equal(got["outlier"], ref["outlier"])


def test_local_ba_marker_corner_edges_stay_out_of_the_gate(ba):
    """A negative Huber width marks a marker-corner edge (local_bundle_adjuster_g2o.cc:246-304: own container, no kernel, information as given):
    neither the chi-square / depth gate after stage 1 nor the outlier list sees it, however large its error."""
    sc = S.ba_scene(num_kf=8, num_lm=500, obs_per_lm=4, num_fixed=2, seed=21)
    rng = np.random.default_rng(3)
    mk = np.flatnonzero(sc["obs_point"] % 11 == 0)               # every edge of some points: "corners"
    sc["obs_huber"][mk] = -1.0
    sc["obs_inv_sigma_sq"][mk] = 1.0
    bad = mk[::6]
    sc["obs_uvr"][bad, 0] += rng.choice([-1, 1], len(bad)).astype(np.float32) * 9.0   # chi-square 81 at information 1: far past 5.99146
    got = ba.optimize_flat(sc)
    ref = O.local_ba(sc)
    assert got["stats"]["num_gated"] == ref["stats"][5] and got["stats"]["iters_stage1"] == ref["stats"][2] and got["stats"]["iters_stage2"] == ref["stats"][3]
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert not got["outlier"][mk].any() and got["outlier"].sum() > 0
    # the same edges with a zero width (no kernel, but ordinary landmark edges) are gated
    sc["obs_huber"][mk] = 0.0
    again = ba.optimize_flat(sc)
    assert again["outlier"][bad].sum() > 0.8 * len(bad)


def test_local_ba_landmark_seen_twice_from_one_keyframe(ba):
    """The flat problem does not forbid two observations of one landmark from one keyframe; their cross terms enter the keyframe's
    diagonal block twice (mirrored pairs), and the pair count computed on the host has to agree with what the device emits."""
    sc = S.ba_scene(num_kf=8, num_lm=600, obs_per_lm=4, num_fixed=2, seed=13)
    rng = np.random.default_rng(5)
    pick = rng.choice(len(sc["obs_pose"]), 150, replace=False)
    for key in ("obs_pose", "obs_point", "obs_inv_sigma_sq", "obs_huber"):
        sc[key] = np.concatenate([sc[key], sc[key][pick]])
    uvr = sc["obs_uvr"][pick].copy()
    uvr[:, :2] += rng.normal(0, 0.5, (len(pick), 2)).astype(np.float32)
    sc["obs_uvr"] = np.concatenate([sc["obs_uvr"], uvr])
    got, ref = ba.optimize_flat(sc), O.local_ba(sc)
    gs, rs = got["stats"], ref["stats"]
    assert gs["iters_stage1"] == rs[2] and gs["iters_stage2"] == rs[3] and gs["num_gated"] == rs[5]
    assert gs["chi2_final"] == pytest.approx(rs[1], rel=1e-6)
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert np.array_equal(got["outlier"], ref["outlier"])


def test_reduced_system_forms_agree(ba, monkeypatch):
    """k_ba_schur_rhs has two forms: a small system runs the right-hand side W Hll^-1 bl as units of its own beside the block shares, a large
    one (>= 8 192 units) sums it inside the shares of the diagonal blocks (pairs (i, i) over a keyframe's observations; W_j is W_i there) and
    walks the units in XCD-contiguous order.  SVGPU_BA_SCHUR_ORDER forces either form: same sums in another grouping, so the LM schedule must
    be identical and the estimates agree far below the parity tolerance -- also with a landmark seen twice from one keyframe (pairs (i, j),
    i != j, inside a diagonal block) and with excluded observations in the second stage (their W is zero, the pair list is reused)."""
    from stella_vslam_amd import optimize
    twice = S.ba_scene(num_kf=8, num_lm=600, obs_per_lm=4, num_fixed=2, seed=13)
    rng = np.random.default_rng(5)
    pick = rng.choice(len(twice["obs_pose"]), 150, replace=False)
    for key in ("obs_pose", "obs_point", "obs_inv_sigma_sq", "obs_huber"):
        twice[key] = np.concatenate([twice[key], twice[key][pick]])
    uvr = twice["obs_uvr"][pick].copy()
    uvr[:, :2] += rng.normal(0, 0.5, (len(pick), 2)).astype(np.float32)
    twice["obs_uvr"] = np.concatenate([twice["obs_uvr"], uvr])
    for sc, run in ((twice, lambda a, s: a.optimize_flat(s)), (S.ba_scene(), lambda a, s: a.optimize_flat(s)),
                    (S.ba_scene_large(num_kf=160, num_lm=40000), lambda a, s: a.optimize_global_flat(s, num_iter=10))):
        out = {}
        for form in ("0", "1"):
            monkeypatch.setenv("SVGPU_BA_SCHUR_ORDER", form)
            out[form] = run(optimize.local_bundle_adjuster(), sc)
        monkeypatch.delenv("SVGPU_BA_SCHUR_ORDER", raising=False)
        a, b = out["0"], out["1"]
        for key in ("iters_stage1", "iters_stage2", "num_gated", "lm_trials", "cholesky_failures"):
            assert a["stats"][key] == b["stats"][key], key
        assert a["stats"]["chi2_final"] == pytest.approx(b["stats"]["chi2_final"], rel=1e-9)
        assert np.abs(a["pose_cw"] - b["pose_cw"]).max() < 1e-8 and np.abs(a["points"] - b["points"]).max() < 1e-8
        if "outlier" in a: assert np.array_equal(a["outlier"], b["outlier"])
    ref = O.local_ba(twice)
    monkeypatch.setenv("SVGPU_BA_SCHUR_ORDER", "1")
    got = optimize.local_bundle_adjuster().optimize_flat(twice)
    monkeypatch.delenv("SVGPU_BA_SCHUR_ORDER", raising=False)
    assert got["stats"]["iters_stage1"] == ref["stats"][2] and got["stats"]["iters_stage2"] == ref["stats"][3] and got["stats"]["num_gated"] == ref["stats"][5]
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert np.array_equal(got["outlier"], ref["outlier"])


def test_local_ba_observations_in_any_order(ba):
    """Observations grouped by landmark (the order the reference creates its edges in) are copied as they are; any other order is
    sorted by landmark first and the outlier flags come back in the caller's order."""
    sc = S.ba_scene(num_kf=9, num_lm=800, obs_per_lm=5, num_fixed=2, seed=17)
    order = np.random.default_rng(3).permutation(len(sc["obs_pose"]))
    sh = dict(sc)
    for key in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma_sq", "obs_huber"):
        sh[key] = np.ascontiguousarray(sc[key][order])
    got, ref, plain = ba.optimize_flat(sh), O.local_ba(sh), ba.optimize_flat(sc)
    gs, rs = got["stats"], ref["stats"]
    assert gs["iters_stage1"] == rs[2] and gs["iters_stage2"] == rs[3] and gs["num_gated"] == rs[5]
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert np.array_equal(got["outlier"], ref["outlier"]) and np.array_equal(got["outlier"], plain["outlier"][order])
    _assert_poses(got["pose_cw"], plain["pose_cw"])


def test_local_ba_repeatable(ba):
    sc = S.ba_scene(num_kf=8, num_lm=600, obs_per_lm=5, num_fixed=2, seed=11)
    a = ba.optimize_flat(sc)
    b = ba.optimize_flat(sc)
    assert np.array_equal(a["pose_cw"], b["pose_cw"]) and np.array_equal(a["points"], b["points"])  # fixed-order reductions


def test_local_ba_degenerate_inputs(ba):
    sc = S.ba_scene(num_kf=4, num_lm=50, obs_per_lm=3, num_fixed=4, seed=2)  # every pose fixed: structure-only BA
    got = ba.optimize_flat(sc)
    ref = O.local_ba(sc)
    assert np.array_equal(got["pose_cw"], sc["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    empty = dict(sc, obs_pose=np.zeros(0, np.int32), obs_point=np.zeros(0, np.int32), obs_uvr=np.zeros((0, 3), np.float32),
                 obs_inv_sigma_sq=np.zeros(0, np.float32), obs_huber=np.zeros(0, np.float32))
    got = ba.optimize_flat(empty)
    assert got["rc"] == 0 and np.array_equal(got["pose_cw"], sc["pose_cw"])


@pytest.mark.parametrize("world", [2, 3])
def test_local_ba_sharded_matches_single(ba, world):
    """svgpu_local_ba_sharded with `world` ranks simulated on ONE GPU (one context + thread per rank, the all-reduce
    callback implemented with a thread barrier and device tensors): every rank must return the same poses/points as
    the single-GPU solve (fp64 summation-order differences only) and the union of the outlier shards must agree."""
    import threading
    import torch
    from stella_vslam_amd import distributed as D, feature, optimize

    sc = S.ba_scene(num_kf=12, num_lm=2000, obs_per_lm=5, num_fixed=3, seed=21)
    single = ba.optimize_flat(sc)
    barrier = threading.Barrier(world)
    slots = [None] * world
    total = [None]

    def make_cb(rank):
        def _cb(user, buf, count, stream):
            try:
                t = torch.as_tensor(D._CudaBuf(buf, count), device="cuda")
                torch.cuda.synchronize()
                slots[rank] = t
                barrier.wait()
                if rank == 0:
                    acc = slots[0].clone()
                    for r in range(1, world):
                        acc += slots[r]
                    total[0] = acc
                    torch.cuda.synchronize()
                barrier.wait()
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait()
                return 0
            except Exception as e:  # pragma: no cover
                print("cb failed", e)
                barrier.abort()
                return 1
        return D.ALLREDUCE_FN(_cb)

    results = [None] * world

    def run(rank):
        adj = optimize.local_bundle_adjuster(ctx=feature.Context())
        cb = make_cb(rank)
        results[rank] = adj.optimize_flat_sharded(D.shard_by_landmark(sc, rank, world), rank, world, cb)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert all(r is not None and r["rc"] == 0 for r in results)
    outl = np.zeros(len(sc["obs_pose"]), np.uint8)
    for rank, r in enumerate(results):
        assert np.array_equal(r["pose_cw"], results[0]["pose_cw"]) and np.array_equal(r["points"], results[0]["points"])
        for k in ("iters_stage1", "iters_stage2", "num_gated", "stage2_entered"):
            assert r["stats"][k] == single["stats"][k], k
        assert r["stats"]["chi2_final"] == pytest.approx(single["stats"]["chi2_final"], rel=1e-9)
        outl[(sc["obs_point"] % world) == rank] = r["outlier"]
    assert _rel(results[0]["pose_cw"], single["pose_cw"]) < 1e-9
    assert _rel(results[0]["points"], single["points"]) < 1e-9
    assert np.array_equal(outl, single["outlier"])


@pytest.mark.parametrize("kw", [dict(num_kf=10, num_lm=800, obs_per_lm=5, num_fixed=1, seed=31),
                                dict(num_kf=40, num_lm=3000, obs_per_lm=6, num_fixed=1, seed=32, loop=True)])
def test_global_ba_matches_oracle(ba, kw):
    """global_bundle_adjuster core: one LM run over the whole graph, only the spanning root fixed.  The second case has
    6 * 39 = 234 reduced unknowns in 780 blocks: beyond the on-chip solvers, it goes through the block envelope Cholesky."""
    sc = S.ba_scene(**kw)
    got = ba.optimize_global_flat(sc, num_iter=10)
    ref = O.local_ba(sc, iters1=10, iters2=0)  # the oracle's gate after stage 1 does not move any vertex
    assert got["stats"]["iters_stage1"] == ref["stats"][2] and got["stats"]["stage2_entered"] == 0
    assert got["stats"]["chi2_initial"] == pytest.approx(ref["stats"][0], rel=1e-9)
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    assert got["stats"]["chi2_final"] < 0.5 * got["stats"]["chi2_initial"]
    assert got["stats"]["stopped_by_terminate_action"] in (0, 1)
    assert got["stats"]["pcg_iterations"] == 0 and got["stats"]["cholesky_failures"] == 0  # direct solves at both sizes


def _hub_scene():
    """A banded (loop) keyframe graph plus one equirectangular hub keyframe that observes landmarks all around the loop -- a dense
    loop-closure row in the reduced camera system -- joined with a second, disconnected loop that has its own fixed keyframe."""
    a = S.ba_scene(num_kf=40, num_lm=3000, obs_per_lm=5, num_fixed=1, seed=41, loop=True)
    hub = 7
    a["intr"][hub] = [0.0, 0.0, 1920.0, 960.0, 0.0]
    T = a["pose_gt"][hub].reshape(3, 4)
    rng = np.random.default_rng(9)
    keep = a["obs_pose"] != hub   # the hub's perspective observations go, equirectangular ones of every 4th landmark come
    pc = a["points_gt"][::4] @ T[:, :3].T + T[:, 3]
    theta = np.arctan2(pc[:, 0], pc[:, 2])
    ok = (np.abs(theta) < 2.8) & (np.linalg.norm(pc, axis=1) > 0.5)
    lm = np.arange(0, len(a["points"]), 4)[ok]
    pc = pc[ok]
    u = 1920.0 * (0.5 + np.arctan2(pc[:, 0], pc[:, 2]) / (2 * np.pi)) + rng.normal(0, 1, len(lm))
    v = 960.0 * (0.5 + np.arcsin(pc[:, 1] / np.linalg.norm(pc, axis=1)) / np.pi) + rng.normal(0, 1, len(lm))
    extra = dict(obs_pose=np.full(len(lm), hub, np.int32), obs_point=lm.astype(np.int32),
                 obs_uvr=np.stack([u, v, np.full(len(lm), -1.0)], 1).astype(np.float32),
                 obs_inv_sigma_sq=np.ones(len(lm), np.float32), obs_huber=np.full(len(lm), np.float32(np.sqrt(5.991)), np.float32))
    for k, x in extra.items():
        a[k] = np.concatenate([a[k][keep], x])
    order = np.argsort(a["obs_point"], kind="stable")   # landmark-major, as the adaptors deliver it
    for k in extra:
        a[k] = a[k][order]
    b = S.ba_scene(num_kf=24, num_lm=1500, obs_per_lm=5, num_fixed=1, seed=42, loop=True)
    out = {}
    for k in a:
        if k == "obs_pose":
            out[k] = np.concatenate([a[k], b[k] + len(a["pose_cw"])]).astype(np.int32)
        elif k == "obs_point":
            out[k] = np.concatenate([a[k], b[k] + len(a["points"])]).astype(np.int32)
        else:
            out[k] = np.concatenate([a[k], b[k]])
    return out


def test_global_ba_envelope_with_hub_and_two_components():
    """The envelope Cholesky on a block graph that is not a band: a dense hub row (what a loop closure looks like after the reverse
    Cuthill-McKee ordering) and two disconnected components (62 free keyframes; at this size the default would be the LDS-resident PCG, the second leg)."""
    from stella_vslam_amd import optimize
    sc = _hub_scene()
    ref = O.local_ba(sc, iters1=10, iters2=0)
    for solver in (optimize.SOLVER_ENVELOPE, optimize.SOLVER_PCG):
        got = optimize.local_bundle_adjuster().set_solver(solver).optimize_global_flat(sc, num_iter=10)
        assert got["stats"]["iters_stage1"] == ref["stats"][2] and got["stats"]["cholesky_failures"] == 0
        assert (got["stats"]["pcg_iterations"] == 0) == (solver == optimize.SOLVER_ENVELOPE)
        _assert_poses(got["pose_cw"], ref["pose_cw"])
        assert _rel(got["points"], ref["points"]) < TOL


@pytest.mark.parametrize("seed,stereo,reset", [(4, False, False), (5, False, True), (6, True, False)])
def test_pose_optimizer_matches_oracle(ba, seed, stereo, reset):
    """pose_optimizer (motion-only BA) as one persistent kernel vs the oracle: same LM iteration count, identical outlier flags,
    pose within 1e-4 relative."""
    from stella_vslam_amd import optimize
    from tests.test_oracle_ba import _pose_problem
    pr = _pose_problem(seed, stereo=stereo)
    po = optimize.pose_optimizer(ctx=ba.ctx, reset_stop_flag_each_round=reset)
    nv, pose, outl, iters = po.optimize_flat(pr["pose_cw"], pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], pr["intr"])
    nvo, poseo, outlo, st = O.pose_optimize(pr["pose_cw"], pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], pr["intr"],
                                            reset_flag_each_round=reset)
    assert iters == st[0] and nv == nvo
    assert np.array_equal(outl, outlo)
    assert _rel(pose, poseo) < TOL
    assert np.abs(pose - pr["pose_gt"]).max() < np.abs(pr["pose_cw"] - pr["pose_gt"]).max()
    nv0, pose0, outl0, it0 = po.optimize_flat(pr["pose_cw"], pr["pos_w"][:3], pr["uvr"][:3], pr["inv_sigma_sq"][:3], pr["huber"][:3], pr["intr"])
    assert nv0 == 0 and np.array_equal(pose0, pr["pose_cw"].reshape(12))
    # observations already resident on the device: identical results
    import torch
    dev = [torch.from_numpy(np.ascontiguousarray(pr[k], t)).cuda() for k, t in (("pos_w", np.float64), ("uvr", np.float32), ("inv_sigma_sq", np.float32), ("huber", np.float32))]
    torch.cuda.synchronize()
    nvd, posed, outld, itd = po.optimize_device(pr["pose_cw"], len(pr["pos_w"]), *dev, pr["intr"])
    assert nvd == nv and itd == iters and np.array_equal(posed, pose) and np.array_equal(outld, outl)


def test_pose_optimizer_equirectangular_matches_oracle():
    """equirectangular_pose_opt_edge.h:64-127 through the same persistent kernel."""
    from stella_vslam_amd import optimize
    sc = S.ba_scene(num_kf=3, num_lm=1000, obs_per_lm=3, num_fixed=0, seed=6, outlier_frac=0.1, equirect=True)
    sel = sc["obs_pose"] == 1
    pr = dict(pose_cw=sc["pose_cw"][1], pose_gt=sc["pose_gt"][1], pos_w=sc["points_gt"][sc["obs_point"][sel]], uvr=sc["obs_uvr"][sel],
              inv_sigma_sq=sc["obs_inv_sigma_sq"][sel], huber=sc["obs_huber"][sel], intr=sc["intr"][1])
    nv, pose, outl, iters = optimize.pose_optimizer().optimize_flat(pr["pose_cw"], pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"],
                                                                    pr["intr"])
    rnv, rpose, routl, rst = O.pose_optimize(pr["pose_cw"], pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], pr["intr"])
    assert nv == rnv and iters == int(rst[0]) and np.array_equal(outl, routl)
    assert _rel(pose, rpose) < TOL
    assert np.abs(pose - pr["pose_gt"]).max() < 0.1 * np.abs(pr["pose_cw"] - pr["pose_gt"]).max()


@pytest.mark.parametrize("solver", ["pcg", "pcg_multi", "dense", "envelope", "cholesky_mfma"])
@pytest.mark.parametrize("kw", [dict(num_kf=10, num_lm=1500, obs_per_lm=5, num_fixed=3, seed=5),
                                dict(num_kf=20, num_lm=10000, obs_per_lm=6, num_fixed=4, seed=1234)])
def test_local_ba_alternative_solvers_match_oracle(kw, solver):
    """A local-BA sized reduced camera system is factored by the dense LL^T in LDS by default.  The other solvers -- the PCG that
    lives in one workgroup's LDS (north_star: "Schur-complement J^T J build + PCG solve"), the one-launch-per-iteration PCG of the
    larger sizes, the dense LL^T on the global-memory image, the block envelope Cholesky of the global-BA sizes -- must walk the same LM schedule to the same poses
    and outliers."""
    from stella_vslam_amd import optimize
    code = dict(pcg=optimize.SOLVER_PCG, pcg_multi=optimize.SOLVER_PCG_MULTI, dense=optimize.SOLVER_DENSE, envelope=optimize.SOLVER_ENVELOPE,
                cholesky_mfma=optimize.SOLVER_CHOLESKY_MFMA)[solver]
    adj = optimize.local_bundle_adjuster().set_solver(code)
    sc = S.ba_scene(**kw)
    got = adj.optimize_flat(sc)
    ref = O.local_ba(sc)
    gs, rs = got["stats"], ref["stats"]
    assert gs["iters_stage1"] == rs[2] and gs["iters_stage2"] == rs[3] and gs["num_gated"] == rs[5]
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert (gs["pcg_iterations"] > 0) == (solver not in ("dense", "envelope", "cholesky_mfma")) and gs["cholesky_failures"] == 0


def test_global_ba_config5_matches_oracle():
    """BASELINE config 5: 500 keyframes on a loop / 200 k landmarks / 1.2 M observations, only the root fixed
    (optimize/global_bundle_adjuster.cc:26-192).  2 994 reduced unknowns in ~6.6 k blocks: block envelope Cholesky on the device (what the
    default takes at this size) and in the oracle; the block-Jacobi PCG (relative residual 1e-10) must land on the same estimate."""
    from stella_vslam_amd import optimize
    sc = S.ba_scene_large()
    adj = optimize.local_bundle_adjuster()
    got = adj.optimize_global_flat(sc, num_iter=10)
    ref = O.local_ba(sc, iters1=10, iters2=0, want_trace=True)  # (the oracle's own "chi2 after" is taken behind its outlier gate)
    gs = got["stats"]
    assert gs["iters_stage1"] == ref["stats"][2] and gs["stage2_entered"] == 0
    assert gs["chi2_initial"] == pytest.approx(ref["stats"][0], rel=1e-9)
    last = ref["trace"][int(ref["stats"][2]) - 1]
    assert gs["chi2_final"] == pytest.approx(last[0], rel=1e-6) and gs["lambda_final"] == pytest.approx(last[1], rel=1e-6)
    _assert_poses(got["pose_cw"], ref["pose_cw"])
    assert _rel(got["points"], ref["points"]) < TOL
    assert gs["pcg_iterations"] == 0 and gs["cholesky_failures"] == 0   # direct solve
    assert gs["chi2_final"] < 0.5 * gs["chi2_initial"]
    again = adj.optimize_global_flat(sc, num_iter=10)
    assert np.array_equal(again["pose_cw"], got["pose_cw"]) and np.array_equal(again["points"], got["points"])  # fixed-order reductions
    pcg = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_PCG).optimize_global_flat(sc, num_iter=10)
    assert pcg["stats"]["pcg_iterations"] > 0 and pcg["stats"]["iters_stage1"] == gs["iters_stage1"]
    _assert_poses(pcg["pose_cw"], got["pose_cw"])
    assert _rel(pcg["points"], got["points"]) < TOL


def test_global_ba_set_up_paths_agree(monkeypatch):
    """The set-up of a global-BA sized call (csrc/svgpu_ba.hip): a team of host threads stages the index arrays first and the measurements
    beside the device's structure work; the solve runs on its own landmark numbering (first observing keyframe) with chunk-major units of the
    Schur kernel.  Every way through that set-up -- one host thread, an odd team, no renumbering, arithmetic shares, observations that arrive
    in random order (the team is abandoned, the permuted copy stages everything) -- must land on the same estimate; the default path is
    repeatable bit for bit (test_global_ba_config5_matches_oracle)."""
    from stella_vslam_amd import optimize
    sc = S.ba_scene_large()
    adj = optimize.local_bundle_adjuster()
    ref = adj.optimize_global_flat(sc, num_iter=10)
    def same(got):
        assert got["stats"]["iters_stage1"] == ref["stats"]["iters_stage1"]
        assert got["stats"]["chi2_final"] == pytest.approx(ref["stats"]["chi2_final"], rel=1e-12)
        assert np.abs(got["pose_cw"] - ref["pose_cw"]).max() < 1e-10 and np.abs(got["points"] - ref["points"]).max() < 1e-10
    for env in ({"SVGPU_BA_HOST_THREADS": "1"}, {"SVGPU_BA_HOST_THREADS": "3"}, {"SVGPU_BA_ONE_THREAD": "1"}, {"SVGPU_BA_NO_RENUMBER": "1"}, {"SVGPU_BA_NO_UNITS": "1"},
                {"SVGPU_BA_CHUNK_SHIFT": "6"}):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            same(adj.optimize_global_flat(sc, num_iter=10))
    perm = np.random.default_rng(7).permutation(len(sc["obs_pose"]))
    shuffled = dict(sc)
    for k in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma_sq", "obs_huber"):
        shuffled[k] = np.ascontiguousarray(sc[k][perm])
    same(adj.optimize_global_flat(shuffled, num_iter=10))
    bad = dict(sc)
    bad["obs_point"] = sc["obs_point"].copy()
    bad["obs_point"][len(bad["obs_point"]) // 2] = len(sc["points"])  # out of range, found by a worker of the team
    pts_before = bad["points"].copy()
    with pytest.raises(Exception):
        adj.optimize_global_flat(bad, num_iter=10)
    assert np.array_equal(bad["points"], pts_before)


def test_global_ba_two_sided_elimination_equals_one_sided(monkeypatch):
    """A long band can be eliminated from both ends by two workgroups (ba_skyline.hip: T | S | B, the Schur complements added on the
    separator) -- the fallback where the segmented plan does not apply, forced here with SVGPU_SKY_SEGMENTS=0; SVGPU_SKY_ONE_SIDED=1 keeps the
    single top-down sweep.  Same blocks, same arithmetic per block, another order of the sums on S: the two must agree far below the parity
    tolerance, with the same LM schedule -- on a ring (the loop closure puts a few wide rows into the band) and on an open chain."""
    from stella_vslam_amd import optimize
    for kw in (dict(num_kf=160, num_lm=40000), dict(num_kf=240, num_lm=30000, obs_per_lm=4), dict(num_kf=200, num_lm=20000, obs_per_lm=2)):  # band widths 11, 7 and a narrow one
        sc = S.ba_scene_large(**kw)
        monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
        monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "0")
        adj = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE)
        two = adj.optimize_global_flat(sc, num_iter=10)
        assert adj.last_envelope_plan()["kind"] == "two-sided"
        monkeypatch.delenv("SVGPU_SKY_SEGMENTS", raising=False)
        monkeypatch.setenv("SVGPU_SKY_ONE_SIDED", "1")
        adj = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE)
        one = adj.optimize_global_flat(sc, num_iter=10)
        assert adj.last_envelope_plan()["kind"] == "one-sided"
        monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
        assert two["stats"]["iters_stage1"] == one["stats"]["iters_stage1"] and two["stats"]["cholesky_failures"] == 0 == one["stats"]["pcg_iterations"]
        assert two["stats"]["chi2_final"] == pytest.approx(one["stats"]["chi2_final"], rel=1e-9)
        assert np.abs(two["pose_cw"] - one["pose_cw"]).max() < 1e-8 and np.abs(two["points"] - one["points"]).max() < 1e-8
        assert two["stats"]["chi2_final"] < 0.6 * two["stats"]["chi2_initial"]


@pytest.mark.parametrize("cuts", [None, 2, 3, 5])
def test_global_ba_segmented_elimination_equals_one_sided(monkeypatch, cuts):
    """The default for a long band: vertex separators cut the RCM-ordered keyframe graph, every connected piece is eliminated by its own
    workgroup towards its separators (k_sky_band in job mode), the separator system collects what they leave and is solved on its own plan,
    the pieces substitute backwards (ba_skyline.hip; the planner is pinned on the CPU by tests/test_sky_segments.py).  Another elimination
    ORDER of the same system: estimates within rounding of the one-sided sweep, the same LM schedule -- on rings of three band widths,
    with the planner's own cut count and forced ones (open chains and disconnected graphs: tests/test_sky_segments.py)."""
    from stella_vslam_amd import optimize
    for kw in (dict(num_kf=160, num_lm=40000), dict(num_kf=240, num_lm=30000, obs_per_lm=4), dict(num_kf=200, num_lm=20000, obs_per_lm=2)):
        sc = S.ba_scene_large(**kw)
        monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
        if cuts is None:
            monkeypatch.delenv("SVGPU_SKY_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("SVGPU_SKY_SEGMENTS", str(cuts))
        adj = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE)
        seg = adj.optimize_global_flat(sc, num_iter=10)
        plan = adj.last_envelope_plan()
        monkeypatch.delenv("SVGPU_SKY_SEGMENTS", raising=False)
        monkeypatch.setenv("SVGPU_SKY_ONE_SIDED", "1")
        one = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE).optimize_global_flat(sc, num_iter=10)
        monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
        if cuts is not None and kw["num_kf"] * 1.0 >= 8 * (plan["widest_column"] + 1):
            assert plan["kind"] == "segmented" and plan["jobs"] >= 2 and plan["jobs_local"] == plan["jobs"], plan
        assert seg["stats"]["iters_stage1"] == one["stats"]["iters_stage1"] and seg["stats"]["cholesky_failures"] == 0 == seg["stats"]["pcg_iterations"]
        assert seg["stats"]["chi2_final"] == pytest.approx(one["stats"]["chi2_final"], rel=1e-9)
        assert np.abs(seg["pose_cw"] - one["pose_cw"]).max() < 1e-8 and np.abs(seg["points"] - one["points"]).max() < 1e-8, plan


def test_stage_boundary_on_the_device_equals_the_host_boundary(monkeypatch):
    """The gate -> second stage hand-over without the host (k_ba_activity + the guarded k_ba_begin) against the host-side boundary
    (SVGPU_BA_HOST_BOUNDARY): same bits, same schedule, same gated count -- on the plain config-3 shaped window, and on one in which a free
    keyframe loses EVERY observation at the gate, so that the pose numbering changes and the enqueued stage must refuse to start and hand
    back to the host path."""
    from stella_vslam_amd import optimize
    adj = optimize.local_bundle_adjuster()
    plain = S.ba_scene(num_kf=14, num_lm=3000, obs_per_lm=5, num_fixed=3, seed=77)
    broken = dict(plain)
    uvr = np.array(plain["obs_uvr"], np.float32, copy=True)
    victim = int(np.flatnonzero(np.asarray(plain["pose_fixed"]) == 0)[2])
    hit = np.asarray(plain["obs_pose"]) == victim
    rng = np.random.default_rng(5)
    uvr[hit, :2] += rng.choice([-1.0, 1.0], (int(hit.sum()), 2)).astype(np.float32) * rng.uniform(60, 90, (int(hit.sum()), 2)).astype(np.float32)
    broken["obs_uvr"] = uvr
    for sc, expect_all_gated in ((plain, False), (broken, True)):
        monkeypatch.delenv("SVGPU_BA_HOST_BOUNDARY", raising=False)
        dev = adj.optimize_flat(sc)
        monkeypatch.setenv("SVGPU_BA_HOST_BOUNDARY", "1")
        host = adj.optimize_flat(sc)
        assert dev["rc"] == 0 and host["rc"] == 0
        assert dev["stats"] == host["stats"], (dev["stats"], host["stats"])
        assert np.array_equal(dev["pose_cw"], host["pose_cw"]) and np.array_equal(dev["points"], host["points"]) and np.array_equal(dev["outlier"], host["outlier"])
        assert dev["stats"]["stage2_entered"] == 1 and dev["stats"]["num_gated"] > 0
        if expect_all_gated:  # every observation of the keyframe is an outlier in the end: it did drop out of the second stage
            assert dev["outlier"][hit].all()


def _simulated_ranks(world, run_rank):
    """`world` ranks as threads on the one test GPU; the all-reduce callback sums the ranks' device buffers through a barrier."""
    import threading
    import torch
    from stella_vslam_amd import distributed as D
    barrier = threading.Barrier(world)
    slots, total = [None] * world, [None]

    def make_cb(rank):
        def _cb(user, buf, count, stream):
            try:
                t = torch.as_tensor(D._CudaBuf(buf, count), device="cuda")
                torch.cuda.synchronize()
                slots[rank] = t
                barrier.wait()
                if rank == 0:
                    acc = slots[0].clone()
                    for r in range(1, world):
                        acc += slots[r]
                    total[0] = acc
                    torch.cuda.synchronize()
                barrier.wait()
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait()
                return 0
            except Exception as e:  # pragma: no cover
                print("cb failed", e)
                barrier.abort()
                return 1
        return D.ALLREDUCE_FN(_cb)

    results = [None] * world

    def run(rank):
        results[rank] = run_rank(rank, make_cb(rank))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    return results


@pytest.mark.parametrize("world", [2, 3, 4])
def test_global_ba_distributed_factorisation_matches_single(world, monkeypatch):
    """svgpu_global_ba_sharded on a system large enough for the segmented plan: every rank eliminates only the jobs it owns, the separator
    contributions and the solution cross ranks through the all-reduce (sums with zeros: an all-gather), the separator system is solved by
    every rank.  `world` ranks simulated on one GPU: every rank owns a strict subset of the jobs, all ranks end bit-identical, and the
    estimate equals the single-GPU solve (whose factorisation is the same segmented one, all jobs on one rank) to 1e-7."""
    from stella_vslam_amd import distributed as D, feature, optimize
    monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    sc = S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)
    adj1 = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE)
    single = adj1.optimize_global_flat(sc, num_iter=10)
    assert adj1.last_envelope_plan()["kind"] == "segmented"
    plans = [None] * world

    def run_rank(rank, cb):
        adj = optimize.local_bundle_adjuster(ctx=feature.Context()).set_solver(optimize.SOLVER_ENVELOPE)
        res = adj.optimize_global_flat_sharded(D.shard_by_landmark(sc, rank, world), rank, world, cb, num_iter=10)
        plans[rank] = adj.last_envelope_plan()
        return res

    results = _simulated_ranks(world, run_rank)
    assert all(r is not None and r["rc"] == 0 for r in results)
    assert all(p["kind"] == "segmented" and 0 < p["jobs_local"] < p["jobs"] for p in plans), plans
    assert sum(p["jobs_local"] for p in plans) == plans[0]["jobs"]
    for r in results[1:]:
        assert np.array_equal(results[0]["pose_cw"], r["pose_cw"]) and np.array_equal(results[0]["points"], r["points"])
    assert results[0]["stats"]["iters_stage1"] == single["stats"]["iters_stage1"]
    _assert_poses(results[0]["pose_cw"], single["pose_cw"], 1e-7)
    assert _rel(results[0]["points"], single["points"]) < 1e-7


@pytest.mark.parametrize("shard", ["landmark", "keyframe_segment"])
def test_global_ba_sharded_through_the_team_set_up(shard, monkeypatch):
    """A sharded solve whose shards go through the global-BA sized set-up (host team, the shard's own landmark numbering, chunk-major Schur
    units): forced onto a small scene with SVGPU_BA_TEAM_MIN_OBS / SVGPU_BA_UNITS_MIN.  The numbering is private to a rank's kernels --
    ownership marks and the final positions cross ranks in the caller's numbering -- so: all ranks bit-identical, equal to the single-GPU
    solve (itself through the same set-up) to 1e-7, and to the plain set-up of the same shards to 1e-7."""
    from stella_vslam_amd import distributed as D, feature, optimize
    monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    world = 2
    sc = S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)
    shard_fn = D.shard_by_landmark if shard == "landmark" else D.shard_by_keyframe_segment

    def run_rank(rank, cb):
        adj = optimize.local_bundle_adjuster(ctx=feature.Context()).set_solver(optimize.SOLVER_ENVELOPE)
        return adj.optimize_global_flat_sharded(shard_fn(sc, rank, world), rank, world, cb, num_iter=10)

    plain = _simulated_ranks(world, run_rank)
    monkeypatch.setenv("SVGPU_BA_TEAM_MIN_OBS", "1000")
    monkeypatch.setenv("SVGPU_BA_UNITS_MIN", "1")
    single = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE).optimize_global_flat(sc, num_iter=10)
    team = _simulated_ranks(world, run_rank)
    assert all(r is not None and r["rc"] == 0 for r in plain + team)
    for r in team[1:]:
        assert np.array_equal(team[0]["pose_cw"], r["pose_cw"]) and np.array_equal(team[0]["points"], r["points"])
    for other in (single, plain[0]):
        assert team[0]["stats"]["iters_stage1"] == other["stats"]["iters_stage1"]
        _assert_poses(team[0]["pose_cw"], other["pose_cw"], 1e-7)
        assert _rel(team[0]["points"], other["points"]) < 1e-7


@pytest.mark.parametrize("world", [2, 3, 4])
def test_global_ba_keyframe_segment_shards_exchange_only_separators(world, monkeypatch):
    """north_star's partition: observations sharded by the keyframe segment that owns them (distributed.shard_by_keyframe_segment over
    svgpu_ba_partition_keyframe_segments).  The solve recognises such shards and, per damping trial, all-reduces only the separator blocks,
    what the jobs leave on the separators and the solution -- not the reduced camera system.  `world` ranks simulated on one GPU: all ranks
    bit-identical, equal to the single-GPU solve to 1e-7 with the same LM schedule, and far fewer bytes than the l % world shards move."""
    from stella_vslam_amd import distributed as D, feature, optimize
    monkeypatch.delenv("SVGPU_SKY_ONE_SIDED", raising=False)
    monkeypatch.delenv("SVGPU_BA_EXCHANGE", raising=False)
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    sc = S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)
    single = optimize.local_bundle_adjuster().set_solver(optimize.SOLVER_ENVELOPE).optimize_global_flat(sc, num_iter=10)
    lm_rank, info = optimize.partition_keyframe_segments(sc, world)
    assert info["segmented"] == 1 and info["jobs"] >= world and 0 < info["separator_keyframes"] < info["free_keyframes"] // 4
    xch = {}

    def run(shard_fn):
        def run_rank(rank, cb):
            adj = optimize.local_bundle_adjuster(ctx=feature.Context()).set_solver(optimize.SOLVER_ENVELOPE)
            res = adj.optimize_global_flat_sharded(shard_fn(sc, rank, world), rank, world, cb, num_iter=10)
            xch[(shard_fn.__name__, rank)] = adj.last_exchange()
            return res
        return _simulated_ranks(world, run_rank)

    seg = run(D.shard_by_keyframe_segment)
    assert all(r is not None and r["rc"] == 0 for r in seg)
    for r in seg[1:]:
        assert np.array_equal(seg[0]["pose_cw"], r["pose_cw"]) and np.array_equal(seg[0]["points"], r["points"])
    assert seg[0]["stats"]["iters_stage1"] == single["stats"]["iters_stage1"] and seg[0]["stats"]["lm_trials"] == single["stats"]["lm_trials"]
    _assert_poses(seg[0]["pose_cw"], single["pose_cw"], 1e-7)
    assert _rel(seg[0]["points"], single["points"]) < 1e-7
    mod = run(D.shard_by_landmark)
    assert all(r is not None and r["rc"] == 0 for r in mod)
    _assert_poses(mod[0]["pose_cw"], single["pose_cw"], 1e-7)
    for rank in range(world):
        a, b = xch[("shard_by_keyframe_segment", rank)], xch[("shard_by_landmark", rank)]
        assert a["mode"] == "keyframe segments" and b["mode"] == "whole reduced system", (a, b)
        assert a["trials"] == b["trials"] and a["separator_bytes"] == b["separator_bytes"] and a["solution_bytes"] == b["solution_bytes"]
        assert a["reduced_system_bytes"] * 4 < b["reduced_system_bytes"], (a, b)


def test_keyframe_segment_exchange_can_be_switched_off(monkeypatch):
    """SVGPU_BA_EXCHANGE=full: the same shards, the whole reduced system per trial (the A/B switch DESIGN section 7 quotes)."""
    from stella_vslam_amd import distributed as D, feature, optimize
    monkeypatch.setenv("SVGPU_SKY_SEGMENTS", "5")
    monkeypatch.setenv("SVGPU_BA_EXCHANGE", "full")
    sc = S.ba_scene_large(num_kf=200, num_lm=24000, obs_per_lm=4)
    modes = [None, None]

    def run_rank(rank, cb):
        adj = optimize.local_bundle_adjuster(ctx=feature.Context()).set_solver(optimize.SOLVER_ENVELOPE)
        res = adj.optimize_global_flat_sharded(D.shard_by_keyframe_segment(sc, rank, 2), rank, 2, cb, num_iter=3)
        modes[rank] = adj.last_exchange()["mode"]
        return res

    res = _simulated_ranks(2, run_rank)
    assert all(r is not None and r["rc"] == 0 for r in res) and modes == ["whole reduced system"] * 2


def test_sharded_through_rccl_communicator_world1():
    """svgpu_comm_init (RCCL resolved with dlopen inside the library) + svgpu_local_ba_sharded with allreduce = NULL: one rank is
    all a single GPU allows (RCCL refuses two ranks on one device), but it drives every collective of the sharded solve through
    ncclAllReduce on the library's stream; the result must equal the plain single-GPU solve bit for bit (a sum over one rank)."""
    import ctypes as C
    from stella_vslam_amd import distributed as D, feature, optimize
    from stella_vslam_amd._lib import lib
    ctx = feature.Context()
    ident = np.zeros(128, np.uint8)
    ctx.check(lib().svgpu_comm_unique_id(C.c_void_p(ident.ctypes.data)), "svgpu_comm_unique_id")
    ctx.check(lib().svgpu_comm_init(ctx.handle, 0, 1, C.c_void_p(ident.ctypes.data)), "svgpu_comm_init")
    adj = optimize.local_bundle_adjuster(ctx=ctx)
    sc = S.ba_scene(num_kf=12, num_lm=2000, obs_per_lm=5, num_fixed=3, seed=21)
    single = adj.optimize_flat(sc)
    shard = adj.optimize_flat_sharded(D.shard_by_landmark(sc, 0, 1), 0, 1, None)
    assert shard["rc"] == 0
    assert np.array_equal(shard["pose_cw"], single["pose_cw"]) and np.array_equal(shard["points"], single["points"])
    assert np.array_equal(shard["outlier"], single["outlier"])
    lib().svgpu_comm_destroy(ctx.handle)


def test_global_ba_sharded_matches_single():
    """svgpu_global_ba_sharded (PCG on the all-reduced blocks) with 2 ranks simulated on one GPU."""
    import threading
    import torch
    from stella_vslam_amd import distributed as D, feature, optimize

    sc = S.ba_scene(num_kf=40, num_lm=3000, obs_per_lm=6, num_fixed=1, seed=32, loop=True)
    single = optimize.local_bundle_adjuster().optimize_global_flat(sc, num_iter=10)
    world = 2
    barrier = threading.Barrier(world)
    slots, total = [None] * world, [None]

    def make_cb(rank):
        def _cb(user, buf, count, stream):
            try:
                t = torch.as_tensor(D._CudaBuf(buf, count), device="cuda")
                torch.cuda.synchronize()
                slots[rank] = t
                barrier.wait()
                if rank == 0:
                    acc = slots[0].clone()
                    for r in range(1, world):
                        acc += slots[r]
                    total[0] = acc
                    torch.cuda.synchronize()
                barrier.wait()
                t.copy_(total[0])
                torch.cuda.synchronize()
                barrier.wait()
                return 0
            except Exception as e:  # pragma: no cover
                print("cb failed", e)
                barrier.abort()
                return 1
        return D.ALLREDUCE_FN(_cb)

    results = [None] * world

    def run(rank):
        adj = optimize.local_bundle_adjuster(ctx=feature.Context())
        cb = make_cb(rank)
        results[rank] = adj.optimize_global_flat_sharded(D.shard_by_landmark(sc, rank, world), rank, world, cb, num_iter=10)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(r is not None and r["rc"] == 0 for r in results)
    assert np.array_equal(results[0]["pose_cw"], results[1]["pose_cw"]) and np.array_equal(results[0]["points"], results[1]["points"])
    assert results[0]["stats"]["iters_stage1"] == single["stats"]["iters_stage1"]
    _assert_poses(results[0]["pose_cw"], single["pose_cw"], 1e-7)
    assert _rel(results[0]["points"], single["points"]) < 1e-7
