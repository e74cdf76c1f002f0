"""Pins the oracle's bundle-adjustment graph elements (oracle/ba_oracle.c: edge_error, reproj_edge_jacobians, the vertex updates, the
information / Huber-width / level conventions) against the REFERENCE's own optimize/internal headers, compiled where they lie into
oracle/_ref/libsvref_opt.so: landmark_vertex.h, se3/shot_vertex.h, se3/perspective_reproj_edge.h, se3/equirectangular_reproj_edge.h,
se3/perspective_pose_opt_edge.h, se3/equirectangular_pose_opt_edge.h and the two wrappers (model -> edge type, information = inv_sigma_sq * I,
Huber delta = sqrt_chi_sq, level 1 = outlier).  g2o itself is absent: the stand-in's SE3Quat forwards to the oracle's SE3 helpers, so what is
pinned is every formula the reference wrote, evaluated on identical camera-frame points; g2o's own SE3 arithmetic and solver are not."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref_opt.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref_opt.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return np.ascontiguousarray(q), rng.uniform(-1, 1, 3)


def _scene(rng, model):
    """One pose, one landmark in front of (or, for the equirectangular model, anywhere around) the camera, one observation."""
    q, t = _random_pose(rng)
    Rr = np.zeros(9)
    O.lib().orc_dbg_quat_to_R(_p(q), _p(Rr))
    R = Rr.reshape(3, 3)
    pc = np.array([rng.uniform(-3, 3), rng.uniform(-2, 2), rng.uniform(0.5, 20)]) if model != 2 else rng.normal(size=3) * rng.uniform(0.5, 10)
    if model != 2 and rng.uniform() < 0.1:
        pc[2] = -pc[2]   # behind the camera: depth_is_positive false
    pw = R.T @ (pc - t)
    return q, t, np.ascontiguousarray(pw)


@pytest.mark.parametrize("model", [0, 1, 2, 3])
@pytest.mark.parametrize("stereo", [0, 1])
def test_reproj_and_pose_opt_edges(ref, model, stereo):
    if model == 2 and stereo:
        pytest.skip("the equirectangular model is monocular only (reproj_edge_wrapper.h:143)")
    rng = np.random.default_rng(100 + 10 * model + stereo)
    cols, rows = 1280, 640
    lib = O.lib()
    for it in range(300):
        q, t, pw = _scene(rng, model)
        intr = np.array([rng.uniform(300, 900), rng.uniform(300, 900), rng.uniform(500, 700), rng.uniform(250, 400), rng.uniform(20, 80)])
        K5 = intr.copy() if model != 2 else np.array([0.0, 0.0, cols, rows, 0.0])
        right = rng.uniform(0, 1200) if (stereo and rng.uniform() < 0.7) else -1.0
        uvr = np.array([rng.uniform(0, cols), rng.uniform(0, rows), right], np.float32)
        inv_sigma_sq = np.float32(1.2 ** (-2 * int(rng.integers(0, 8))))
        sqrt_chi = np.float32(np.sqrt(5.991) if right < 0 else np.sqrt(7.815))
        # oracle
        err_o, A_o, B_o = np.zeros(3), np.zeros(9), np.zeros(18)
        lib.orc_dbg_reproj_edge.restype = C.c_int
        D_o = lib.orc_dbg_reproj_edge(_p(q), _p(t), _p(pw), _p(K5), _p(uvr), _p(err_o), _p(A_o), _p(B_o))
        # reference: binary edge
        err_r, A_r, B_r, meta = np.zeros(3), np.zeros(9), np.zeros(18), np.zeros(6)
        use_huber = int(rng.uniform() < 0.8)
        ref.svref_reproj_edge.restype = C.c_int
        D_r = ref.svref_reproj_edge(model, stereo, cols, rows, _p(intr), _p(q), _p(t), _p(pw), _p(uvr), C.c_float(inv_sigma_sq), C.c_float(sqrt_chi),
                                    use_huber, _p(err_r), _p(A_r), _p(B_r), _p(meta))
        assert D_r == D_o == (3 if right >= 0 else 2)
        np.testing.assert_array_equal(err_r[:D_r], err_o[:D_r])
        np.testing.assert_array_equal(A_r[:3 * D_r], A_o[:3 * D_r])
        np.testing.assert_array_equal(B_r[:6 * D_r], B_o[:6 * D_r])
        assert meta[0] == float(inv_sigma_sq)                       # information = inv_sigma_sq * Identity
        assert meta[1] == (float(sqrt_chi) if use_huber else -1.0)   # Huber delta
        chi_o = float(np.dot(err_o[:D_o], float(inv_sigma_sq) * err_o[:D_o]))
        assert abs(meta[2] - chi_o) <= 1e-12 * max(chi_o, 1.0)
        # depth gate as the oracle applies it (depth_ok): camera-frame z, always true for the equirectangular model
        pc = np.zeros(3)
        lib.orc_dbg_se3_map(_p(q), _p(t), _p(pw), _p(pc))
        assert bool(meta[3]) == (model == 2 or pc[2] > 0)
        assert meta[4] == 11 and meta[5] == 10                        # outlier = level 1, inlier = level 0
        # reference: unary (pose-only) edge -- same error and the same pose block
        err_u, B_u, meta_u = np.zeros(3), np.zeros(18), np.zeros(6)
        ref.svref_pose_opt_edge.restype = C.c_int
        D_u = ref.svref_pose_opt_edge(model, stereo, cols, rows, _p(intr), _p(q), _p(t), _p(pw), _p(uvr), C.c_float(inv_sigma_sq), C.c_float(sqrt_chi),
                                      _p(err_u), _p(B_u), _p(meta_u))
        assert D_u == D_o
        np.testing.assert_array_equal(err_u[:D_u], err_o[:D_u])
        np.testing.assert_array_equal(B_u[:6 * D_u], B_o[:6 * D_u])
        assert meta_u[0] == float(inv_sigma_sq) and meta_u[1] == float(sqrt_chi)
        assert bool(meta_u[3]) == (model == 2 or pc[2] > 0)
        assert meta_u[4] == 11 and meta_u[5] == 10


def test_vertex_updates(ref):
    rng = np.random.default_rng(7)
    lib = O.lib()
    for it in range(200):
        q, t = _random_pose(rng)
        upd6 = rng.normal(size=6) * rng.choice([1e-9, 1e-3, 0.3])
        pos, upd3 = rng.uniform(-5, 5, 3), rng.normal(size=3)
        q_r, t_r, p_r, org = np.zeros(4), np.zeros(3), np.zeros(3), np.zeros(10)
        ref.svref_vertex_oplus(_p(q), _p(t), _p(upd6), _p(q_r), _p(t_r), _p(pos), _p(upd3), _p(p_r), _p(org))
        q_o, t_o = np.zeros(4), np.zeros(3)
        lib.orc_dbg_se3_exp_mul(_p(upd6), _p(q), _p(t), _p(q_o), _p(t_o))   # exp(update) * estimate, shot_vertex.h:54
        np.testing.assert_array_equal(q_r, q_o)
        np.testing.assert_array_equal(t_r, t_o)
        np.testing.assert_array_equal(p_r, pos + upd3)                        # landmark_vertex.h:51
        np.testing.assert_array_equal(org, [0, 0, 0, 1, 0, 0, 0, 0, 0, 0])


def test_terminate_action(ref):
    """optimize/terminate_action.cc compiled from the reference (over a scripted optimizer that reports the chi2 of each iteration) against
    the oracle's post-iteration rule: iteration 0 stores the chi2, later iterations raise the force-stop flag when
    0 <= (last - now) / now < gain threshold -- through the caller's flag or, when the optimizer has none, g2o's own."""
    rng = np.random.default_rng(11)
    lib = O.lib()
    for trial in range(200):
        n = int(rng.integers(2, 16))
        it = np.arange(n, dtype=np.int32)
        chi = np.empty(n)
        chi[0] = rng.uniform(1e2, 1e6)
        for k in range(1, n):   # mostly small gains around the threshold, sometimes an increase
            chi[k] = chi[k - 1] * (1.0 - rng.choice([rng.uniform(-2e-3, 4e-3), rng.uniform(0, 0.5), 0.0, 1e-3]))
        thr = float(rng.choice([1e-3, 1e-6, 0.05]))
        raised_o, last_o = np.zeros(n, np.uint8), np.zeros(n)
        lib.orc_dbg_terminate(n, _p(it), _p(chi), C.c_double(thr), _p(raised_o), _p(last_o))
        for install in (0, 1):
            raised_r, last_r, by_r = np.zeros(n, np.uint8), np.zeros(n), np.zeros(n, np.uint8)
            ref.svref_terminate_sequence(n, _p(it), _p(chi), C.c_double(thr), install, _p(raised_r), _p(last_r), _p(by_r))
            np.testing.assert_array_equal(raised_r, raised_o)
            np.testing.assert_array_equal(last_r, last_o)
            assert by_r[-1] == (1 if raised_o.any() else 0)   # sticky until a "reset" call
    # the reset call (iteration < 0) lowers the flag and clears stopped_by_terminate_action_
    it = np.array([0, 1, -1, 0], np.int32)
    chi = np.array([100.0, 99.99, 50.0, 50.0])
    raised, last, by = np.zeros(4, np.uint8), np.zeros(4), np.zeros(4, np.uint8)
    ref.svref_terminate_sequence(4, _p(it), _p(chi), C.c_double(1e-3), 1, _p(raised), _p(last), _p(by))
    assert list(raised) == [0, 1, 0, 0] and list(by) == [0, 1, 0, 0]


def _pose_scene(rng, model, stereo, n):
    """A camera pose, n keypoints with landmarks around the true pose, a share of gross outliers, some keypoints without a landmark
    and some whose landmark is about to be erased."""
    q, t = _random_pose(rng)
    Rr = np.zeros(9)
    O.lib().orc_dbg_quat_to_R(_p(q), _p(Rr))
    R = Rr.reshape(3, 3)
    cols, rows = (1920, 960) if model == 2 else (1280, 720)
    intr = np.array([600.0, 610.0, 640.0, 360.0, 45.0 if stereo else 0.0])
    pc = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(2, 15, n)], 1)
    if model == 2:
        pc = rng.normal(size=(n, 3)) * rng.uniform(2, 10, (n, 1))
        pc = pc[np.abs(np.arctan2(pc[:, 0], pc[:, 2])) < 2.6][: n]
        n = len(pc)
    pw = (pc - t) @ R   # R^T (pc - t)
    octave = rng.integers(0, 8, n).astype(np.int32)
    sig = 1.2 ** octave
    if model == 2:
        L = np.linalg.norm(pc, axis=1)
        u = cols * (0.5 + np.arctan2(pc[:, 0], pc[:, 2]) / (2 * np.pi))
        v = rows * (0.5 + np.arcsin(pc[:, 1] / L) / np.pi)
    else:
        u = intr[0] * pc[:, 0] / pc[:, 2] + intr[2]
        v = intr[1] * pc[:, 1] / pc[:, 2] + intr[3]
    kp = np.stack([u + rng.normal(0, 1, n) * sig, v + rng.normal(0, 1, n) * sig], 1)
    bad = rng.uniform(size=n) < 0.15
    kp[bad] += rng.uniform(15, 60, (int(bad.sum()), 2)) * rng.choice([-1, 1], (int(bad.sum()), 2))
    xr = None
    if stereo:
        xr = (kp[:, 0] - intr[4] / pc[:, 2] + rng.normal(0, 1, n) * sig).astype(np.float32)
        xr[rng.uniform(size=n) < 0.3] = -1.0   # no stereo match for this keypoint
    state = np.ones(n, np.uint8)
    state[rng.uniform(size=n) < 0.1] = 0
    state[rng.uniform(size=n) < 0.05] = 2
    # the initial pose: the true one, perturbed
    dq = np.concatenate([rng.normal(0, 0.01, 3), [1.0]])
    q0 = np.array([dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1], dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0],
                   dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3], dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]])
    q0 /= np.linalg.norm(q0)
    R0 = np.zeros(9)
    O.lib().orc_dbg_quat_to_R(_p(np.ascontiguousarray(q0)), _p(R0))
    pose = np.concatenate([R0.reshape(3, 3), (t + rng.normal(0, 0.05, 3))[:, None]], 1).reshape(-1)
    return dict(cols=cols, rows=rows, intr=intr, pose=np.ascontiguousarray(pose), kp=np.ascontiguousarray(kp, dtype=np.float32), octave=octave, xr=xr,
                pw=np.ascontiguousarray(pw), state=state, n=n)


@pytest.mark.parametrize("model,stereo", [(0, 0), (0, 1), (1, 0), (2, 0), (3, 1)])
def test_pose_optimizer_schedule(ref, model, stereo):
    """optimize/pose_optimizer_g2o.cc compiled from the reference (all three overloads), with g2o's optimize() played by the oracle's own
    pose-only Levenberg-Marquardt: which keypoints get an edge, the rounds, the chi-square classification with its float thresholds,
    the kernel removal after the robust rounds, the early exit below five inliers and the returned count must equal the oracle's
    orc_pose_optimize -- pose and outlier flags bit for bit, for both readings of g2o's "iteration -1" call."""
    rng = np.random.default_rng(300 + 10 * model + stereo)
    lib = O.lib()
    for trial in range(40):
        n = int(rng.choice([3, 6, 40, 300]))
        sc = _pose_scene(rng, model, stereo, n)
        n = sc["n"]
        tr, tn, each = [(2, 2, 10), (0, 3, 5), (2, 0, 10), (1, 1, 3)][trial % 4]
        for reset in (0, 1):
            pose_r, flags_r, its_r = np.zeros(12), np.zeros(n, np.uint8), C.c_int(0)
            ref.svref_pose_optimize.restype = C.c_int
            valid_r = ref.svref_pose_optimize(model, stereo, sc["cols"], sc["rows"], _p(sc["intr"]), _p(sc["pose"]), n, _p(sc["kp"]), _p(sc["octave"]),
                                              None if sc["xr"] is None else _p(sc["xr"]), _p(sc["pw"]), _p(sc["state"]), C.c_float(1.2), 8, tr, tn, each,
                                              reset, trial % 3, _p(pose_r), _p(flags_r), C.byref(its_r))
            # the oracle takes the observations that have a live landmark, as the adaptors gather them
            keep = np.flatnonzero(sc["state"] == 1)
            m = len(keep)
            uvr = np.concatenate([sc["kp"][keep], (sc["xr"][keep] if sc["xr"] is not None else np.full(m, -1, np.float32))[:, None]], 1).astype(np.float32)
            w = np.array([O.scale_tables(1.2, 8)[3][o] for o in sc["octave"][keep]], np.float32)
            pw_k, uvr = np.ascontiguousarray(sc["pw"][keep]), np.ascontiguousarray(uvr)   # (kept alive across the call)
            hub = np.full(m, np.float32(np.sqrt(np.float32(7.81473))) if stereo else np.float32(np.sqrt(np.float32(5.99146))), np.float32)
            K5 = sc["intr"].copy() if model != 2 else np.array([0.0, 0.0, sc["cols"], sc["rows"], 0.0])
            pose_o, out_o, stats = np.zeros(12), np.zeros(max(m, 1), np.uint8), np.zeros(4)
            lib.orc_pose_optimize.restype = C.c_int
            valid_o = lib.orc_pose_optimize(_p(sc["pose"]), m, _p(pw_k), _p(uvr), _p(w), _p(hub), _p(K5),
                                            tr, tn, each, C.c_double(1e-3), reset, _p(pose_o), _p(out_o), _p(stats))
            assert valid_r == valid_o
            if m >= 5:
                np.testing.assert_array_equal(flags_r[keep], out_o[:m])
                assert not flags_r[sc["state"] != 1].any()
                np.testing.assert_allclose(pose_r, pose_o, rtol=0, atol=1e-15)
                assert its_r.value == int(stats[0])
            else:
                np.testing.assert_array_equal(pose_r, sc["pose"])   # fewer than five observations: nothing is touched
