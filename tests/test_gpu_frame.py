"""GPU parity: device frame observation (undistort / bearings / grid), landmark reprojection (frame::can_observe) and the fused
can_observe + projection::match_frame_and_landmarks pass, through the C ABI, against the CPU oracle.
Bit-exact wherever the arithmetic is + - * / sqrt (perspective, radial_division, fisheye's projective part); the members that
call libm (fisheye tan, equirectangular asin / atan2 / sin / cos, logf in predict_scale_level) are compared with the tolerance
written at each assertion."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu

EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0))
TUMVI = dict(fx=190.97847715128717, fy=190.9733070521226, cx=254.93170605935475, cy=256.8974428996504,
             k=(0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182))


@pytest.fixture(scope="module")
def mods():
    from stella_vslam_amd import camera, data, feature, match
    return camera, data, feature, match


def _cameras(cam_mod, ctx):
    p = cam_mod.perspective("euroc", "Stereo", "Gray", 752, 480, 20.0, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], *EUROC["k"],
                            focal_x_baseline=47.9, ctx=ctx)
    f = cam_mod.fisheye("tumvi", "Monocular", "Gray", 512, 512, 20.0, TUMVI["fx"], TUMVI["fy"], TUMVI["cx"], TUMVI["cy"], *TUMVI["k"], ctx=ctx)
    e = cam_mod.equirectangular("theta", "RGB", 1920, 960, 30.0, ctx=ctx)
    r = cam_mod.radial_division("rd", "Monocular", "Gray", 640, 480, 30.0, 320.0, 318.0, 319.5, 239.5, -0.25, ctx=ctx)
    op = O.make_camera(O.CAM_PERSPECTIVE, 752, 480, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], EUROC["k"], 47.9)
    of = O.make_camera(O.CAM_FISHEYE, 512, 512, TUMVI["fx"], TUMVI["fy"], TUMVI["cx"], TUMVI["cy"], TUMVI["k"])
    oe = O.make_camera(O.CAM_EQUIRECTANGULAR, 1920, 960)
    orr = O.make_camera(O.CAM_RADIAL_DIVISION, 640, 480, 320.0, 318.0, 319.5, 239.5, (-0.25,))
    return [("perspective", p, op), ("fisheye", f, of), ("equirectangular", e, oe), ("radial_division", r, orr)]


def _keypoints(rng, cols, rows, n):
    k = np.zeros(n, O.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(0, cols, n).astype(np.float32), rng.uniform(0, rows, n).astype(np.float32)
    k["size"], k["angle"], k["response"] = 31.0, rng.uniform(0, 360, n), rng.uniform(20, 200, n)
    k["octave"], k["class_id"] = rng.integers(0, 8, n), -1
    return k


def test_image_bounds_undistort_bearings_grid(mods):
    cam_mod, data, feature, _ = mods
    ctx = feature.Context(0)
    rng = np.random.default_rng(0)
    for name, cam, ocam in _cameras(cam_mod, ctx):
        assert cam.img_bounds_.as_tuple() == (ocam.min_x, ocam.max_x, ocam.min_y, ocam.max_y), name
        kps = _keypoints(rng, cam.cols_, cam.rows_, 3000)
        obs = data.frame_observation(cam, kps, np.zeros((len(kps), 32), np.uint8))
        xy = np.stack([kps["x"], kps["y"]], 1)
        und = O.undistort_keypoints(ocam, xy)
        got = np.stack([obs.undist_keypts_["x"], obs.undist_keypts_["y"]], 1)
        if name == "fisheye":  # tan() from two math libraries: identical floats except at rounding ties
            assert np.abs(got - und).max() <= 6.2e-5 and (got != und).mean() < 1e-3
            und = got
        else:
            assert np.array_equal(got, und), name
        for f in ("size", "angle", "octave"):
            assert np.array_equal(obs.undist_keypts_[f], kps[f])
        if name == "equirectangular":  # keypoints copied whole (equirectangular.cc:129-131)
            assert np.array_equal(obs.undist_keypts_["response"], kps["response"])
        else:                          # a fresh cv::KeyPoint: response 0, class_id -1 (camera/base.cc:130-148)
            assert (obs.undist_keypts_["response"] == 0).all() and (obs.undist_keypts_["class_id"] == -1).all()
        brg = O.keypoints_to_bearings(ocam, und)
        if name == "equirectangular":  # sin / cos: <= 2 ulp of fp64
            assert np.abs(obs.bearings_ - brg).max() < 5e-16
        else:
            assert np.array_equal(obs.bearings_, brg), name
        assert np.array_equal(cam.convert_keypoints_to_bearings(obs.undist_keypts_), obs.bearings_)
        assert np.array_equal(cam.undistort_keypoints(kps), obs.undist_keypts_)
        off, items = O.assign_keypoints_to_grid(und[:, 0], und[:, 1], cam.img_bounds_.as_tuple(), 64, 48)
        assert np.array_equal(obs.cell_off_, off) and np.array_equal(obs.cell_items_, items), name
        assert 0.5 * len(kps) < off[-1] <= len(kps)
    # empty input
    e = data.frame_observation(cam, np.zeros(0, O.KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8))
    assert len(e.undist_keypts_) == 0 and e.cell_off_[-1] == 0


def test_matcher_grid_crowded_cells_outside_keypoints_and_large_frames(mods):
    """The one-launch grid build (k_grid_frame_one) orders a cell's keypoints by one thread while the cell is small and by a rank sort of the
    whole workgroup when it is crowded (a dense patch under a coarse grid; here up to every keypoint in ONE cell), drops keypoints outside the
    image bounds, and hands frames beyond 8 192 keypoints to the four-launch form: the cell lists equal the oracle's
    assign_keypoints_to_grid (data/common.cc) in every case, on the equirectangular model (undistortion = identity)."""
    cam_mod, data, feature, _ = mods
    ctx = feature.Context(0)
    cam = cam_mod.equirectangular("theta", "RGB", 1920, 960, 30.0, ctx=ctx)
    bounds = cam.img_bounds_.as_tuple()
    rng = np.random.default_rng(77)

    def check(x, y):
        k = np.zeros(len(x), O.KEYPOINT_DTYPE)
        k["x"], k["y"], k["size"], k["class_id"] = np.asarray(x, np.float32), np.asarray(y, np.float32), 31.0, -1
        obs = data.frame_observation(cam, k, np.zeros((len(k), 32), np.uint8))
        off, items = O.assign_keypoints_to_grid(k["x"], k["y"], bounds, 64, 48)
        assert np.array_equal(obs.cell_off_, off) and np.array_equal(obs.cell_items_, items)
        return off

    # 2 500 keypoints in one 4 x 4 px patch (one cell, far beyond the one-thread limit) among 500 scattered ones, in shuffled order
    x = np.concatenate([rng.uniform(100, 104, 2500), rng.uniform(0, 1920, 500)])
    y = np.concatenate([rng.uniform(50, 54, 2500), rng.uniform(0, 960, 500)])
    perm = rng.permutation(len(x))
    off = check(x[perm], y[perm])
    assert np.diff(off).max() >= 2500
    # several crowded cells of different sizes next to each other + keypoints outside the bounds (negative, beyond the far edges)
    xs, ys = [], []
    for cx, cy, n in ((15, 10, 30), (45, 10, 25), (75, 10, 24), (105, 10, 400), (15, 30, 1000)):
        xs.append(rng.uniform(cx - 5, cx + 5, n)), ys.append(rng.uniform(cy - 5, cy + 5, n))
    xs.append(rng.uniform(-50, -1, 200)), ys.append(rng.uniform(0, 960, 200))
    xs.append(rng.uniform(1921, 2100, 200)), ys.append(rng.uniform(961, 1200, 200))
    x, y = np.concatenate(xs), np.concatenate(ys)
    perm = rng.permutation(len(x))
    off = check(x[perm], y[perm])
    assert off[-1] == len(x) - 400
    # every keypoint in ONE cell at the one-launch limit, and a frame beyond it (four launches)
    check(rng.uniform(600, 610, 8192), rng.uniform(300, 310, 8192))
    check(rng.uniform(0, 1920, 9000), rng.uniform(0, 960, 9000))


def test_wide_fisheye_bounds_and_reference_cell_vectors(mods):
    cam_mod, data, feature, _ = mods
    ctx = feature.Context(0)
    # fov beyond 180 degrees: the fisheye.cc:83-116 branch (corner angle > pi/2)
    k = (-0.01, 0.002, -0.0005, 0.00005)
    w = cam_mod.fisheye("wide", "Monocular", "Gray", 1280, 1280, 30.0, 300.0, 300.0, 640.0, 640.0, *k, ctx=ctx)
    ow = O.make_camera(O.CAM_FISHEYE, 1280, 1280, 300.0, 300.0, 640.0, 640.0, k)
    assert math.hypot(640 / 300.0, 640 / 300.0) > math.pi / 2
    got, want = np.array(w.img_bounds_.as_tuple()), np.array([ow.min_x, ow.max_x, ow.min_y, ow.max_y])
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max() and got[0] < 0 < got[1]
    # the reference's get_cell_indices vectors (test/stella_vslam/data/common_get_cell_indices.cc:16-61) on the device grid
    cam = cam_mod.perspective("perspective", "Monocular", "RGB", 2000, 1000, 30.0, 1000.0, 1000.0, 1000.0, 500.0, float(np.float32(-0.1)),
                              float(np.float32(0.1)), 0.0, 0.0, 0.0, ctx=ctx)
    mnx, mxx, mny, mxy = cam.img_bounds_.as_tuple()
    f, eps = np.float32, np.float32(0.01)
    cases = [((mnx, mny), (0, 0)), ((f(mxx) - eps, mny), (63, 0)), ((mnx, f(mxy) - eps), (0, 47)), ((f(mxx) - eps, f(mxy) - eps), (63, 47)),
             ((1000.0, mny), (31, 0)), ((1000.0, f(mxy) - eps), (31, 47)), ((mnx, 500.0), (0, 23)), ((f(mxx) - eps, 500.0), (63, 23)),
             ((f(mnx) - eps, f(mny) - eps), None), ((mxx, mxy), None), ((1000.0, mxy), None), ((mxx, 500.0), None)]
    kps = np.zeros(len(cases), O.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = [c[0][0] for c in cases], [c[0][1] for c in cases]
    plain = cam_mod.radial_division("undistorted view", "Monocular", "RGB", 2000, 1000, 30.0, 1000.0, 1000.0, 1000.0, 500.0, 0.0, ctx=ctx)
    plain.c_.min_x, plain.c_.max_x, plain.c_.min_y, plain.c_.max_y = mnx, mxx, mny, mxy  # the points above are already undistorted
    obs = data.frame_observation(plain, kps, np.zeros((len(kps), 32), np.uint8))
    assert np.array_equal(obs.undist_keypts_["x"], kps["x"]) and np.array_equal(obs.undist_keypts_["y"], kps["y"])
    for i, (_, want_cell) in enumerate(cases):
        where = [(c // 48, c % 48) for c in range(64 * 48) if i in obs.cell_items_[obs.cell_off_[c]:obs.cell_off_[c + 1]]]
        assert where == ([want_cell] if want_cell else []), (i, where)


def _scene(rng, n, R, t):
    twc = -R.T @ t
    pc = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-2, 12, n)], 1)
    pw = (pc - t) @ R
    d = np.linalg.norm(pw - twc, axis=1)
    nv = pw - (twc + rng.normal(0, 3.0, (n, 3)))
    nv /= np.linalg.norm(nv, axis=1, keepdims=True)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    mx = (d * rng.uniform(0.4, 2.5, n) * sf[rng.integers(0, 8, n)]).astype(np.float32)
    return pw, nv, (mx / sf[7]).astype(np.float32), mx


def _pose():
    a, b = 0.3, 0.1
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]) @ np.array(
        [[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    return R, np.array([0.3, -0.2, 0.5])


def test_reproject_landmarks_all_models(mods):
    cam_mod, _, feature, _ = mods
    ctx = feature.Context(0)
    rng = np.random.default_rng(7)
    R, t = _pose()
    lsf = float(np.log(np.float32(1.2)))
    for name, cam, ocam in _cameras(cam_mod, ctx):
        pw, nv, mn, mx = _scene(rng, 20000, R, t)
        skip = (rng.uniform(size=len(pw)) < 0.1).astype(np.uint8)
        vis, rp, xr, lv = cam.reproject_landmarks(R, t, pw, nv, mn, mx, 0.5, 8, lsf, skip=skip)
        ovis, orp, oxr, olv = O.can_observe(ocam, R, t, pw, nv, mn, mx, 0.5, 8, lsf)
        ovis = ovis & (1 - skip)
        orp[ovis == 0], oxr[ovis == 0], olv[ovis == 0] = 0, 0, -1
        assert 0.03 < ovis.mean() < 0.7, name
        assert np.array_equal(vis, ovis), name
        # predict_scale_level goes through logf: identical unless log(ratio) / log(1.2) sits within an ulp of an integer
        q = np.log(mx.astype(np.float64) / np.linalg.norm(pw - (-R.T @ t), axis=1)) / lsf
        near_int = np.abs(q - np.round(q)) < 1e-5
        assert np.array_equal(lv[~near_int], olv[~near_int]) and near_int.sum() < 5, name
        if name == "equirectangular":  # asin / atan2: <= a few ulp of fp64 on a value of magnitude <= 1920
            assert np.abs(rp - orp).max() < 1e-11
        else:
            assert np.array_equal(rp, orp) and np.array_equal(xr, oxr), name


def _tracked_frame(mods, ctx, seed, stereo):
    """A 752x480 frame extracted on the device, observed through the EuRoC camera, and landmarks scattered along the bearings
    of its keypoints (noisy descriptors) plus clutter -- the inputs of tracking_module::search_local_landmarks."""
    cam_mod, data, feature, _ = mods
    rng = np.random.default_rng(seed)
    img = S.frame(752, 480, 100 + seed)
    ext = feature.orb_extractor(feature.orb_params(), ctx=ctx)
    kps, desc = ext.extract(img)
    cam = cam_mod.perspective("euroc", "Stereo", "Gray", 752, 480, 20.0, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], *EUROC["k"],
                              focal_x_baseline=47.9, ctx=ctx)
    ocam = O.make_camera(O.CAM_PERSPECTIVE, 752, 480, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], EUROC["k"], 47.9)
    obs = data.frame_observation(cam, kps, desc)
    R, t = _pose()
    twc = -R.T @ t
    n_kp = len(kps)
    pick = rng.permutation(n_kp)[: int(0.7 * n_kp)]
    depth = rng.uniform(2.0, 9.0, len(pick))
    pc = obs.bearings_[pick] / obs.bearings_[pick, 2:3] * depth[:, None]
    pc[:, :2] += rng.normal(0, 0.004, (len(pick), 2)) * depth[:, None]        # ~2 px of reprojection noise
    pw_a = (pc - t) @ R
    dsc_a = desc[pick].copy()
    flips = rng.integers(0, 256, (len(pick), 12))
    for j in range(12):
        dsc_a[np.arange(len(pick)), flips[:, j] // 8] ^= (1 << (flips[:, j] % 8)).astype(np.uint8)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    d_a = np.linalg.norm(pw_a - twc, axis=1)
    mx_a = (d_a * rng.uniform(0.85, 1.15, len(pick)) * sf[kps["octave"][pick]]).astype(np.float32)
    nv_a = pw_a - (twc + rng.normal(0, 0.5, (len(pick), 3)))
    pw_b, nv_b, _, mx_b = _scene(rng, 1200, R, t)
    dsc_b = rng.integers(0, 256, (1200, 32), dtype=np.uint8)
    pw, nv = np.concatenate([pw_a, pw_b]), np.concatenate([nv_a, nv_b])
    nv /= np.linalg.norm(nv, axis=1, keepdims=True)
    mx = np.concatenate([mx_a, mx_b])
    mn = (mx / sf[7]).astype(np.float32)
    lm_desc = np.concatenate([dsc_a, dsc_b])
    order = rng.permutation(len(pw))
    pw, nv, mx, mn, lm_desc = pw[order], nv[order], mx[order], mn[order], lm_desc[order]
    if stereo:
        xr = np.full(n_kp, -1.0, np.float32)
        xr[pick] = (obs.undist_keypts_["x"][pick] - 47.9 / depth + rng.normal(0, 2.0, len(pick))).astype(np.float32)
        xr[rng.uniform(size=n_kp) < 0.3] = -1.0
        obs.stereo_x_right_ = xr
    return cam, ocam, obs, R, t, pw, nv, mn, mx, lm_desc, sf


def _oracle_match_frame_and_landmarks(ocam, obs, R, t, pw, nv, mn, mx, lm_desc, sf, margin, skip, occupied, lowe_ratio):
    lsf = float(np.log(np.float32(1.2)))
    vis, rp, xr, lv = O.can_observe(ocam, R, t, pw, nv, mn, mx, 0.5, 8, lsf)
    valid = vis & (1 - skip)
    u = obs.undist_keypts_
    bounds = (ocam.min_x, ocam.max_x, ocam.min_y, ocam.max_y)
    off_g, items = O.assign_keypoints_to_grid(u["x"], u["y"], bounds)
    lvq = np.where(valid == 1, lv, 0)
    q_margin = (np.float32(margin) * sf[lvq]).astype(np.float32)
    cand_off, cand_idx = [0], []
    for q in range(len(pw)):
        if valid[q]:
            c = O.get_keypoints_in_cell(u["x"], u["y"], u["octave"], off_g, items, bounds, float(np.float32(rp[q, 0])), float(np.float32(rp[q, 1])),
                                        float(q_margin[q]), max(0, int(lv[q]) - 1), min(7, int(lv[q]) + 1))
            cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    kw = {}
    if obs.stereo_x_right_ is not None:
        kw = dict(q_xright=xr, t_xright=obs.stereo_x_right_, q_xr_tol=q_margin)
    exp = O.match_candidates(lm_desc, obs.descriptors_, cand_off, cand_idx, check_orientation=False, thr=100, lowe_ratio=lowe_ratio, mode=1,
                             t_octave=u["octave"], q_valid=valid.astype(np.uint8), occupied=occupied, **kw)
    return exp, valid, rp, xr, lv


@pytest.mark.parametrize("stereo,margin,seed", [(False, 5.0, 0), (True, 5.0, 1), (False, 20.0, 2)])
def test_match_frame_and_landmarks_fused(mods, stereo, margin, seed):
    _, _, feature, match = mods
    ctx = feature.Context(0)
    cam, ocam, obs, R, t, pw, nv, mn, mx, lm_desc, sf = _tracked_frame(mods, ctx, seed, stereo)
    rng = np.random.default_rng(50 + seed)
    skip = (rng.uniform(size=len(pw)) < 0.1).astype(np.uint8)
    occupied = (rng.uniform(size=len(obs.descriptors_)) < 0.05).astype(np.uint8)
    lsf = float(np.log(np.float32(1.2)))
    M = match.projection(0.8, True, ctx)
    got, num, vis, rp, xr, lv = M.match_frame_and_landmarks(cam, R, t, pw, nv, mn, mx, lm_desc, obs, sf, lsf, margin=margin, skip=skip, occupied=occupied)
    exp, valid, orp, oxr, olv = _oracle_match_frame_and_landmarks(ocam, obs, R, t, pw, nv, mn, mx, lm_desc, sf, margin, skip, occupied, 0.8)
    assert np.array_equal(vis, valid)
    v = valid == 1
    assert np.array_equal(lv[v], olv[v]) and np.array_equal(rp[v], orp[v]) and np.array_equal(xr[v], oxr[v])
    assert (exp >= 0).sum() > (250 if stereo else 400), (exp >= 0).sum()
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()
    # the two-step form (reproject, then the cell matcher on host-made query arrays) gives the same list
    vis2, rp2, xr2, lv2 = cam.reproject_landmarks(R, t, pw, nv, mn, mx, 0.5, 8, lsf, skip=skip)
    lvq = np.where(vis2 == 1, lv2, 0)
    u = obs.undist_keypts_
    kw = dict(q_xright=xr2, t_xright=obs.stereo_x_right_, q_xr_tol=(np.float32(margin) * sf[lvq]).astype(np.float32)) if stereo else {}
    two, _ = match.projection(0.8, False, ctx).match_in_cells(lm_desc, rp2.astype(np.float32), (np.float32(margin) * sf[lvq]).astype(np.float32), obs.descriptors_,
                                                              np.stack([u["x"], u["y"]], 1), u["octave"], cam.img_bounds_.as_tuple(), 1, 100,
                                                              q_min_level=np.maximum(0, lvq - 1), q_max_level=np.minimum(7, lvq + 1), q_valid=vis2,
                                                              occupied=occupied, **kw)
    assert np.array_equal(two, exp)


def test_frame_entry_points_reject_bad_arguments(mods):
    cam_mod, _, feature, _ = mods
    from stella_vslam_amd._lib import lib, SvgpuError
    ctx = feature.Context(0)
    cam = cam_mod.perspective("p", "Monocular", "Gray", 640, 480, 30.0, 500.0, 500.0, 320.0, 240.0, 0, 0, 0, 0, 0, ctx=ctx)
    bad = cam_mod.svgpu_camera.from_buffer_copy(cam.c_)
    bad.model = 7
    assert lib().svgpu_camera_image_bounds(ctx.handle, C.byref(bad)) != 0
    assert lib().svgpu_frame_observation(ctx.handle, C.byref(bad), None, 0, 64, 48, None, None, None, None) != 0
    assert lib().svgpu_frame_observation(ctx.handle, C.byref(cam.c_), None, 5, 64, 48, None, None, None, None) != 0   # n > 0 without keypoints
    nob = cam_mod.svgpu_camera.from_buffer_copy(cam.c_)
    nob.min_x = nob.max_x = 0.0
    off = np.zeros(64 * 48 + 1, np.int32)
    k = np.zeros(4, O.KEYPOINT_DTYPE)
    assert lib().svgpu_frame_observation(ctx.handle, C.byref(nob), k.ctypes.data_as(C.c_void_p), 4, 64, 48, None, None, off.ctypes.data_as(C.c_void_p), None) != 0
    with pytest.raises(SvgpuError):
        cam.reproject_landmarks(np.eye(3), np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3)), np.ones(3), np.ones(3), num_levels=99)
    # zero landmarks / zero keypoints are fine
    vis, rp, xr, lv = cam.reproject_landmarks(np.eye(3), np.zeros(3), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0))
    assert len(vis) == 0
