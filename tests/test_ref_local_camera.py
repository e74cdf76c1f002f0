"""Pins the oracle's camera models (oracle/frame_oracle.c, match2_oracle.c) against the REFERENCE's own camera code: camera/base.cc,
perspective.cc, fisheye.cc, equirectangular.cc and radial_division.cc are compiled where they lie over the stand-in headers of
oracle/ref_local/shim into oracle/_ref/libsvref_cam.so.  The two OpenCV undistortion solvers behind the stand-in cv::undistortPoints /
cv::fisheye::undistortPoints are the oracle's restatements, so what is pinned is everything the reference itself wrote: the float camera
matrix it hands to OpenCV, the keypoint marshalling, compute_image_bounds (incl. the fisheye 5-degree rule), the radial-division and
equirectangular closed forms, convert_keypoints_to_bearings, reproject_to_image and reproject_to_bearing of the four models."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref_cam.so")

CAMS = {
    "perspective": dict(model=O.CAM_PERSPECTIVE, cols=752, rows=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375,
                        dist=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0), fxb=50.0),
    "perspective_nodist": dict(model=O.CAM_PERSPECTIVE, cols=1241, rows=376, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, dist=(0, 0, 0, 0, 0), fxb=386.1448),
    "fisheye": dict(model=O.CAM_FISHEYE, cols=640, rows=480, fx=285.7, fy=286.1, cx=320.5, cy=240.2, dist=(-0.0075, 0.044, -0.041, 0.0076), fxb=0.0),
    "equirectangular": dict(model=O.CAM_EQUIRECTANGULAR, cols=1920, rows=960, fx=0, fy=0, cx=0, cy=0, dist=(), fxb=0.0),
    "radial_division": dict(model=O.CAM_RADIAL_DIVISION, cols=640, rows=480, fx=520.9, fy=521.0, cx=325.1, cy=249.7, dist=(-2.5e-7,), fxb=0.0),
}


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref_cam.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _args(c):
    d = np.zeros(5)
    d[:len(c["dist"])] = c["dist"]
    return [c["model"], int(c["fxb"] != 0), c["cols"], c["rows"], C.c_double(c["fx"]), C.c_double(c["fy"]), C.c_double(c["cx"]), C.c_double(c["cy"]), _p(d),
            C.c_double(c["fxb"])], d


@pytest.mark.parametrize("name", list(CAMS))
def test_bounds_undistortion_and_bearings(ref, name):
    c = CAMS[name]
    cam = O.make_camera(c["model"], c["cols"], c["rows"], c["fx"], c["fy"], c["cx"], c["cy"], c["dist"], c["fxb"])
    rng = np.random.default_rng(1)
    n = 3000
    xy = np.stack([rng.uniform(0, c["cols"], n), rng.uniform(0, c["rows"], n)], 1).astype(np.float32)
    xy[:4] = [[0, 0], [c["cols"], 0], [0, c["rows"]], [c["cols"], c["rows"]]]
    args, keep = _args(c)
    bounds, und, bear, tb = np.zeros(4, np.float32), np.zeros((n, 2), np.float32), np.zeros((n, 3)), C.c_double(0)
    ref.svref_camera_observation(*args, n, _p(xy), _p(bounds), _p(und), _p(bear), C.byref(tb))
    assert np.array_equal(bounds, np.array([cam.min_x, cam.max_x, cam.min_y, cam.max_y], np.float32))
    exp_u = O.undistort_keypoints(cam, xy)
    assert np.array_equal(und.view(np.uint32), exp_u.view(np.uint32))
    exp_b = O.keypoints_to_bearings(cam, exp_u)
    if name == "equirectangular":  # libm sin / cos on both sides, but gcc may evaluate sincos() differently in the two translation units
        assert np.abs(bear - exp_b).max() < 1e-15
    else:
        assert np.array_equal(bear.view(np.uint64), exp_b.view(np.uint64))
    if c["fx"]:
        assert tb.value == c["fxb"] / c["fx"]


@pytest.mark.parametrize("name", list(CAMS))
def test_reprojection(ref, name):
    c = CAMS[name]
    cam = O.make_camera(c["model"], c["cols"], c["rows"], c["fx"], c["fy"], c["cx"], c["cy"], c["dist"], c["fxb"])
    rng = np.random.default_rng(2)
    n = 4000
    w = rng.normal(0, 0.2, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    t = rng.normal(0, 0.3, 3)
    pw = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(-2, 12, n)], 1)   # also behind the camera and outside the image
    args, keep = _args(c)
    ok, rp, xr, bok, bear = np.zeros(n, np.uint8), np.zeros((n, 2)), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros((n, 3))
    ref.svref_camera_reproject(*args, _p(np.ascontiguousarray(R)), _p(t), n, _p(pw), _p(ok), _p(rp), _p(xr), _p(bok), _p(bear))
    L = O.lib()
    e_ok, e_rp, e_xr, e_bok, e_bear = np.zeros(n, np.uint8), np.zeros((n, 2)), np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros((n, 3))
    Rc = np.ascontiguousarray(R)
    for i in range(n):
        r2, x1, b3 = np.zeros(2), C.c_float(0), np.zeros(3)
        e_ok[i] = L.orc_reproject_to_image(C.byref(cam), _p(Rc), _p(t), _p(pw[i]), _p(r2), C.byref(x1))
        e_rp[i], e_xr[i] = r2, x1.value
        e_bok[i] = L.orc_reproject_to_bearing(C.byref(cam), _p(Rc), _p(t), _p(pw[i]), _p(b3))
        e_bear[i] = b3
    assert 0.1 * n < ok.sum() < 0.95 * n or name == "equirectangular"
    assert np.array_equal(ok, e_ok) and np.array_equal(bok, e_bok)
    tol = 1e-9 if name in ("equirectangular", "fisheye") else 0.0   # these two call libm (atan2 / asin / sqrt chains): same formulas, last-bit freedom
    v = ok.astype(bool)
    assert np.abs(rp[v] - e_rp[v]).max() <= tol * max(c["cols"], 1)
    if name in ("perspective", "perspective_nodist", "radial_division"):
        assert np.array_equal(rp[v].view(np.uint64), e_rp[v].view(np.uint64)) and np.array_equal(xr[v].view(np.uint32), e_xr[v].view(np.uint32))
        bv = bok.astype(bool)
        assert np.array_equal(bear[bv].view(np.uint64), e_bear[bv].view(np.uint64))
    else:
        bv = bok.astype(bool)
        assert np.abs(bear[bv] - e_bear[bv]).max() <= 1e-14


@pytest.mark.parametrize("name", ["perspective", "fisheye", "equirectangular"])
def test_grid_assignment_and_lookup(ref, name):
    """data::assign_keypoints_to_grid + data::get_keypoints_in_cell (data/common.cc, the reference's own code): the candidate lists every
    projection matcher starts from -- same indices in the same order as the oracle's restatement, for windows inside, across and outside
    the image bounds, with and without level limits."""
    c = CAMS[name]
    cam = O.make_camera(c["model"], c["cols"], c["rows"], c["fx"], c["fy"], c["cx"], c["cy"], c["dist"], c["fxb"])
    rng = np.random.default_rng(3)
    n, nq = 2500, 1500
    bounds = (cam.min_x, cam.max_x, cam.min_y, cam.max_y)
    w, h = cam.max_x - cam.min_x, cam.max_y - cam.min_y
    xy = np.stack([rng.uniform(cam.min_x - 0.02 * w, cam.max_x + 0.02 * w, n), rng.uniform(cam.min_y - 0.02 * h, cam.max_y + 0.02 * h, n)], 1).astype(np.float32)
    xy[:8] = [[cam.min_x, cam.min_y], [cam.max_x, cam.max_y], [cam.min_x, cam.max_y], [cam.max_x, cam.min_y], [cam.min_x + w / 64, cam.min_y + h / 48],
              [cam.min_x + w / 2, cam.min_y + h / 2], [cam.max_x - 1e-3, cam.max_y - 1e-3], [cam.min_x - 1e-3, cam.min_y]]
    octave = rng.integers(0, 8, n).astype(np.int32)
    q = np.stack([rng.uniform(cam.min_x - 0.1 * w, cam.max_x + 0.1 * w, nq), rng.uniform(cam.min_y - 0.1 * h, cam.max_y + 0.1 * h, nq),
                  rng.choice([3.0, 7.5, 15.0, 40.0, 0.5 * w], nq)], 1).astype(np.float32)
    lv = np.stack([rng.integers(-1, 6, nq), rng.integers(-1, 8, nq)], 1).astype(np.int32)
    args, keep = _args(c)
    off, idx = np.zeros(nq + 1, np.int32), np.zeros(4_000_000, np.int32)
    tot = ref.svref_grid_lookup(args[0], args[2], args[3], *args[4:9], n, _p(xy), _p(octave), 64, 48, nq, _p(q), _p(lv), _p(off), _p(idx), len(idx))
    assert tot > 10000
    kx, ky = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    goff, items = O.assign_keypoints_to_grid(kx, ky, bounds)
    for i in range(nq):
        exp = O.get_keypoints_in_cell(kx, ky, octave, goff, items, bounds, float(q[i, 0]), float(q[i, 1]), float(q[i, 2]), int(lv[i, 0]), int(lv[i, 1]))
        assert np.array_equal(idx[off[i]:off[i + 1]], exp), i


_SO_FRM = os.path.join(os.path.dirname(_SO), "libsvref_frm.so")


@pytest.mark.parametrize("name", list(CAMS))
def test_frame_can_observe(name):
    """data/frame.cc compiled from the reference with its real data/frame.h and data/landmark.h (libsvref_frm.so): frame::set_pose_cw and
    frame::can_observe -- in-image test through the camera's reproject_to_image, the valid-distance gate with its 1.3 margins, the ray
    cosine against the landmark's mean normal, the predicted scale level -- against the oracle's can_observe fed with the landmark state
    the reference's own landmark refresh produced."""
    if not os.path.exists(_SO_FRM):
        pytest.skip("oracle/_ref/libsvref_frm.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    ref = C.CDLL(_SO_FRM)
    c = CAMS[name]
    cam = O.make_camera(c["model"], c["cols"], c["rows"], c["fx"], c["fy"], c["cx"], c["cy"], (0,) * len(c["dist"]), c["fxb"])
    rng = np.random.default_rng(23)
    n = 6000
    w = rng.normal(0, 0.3, 3)
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    R = np.ascontiguousarray(np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx)
    t = rng.normal(0, 0.5, 3)
    pw = np.ascontiguousarray(np.stack([rng.uniform(-8, 8, n), rng.uniform(-5, 5, n), rng.uniform(-3, 14, n)], 1))
    ref_twc = np.ascontiguousarray(pw + rng.normal(0, 1, (n, 3)) * rng.uniform(0.5, 6, (n, 1)))   # where each landmark was seen from
    octv = rng.integers(0, 8, n).astype(np.int32)
    pose = np.ascontiguousarray(np.concatenate([R, t[:, None]], 1).reshape(-1))
    intr = np.array([c["fx"], c["fy"], c["cx"], c["cy"], c["fxb"]])
    vis, rp, xr, lv = np.zeros(n, np.uint8), np.zeros((n, 2)), np.zeros(n, np.float32), np.zeros(n, np.int32)
    nrm, mn, mx, twc = np.zeros((n, 3)), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(3)
    for thr in (0.5, 0.7):   # the two thresholds the tracker uses (projection.cc / local-map search)
        ref.svref_frame_can_observe(c["model"], int(c["fxb"] != 0), c["cols"], c["rows"], _p(intr), _p(pose), n, _p(pw), _p(ref_twc), _p(octv), C.c_float(thr),
                                    C.c_float(1.2), 8, _p(vis), _p(rp), _p(xr), _p(lv), _p(nrm), _p(mn), _p(mx), _p(twc))
        np.testing.assert_allclose(twc, -R.T @ t, rtol=0, atol=1e-15)
        e_vis, e_rp, e_xr, e_lv = np.zeros(n, np.uint8), np.zeros((n, 2)), np.zeros(n, np.float32), np.zeros(n, np.int32)
        O.lib().orc_can_observe(C.byref(cam), _p(R), _p(t), _p(twc), n, _p(pw), _p(nrm), _p(mn), _p(mx), C.c_float(thr), C.c_uint(8),
                                C.c_float(float(np.log(np.float32(1.2)))), _p(e_vis), _p(e_rp), _p(e_xr), _p(e_lv))
        assert 50 < vis.sum() < 0.9 * n
        assert np.array_equal(vis, e_vis)
        v = vis.astype(bool)
        assert np.array_equal(lv[v], e_lv[v])
        tol = 1e-9 if name in ("equirectangular", "fisheye") else 0.0
        assert np.abs(rp[v] - e_rp[v]).max() <= tol * c["cols"]
        if tol == 0.0:
            assert np.array_equal(xr[v].view(np.uint32), e_xr[v].view(np.uint32))
