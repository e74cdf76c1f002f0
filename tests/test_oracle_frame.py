"""CPU: the oracle's frame-observation / reprojection restatement (oracle/frame_oracle.c) against
  * the reference's own test vectors that cross it: test/stella_vslam/data/common_get_cell_indices.cc (a distorted perspective
    camera's img_bounds_ -- i.e. cv::undistortPoints on the corners -- feeding get_cell_indices),
  * the forward distortion models (distort -> undistort round trips pin the two restated OpenCV solvers),
  * literal numpy restatements of the closed-form members (radial_division, bearings, frame::can_observe)."""
import math

import numpy as np
import pytest

from oracle import oracle as O


def _ref_test_camera(cols, rows, k1=0.0, k2=0.0):
    # create_perspective_camera of common_get_cell_indices.cc:8-14 (k1, k2 arrive as floats)
    return O.make_camera(O.CAM_PERSPECTIVE, cols, rows, float(rows), float(rows), cols / 2.0, rows / 2.0,
                         (float(np.float32(k1)), float(np.float32(k2)), 0.0, 0.0, 0.0))


def _cell_of(cam, x, y, cols=64, rows=48):
    b = (cam.min_x, cam.max_x, cam.min_y, cam.max_y)
    off, items = O.assign_keypoints_to_grid(np.array([x], np.float32), np.array([y], np.float32), b, cols, rows)
    if off[-1] == 0:
        return None
    c = int(np.nonzero(np.diff(off))[0][0])
    return (c // rows, c % rows)


def test_reference_get_cell_indices_valid_cases_1_and_invalid_cases():
    """common_get_cell_indices.cc:16-61 and :121-158 -- the expectations depend on img_bounds_ of the k1=-0.1, k2=0.1 camera."""
    cols, rows, C, R = 2000, 1000, 64, 48
    cam = _ref_test_camera(cols, rows, -0.1, 0.1)
    mnx, mxx, mny, mxy = cam.min_x, cam.max_x, cam.min_y, cam.max_y
    assert 0 < mnx < 40 and cols - 40 < mxx < cols and 0 < mny < 20 and rows - 20 < mxy < rows  # pincushion at the corners (1 + k1 r2 + k2 r4 > 1)
    f = np.float32
    eps = f(0.01)
    valid = [((mnx, mny), (0, 0)), ((f(mxx) - eps, mny), (C - 1, 0)), ((mnx, f(mxy) - eps), (0, R - 1)),
             ((f(mxx) - eps, f(mxy) - eps), (C - 1, R - 1)), ((cols / 2.0, mny), (C // 2 - 1, 0)),
             ((cols / 2.0, f(mxy) - eps), (C // 2 - 1, R - 1)), ((mnx, rows / 2.0), (0, R // 2 - 1)),
             ((f(mxx) - eps, rows / 2.0), (C - 1, R // 2 - 1))]
    for (x, y), want in valid:
        assert _cell_of(cam, x, y) == want, (x, y)
    invalid = [(f(mnx) - eps, f(mny) - eps), (mxx, f(mny) - eps), (f(mnx) - eps, mxy), (mxx, mxy), (cols / 2.0, f(mny) - eps),
               (cols / 2.0, mxy), (f(mnx) - eps, rows / 2.0), (mxx, rows / 2.0)]
    for x, y in invalid:
        assert _cell_of(cam, x, y) is None, (x, y)


def test_reference_get_cell_indices_valid_cases_2():
    """common_get_cell_indices.cc:63-119: distortion-free camera, the four inner corners of every cell."""
    cols, rows, C, R = 2000, 1000, 64, 48
    cam = _ref_test_camera(cols, rows)
    assert (cam.min_x, cam.max_x, cam.min_y, cam.max_y) == (0.0, 2000.0, 0.0, 1000.0)
    iw, ih = C / (np.float32(cam.max_x) - np.float32(cam.min_x)), R / (np.float32(cam.max_y) - np.float32(cam.min_y))
    eps = np.float32(0.01)
    for ix in range(C):
        for iy in range(0, R, 5):
            for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                x = np.float32((ix + dx) * (1.0 / float(iw)) + (float(eps) if dx == 0 else -float(eps)))
                y = np.float32((iy + dy) * (1.0 / float(ih)) + (float(eps) if dy == 0 else -float(eps)))
                assert _cell_of(cam, x, y) == (ix, iy)


EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, dist=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0))


def test_perspective_undistort_round_trip_and_identity():
    rng = np.random.default_rng(1)
    cam = O.make_camera(O.CAM_PERSPECTIVE, 752, 480, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], EUROC["dist"])
    # true normalised points -> distorted pixels (forward Brown model) -> undistort -> K * point
    xy = np.stack([rng.uniform(-0.75, 0.75, 4000), rng.uniform(-0.5, 0.5, 4000)], 1)
    pix = O.distort_points_for_tests(cam, xy)
    und = O.undistort_keypoints(cam, pix)
    want = xy * [np.float32(EUROC["fx"]), np.float32(EUROC["fy"])] + [np.float32(EUROC["cx"]), np.float32(EUROC["cy"])]
    # termination: reprojection error < 1e-6 px on the (float-rounded) input pixel; the float rounding of the distorted pixel
    # (<= 3e-5 px) is amplified by at most ~1.5 when undone
    assert np.abs(und - want).max() < 2e-4
    # bounds of a barrel-distorted image exceed the sensor
    assert cam.min_x < 0 and cam.max_x > 752 and cam.min_y < 0 and cam.max_y > 480
    # zero distortion: one iteration, the point itself (to float rounding of (u - cx) / fx * fx + cx in double)
    cam0 = O.make_camera(O.CAM_PERSPECTIVE, 752, 480, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], (0, 0, 0, 0, 0))
    pts = np.stack([rng.uniform(0, 752, 5000), rng.uniform(0, 480, 5000)], 1).astype(np.float32)
    assert np.array_equal(O.undistort_keypoints(cam0, pts), pts)
    assert (cam0.min_x, cam0.max_x, cam0.min_y, cam0.max_y) == (0.0, 752.0, 0.0, 480.0)


def test_fisheye_undistort_round_trip():
    rng = np.random.default_rng(2)
    k = (0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182)
    cam = O.make_camera(O.CAM_FISHEYE, 512, 512, 190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504, k)
    ang = rng.uniform(0, 2 * np.pi, 3000)
    r = np.tan(rng.uniform(0.0, 1.2, 3000))  # up to ~69 degrees off axis
    xy = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
    pix = O.distort_points_for_tests(cam, xy)
    und = O.undistort_keypoints(cam, pix)
    fx, fy, cx, cy = np.float32(cam.fx), np.float32(cam.fy), np.float32(cam.cx), np.float32(cam.cy)
    want = xy * [fx, fy] + [cx, cy]
    # float rounding of the distorted pixel (4e-5 px) / f, amplified by d tan(theta)/d theta_d * f  (<= ~8 at 69 degrees)
    assert (np.abs(und - want) / (1 + r[:, None] ** 2)).max() < 5e-4
    # principal point: theta_d < eps -> scale 0 -> exactly (cx, cy) as floats
    c = O.undistort_keypoints(cam, np.array([[cx, cy]], np.float32))
    assert np.array_equal(c, np.array([[cx, cy]], np.float32))
    assert cam.min_x < 0 and cam.max_x > 512


def test_radial_division_and_bearings_against_literal_numpy():
    rng = np.random.default_rng(3)
    fx, fy, cx, cy, d = 320.0, 318.0, 319.5, 239.5, -0.25
    cam = O.make_camera(O.CAM_RADIAL_DIVISION, 640, 480, fx, fy, cx, cy, (d,))
    pts = np.stack([rng.uniform(0, 640, 3000), rng.uniform(0, 480, 3000)], 1).astype(np.float32)
    und = O.undistort_keypoints(cam, pts)
    x, y = (pts[:, 0].astype(np.float64) - cx) / fx, (pts[:, 1].astype(np.float64) - cy) / fy
    u = 1.0 + d * (x * x + y * y)
    want = np.stack([(x / u) * fx + cx, (y / u) * fy + cy], 1).astype(np.float32)
    assert np.array_equal(und, want)
    b = O.keypoints_to_bearings(cam, und)
    xn, yn = (und[:, 0].astype(np.float64) - cx) / fx, (und[:, 1].astype(np.float64) - cy) / fy
    l2 = np.sqrt(xn * xn + yn * yn + 1.0)
    assert np.array_equal(b, np.stack([xn / l2, yn / l2, 1.0 / l2], 1))
    assert np.abs(np.linalg.norm(b, axis=1) - 1).max() < 1e-15


def test_equirectangular_bearings_round_trip():
    rng = np.random.default_rng(4)
    cam = O.make_camera(O.CAM_EQUIRECTANGULAR, 1920, 960)
    assert (cam.min_x, cam.max_x, cam.min_y, cam.max_y) == (0.0, 1920.0, 0.0, 960.0)
    pts = np.stack([rng.uniform(1, 1919, 2000), rng.uniform(1, 959, 2000)], 1).astype(np.float32)
    assert np.array_equal(O.undistort_keypoints(cam, pts), pts)
    b = O.keypoints_to_bearings(cam, pts)
    assert np.abs(np.linalg.norm(b, axis=1) - 1).max() < 1e-15
    # convert_bearing_to_point (equirectangular.cc:50-56) inverts it
    lat, lon = -np.arcsin(b[:, 1]), np.arctan2(b[:, 0], b[:, 2])
    back = np.stack([1920 * (0.5 + lon / (2 * np.pi)), 960 * (0.5 - lat / np.pi)], 1)
    assert np.abs(back - pts).max() < 1e-3
    # a landmark along the bearing reprojects onto the keypoint
    vis, rp, xr, lv = O.can_observe(cam, np.eye(3), np.zeros(3), b * 3.0, b, np.full(len(b), 1.0, np.float32), np.full(len(b), 10.0, np.float32))
    assert vis.all() and np.abs(rp - pts).max() < 1e-3 and (xr == 0).all()


def _can_observe_literal(cam, R, t, pw, nv, mn, mx, thr, num_levels, lsf):
    """data/frame.cc:59-85 with the reference's types spelled out in numpy scalars (perspective model)."""
    f32, f64 = np.float32, np.float64
    twc = -R.T @ t
    out = []
    for i in range(len(pw)):
        pc = R @ pw[i] + t
        if pc[2] <= 0.0:
            out.append((0, -1))
            continue
        zi = 1.0 / pc[2]
        u, v = cam.fx * pc[0] * zi + cam.cx, cam.fy * pc[1] * zi + cam.cy
        if not (cam.min_x < u < cam.max_x and cam.min_y < v < cam.max_y):
            out.append((0, -1))
            continue
        vec = pw[i] - twc
        dist = f64(np.sqrt(vec @ vec))
        fd = f32(dist)
        if not (f32(f32(1.0 / 1.3) * mn[i]) <= fd <= f32(f32(1.3) * mx[i])):
            out.append((0, -1))
            continue
        if (vec @ nv[i]) / dist < thr:
            out.append((0, -1))
            continue
        ratio = f32(mx[i] / fd)
        lvl = int(math.ceil(f32(f32(math.log(float(ratio))) / f32(lsf))))
        lvl = 0 if lvl < 0 else (num_levels - 1 if num_levels <= lvl else lvl)
        out.append((1, lvl))
    return np.array(out)


def _landmark_scene(rng, n, cam_fx=458.654):
    # camera at a generic pose; landmarks in front, behind, off-image, too near / far, seen from the side
    ang = 0.3
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]]) @ np.array(
        [[1, 0, 0], [0, math.cos(0.1), -math.sin(0.1)], [0, math.sin(0.1), math.cos(0.1)]])
    t = np.array([0.3, -0.2, 0.5])
    twc = -R.T @ t
    pc = np.stack([rng.uniform(-6, 6, n), rng.uniform(-4, 4, n), rng.uniform(-2, 12, n)], 1)
    pw = (pc - t) @ R  # R^T (pc - t)
    d = np.linalg.norm(pw - twc, axis=1)
    # mean viewing direction: from a previous camera somewhere around
    nv = pw - (twc + rng.normal(0, 3.0, (n, 3)))
    nv /= np.linalg.norm(nv, axis=1, keepdims=True)
    lvl = rng.integers(0, 8, n)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    ref_d = d * rng.uniform(0.4, 2.5, n)  # distance at which the landmark was first seen
    mx = (ref_d * sf[lvl]).astype(np.float32)
    mn = (mx / sf[7]).astype(np.float32)
    return R, t, pw, nv, mn, mx


def test_can_observe_against_literal_restatement():
    rng = np.random.default_rng(5)
    cam = O.make_camera(O.CAM_PERSPECTIVE, 752, 480, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], EUROC["dist"], focal_x_baseline=50.0)
    R, t, pw, nv, mn, mx = _landmark_scene(rng, 6000)
    lsf = float(np.log(np.float32(1.2)))
    vis, rp, xr, lv = O.can_observe(cam, R, t, pw, nv, mn, mx, 0.5, 8, lsf)
    lit = _can_observe_literal(cam, R, t, pw, nv, mn, mx, 0.5, 8, lsf)
    assert np.array_equal(vis, lit[:, 0]) and np.array_equal(lv, lit[:, 1])
    assert 0.05 < vis.mean() < 0.6 and len(np.unique(lv[vis == 1])) == 8  # every gate and every level exercised
    pc = pw @ R.T + t
    v = vis == 1
    assert np.allclose(rp[v, 0], cam.fx * pc[v, 0] / pc[v, 2] + cam.cx, rtol=1e-12)
    assert np.allclose(xr[v], rp[v, 0] - 50.0 / pc[v, 2], rtol=1e-6)
    assert (rp[~v] == 0).all() and (lv[~v] == -1).all()
