"""Pins the oracle's landmark refresh (oracle/landmark_oracle.c) against the REFERENCE's own data/landmark.cc, compiled where it lies
(the reference's real data/landmark.h over stand-in keyframe / map_database headers) into oracle/_ref/libsvref_lm.so:
landmark::compute_descriptor (k x k Hamming, lower median through std::sort, first minimum wins) and
landmark::update_mean_normal_and_obs_scale_variance (Eigen normalized() sums in observation order, the float / double mix of the
valid-distance range)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref_lm.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref_lm.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def test_landmark_refresh(ref):
    rng = np.random.default_rng(5)
    n = 1500
    k = np.concatenate([rng.integers(1, 12, n - 20), rng.integers(40, 200, 20)])   # observations per landmark, incl. long rows and k = 1, 2
    off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
    m = int(off[-1])
    base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc = np.repeat(base, k, axis=0)
    flips = rng.integers(0, 256, (m, 12))
    for j in range(12):   # up to 12 flipped bits per observation; identical rows (median ties) stay possible
        on = rng.uniform(0, 1, m) < 0.6
        desc[np.flatnonzero(on), flips[on, j] // 8] ^= (1 << (flips[on, j] % 8)).astype(np.uint8)
    pos = rng.uniform(-5, 5, (n, 3))
    centres = np.repeat(pos, k, axis=0) + rng.normal(0, 1, (m, 3)) * np.array([4.0, 1.0, 4.0])
    octave = rng.integers(0, 8, m).astype(np.int32)
    ref_in_row = (rng.uniform(0, 1, n) * k).astype(np.int32)
    sf, isf, _, _ = O.scale_tables(1.2, 8)
    ref_obs = off[:-1] + ref_in_row
    best, exp_desc = O.landmarks_compute_descriptor(off, desc)
    exp_nrm, exp_mx, exp_mn = O.landmarks_update_geometry(off, centres, pos, centres[ref_obs], np.asarray(sf, np.float32)[octave[ref_obs]], float(isf[7]))
    d, nrm, mx, mn = np.zeros((n, 32), np.uint8), np.zeros((n, 3)), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ref.svref_landmarks_refresh(n, _p(off), _p(desc), _p(np.ascontiguousarray(centres)), _p(octave), _p(ref_in_row), _p(np.ascontiguousarray(pos)), C.c_float(1.2), 8,
                                _p(d), _p(nrm), _p(mx), _p(mn))
    assert np.array_equal(d, exp_desc) and np.array_equal(d, desc[off[:-1] + best])
    assert np.array_equal(nrm.view(np.uint64), exp_nrm.view(np.uint64))
    assert np.array_equal(mx.view(np.uint32), exp_mx.view(np.uint32)) and np.array_equal(mn.view(np.uint32), exp_mn.view(np.uint32))


def test_can_observe_scale_gates(ref):
    """frame::can_observe (data/frame.cc:59-85) = the camera's reproject_to_image (pinned in test_ref_local_camera.py) + two landmark members:
    is_inside_in_orb_scale (the float products margin x valid distance) and predict_scale_level (std::log of a float ratio, ceil, clamps).
    Both run here from the reference's own landmark code; the oracle's can_observe has to take the same decisions on landmarks straight
    ahead of an identity camera (so that only these two gates decide), including distances exactly on the boundaries."""
    rng = np.random.default_rng(8)
    n = 20000
    pos = np.zeros((n, 3))
    pos[:, 2] = rng.uniform(0.5, 60.0, n)
    ref_c = pos + rng.normal(0, 1, (n, 3)) * np.array([0.3, 0.3, 6.0])
    octave = rng.integers(0, 8, n).astype(np.int32)
    dist = pos[:, 2].copy()                      # camera at the origin: cam_to_lm_dist = z
    inside, level, mx, mn = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ref.svref_landmark_scale_queries(n, _p(pos), _p(ref_c), _p(octave), _p(dist), C.c_float(1.2), 8, _p(inside), _p(level), _p(mx), _p(mn))
    # put a third of the landmarks exactly on a boundary of their range (float products, as the member computes them)
    edge = np.arange(n) % 3 == 0
    far = (np.float32(1.3) * mx).astype(np.float32)
    near = (np.float32(1.0 / 1.3) * mn).astype(np.float32)
    pos[edge, 2] = np.where(np.arange(n)[edge] % 2 == 0, far[edge], near[edge]).astype(np.float64)
    dist = pos[:, 2].copy()
    ref.svref_landmark_scale_queries(n, _p(pos), _p(ref_c), _p(octave), _p(dist), C.c_float(1.2), 8, _p(inside), _p(level), _p(mx), _p(mn))
    cam = O.make_camera(O.CAM_PERSPECTIVE, 4000, 4000, 500.0, 500.0, 2000.0, 2000.0, (0, 0, 0, 0, 0), 0.0)
    normal = np.tile([0.0, 0.0, 1.0], (n, 1))
    vis, _, _, lv = O.can_observe(cam, np.eye(3), np.zeros(3), pos, normal, mn, mx, 0.5, 8, float(np.log(np.float32(1.2))))
    assert 0.2 * n < inside.sum() < 0.9 * n
    assert np.array_equal(vis, inside)
    assert np.array_equal(lv[vis == 1], level[vis == 1])
