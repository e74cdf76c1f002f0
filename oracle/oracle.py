"""ctypes bindings of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under stella_vslam_amd/.
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> pathlib.Path:
    so = _HERE / "liboracle.so"
    srcs = [_HERE / n for n in ("orb_oracle.c", "match_oracle.c", "ba_oracle.c", "frame_oracle.c", "landmark_oracle.c", "bow_oracle.c", "match2_oracle.c", "orb_pattern_i8.inc", "Makefile")]
    if force or not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.orc_util_cos.restype = C.c_float
        _LIB.orc_util_cos.argtypes = [C.c_float]
        _LIB.orc_util_sin.restype = C.c_float
        _LIB.orc_util_sin.argtypes = [C.c_float]
        _LIB.orc_ic_angle.restype = C.c_float
        _LIB.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        _LIB.orc_angle_diff.restype = C.c_float
        _LIB.orc_angle_diff.argtypes = [C.c_float, C.c_float]
        _LIB.orc_hamming_32.restype = C.c_uint
        _LIB.orc_hamming_64.restype = C.c_uint
    return _LIB


def _p(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


# ------------------------------------------------------------------------------------------- ORB

def scale_tables(scale_factor: float, num_levels: int):
    out = [np.zeros(num_levels, np.float32) for _ in range(4)]
    lib().orc_orb_scale_tables(C.c_float(scale_factor), num_levels, *[_p(o) for o in out])
    return out


def level_size(w0: int, h0: int, scale: float):
    w, h = C.c_int(), C.c_int()
    lib().orc_level_size(w0, h0, C.c_float(scale), C.byref(w), C.byref(h))
    return w.value, h.value


def level_sizes(w0: int, h0: int, scale_factor: float = 1.2, num_levels: int = 8):
    sf = scale_tables(scale_factor, num_levels)[0]
    return [(w0, h0)] + [level_size(w0, h0, float(sf[l])) for l in range(1, num_levels)]


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gauss_taps(n: int = 7, sigma: float = 2.0):
    t = np.zeros(n, np.int32)
    lib().orc_gauss_taps_q8(n, C.c_double(sigma), _p(t))
    return t


def gaussian_blur7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.empty_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast9_16(img: np.ndarray, threshold: int, nms: bool = True) -> np.ndarray:
    """(n,3) int32 rows (x, y, score) in emission order; img may be a strided view (ROI)."""
    assert img.dtype == np.uint8 and img.strides[1] == 1
    cap = img.shape[0] * img.shape[1]
    out = np.zeros((max(cap, 1), 3), np.int32)
    n = lib().orc_fast9_16(C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0], img.strides[0], threshold, int(nms),
                           _p(out), cap)
    return out[:n].copy()


def umax():
    t = np.zeros(16, np.int32)
    lib().orc_umax(_p(t))
    return t


def ic_angle(img: np.ndarray, x: float, y: float) -> float:
    assert img.dtype == np.uint8 and img.strides[1] == 1
    return float(lib().orc_ic_angle(C.c_void_p(img.ctypes.data), img.strides[0], C.c_float(x), C.c_float(y)))


def orb_descriptor(img: np.ndarray, x: float, y: float, angle_deg: float) -> np.ndarray:
    assert img.dtype == np.uint8 and img.strides[1] == 1
    d = np.zeros(32, np.uint8)
    lib().orc_compute_orb_descriptor(C.c_void_p(img.ctypes.data), img.strides[0], C.c_float(x), C.c_float(y),
                                     C.c_float(angle_deg), _p(d))
    return d


def orb_extract(img: np.ndarray, mask: np.ndarray | None = None, scale_factor: float = 1.2, num_levels: int = 8,
                ini_thr: int = 20, min_thr: int = 7, min_area: int = 800, cap: int = 20000, want_pyramid: bool = False):
    """Returns (keypoints[KEYPOINT_DTYPE], descriptors[n,32] u8, level_counts[L] (, pyramid list))."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.strides[1] == 1
    h, w = img.shape
    kps = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    counts = np.zeros(num_levels, np.int32)
    sizes = level_sizes(w, h, scale_factor, num_levels)
    pyr = np.zeros(sum(a * b for a, b in sizes[1:]) + 1, np.uint8) if want_pyramid else None
    if mask is not None:
        assert mask.dtype == np.uint8 and mask.shape == img.shape and mask.strides[1] == 1
    rc = lib().orc_orb_extract(C.c_void_p(img.ctypes.data), w, h, img.strides[0],
                               None if mask is None else C.c_void_p(mask.ctypes.data),
                               0 if mask is None else mask.strides[0],
                               C.c_float(scale_factor), num_levels, ini_thr, min_thr, C.c_uint(min_area),
                               _p(kps), _p(desc), cap, C.byref(n), _p(pyr), _p(counts))
    if rc != 0:
        raise RuntimeError(f"orc_orb_extract rc={rc} n={n.value}")
    res = (kps[:n.value].copy(), desc[:n.value].copy(), counts)
    if want_pyramid:
        levels, off = [img], 0
        for (lw, lh) in sizes[1:]:
            levels.append(pyr[off:off + lw * lh].reshape(lh, lw).copy())
            off += lw * lh
        res = res + (levels,)
    return res


# ------------------------------------------------------------------------------------------- match

def hamming(a: np.ndarray, b: np.ndarray, bits64: bool = False) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    f = lib().orc_hamming_64 if bits64 else lib().orc_hamming_32
    return int(f(_p(a), _p(b)))


def hamming_matrix(desc1: np.ndarray, desc2: np.ndarray) -> np.ndarray:
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    desc2 = np.ascontiguousarray(desc2, np.uint8)
    out = np.zeros((len(desc2), len(desc1)), np.uint16)
    lib().orc_hamming_matrix(_p(desc1), len(desc1), _p(desc2), len(desc2), _p(out))
    return out


def angle_diff(a: float, b: float) -> float:
    return float(lib().orc_angle_diff(C.c_float(a), C.c_float(b)))


def brute_force_match(desc1, angle1, desc2, angle2, valid2=None, lowe_ratio: float = 0.75, check_orientation: bool = True):
    """robust::brute_force_match; returns matched_2_in_1 (len n1, -1 = unmatched)."""
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    desc2 = np.ascontiguousarray(desc2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32)
    angle2 = np.ascontiguousarray(angle2, np.float32)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    out = np.full(len(desc1), -1, np.int32)
    lib().orc_brute_force_match(_p(desc1), _p(angle1), len(desc1), _p(desc2), _p(angle2), _p(v2), len(desc2),
                                C.c_float(lowe_ratio), int(check_orientation), _p(out))
    return out


MODE_BEST_ONLY, MODE_RATIO_SAME_OCTAVE, MODE_RATIO, MODE_TRIANGULATION, MODE_AREA = 0, 1, 2, 3, 4


def match_candidates(qdesc, tdesc, cand_off, cand_idx, cand_skip=None, t_octave=None, q_valid=None, occupied=None, q_angle=None,
                     t_angle=None, check_orientation=False, q_xright=None, t_xright=None, q_xr_tol=None, thr=100,
                     lowe_ratio=0.8, mode=MODE_BEST_ONLY):
    qdesc = np.ascontiguousarray(qdesc, np.uint8)
    tdesc = np.ascontiguousarray(tdesc, np.uint8)
    cand_off = np.ascontiguousarray(cand_off, np.int32)
    cand_idx = np.ascontiguousarray(cand_idx, np.int32)
    cv = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    t_octave, q_valid, occupied = cv(t_octave, np.int32), cv(q_valid, np.uint8), cv(occupied, np.uint8)
    cand_skip = cv(cand_skip, np.uint8)
    q_angle, t_angle = cv(q_angle, np.float32), cv(t_angle, np.float32)
    q_xright, t_xright, q_xr_tol = cv(q_xright, np.float32), cv(t_xright, np.float32), cv(q_xr_tol, np.float32)
    out = np.full(len(qdesc), -1, np.int32)
    lib().orc_match_candidates(_p(qdesc), len(qdesc), _p(tdesc), _p(t_octave), len(tdesc), _p(cand_off), _p(cand_idx),
                               _p(cand_skip), _p(q_valid), _p(occupied), _p(q_angle), _p(t_angle), int(check_orientation),
                               _p(q_xright), _p(t_xright), _p(q_xr_tol), C.c_uint(thr), C.c_float(lowe_ratio), mode,
                               _p(out))
    return out


def assign_keypoints_to_grid(kx, ky, bounds, cols=64, rows=48):
    kx = np.ascontiguousarray(kx, np.float32)
    ky = np.ascontiguousarray(ky, np.float32)
    off = np.zeros(cols * rows + 1, np.int32)
    items = np.zeros(max(len(kx), 1), np.int32)
    lib().orc_assign_keypoints_to_grid(_p(kx), _p(ky), len(kx), *[C.c_float(b) for b in bounds], cols, rows, _p(off),
                                       _p(items))
    return off, items[:off[-1]]


def get_keypoints_in_cell(kx, ky, octave, cell_off, cell_items, bounds, ref_x, ref_y, margin, min_level=-1,
                          max_level=-1, cols=64, rows=48):
    kx = np.ascontiguousarray(kx, np.float32)
    ky = np.ascontiguousarray(ky, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    out = np.zeros(max(len(kx), 1), np.int32)
    n = lib().orc_get_keypoints_in_cell(_p(kx), _p(ky), _p(octave), _p(cell_off), _p(cell_items),
                                        *[C.c_float(b) for b in bounds], cols, rows, C.c_float(ref_x), C.c_float(ref_y),
                                        C.c_float(margin), min_level, max_level, _p(out), len(out))
    return out[:n].copy()


def stereo_match(kps_left, desc_left, kps_right, desc_right, pyr_left, pyr_right, focal_x_baseline, true_baseline,
                 scale_factor=1.2):
    """match::stereo::compute: returns (stereo_x_right, depths)."""
    L = len(pyr_left)
    sf, isf, _, _ = scale_tables(scale_factor, L)
    kl = np.ascontiguousarray(kps_left)
    kr = np.ascontiguousarray(kps_right)
    dl = np.ascontiguousarray(desc_left, np.uint8)
    dr = np.ascontiguousarray(desc_right, np.uint8)
    pl = [np.ascontiguousarray(a) for a in pyr_left]
    pr = [np.ascontiguousarray(a) for a in pyr_right]
    PL = (C.c_void_p * L)(*[a.ctypes.data for a in pl])
    PR = (C.c_void_p * L)(*[a.ctypes.data for a in pr])
    lw = np.array([a.shape[1] for a in pl], np.int32)
    lh = np.array([a.shape[0] for a in pl], np.int32)
    lsl = np.array([a.strides[0] for a in pl], np.int32)
    lsr = np.array([a.strides[0] for a in pr], np.int32)
    xr = np.zeros(max(len(kl), 1), np.float32)
    dp = np.zeros(max(len(kl), 1), np.float32)
    lib().orc_stereo_match(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), PL, PR, _p(lw), _p(lh), _p(lsl), _p(lsr), _p(sf), _p(isf), L,
                           C.c_float(focal_x_baseline), C.c_float(true_baseline), _p(xr), _p(dp))
    return xr[:len(kl)].copy(), dp[:len(kl)].copy()


# ------------------------------------------------------------------------------------------- frame observation

class Camera(C.Structure):
    """camera::base + model parameters, laid out as include/svgpu.h's svgpu_camera."""
    _fields_ = [("model", C.c_int32), ("pad_", C.c_int32), ("cols", C.c_double), ("rows", C.c_double), ("fx", C.c_double),
                ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("dist", C.c_double * 5),
                ("focal_x_baseline", C.c_double), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float)]


CAM_PERSPECTIVE, CAM_FISHEYE, CAM_EQUIRECTANGULAR, CAM_RADIAL_DIVISION = 0, 1, 2, 3


def make_camera(model, cols, rows, fx=0.0, fy=0.0, cx=0.0, cy=0.0, dist=(), focal_x_baseline=0.0):
    """Camera with img_bounds_ filled by compute_image_bounds, as the reference's constructors do."""
    cam = Camera()
    cam.model, cam.cols, cam.rows = int(model), float(cols), float(rows)
    cam.fx, cam.fy, cam.cx, cam.cy = float(fx), float(fy), float(cx), float(cy)
    for i, d in enumerate(dist):
        cam.dist[i] = float(d)
    cam.focal_x_baseline = float(focal_x_baseline)
    b = np.zeros(4, np.float32)
    lib().orc_image_bounds(C.byref(cam), _p(b))
    cam.min_x, cam.max_x, cam.min_y, cam.max_y = [float(v) for v in b]
    return cam


def undistort_keypoints(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    lib().orc_undistort_keypoints(C.byref(cam), len(xy), _p(xy), _p(out))
    return out


def keypoints_to_bearings(cam, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros((len(xy), 3), np.float64)
    lib().orc_keypoints_to_bearings(C.byref(cam), len(xy), _p(xy), _p(out))
    return out


def distort_points_for_tests(cam, xy_norm):
    xy_norm = np.ascontiguousarray(xy_norm, np.float64).reshape(-1, 2)
    out = np.zeros((len(xy_norm), 2), np.float32)
    lib().orc_test_distort_points(C.byref(cam), len(xy_norm), _p(xy_norm), _p(out))
    return out


def can_observe(cam, rot_cw, trans_cw, pos_w, mean_normal, min_valid_dist, max_valid_dist, ray_cos_thr=0.5, num_levels=8,
                log_scale_factor=float(np.log(np.float32(1.2)))):
    """data::frame::can_observe over n landmarks: (visible u8, reproj n x 2 f64, x_right f32, pred_scale_level i32)."""
    R = np.ascontiguousarray(rot_cw, np.float64).reshape(3, 3)
    t = np.ascontiguousarray(trans_cw, np.float64).reshape(3)
    twc = np.ascontiguousarray(-R.T @ t)  # frame::update_pose_params: trans_wc_ = -rot_wc_ * trans_cw_
    pw = np.ascontiguousarray(pos_w, np.float64).reshape(-1, 3)
    nv = np.ascontiguousarray(mean_normal, np.float64).reshape(-1, 3)
    mn = np.ascontiguousarray(min_valid_dist, np.float32)
    mx = np.ascontiguousarray(max_valid_dist, np.float32)
    n = len(pw)
    vis = np.zeros(n, np.uint8)
    rp = np.zeros((n, 2), np.float64)
    xr = np.zeros(n, np.float32)
    lv = np.zeros(n, np.int32)
    lib().orc_can_observe(C.byref(cam), _p(R), _p(t), _p(twc), n, _p(pw), _p(nv), _p(mn), _p(mx), C.c_float(ray_cos_thr),
                          C.c_uint(num_levels), C.c_float(log_scale_factor), _p(vis), _p(rp), _p(xr), _p(lv))
    return vis, rp, xr, lv


# ------------------------------------------------------------------------------------------- BoW

def bow_transform(tree, desc, node_level):
    """Vocabulary-tree descent per descriptor: (word_id, weight, node id at `node_level`).  tree = dict(child_off, children,
    node_desc, node_weight, word_id)."""
    off = np.ascontiguousarray(tree["child_off"], np.int32)
    ch = np.ascontiguousarray(tree["children"], np.int32)
    nd = np.ascontiguousarray(tree["node_desc"], np.uint8)
    nw = np.ascontiguousarray(tree["node_weight"], np.float32)
    wi = np.ascontiguousarray(tree["word_id"], np.int32)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    ow, owt, on = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib().orc_bow_transform(len(off) - 1, _p(off), _p(ch), _p(nd), _p(nw), _p(wi), int(node_level), n, _p(d), _p(ow), _p(owt), _p(on))
    return ow, owt, on


def fbow_transform(tree, desc, store_level, k):
    """fbow::Vocabulary::transform(features, level, r, r2) per descriptor: (word_id, weight, r2 key = path code of the node at `store_level`
    counted from the root, or of the block in which a leaf was met above it)."""
    off = np.ascontiguousarray(tree["child_off"], np.int32)
    ch = np.ascontiguousarray(tree["children"], np.int32)
    nd = np.ascontiguousarray(tree["node_desc"], np.uint8)
    nw = np.ascontiguousarray(tree["node_weight"], np.float32)
    wi = np.ascontiguousarray(tree["word_id"], np.int32)
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(d)
    ow, owt, on = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
    lib().orc_fbow_transform(len(off) - 1, _p(off), _p(ch), _p(nd), _p(nw), _p(wi), int(store_level), int(k), n, _p(d), _p(ow), _p(owt), _p(on))
    return ow, owt, on


# ------------------------------------------------------------------------------------------- landmark refresh

def landmarks_compute_descriptor(obs_off, obs_desc):
    """landmark::compute_descriptor per CSR row: (best index inside the row, n x 32 representative descriptors)."""
    off = np.ascontiguousarray(obs_off, np.int32)
    d = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32)
    n = len(off) - 1
    best = np.zeros(n, np.int32)
    out = np.zeros((n, 32), np.uint8)
    lib().orc_landmarks_compute_descriptor(n, _p(off), _p(d), _p(best), _p(out))
    return best, out


def landmarks_update_geometry(obs_off, obs_trans_wc, pos_w, ref_trans_wc, ref_scale_factor, inv_scale_factor_last):
    """landmark::update_mean_normal_and_obs_scale_variance per CSR row: (mean_normal n x 3, max_valid_dist, min_valid_dist)."""
    off = np.ascontiguousarray(obs_off, np.int32)
    c = np.ascontiguousarray(obs_trans_wc, np.float64).reshape(-1, 3)
    p = np.ascontiguousarray(pos_w, np.float64).reshape(-1, 3)
    r = np.ascontiguousarray(ref_trans_wc, np.float64).reshape(-1, 3)
    sfr = np.ascontiguousarray(ref_scale_factor, np.float32)
    n = len(off) - 1
    mnrm, mx, mn = np.zeros((n, 3), np.float64), np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().orc_landmarks_update_geometry(n, _p(off), _p(c), _p(p), _p(r), _p(sfr), C.c_float(inv_scale_factor_last), _p(mnrm), _p(mx), _p(mn))
    return mnrm, mx, mn


# ------------------------------------------------------------------------------------------- BA

def local_ba(scene: dict, iters1: int = 5, iters2: int = 10, gain_thr: float = 1e-3, stop=None, want_trace=False):
    """scene: dict as produced by stella_vslam_amd.synthetic.ba_scene.  `stop`: None or np.uint8[1]."""
    P, L, E = len(scene["pose_cw"]), len(scene["points"]), len(scene["obs_pose"])
    pose = np.ascontiguousarray(scene["pose_cw"], np.float64)
    pts = np.ascontiguousarray(scene["points"], np.float64)
    pose_out, pts_out = np.zeros_like(pose), np.zeros_like(pts)
    outl = np.zeros(max(E, 1), np.uint8)
    stats = np.zeros(8)
    trace = np.zeros(2 * (iters1 + iters2) + 2) if want_trace else None
    a = lambda k, t: np.ascontiguousarray(scene[k], t)
    pf = scene.get("point_fixed")
    pf = None if pf is None else np.ascontiguousarray(pf, np.uint8)
    args = [a("pose_fixed", np.uint8), pf, a("obs_pose", np.int32), a("obs_point", np.int32), a("obs_uvr", np.float32),
            a("obs_inv_sigma_sq", np.float32), a("obs_huber", np.float32), a("intr", np.float64)]
    rc = lib().orc_local_ba(P, L, E, _p(pose), _p(args[0]), _p(pts), _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]),
                            _p(args[5]), _p(args[6]), _p(args[7]), iters1, iters2, C.c_double(gain_thr),
                            None if stop is None else C.c_void_p(stop.ctypes.data), _p(pose_out), _p(pts_out), _p(outl),
                            _p(stats), _p(trace))
    res = dict(rc=rc, pose_cw=pose_out, points=pts_out, outlier=outl[:E].copy(), stats=stats)
    if want_trace:
        res["trace"] = trace[:-2].reshape(-1, 2)
    return res


def pose_optimize(pose_cw, pos_w, uvr, inv_sigma_sq, huber, intr, num_trials_robust=2, num_trials=2, num_each_iter=10,
                  gain_thr=1e-3, reset_flag_each_round=False):
    """pose_optimizer_g2o::optimize on flat arrays; returns (num_valid, pose 3x4 flat, outlier flags, stats)."""
    pose = np.ascontiguousarray(pose_cw, np.float64).reshape(12)
    pw = np.ascontiguousarray(pos_w, np.float64)
    uv = np.ascontiguousarray(uvr, np.float32)
    w = np.ascontiguousarray(inv_sigma_sq, np.float32)
    hb = np.ascontiguousarray(huber, np.float32)
    K = np.ascontiguousarray(intr, np.float64).reshape(5)
    n = len(pw)
    out = np.zeros(12)
    outl = np.zeros(max(n, 1), np.uint8)
    st = np.zeros(4)
    lib().orc_pose_optimize.restype = C.c_int
    nv = lib().orc_pose_optimize(_p(pose), n, _p(pw), _p(uv), _p(w), _p(hb), _p(K), num_trials_robust, num_trials, num_each_iter,
                                 C.c_double(gain_thr), int(reset_flag_each_round), _p(out), _p(outl), _p(st))
    return nv, out, outl[:n].copy(), st


# ------------------------------------------------------------------------------------------- per-method matcher oracles (match2_oracle.c)

def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, np.float64)
    return a if shape is None else a.reshape(shape)


def _c(a, t):
    return None if a is None else np.ascontiguousarray(a, t)


def reproject_to_bearing(cam, rot_cw, trans_cw, pos_w):
    b = np.zeros(3)
    v = lib().orc_reproject_to_bearing(C.byref(cam), _p(_f64(rot_cw)), _p(_f64(trans_cw)), _p(_f64(pos_w)), _p(b))
    return b, bool(v)


def match_for_triangulation(lowe_ratio, check_orientation, desc1, angle1, octave1, bearings1, has_lm1, desc2, angle2, bearings2, has_lm2, E_12,
                            epipole_in_2, valid_epipole, scale_factors, residual_rad_thr, xright1=None, xright2=None, node1=None, node2=None):
    d1, d2, sf = _c(desc1, np.uint8), _c(desc2, np.uint8), _c(scale_factors, np.float32)
    out = np.full(len(d1), -1, np.int32)
    num = lib().orc_match_for_triangulation(
        _p(d1), _p(_c(angle1, np.float32)), _p(_c(octave1, np.int32)), _p(_f64(bearings1)), _p(_c(has_lm1, np.uint8)), _p(_c(xright1, np.float32)), len(d1),
        _p(d2), _p(_c(angle2, np.float32)), _p(_f64(bearings2)), _p(_c(has_lm2, np.uint8)), _p(_c(xright2, np.float32)), len(d2), _p(_c(node1, np.int32)),
        _p(_c(node2, np.int32)), _p(_f64(E_12)), _p(_f64(epipole_in_2)), int(valid_epipole), _p(sf), C.c_float(residual_rad_thr), C.c_float(lowe_ratio),
        int(check_orientation), _p(out))
    return out, num


def bow_match(lowe_ratio, check_orientation, desc1, angle1, valid1, node1, desc2, angle2, node2, valid2=None, occupied2=None):
    d1, d2 = _c(desc1, np.uint8), _c(desc2, np.uint8)
    out = np.full(len(d1), -1, np.int32)
    num = lib().orc_bow_match(_p(d1), _p(_c(angle1, np.float32)), _p(_c(valid1, np.uint8)), _p(_c(node1, np.int32)), len(d1), _p(d2),
                              _p(_c(angle2, np.float32)), _p(_c(valid2, np.uint8)), _p(_c(node2, np.int32)), len(d2), _p(_c(occupied2, np.uint8)),
                              C.c_float(lowe_ratio), int(check_orientation), _p(out))
    return out, num


def match_current_and_last_frames(check_orientation, cam, rot_cw, trans_cw, rot_lw, trans_lw, pos_w, valid, lm_desc, octave_last, angle_last, scale_factors,
                                  margin, tdesc, t_xy, t_octave, t_angle, occupied=None, t_xright=None, lm_has_observation=None, is_monocular=True,
                                  true_baseline=0.0, grid_cols=64, grid_rows=48):
    pw = _f64(pos_w, (-1, 3))
    n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
    out = np.full(n, -1, np.int32)
    num = lib().orc_match_current_and_last_frames(
        C.byref(cam), _p(_f64(rot_cw)), _p(_f64(trans_cw)), _p(_f64(rot_lw)), _p(_f64(trans_lw)), int(is_monocular), C.c_float(true_baseline), n, _p(pw),
        _p(_c(valid, np.uint8)), _p(_c(lm_desc, np.uint8)), _p(_c(octave_last, np.int32)), _p(_c(angle_last, np.float32)), _p(_c(lm_has_observation, np.uint8)),
        len(sf), _p(sf), C.c_float(margin), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)), _p(_c(t_angle, np.float32)), len(td),
        _p(_c(occupied, np.uint8)), _p(_c(t_xright, np.float32)), grid_cols, grid_rows, int(check_orientation), _p(out))
    return out, num


def match_frame_and_landmarks(cam, visible, reproj, x_right, pred_scale_level, lm_desc, scale_factors, margin, lowe_ratio, tdesc, t_xy, t_octave,
                              occupied=None, t_xright=None, lm_has_observation=None, grid_cols=64, grid_rows=48):
    """projection::match_frame_and_landmarks (match/projection.cc:13-93) on the outputs of can_observe: (match_lm, num_matches)."""
    vis = _c(visible, np.uint8)
    n, sf, td = len(vis), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
    out = np.full(n, -1, np.int32)
    num = lib().orc_match_frame_and_landmarks(
        C.byref(cam), n, _p(vis), _p(_f64(reproj)), _p(_c(x_right, np.float32)), _p(_c(pred_scale_level, np.int32)), _p(_c(lm_desc, np.uint8)),
        _p(_c(lm_has_observation, np.uint8)), len(sf), _p(sf), C.c_float(margin), C.c_float(lowe_ratio), _p(td), _p(_c(t_xy, np.float32)),
        _p(_c(t_octave, np.int32)), len(td), _p(_c(occupied, np.uint8)), _p(_c(t_xright, np.float32)), grid_cols, grid_rows, _p(out))
    return out, num


def match_frame_and_keyframe_projection(check_orientation, cam, rot_cw, trans_cw, pos_w, valid, min_valid_dist, max_valid_dist, lm_desc, angle_kf,
                                        scale_factors, log_scale_factor, margin, hamm_dist_thr, tdesc, t_xy, t_octave, t_angle, occupied=None,
                                        grid_cols=64, grid_rows=48):
    pw = _f64(pos_w, (-1, 3))
    n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
    out = np.full(n, -1, np.int32)
    num = lib().orc_match_frame_and_keyframe_projection(
        C.byref(cam), _p(_f64(rot_cw)), _p(_f64(trans_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)),
        _p(_c(max_valid_dist, np.float32)), _p(_c(lm_desc, np.uint8)), _p(_c(angle_kf, np.float32)), len(sf), _p(sf), C.c_float(log_scale_factor),
        C.c_float(margin), C.c_uint(hamm_dist_thr), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)), _p(_c(t_angle, np.float32)), len(td),
        _p(_c(occupied, np.uint8)), grid_cols, grid_rows, int(check_orientation), _p(out))
    return out, num


def match_by_sim3_transform(cam, sim3_cw, pos_w, valid, min_valid_dist, max_valid_dist, mean_normal, lm_desc, scale_factors, log_scale_factor, margin, tdesc,
                            t_xy, t_octave, occupied=None, grid_cols=64, grid_rows=48):
    pw = _f64(pos_w, (-1, 3))
    n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
    out = np.full(n, -1, np.int32)
    num = lib().orc_match_by_sim3_transform(
        C.byref(cam), _p(_f64(sim3_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)), _p(_c(max_valid_dist, np.float32)),
        _p(_f64(mean_normal)), _p(_c(lm_desc, np.uint8)), len(sf), _p(sf), C.c_float(log_scale_factor), C.c_float(margin), _p(td), _p(_c(t_xy, np.float32)),
        _p(_c(t_octave, np.int32)), len(td), _p(_c(occupied, np.uint8)), grid_cols, grid_rows, _p(out))
    return out, num


def match_keyframes_mutually(cam1, cam2, rot_1w, trans_1w, rot_2w, trans_2w, s_12, rot_12, trans_12, kf1, kf2, scale_factors, log_scale_factor, margin,
                             grid_cols=64, grid_rows=48):
    sf = _c(scale_factors, np.float32)

    def side(k):
        pw = _f64(k["pos_w"], (-1, 3))
        return [len(pw), _p(pw), _p(_c(k["valid"], np.uint8)), _p(_c(k["min_valid_dist"], np.float32)), _p(_c(k["max_valid_dist"], np.float32)),
                _p(_c(k["lm_desc"], np.uint8)), _p(_c(k["desc"], np.uint8)), _p(_c(k["xy"], np.float32)), _p(_c(k["octave"], np.int32))]
    n1, n2 = len(kf1["pos_w"]), len(kf2["pos_w"])
    m21, m12, mut = np.full(n1, -1, np.int32), np.full(n2, -1, np.int32), np.full(n1, -1, np.int32)
    num = lib().orc_match_keyframes_mutually(
        C.byref(cam1), C.byref(cam2), _p(_f64(rot_1w)), _p(_f64(trans_1w)), _p(_f64(rot_2w)), _p(_f64(trans_2w)), C.c_float(s_12), _p(_f64(rot_12)),
        _p(_f64(trans_12)), *side(kf1), *side(kf2), len(sf), _p(sf), C.c_float(log_scale_factor), C.c_float(margin), grid_cols, grid_rows, _p(m21), _p(m12),
        _p(mut))
    return m21, m12, mut, num


def fuse_detect_duplication(cam, rot_cw, trans_cw, pos_w, valid, min_valid_dist, max_valid_dist, mean_normal, lm_desc, scale_factors, inv_level_sigma_sq,
                            log_scale_factor, margin, tdesc, t_xy, t_octave, t_xright=None, do_reprojection_matching=False, grid_cols=64, grid_rows=48):
    pw = _f64(pos_w, (-1, 3))
    n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
    out = np.full(n, -1, np.int32)
    num = lib().orc_fuse_detect_duplication(
        C.byref(cam), _p(_f64(rot_cw)), _p(_f64(trans_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)),
        _p(_c(max_valid_dist, np.float32)), _p(_f64(mean_normal)), _p(_c(lm_desc, np.uint8)), len(sf), _p(sf), _p(_c(inv_level_sigma_sq, np.float32)),
        C.c_float(log_scale_factor), C.c_float(margin), int(do_reprojection_matching), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)),
        _p(_c(t_xright, np.float32)), len(td), grid_cols, grid_rows, _p(out))
    return out, num
