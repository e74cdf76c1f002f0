"""Python mirror of stella_vslam::match::* over the C ABI (flat-array form).

The reference matchers take frame / keyframe / landmark objects (match/robust.h, match/projection.h).
Here a "frame observation" is the flat part of data::frame_observation (data/frame_observation.h:12-38):
`descriptors` (N x 32 uint8) and `keypts` (structured array with at least `angle`, `octave`).
Constructor arguments and thresholds are the reference's (match/base.h:15-17,81-91).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .feature import Context

HAMMING_DIST_THR_LOW = 50
HAMMING_DIST_THR_HIGH = 100
MAX_HAMMING_DIST = 256

MATCH_BEST_ONLY = 0
MATCH_RATIO_SAME_OCTAVE = 1
MATCH_RATIO = 2          # bow_tree::match_frame_and_keyframe / match_keyframes
MATCH_TRIANGULATION = 3  # bow_tree / robust ::match_for_triangulation
MATCH_AREA = 4           # area::match_in_consistent_area


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, t):
    return None if a is None else np.ascontiguousarray(a, t)


def compute_descriptor_distance_32(ctx: Context, desc_1: np.ndarray, desc_2: np.ndarray) -> np.ndarray:
    """match/base.h:20-41 for n row pairs."""
    a = np.ascontiguousarray(desc_1, np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(desc_2, np.uint8).reshape(-1, 32)
    out = np.zeros(len(a), np.uint32)
    ctx.check(lib().svgpu_hamming_distance(ctx.handle, _p(a), _p(b), len(a), _p(out)), "svgpu_hamming_distance")
    return out


def hamming_matrix(ctx: Context, desc1: np.ndarray, desc2: np.ndarray) -> np.ndarray:
    d1, d2 = _c(desc1, np.uint8), _c(desc2, np.uint8)
    out = np.zeros((len(d2), len(d1)), np.uint16)
    ctx.check(lib().svgpu_hamming_matrix(ctx.handle, _p(d1), len(d1), _p(d2), len(d2), _p(out)), "svgpu_hamming_matrix")
    return out


class base:
    def __init__(self, lowe_ratio: float, check_orientation: bool, ctx: Context | None = None):
        self.lowe_ratio_ = float(lowe_ratio)
        self.check_orientation_ = bool(check_orientation)
        self.ctx = ctx or Context()


class robust(base):
    """match/robust.h.  brute_force_match (match/robust.cc:232-328): returns the (idx_1, idx_2) pairs sorted by idx_1."""

    def brute_force_match(self, desc_1, angle_1, desc_2, angle_2, lm_valid_2=None):
        d1, d2 = _c(desc_1, np.uint8), _c(desc_2, np.uint8)
        a1, a2 = _c(angle_1, np.float32), _c(angle_2, np.float32)
        v2 = _c(lm_valid_2, np.uint8)
        out = np.full(len(d1), -1, np.int32)
        num = C.c_int(0)
        self.ctx.check(lib().svgpu_match_bruteforce(self.ctx.handle, _p(d1), _p(a1), len(d1), _p(d2), _p(a2), _p(v2), len(d2),
                                                    C.c_float(self.lowe_ratio_), int(self.check_orientation_), _p(out),
                                                    C.byref(num)), "svgpu_match_bruteforce")
        idx1 = np.flatnonzero(out >= 0)
        assert len(idx1) == num.value
        return [(int(i), int(out[i])) for i in idx1], out


class projection(base):
    """match/projection.h on flattened inputs: the caller supplies, per query (landmark / last-frame keypoint),
    the candidate keypoint indices that get_keypoints_in_cell returned (data/common.cc:127-190), as CSR."""

    def match_candidates(self, qdesc, tdesc, cand_off, cand_idx, mode, thr, cand_skip=None, t_octave=None, q_valid=None, occupied=None,
                         q_angle=None, t_angle=None, q_xright=None, t_xright=None, q_xr_tol=None):
        qd, td = _c(qdesc, np.uint8), _c(tdesc, np.uint8)
        off, idx = _c(cand_off, np.int32), _c(cand_idx, np.int32)
        toct, qv, occ = _c(t_octave, np.int32), _c(q_valid, np.uint8), _c(occupied, np.uint8)
        skip = _c(cand_skip, np.uint8)
        qa, ta = _c(q_angle, np.float32), _c(t_angle, np.float32)
        qx, tx, qt = _c(q_xright, np.float32), _c(t_xright, np.float32), _c(q_xr_tol, np.float32)
        out = np.full(len(qd), -1, np.int32)
        num = C.c_int(0)
        self.ctx.check(lib().svgpu_match_candidates(self.ctx.handle, _p(qd), len(qd), _p(td), _p(toct), len(td), _p(off), _p(idx),
                                                    _p(skip), _p(qv), _p(occ), _p(qa), _p(ta), int(self.check_orientation_), _p(qx),
                                                    _p(tx), _p(qt), C.c_uint(thr), C.c_float(self.lowe_ratio_), mode, _p(out),
                                                    C.byref(num)), "svgpu_match_candidates")
        return out, num.value


    def match_in_cells(self, qdesc, q_xy, q_margin, tdesc, t_xy, t_octave, bounds, mode, thr, q_min_level=None, q_max_level=None,
                       q_valid=None, occupied=None, q_angle=None, t_angle=None, q_xright=None, t_xright=None, q_xr_tol=None,
                       grid_cols=64, grid_rows=48):
        """Same matcher, candidate lists built on the device: query q scans frm.get_keypoints_in_cell(q_xy[q], q_margin[q],
        q_min_level[q], q_max_level[q]) (data/common.cc:127-190) over the grid of data::assign_keypoints_to_grid."""
        qd, td = _c(qdesc, np.uint8), _c(tdesc, np.uint8)
        qxy, txy, qm = _c(q_xy, np.float32), _c(t_xy, np.float32), _c(q_margin, np.float32)
        toct, qlo, qhi = _c(t_octave, np.int32), _c(q_min_level, np.int32), _c(q_max_level, np.int32)
        qv, occ = _c(q_valid, np.uint8), _c(occupied, np.uint8)
        qa, ta = _c(q_angle, np.float32), _c(t_angle, np.float32)
        qx, tx, qt = _c(q_xright, np.float32), _c(t_xright, np.float32), _c(q_xr_tol, np.float32)
        out = np.full(len(qd), -1, np.int32)
        num = C.c_int(0)
        self.ctx.check(lib().svgpu_match_in_cells(self.ctx.handle, _p(qd), len(qd), _p(qxy), _p(qm), _p(qlo), _p(qhi), _p(qv), _p(qa), _p(qx),
                                                  _p(qt), _p(td), _p(txy), _p(toct), len(td), _p(occ), _p(ta), _p(tx),
                                                  C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]),
                                                  grid_cols, grid_rows, int(self.check_orientation_ and qa is not None and ta is not None), C.c_uint(thr),
                                                  C.c_float(self.lowe_ratio_), mode, _p(out), C.byref(num)), "svgpu_match_in_cells")
        return out, num.value


    def match_frame_and_landmarks(self, camera, rot_cw, trans_cw, pos_w, mean_normal, min_valid_dist, max_valid_dist, lm_desc,
                                  frm_obs, scale_factors, log_scale_factor, margin=5.0, skip=None, occupied=None, ray_cos_thr=0.5,
                                  trans_wc=None):
        """tracking_module::search_local_landmarks' observability loop (tracking_module.cc:554-594 -> data/frame.cc:59-85)
        followed by projection::match_frame_and_landmarks (match/projection.cc:13-93) as ONE device pass: the landmarks come in
        as flat arrays (position, mean viewing direction, valid distance range, representative descriptor), `skip` marks the
        ones the caller's object graph excludes, `occupied` the keypoints that already hold an observed landmark.
        Returns (match_lm, num_matches, visible, reproj, x_right, pred_scale_level); match_lm[i] = keypoint index given to
        frm.add_landmark(lm_i, idx) or -1."""
        R = np.ascontiguousarray(rot_cw, np.float64).reshape(3, 3)
        t = np.ascontiguousarray(trans_cw, np.float64).reshape(3)
        twc = np.ascontiguousarray(-R.T @ t if trans_wc is None else trans_wc, np.float64)
        pw = np.ascontiguousarray(pos_w, np.float64).reshape(-1, 3)
        nv = np.ascontiguousarray(mean_normal, np.float64).reshape(-1, 3)
        mn, mx = _c(min_valid_dist, np.float32), _c(max_valid_dist, np.float32)
        qd = _c(lm_desc, np.uint8)
        sk, occ = _c(skip, np.uint8), _c(occupied, np.uint8)
        sf = _c(scale_factors, np.float32)
        td = _c(frm_obs.descriptors_, np.uint8)
        txy = np.ascontiguousarray(np.stack([frm_obs.undist_keypts_["x"], frm_obs.undist_keypts_["y"]], 1), np.float32)
        toct = _c(frm_obs.undist_keypts_["octave"], np.int32)
        tx = _c(frm_obs.stereo_x_right_, np.float32)
        n = len(pw)
        out = np.full(n, -1, np.int32)
        num = C.c_int(0)
        vis, rp, xr, lv = np.zeros(n, np.uint8), np.zeros((n, 2), np.float64), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.ctx.check(lib().svgpu_match_frame_and_landmarks(
            self.ctx.handle, C.byref(camera.c_), _p(R), _p(t), _p(twc), n, _p(pw), _p(nv), _p(mn), _p(mx), _p(sk), _p(qd),
            C.c_float(ray_cos_thr), len(sf), _p(sf), C.c_float(log_scale_factor), C.c_float(margin), _p(td), _p(txy), _p(toct), len(td),
            _p(occ), _p(tx), frm_obs.num_grid_cols_, frm_obs.num_grid_rows_, C.c_uint(HAMMING_DIST_THR_HIGH), C.c_float(self.lowe_ratio_),
            _p(out), C.byref(num), _p(vis), _p(rp), _p(xr), _p(lv)), "svgpu_match_frame_and_landmarks")
        return out, num.value, vis, rp, xr, lv


class area(base):
    """match/area.h (the monocular initialiser's matcher).  match_in_consistent_area (match/area.cc:8-98) on flattened inputs:
    level-0 keypoints of frame 1 are the queries, the candidate list of query idx_1 is
    frm_2.get_keypoints_in_cell(prev_matched_pts[idx_1], margin, 0, 0) (empty for keypoints of higher levels), as CSR.
    Returns matched_indices_2_in_frm_1 and the number of matches; the caller then refreshes prev_matched_pts (:91-95)."""

    def match_in_consistent_area(self, desc_1, angle_1, desc_2, angle_2, cand_off, cand_idx):
        out, num = projection.match_candidates(self, desc_1, desc_2, cand_off, cand_idx, MATCH_AREA, 50, q_angle=angle_1, t_angle=angle_2)
        return out, num


class stereo:
    """match/stereo.h: stereo(left_pyramid, right_pyramid, keypts_left, keypts_right, descs_left, descs_right, scale_factors,
    inv_scale_factors, focal_x_baseline, true_baseline).compute() -> (stereo_x_right, depths).  The two pyramids are taken
    from the extractors that produced the keypoints (their last extract call), as system.cc:443-447 does."""

    def __init__(self, extractor_left, extractor_right, keypts_left, keypts_right, descs_left, descs_right,
                 focal_x_baseline: float, true_baseline: float):
        self.el, self.er = extractor_left, extractor_right
        self.kl, self.kr = np.ascontiguousarray(keypts_left), np.ascontiguousarray(keypts_right)
        self.dl, self.dr = _c(descs_left, np.uint8), _c(descs_right, np.uint8)
        self.focal_x_baseline_, self.true_baseline_ = float(focal_x_baseline), float(true_baseline)

    def compute(self):
        n = len(self.kl)
        xr, dp = np.full(max(n, 1), -1, np.float32), np.full(max(n, 1), -1, np.float32)
        ctx = self.el.ctx
        ctx.check(lib().svgpu_stereo_match(ctx.handle, self.er.ctx.handle, _p(self.kl), _p(self.dl), n, _p(self.kr), _p(self.dr),
                                           len(self.kr), C.c_float(self.focal_x_baseline_), C.c_float(self.true_baseline_),
                                           _p(xr), _p(dp)), "svgpu_stereo_match")
        return xr[:n].copy(), dp[:n].copy()
