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
                       grid_cols=64, grid_rows=48, q_blocks=None):
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
        qb = _c(q_blocks, np.uint8)  # 0 = an accepted query does not close its keypoint (its landmark has no observation, projection.cc:52-55)
        if qb is not None:
            self.ctx.check(lib().svgpu_match_set_query_blocks(self.ctx.handle, _p(qb)), "svgpu_match_set_query_blocks")
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


# ------------------------------------------------------------------------------------------- function-specific matchers
# Flat-array mirrors of the reference methods whose candidate generation and gates run on the device (include/svgpu.h,
# "function-specific matchers").  Argument names are the C ABI's; `cam` is a camera.base (its svgpu_camera carries img_bounds_).

def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, np.float64)
    return a if shape is None else a.reshape(shape)


def _call(ctx, fn_name, args):
    ctx.check(getattr(lib(), fn_name)(ctx.handle, *args), fn_name)


class projection_flat(base):
    """match::projection's methods other than match_frame_and_landmarks, on flat arrays (match/projection.cc:95-629)."""

    def match_current_and_last_frames(self, cam, rot_cw, trans_cw, rot_lw, trans_lw, pos_w, valid, lm_desc, octave_last, angle_last,
                                      scale_factors, margin, tdesc, t_xy, t_octave, t_angle, occupied=None, t_xright=None,
                                      lm_has_observation=None, is_monocular=True, true_baseline=0.0, grid_cols=64, grid_rows=48):
        pw = _f64(pos_w, (-1, 3))
        n, sf = len(pw), _c(scale_factors, np.float32)
        td = _c(tdesc, np.uint8)
        out, num = np.full(n, -1, np.int32), C.c_int(0)
        a = [C.byref(cam.c_), _p(_f64(rot_cw)), _p(_f64(trans_cw)), _p(_f64(rot_lw)), _p(_f64(trans_lw)), int(is_monocular), C.c_float(true_baseline),
             n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(lm_desc, np.uint8)), _p(_c(octave_last, np.int32)), _p(_c(angle_last, np.float32)),
             _p(_c(lm_has_observation, np.uint8)), len(sf), _p(sf), C.c_float(margin), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)),
             _p(_c(t_angle, np.float32)), len(td), _p(_c(occupied, np.uint8)), _p(_c(t_xright, np.float32)), grid_cols, grid_rows,
             int(self.check_orientation_), _p(out), C.byref(num)]
        keep = a  # noqa: F841  (the temporaries stay alive until the call returns)
        _call(self.ctx, "svgpu_match_current_and_last_frames", a)
        return out, num.value

    def match_frame_and_keyframe(self, cam, rot_cw, trans_cw, pos_w, valid, min_valid_dist, max_valid_dist, lm_desc, angle_kf, scale_factors,
                                 log_scale_factor, margin, hamm_dist_thr, tdesc, t_xy, t_octave, t_angle, occupied=None, grid_cols=64, grid_rows=48):
        pw = _f64(pos_w, (-1, 3))
        n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
        out, num = np.full(n, -1, np.int32), C.c_int(0)
        _call(self.ctx, "svgpu_match_frame_and_keyframe_projection",
              [C.byref(cam.c_), _p(_f64(rot_cw)), _p(_f64(trans_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)),
               _p(_c(max_valid_dist, np.float32)), _p(_c(lm_desc, np.uint8)), _p(_c(angle_kf, np.float32)), len(sf), _p(sf), C.c_float(log_scale_factor),
               C.c_float(margin), C.c_uint(hamm_dist_thr), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)), _p(_c(t_angle, np.float32)),
               len(td), _p(_c(occupied, np.uint8)), grid_cols, grid_rows, int(self.check_orientation_), _p(out), C.byref(num)])
        return out, num.value

    def match_by_Sim3_transform(self, cam, sim3_cw, pos_w, valid, min_valid_dist, max_valid_dist, mean_normal, lm_desc, scale_factors, log_scale_factor,
                                margin, tdesc, t_xy, t_octave, occupied=None, grid_cols=64, grid_rows=48):
        pw = _f64(pos_w, (-1, 3))
        n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
        out, num = np.full(n, -1, np.int32), C.c_int(0)
        _call(self.ctx, "svgpu_match_by_sim3_transform",
              [C.byref(cam.c_), _p(_f64(sim3_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)), _p(_c(max_valid_dist, np.float32)),
               _p(_f64(mean_normal)), _p(_c(lm_desc, np.uint8)), len(sf), _p(sf), C.c_float(log_scale_factor), C.c_float(margin), _p(td),
               _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)), len(td), _p(_c(occupied, np.uint8)), grid_cols, grid_rows, _p(out), C.byref(num)])
        return out, num.value

    def match_keyframes_mutually(self, cam1, cam2, rot_1w, trans_1w, rot_2w, trans_2w, s_12, rot_12, trans_12, kf1, kf2, scale_factors,
                                 log_scale_factor, margin, grid_cols=64, grid_rows=48):
        """kf1 / kf2: dicts with pos_w, valid, min_valid_dist, max_valid_dist, lm_desc (the keyframe's landmarks, one per keypoint) and
        desc, xy, octave (its keypoints)."""
        sf = _c(scale_factors, np.float32)

        def side(k):
            pw = _f64(k["pos_w"], (-1, 3))
            return [len(pw), _p(pw), _p(_c(k["valid"], np.uint8)), _p(_c(k["min_valid_dist"], np.float32)), _p(_c(k["max_valid_dist"], np.float32)),
                    _p(_c(k["lm_desc"], np.uint8)), _p(_c(k["desc"], np.uint8)), _p(_c(k["xy"], np.float32)), _p(_c(k["octave"], np.int32))]
        n1, n2 = len(kf1["pos_w"]), len(kf2["pos_w"])
        m21, m12, mut, num = np.full(n1, -1, np.int32), np.full(n2, -1, np.int32), np.full(n1, -1, np.int32), C.c_int(0)
        _call(self.ctx, "svgpu_match_keyframes_mutually",
              [C.byref(cam1.c_), C.byref(cam2.c_), _p(_f64(rot_1w)), _p(_f64(trans_1w)), _p(_f64(rot_2w)), _p(_f64(trans_2w)), C.c_float(s_12), _p(_f64(rot_12)),
               _p(_f64(trans_12))] + side(kf1) + side(kf2) + [len(sf), _p(sf), C.c_float(log_scale_factor), C.c_float(margin), grid_cols, grid_rows, _p(m21),
                                                             _p(m12), _p(mut), C.byref(num)])
        return m21, m12, mut, num.value


class fuse:
    """match/fuse.h: detect_duplication on flat arrays (match/fuse.cc:11-154)."""

    def __init__(self, lowe_ratio: float = 0.6, ctx: Context | None = None):
        self.lowe_ratio_ = float(lowe_ratio)
        self.ctx = ctx or Context()

    def detect_duplication(self, cam, rot_cw, trans_cw, pos_w, valid, min_valid_dist, max_valid_dist, mean_normal, lm_desc, scale_factors,
                           inv_level_sigma_sq, log_scale_factor, margin, tdesc, t_xy, t_octave, t_xright=None, do_reprojection_matching=False,
                           grid_cols=64, grid_rows=48):
        pw = _f64(pos_w, (-1, 3))
        n, sf, td = len(pw), _c(scale_factors, np.float32), _c(tdesc, np.uint8)
        out, num = np.full(n, -1, np.int32), C.c_int(0)
        _call(self.ctx, "svgpu_fuse_detect_duplication",
              [C.byref(cam.c_), _p(_f64(rot_cw)), _p(_f64(trans_cw)), n, _p(pw), _p(_c(valid, np.uint8)), _p(_c(min_valid_dist, np.float32)),
               _p(_c(max_valid_dist, np.float32)), _p(_f64(mean_normal)), _p(_c(lm_desc, np.uint8)), len(sf), _p(sf), _p(_c(inv_level_sigma_sq, np.float32)),
               C.c_float(log_scale_factor), C.c_float(margin), int(do_reprojection_matching), _p(td), _p(_c(t_xy, np.float32)), _p(_c(t_octave, np.int32)),
               _p(_c(t_xright, np.float32)), len(td), grid_cols, grid_rows, _p(out), C.byref(num)])
        return out, num.value


def reproject_to_bearing(cam, rot_cw, trans_cw, pos_w):
    """camera::*::reproject_to_bearing -> (bearing 3, valid)."""
    b, v = np.zeros(3), C.c_int(0)
    lib().svgpu_reproject_to_bearing(C.byref(cam.c_), _p(_f64(rot_cw)), _p(_f64(trans_cw)), _p(_f64(pos_w)), _p(b), C.byref(v))
    return b, bool(v.value)


def match_for_triangulation(ctx, lowe_ratio, check_orientation, desc1, angle1, octave1, bearings1, has_lm1, desc2, angle2, bearings2, has_lm2, E_12,
                            epipole_in_2, valid_epipole, scale_factors, residual_rad_thr, xright1=None, xright2=None, node1=None, node2=None):
    """robust::match_for_triangulation (node1 = node2 = None) / bow_tree::match_for_triangulation (bow_feat_vec_ node ids given)."""
    d1, d2, sf = _c(desc1, np.uint8), _c(desc2, np.uint8), _c(scale_factors, np.float32)
    out, num = np.full(len(d1), -1, np.int32), C.c_int(0)
    ctx.check(lib().svgpu_match_for_triangulation(
        ctx.handle, _p(d1), _p(_c(angle1, np.float32)), _p(_c(octave1, np.int32)), _p(_f64(bearings1)), _p(_c(has_lm1, np.uint8)), _p(_c(xright1, np.float32)),
        len(d1), _p(d2), _p(_c(angle2, np.float32)), _p(_f64(bearings2)), _p(_c(has_lm2, np.uint8)), _p(_c(xright2, np.float32)), len(d2),
        _p(_c(node1, np.int32)), _p(_c(node2, np.int32)), _p(_f64(E_12)), _p(_f64(epipole_in_2)), int(valid_epipole), _p(sf), len(sf),
        C.c_float(residual_rad_thr), C.c_float(lowe_ratio), int(check_orientation), _p(out), C.byref(num)), "svgpu_match_for_triangulation")
    return out, num.value


class bow_tree(base):
    """match/bow_tree.h on flat arrays: side 1 hands its landmarks over (the keyframe), side 2 receives them (frame / other keyframe);
    node ids = bow_feat_vec_ membership of every keypoint (data.bow_vocabulary.descend)."""

    def match(self, desc1, angle1, valid1, node1, desc2, angle2, node2, valid2=None, occupied2=None):
        d1, d2 = _c(desc1, np.uint8), _c(desc2, np.uint8)
        out, num = np.full(len(d1), -1, np.int32), C.c_int(0)
        self.ctx.check(lib().svgpu_bow_match(self.ctx.handle, _p(d1), _p(_c(angle1, np.float32)), _p(_c(valid1, np.uint8)), _p(_c(node1, np.int32)),
                                             len(d1), _p(d2), _p(_c(angle2, np.float32)), _p(_c(valid2, np.uint8)), _p(_c(node2, np.int32)), len(d2),
                                             _p(_c(occupied2, np.uint8)), C.c_float(self.lowe_ratio_), int(self.check_orientation_), _p(out),
                                             C.byref(num)), "svgpu_bow_match")
        return out, num.value

    # bow_tree.cc:169-256: matched_lms_in_frm[match[i]] = keyfrm landmark i
    match_frame_and_keyframe = match
    # bow_tree.cc:258-366 (valid2 = keypoints of keyframe 2 that hold a live landmark)
    match_keyframes = match

    def match_for_triangulation(self, **kw):
        return match_for_triangulation(self.ctx, self.lowe_ratio_, self.check_orientation_, **kw)
