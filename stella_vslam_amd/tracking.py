"""Python mirror of the tracked-frame chain over the C ABI (include/svgpu.h svgpu_map_* / svgpu_tracker_* / svgpu_track_*).

`landmark_table` is the device-resident local map (what tracking reads of data::landmark, indexed by landmark id); `tracker` runs the two
halves of tracking_module's per-frame chain -- frame_tracker::motion_based_track (module/frame_tracker.cc:22-60) and
search_local_landmarks + the second pose optimisation (tracking_module.cc:533-608, 441-446) -- as one submission each.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import camera as _camera
from ._lib import lib
from .data import resident_frame
from .feature import KEYPOINT_DTYPE, Context

LM_PRESENT, LM_HAS_OBSERVATION, LM_HAS_DESCRIPTOR = 1, 2, 4
LANDMARK_RECORD_DTYPE = np.dtype([("pos_w", "<f8", 3), ("mean_normal", "<f8", 3), ("min_valid_dist", "<f4"), ("max_valid_dist", "<f4"),
                                  ("descriptor", "u1", 32), ("flags", "<u4"), ("reserved", "<u4")])
assert LANDMARK_RECORD_DTYPE.itemsize == 96


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def landmark_records(pos_w, mean_normal, min_valid_dist, max_valid_dist, descriptor, flags=None) -> np.ndarray:
    """flat arrays -> svgpu_landmark_record[n]; flags default to present + has_observation + has_descriptor"""
    n = len(pos_w)
    r = np.zeros(n, LANDMARK_RECORD_DTYPE)
    r["pos_w"], r["mean_normal"] = pos_w, mean_normal
    r["min_valid_dist"], r["max_valid_dist"] = min_valid_dist, max_valid_dist
    r["descriptor"] = np.asarray(descriptor, np.uint8).reshape(n, 32)
    r["flags"] = (LM_PRESENT | LM_HAS_OBSERVATION | LM_HAS_DESCRIPTOR) if flags is None else flags
    return r


class landmark_table:
    """svgpu_map: landmark records on the device, indexed by data::landmark::id_"""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = C.c_void_p()
        ctx.check(lib().svgpu_map_create(ctx.handle, C.byref(self._h)), "svgpu_map_create")

    def upsert(self, ids, records, ctx: Context | None = None):
        c = ctx or self.ctx
        i = np.ascontiguousarray(ids, np.uint32)
        r = np.ascontiguousarray(records, LANDMARK_RECORD_DTYPE)
        assert len(i) == len(r)
        c.check(lib().svgpu_map_upsert(c.handle, self._h, len(i), _p(i), _p(r)), "svgpu_map_upsert")
        return self

    def erase(self, ids, ctx: Context | None = None):
        c = ctx or self.ctx
        i = np.ascontiguousarray(ids, np.uint32)
        c.check(lib().svgpu_map_erase(c.handle, self._h, len(i), _p(i)), "svgpu_map_erase")
        return self

    def download(self, ids, ctx: Context | None = None) -> np.ndarray:
        c = ctx or self.ctx
        i = np.ascontiguousarray(ids, np.uint32)
        r = np.zeros(len(i), LANDMARK_RECORD_DTYPE)
        c.check(lib().svgpu_map_download(c.handle, self._h, len(i), _p(i), _p(r)), "svgpu_map_download")
        return r

    @property
    def capacity(self) -> int:
        return int(lib().svgpu_map_capacity(self._h))

    def close(self):
        if self._h:
            lib().svgpu_map_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _TrackConfig(C.Structure):
    _fields_ = [("num_levels", C.c_int), ("scale_factors", C.c_float * 16), ("inv_level_sigma_sq", C.c_float * 16), ("log_scale_factor", C.c_float),
                ("grid_cols", C.c_int), ("grid_rows", C.c_int), ("is_monocular", C.c_int), ("true_baseline", C.c_float),
                ("po_num_trials_robust", C.c_int), ("po_num_trials", C.c_int), ("po_num_each_iter", C.c_int), ("po_reset_stop_flag_each_round", C.c_int)]


class _TrackResult(C.Structure):
    _fields_ = [("n_keypoints", C.c_int), ("num_matches", C.c_int), ("num_valid", C.c_int), ("lm_iterations", C.c_int), ("num_observations", C.c_int),
                ("num_candidates", C.c_int), ("replay_sweeps", C.c_int), ("reserved", C.c_int), ("pose_cw", C.c_double * 12)]

    def as_dict(self):
        d = {f: getattr(self, f) for f, _ in self._fields_ if f != "pose_cw"}
        d["pose_cw"] = np.array(self.pose_cw, np.float64)
        return d


class tracker:
    """svgpu_tracker.  `ctx` is the context the chain's launches go to (configured for ORB when `track_motion` is given an image)."""

    def __init__(self, ctx: Context, table: landmark_table, camera: _camera.base, scale_factors, inv_level_sigma_sq, log_scale_factor: float,
                 num_grid_cols: int = 64, num_grid_rows: int = 48, is_monocular: bool = True, true_baseline: float = 0.0, num_trials_robust: int = 2,
                 num_trials: int = 2, num_each_iter: int = 10, reset_stop_flag_each_round: bool = False):
        self.ctx, self.table, self.camera = ctx, table, camera
        cfg = _TrackConfig()
        cfg.num_levels = len(scale_factors)
        for l in range(cfg.num_levels):
            cfg.scale_factors[l] = float(scale_factors[l])
            cfg.inv_level_sigma_sq[l] = float(inv_level_sigma_sq[l])
        cfg.log_scale_factor = float(log_scale_factor)
        cfg.grid_cols, cfg.grid_rows = num_grid_cols, num_grid_rows
        cfg.is_monocular, cfg.true_baseline = int(is_monocular), float(true_baseline)
        cfg.po_num_trials_robust, cfg.po_num_trials, cfg.po_num_each_iter = num_trials_robust, num_trials, num_each_iter
        cfg.po_reset_stop_flag_each_round = int(reset_stop_flag_each_round)
        self._cfg = cfg
        self._h = C.c_void_p()
        ctx.check(lib().svgpu_tracker_create(ctx.handle, table._h, C.byref(camera.c_), C.byref(cfg), C.byref(self._h)), "svgpu_tracker_create")

    def track_motion(self, cur: resident_frame, last: resident_frame, last_lm_ids, pose_guess_cw, pose_last_cw, margin: float, check_orientation: bool = True,
                     img: np.ndarray | None = None):
        """-> dict(match_last, outlier, result[, keypts, descriptors, undist_keypts, bearings when `img` is given])"""
        ids = np.ascontiguousarray(last_lm_ids, np.int32)
        assert len(ids) == last.size
        guess = np.ascontiguousarray(pose_guess_cw, np.float64).reshape(12)
        plast = np.ascontiguousarray(pose_last_cw, np.float64).reshape(12)
        match = np.full(max(len(ids), 1), -1, np.int32)
        res = _TrackResult()
        out = {}
        if img is not None:
            im = np.ascontiguousarray(img, np.uint8)
            cap = max(int(lib().svgpu_orb_max_keypoints(self.ctx.handle)), 1)
            kps, desc = np.zeros(cap, KEYPOINT_DTYPE), np.zeros((cap, 32), np.uint8)
            und, brg = np.zeros(cap, KEYPOINT_DTYPE), np.zeros((cap, 3), np.float64)
            outl = np.zeros(cap, np.uint8)
            self.ctx.check(lib().svgpu_track_motion(self._h, cur._h, _p(im), im.strides[0], last._h, _p(ids), _p(guess), _p(plast), C.c_float(margin),
                                                    int(check_orientation), _p(kps), _p(desc), _p(und), _p(brg), cap, _p(match), _p(outl), C.byref(res)),
                           "svgpu_track_motion")
            n = res.n_keypoints
            out.update(keypts=kps[:n].copy(), descriptors=desc[:n].copy(), undist_keypts=und[:n].copy(), bearings=brg[:n].copy())
        else:
            outl = np.zeros(max(cur.size, 1), np.uint8)
            self.ctx.check(lib().svgpu_track_motion(self._h, cur._h, None, 0, last._h, _p(ids), _p(guess), _p(plast), C.c_float(margin), int(check_orientation),
                                                    None, None, None, None, 0, _p(match), _p(outl), C.byref(res)), "svgpu_track_motion")
            n = cur.size
        out.update(match_last=match[:len(ids)].copy(), outlier=outl[:n].copy(), result=res.as_dict())
        return out

    def track_motion_rgbd(self, cur: resident_frame, last: resident_frame, last_lm_ids, pose_guess_cw, pose_last_cw, margin: float, img: np.ndarray,
                          depth: np.ndarray, check_orientation: bool = True):
        """An RGB-D frame in one submission (svgpu_track_motion_rgbd): `depth` = float32 metres, same shape as `img`.  Same outputs as track_motion_stereo."""
        return self.track_motion_stereo(cur, last, last_lm_ids, pose_guess_cw, pose_last_cw, margin, img, None, None, check_orientation, depth=depth)

    def track_motion_stereo(self, cur: resident_frame, last: resident_frame, last_lm_ids, pose_guess_cw, pose_last_cw, margin: float, img_left: np.ndarray,
                            img_right: np.ndarray, ctx_right: Context, check_orientation: bool = True, depth: np.ndarray | None = None):
        """A stereo frame in one submission (svgpu_track_motion_stereo): both extractions, match::stereo::compute, the left observation, matcher and
        optimiser.  -> dict(match_last, outlier, result, keypts, descriptors, undist_keypts, bearings, stereo_x_right, depths)"""
        from .feature import KEYPOINT_DTYPE as KP
        ids = np.ascontiguousarray(last_lm_ids, np.int32)
        assert len(ids) == last.size
        guess = np.ascontiguousarray(pose_guess_cw, np.float64).reshape(12)
        plast = np.ascontiguousarray(pose_last_cw, np.float64).reshape(12)
        match = np.full(max(len(ids), 1), -1, np.int32)
        res = _TrackResult()
        il = np.ascontiguousarray(img_left, np.uint8)
        cap = max(int(lib().svgpu_orb_max_keypoints(self.ctx.handle)), 1)
        outl = np.zeros(cap, np.uint8)
        if depth is not None:
            dm = np.ascontiguousarray(depth, np.float32)
            assert dm.shape == il.shape
            self.ctx.check(lib().svgpu_track_motion_rgbd(self._h, cur._h, _p(il), il.strides[0], _p(dm), dm.strides[0] // 4, last._h, _p(ids), _p(guess), _p(plast),
                                                         C.c_float(margin), int(check_orientation), cap, _p(match), _p(outl), C.byref(res)), "svgpu_track_motion_rgbd")
        else:
            ir = np.ascontiguousarray(img_right, np.uint8)
            self.ctx.check(lib().svgpu_track_motion_stereo(self._h, ctx_right.handle, cur._h, _p(il), il.strides[0], _p(ir), ir.strides[0], last._h, _p(ids), _p(guess),
                                                           _p(plast), C.c_float(margin), int(check_orientation), cap, _p(match), _p(outl), C.byref(res)),
                           "svgpu_track_motion_stereo")
        n = res.n_keypoints
        pk, pd, pu, pb, px, pz = (C.c_void_p() for _ in range(6))
        got = lib().svgpu_tracker_observation(self._h, C.byref(pk), C.byref(pd), C.byref(pu), C.byref(pb))
        assert got == n and lib().svgpu_tracker_observation_stereo(self._h, C.byref(px), C.byref(pz)) == n

        def view(ptr, dtype, shape):
            if n == 0:
                return np.zeros(shape, dtype)
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * nbytes).from_address(ptr.value), dtype=dtype).reshape(shape).copy()
        return dict(match_last=match[:len(ids)].copy(), outlier=outl[:n].copy(), result=res.as_dict(), keypts=view(pk, KP, (n,)), descriptors=view(pd, np.uint8, (n, 32)),
                    undist_keypts=view(pu, KP, (n,)), bearings=view(pb, np.float64, (n, 3)), stereo_x_right=view(px, np.float32, (n,)),
                    depths=view(pz, np.float32, (n,)))

    def track_local_map(self, cur: resident_frame, cur_lm_ids, local_ids, margin: float = 5.0, lowe_ratio: float = 0.8, ray_cos_thr: float = 0.5,
                        pose_cw=None):
        """-> dict(match_local, visible, outlier, result)"""
        cl = np.ascontiguousarray(cur_lm_ids, np.int32)
        assert len(cl) == cur.size
        li = np.ascontiguousarray(local_ids, np.int32)
        pose = None if pose_cw is None else np.ascontiguousarray(pose_cw, np.float64).reshape(12)
        match, vis = np.full(max(len(li), 1), -1, np.int32), np.zeros(max(len(li), 1), np.uint8)
        outl = np.zeros(max(cur.size, 1), np.uint8)
        res = _TrackResult()
        self.ctx.check(lib().svgpu_track_local_map(self._h, cur._h, _p(cl), len(li), _p(li), _p(pose), C.c_float(margin), C.c_float(lowe_ratio),
                                                   C.c_float(ray_cos_thr), _p(match), _p(vis), _p(outl), C.byref(res)), "svgpu_track_local_map")
        self._n_local = len(li)
        return dict(match_local=match[:len(li)].copy(), visible=vis[:len(li)].copy(), outlier=outl[:cur.size].copy(), result=res.as_dict())

    def local_map_observability(self):
        """lm_to_reproj / lm_to_x_right / lm_to_scale of the last track_local_map as arrays over its local landmarks"""
        n = self._n_local
        rp, xr, lv = np.zeros((max(n, 1), 2), np.float64), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32)
        self.ctx.check(lib().svgpu_track_local_map_observability(self._h, n, _p(rp), _p(xr), _p(lv)), "svgpu_track_local_map_observability")
        return rp[:n], xr[:n], lv[:n]

    def counters(self):
        a, b = C.c_longlong(0), C.c_longlong(0)
        lib().svgpu_tracker_counters(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().svgpu_tracker_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
