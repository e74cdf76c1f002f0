"""Python mirror of stella_vslam::camera::* over the C ABI (the batched members the tracking front end calls).

Constructor arguments are the reference's (camera/perspective.h:17-21, fisheye.h:17-21, equirectangular.h:14-16,
radial_division.h:16-19); `img_bounds_` is filled in the constructor as the reference does (compute_image_bounds).
All arithmetic runs in libsvgpu.so -- there is no host fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .feature import Context, KEYPOINT_DTYPE

MODEL_PERSPECTIVE, MODEL_FISHEYE, MODEL_EQUIRECTANGULAR, MODEL_RADIAL_DIVISION = 0, 1, 2, 3  # camera/base.h:24-29


class svgpu_camera(C.Structure):
    """include/svgpu.h svgpu_camera."""
    _fields_ = [("model", C.c_int32), ("pad_", C.c_int32), ("cols", C.c_double), ("rows", C.c_double), ("fx", C.c_double),
                ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("dist", C.c_double * 5),
                ("focal_x_baseline", C.c_double), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class image_bounds:
    def __init__(self, min_x, max_x, min_y, max_y):
        self.min_x_, self.max_x_, self.min_y_, self.max_y_ = float(min_x), float(max_x), float(min_y), float(max_y)

    def as_tuple(self):
        return (self.min_x_, self.max_x_, self.min_y_, self.max_y_)


class base:
    model_type_ = None

    def __init__(self, name, setup_type, color_order, cols, rows, fps, focal_x_baseline=0.0, true_baseline=0.0, depth_thr=0.0,
                 ctx: Context | None = None, fx=0.0, fy=0.0, cx=0.0, cy=0.0, dist=()):
        self.name_, self.setup_type_, self.color_order_ = name, setup_type, color_order
        self.cols_, self.rows_, self.fps_ = int(cols), int(rows), float(fps)
        self.focal_x_baseline_, self.true_baseline_, self.depth_thr_ = float(focal_x_baseline), float(true_baseline), float(depth_thr)
        self.ctx = ctx or Context()
        c = svgpu_camera()
        c.model, c.cols, c.rows = self.model_type_, float(self.cols_), float(self.rows_)
        c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
        for i, d in enumerate(dist):
            c.dist[i] = float(d)
        c.focal_x_baseline = self.focal_x_baseline_
        self.c_ = c
        self.img_bounds_ = self.compute_image_bounds()

    # camera::*::compute_image_bounds
    def compute_image_bounds(self) -> image_bounds:
        self.ctx.check(lib().svgpu_camera_image_bounds(self.ctx.handle, C.byref(self.c_)), "svgpu_camera_image_bounds")
        return image_bounds(self.c_.min_x, self.c_.max_x, self.c_.min_y, self.c_.max_y)

    def _observe(self, keypts, grid=None, want_undist=True, want_bearings=True):
        k = np.ascontiguousarray(keypts, KEYPOINT_DTYPE)
        n = len(k)
        und = np.zeros(n, KEYPOINT_DTYPE) if want_undist else None
        brg = np.zeros((n, 3), np.float64) if want_bearings else None
        cols, rows = grid if grid else (1, 1)
        off = np.zeros(cols * rows + 1, np.int32) if grid else None
        items = np.zeros(max(n, 1), np.int32) if grid else None
        self.ctx.check(lib().svgpu_frame_observation(self.ctx.handle, C.byref(self.c_), _p(k), n, cols, rows, _p(und), _p(brg), _p(off), _p(items)),
                       "svgpu_frame_observation")
        return und, brg, off, (items[:off[-1]] if grid else None)

    # camera::base::undistort_keypoints (base.cc:139-148) / overrides
    def undistort_keypoints(self, dist_keypts):
        return self._observe(dist_keypts, want_bearings=False)[0]

    # camera::base::convert_keypoints_to_bearings (base.cc:160-164): takes UNDISTORTED keypoints
    def convert_keypoints_to_bearings(self, undist_keypts):
        k = np.ascontiguousarray(undist_keypts, KEYPOINT_DTYPE)
        brg = np.zeros((len(k), 3), np.float64)
        self.ctx.check(lib().svgpu_keypoints_to_bearings(self.ctx.handle, C.byref(self.c_), _p(k), len(k), _p(brg)), "svgpu_keypoints_to_bearings")
        return brg

    # data::frame::can_observe over many landmarks (reproject_to_image + scale / angle gates)
    def reproject_landmarks(self, rot_cw, trans_cw, pos_w, mean_normal, min_valid_dist, max_valid_dist, ray_cos_thr=0.5, num_levels=8,
                            log_scale_factor=None, skip=None, trans_wc=None):
        R = np.ascontiguousarray(rot_cw, np.float64).reshape(3, 3)
        t = np.ascontiguousarray(trans_cw, np.float64).reshape(3)
        twc = np.ascontiguousarray(-R.T @ t if trans_wc is None else trans_wc, np.float64)
        pw = np.ascontiguousarray(pos_w, np.float64).reshape(-1, 3)
        nv = np.ascontiguousarray(mean_normal, np.float64).reshape(-1, 3)
        mn, mx = np.ascontiguousarray(min_valid_dist, np.float32), np.ascontiguousarray(max_valid_dist, np.float32)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        n = len(pw)
        lsf = float(np.log(np.float32(1.2))) if log_scale_factor is None else float(log_scale_factor)
        vis, rp, xr, lv = np.zeros(n, np.uint8), np.zeros((n, 2), np.float64), np.zeros(n, np.float32), np.zeros(n, np.int32)
        self.ctx.check(lib().svgpu_reproject_landmarks(self.ctx.handle, C.byref(self.c_), _p(R), _p(t), _p(twc), n, _p(pw), _p(nv), _p(mn), _p(mx),
                                                       _p(sk), C.c_float(ray_cos_thr), int(num_levels), C.c_float(lsf), _p(vis), _p(rp), _p(xr),
                                                       _p(lv)), "svgpu_reproject_landmarks")
        return vis, rp, xr, lv


class perspective(base):
    """camera/perspective.h."""
    model_type_ = MODEL_PERSPECTIVE

    def __init__(self, name, setup_type, color_order, cols, rows, fps, fx, fy, cx, cy, k1, k2, p1, p2, k3, focal_x_baseline=0.0,
                 depth_thr=0.0, ctx=None):
        self.fx_, self.fy_, self.cx_, self.cy_ = fx, fy, cx, cy
        self.k1_, self.k2_, self.p1_, self.p2_, self.k3_ = k1, k2, p1, p2, k3
        super().__init__(name, setup_type, color_order, cols, rows, fps, focal_x_baseline, focal_x_baseline / fx, depth_thr, ctx, fx, fy, cx, cy,
                         (k1, k2, p1, p2, k3))


class fisheye(base):
    """camera/fisheye.h."""
    model_type_ = MODEL_FISHEYE

    def __init__(self, name, setup_type, color_order, cols, rows, fps, fx, fy, cx, cy, k1, k2, k3, k4, focal_x_baseline=0.0,
                 depth_thr=0.0, ctx=None):
        self.fx_, self.fy_, self.cx_, self.cy_ = fx, fy, cx, cy
        self.k1_, self.k2_, self.k3_, self.k4_ = k1, k2, k3, k4
        super().__init__(name, setup_type, color_order, cols, rows, fps, focal_x_baseline, focal_x_baseline / fx, depth_thr, ctx, fx, fy, cx, cy,
                         (k1, k2, k3, k4))


class equirectangular(base):
    """camera/equirectangular.h."""
    model_type_ = MODEL_EQUIRECTANGULAR

    def __init__(self, name, color_order, cols, rows, fps, ctx=None):
        super().__init__(name, "Monocular", color_order, cols, rows, fps, 0.0, 0.0, 0.0, ctx)


class radial_division(base):
    """camera/radial_division.h."""
    model_type_ = MODEL_RADIAL_DIVISION

    def __init__(self, name, setup_type, color_order, cols, rows, fps, fx, fy, cx, cy, distortion, focal_x_baseline=0.0, depth_thr=0.0,
                 ctx=None):
        self.fx_, self.fy_, self.cx_, self.cy_, self.distortion_ = fx, fy, cx, cy, distortion
        super().__init__(name, setup_type, color_order, cols, rows, fps, focal_x_baseline, focal_x_baseline / fx, depth_thr, ctx, fx, fy, cx, cy,
                         (distortion,))
