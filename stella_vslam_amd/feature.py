"""Python mirror of stella_vslam::feature::{orb_params, orb_extractor} over the C ABI.

Same names, argument meaning and error behaviour as the reference classes
(feature/orb_params.h, feature/orb_extractor.h:46-71); the C++ adaptor with the identical C++
signature lives in stella_vslam_amd/host/orb_extractor.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import SvgpuError, lib

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])


class Context:
    """One svgpu_ctx: a device, a stream and the workspaces configured on it."""

    def __init__(self, device: int = 0, priority: int = 0):
        self._h = C.c_void_p()
        rc = lib().svgpu_create_with_priority(device, priority, C.byref(self._h))
        if rc:
            raise SvgpuError(rc, "svgpu_create")
        self.device = device

    def check(self, rc: int, where: str, ok=(0,)):
        if rc not in ok:
            raise SvgpuError(rc, where, lib().svgpu_last_error(self._h).decode())
        return rc

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return lib().svgpu_stream(self._h)

    def synchronize(self):
        self.check(lib().svgpu_synchronize(self._h), "svgpu_synchronize")

    def close(self):
        if self._h:
            lib().svgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class orb_params:
    """feature/orb_params.cc:12-71 (defaults 1.2 / 8 / 20 / 7)."""

    def __init__(self, name: str = "default ORB feature extraction setting", scale_factor: float = 1.2,
                 num_levels: int = 8, ini_fast_thr: int = 20, min_fast_thr: int = 7):
        self.name_ = name
        self.scale_factor_ = float(np.float32(scale_factor))
        self.log_scale_factor_ = float(np.log(np.float32(scale_factor)))
        self.num_levels_ = int(num_levels)
        self.ini_fast_thr_ = int(ini_fast_thr)
        self.min_fast_thr_ = int(min_fast_thr)
        tabs = [np.zeros(num_levels, np.float32) for _ in range(4)]
        rc = lib().svgpu_orb_scale_tables(C.c_float(scale_factor), num_levels, *[t.ctypes.data_as(C.c_void_p) for t in tabs])
        if rc:
            raise SvgpuError(rc, "svgpu_orb_scale_tables")
        self.scale_factors_, self.inv_scale_factors_, self.level_sigma_sq_, self.inv_level_sigma_sq_ = tabs


def rectangle_mask(mask_rects, cols: int, rows: int) -> np.ndarray:
    """create_rectangle_mask (orb_extractor.cc:138-151): 255 background, every [x_min, x_max, y_min, y_max] ratio rectangle filled with 0,
    both corner points inclusive (cv::rectangle).  The corners are std::round of a FLOAT product: half away from zero."""
    def rnd(n, f):
        x = np.float32(n) * np.float32(f)
        r = np.floor(x)
        return int(r) + (1 if x - r >= np.float32(0.5) else 0)
    m = np.full((rows, cols), 255, np.uint8)
    for r in mask_rects:
        x_min, x_max = rnd(cols, r[0]), rnd(cols, r[1])
        y_min, y_max = rnd(rows, r[2]), rnd(rows, r[3])
        m[max(y_min, 0):y_max + 1, max(x_min, 0):x_max + 1] = 0
    return m


class orb_extractor:
    """feature/orb_extractor.h:46-71.  extract(image, mask) -> (keypoints, descriptors).

    The image geometry is bound at the first extract() (workspaces are sized for it) and re-bound
    when it changes.  `image_pyramid_` is downloaded lazily, as match::stereo needs it.
    """

    def __init__(self, orb_params_: orb_params, min_area: int = 800, mask_rects=(), ctx: Context | None = None,
                 max_batch: int = 1):
        self.orb_params_ = orb_params_
        self.mask_rects_ = [list(r) for r in mask_rects]
        self.min_area_ = int(min_area)
        self.ctx = ctx or Context()
        self.max_batch = max_batch
        self._geom = None
        self._rect_mask = None

    # -- geometry
    def _configure(self, w: int, h: int):
        if self._geom == (w, h):
            return
        p = self.orb_params_
        self.ctx.check(lib().svgpu_orb_configure(self.ctx.handle, w, h, self.max_batch, C.c_float(p.scale_factor_),
                                                 p.num_levels_, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(self.min_area_)),
                       "svgpu_orb_configure")
        self._geom = (w, h)
        self._rect_mask = None

    def max_keypoints(self) -> int:
        return lib().svgpu_orb_max_keypoints(self.ctx.handle)

    def level_size(self, level: int):
        w, h = C.c_int(), C.c_int()
        self.ctx.check(lib().svgpu_orb_level_size(self.ctx.handle, level, C.byref(w), C.byref(h)), "svgpu_orb_level_size")
        return w.value, h.value

    def _rectangle_mask(self, cols: int, rows: int):
        if self._rect_mask is None:
            self._rect_mask = rectangle_mask(self.mask_rects_, cols, rows)
        return self._rect_mask

    # -- extract
    def extract(self, in_image: np.ndarray, in_image_mask: np.ndarray | None = None):
        if in_image is None or in_image.size == 0:  # orb_extractor.cc:30-32: silent return
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        if in_image.dtype != np.uint8 or in_image.ndim != 2:
            raise TypeError("image must be CV_8UC1 (2-D uint8)")  # assert(image.type() == CV_8UC1)
        if in_image.strides[1] != 1:
            in_image = np.ascontiguousarray(in_image)
        h, w = in_image.shape
        self._configure(w, h)
        mask = in_image_mask
        if mask is not None and mask.size == 0:
            mask = None
        if mask is None and self.mask_rects_:
            mask = self._rectangle_mask(w, h)
        if mask is not None:
            if mask.dtype != np.uint8 or mask.shape != in_image.shape:
                raise TypeError("mask must be CV_8UC1 of the image size")
            if mask.strides[1] != 1:
                mask = np.ascontiguousarray(mask)
        cap = max(self.max_keypoints(), 1)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        counts = np.zeros(self.orb_params_.num_levels_, np.int32)
        self.ctx.check(lib().svgpu_orb_extract(self.ctx.handle, C.c_void_p(in_image.ctypes.data), in_image.strides[0],
                                               None if mask is None else C.c_void_p(mask.ctypes.data),
                                               0 if mask is None else mask.strides[0],
                                               kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                               C.byref(n), counts.ctypes.data_as(C.c_void_p)), "svgpu_orb_extract")
        self.level_counts_ = counts
        return kps[:n.value].copy(), desc[:n.value].copy()

    @property
    def image_pyramid_(self):
        """Levels 0..L-1 of the last extracted frame (feature/orb_extractor.h:71)."""
        out = []
        for l in range(self.orb_params_.num_levels_):
            w, h = self.level_size(l)
            a = np.zeros((h, w), np.uint8)
            self.ctx.check(lib().svgpu_orb_pyramid_download(self.ctx.handle, 0, l, a.ctypes.data_as(C.c_void_p), w),
                           "svgpu_orb_pyramid_download")
            out.append(a)
        return out

    def blurred_pyramid(self):
        out = []
        for l in range(self.orb_params_.num_levels_):
            w, h = self.level_size(l)
            a = np.zeros((h, w), np.uint8)
            self.ctx.check(lib().svgpu_orb_blurred_download(self.ctx.handle, 0, l, a.ctypes.data_as(C.c_void_p), w),
                           "svgpu_orb_blurred_download")
            out.append(a)
        return out
