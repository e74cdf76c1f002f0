"""Device-resident batches: thin torch-backed holders around the *_batch_device entry points of the C ABI.

PyTorch is only the owner of the HBM buffers and streams here (the data path is libsvgpu).  Used by bench.py and the GPU tests:
  BatchExtractor     svgpu_orb_extract_batch_device for B frames of one geometry -> keypoints / descriptors / counts in HBM
  stereo_batch       svgpu_stereo_match_batch_device over two BatchExtractors (left / right images of B stereo pairs)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib
from .feature import Context, orb_params

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class BatchExtractor:
    def __init__(self, width: int, height: int, batch: int, params: orb_params | None = None, ctx: Context | None = None, device: int = 0,
                 priority: int = 0, min_area: int = 800):
        import torch
        self.torch = torch
        self.ctx = ctx or Context(device, priority=priority)
        self.params = params or orb_params()
        self.W, self.H, self.B = width, height, batch
        L = lib()
        p = self.params
        self.ctx.check(L.svgpu_orb_configure(self.ctx.handle, width, height, batch, C.c_float(p.scale_factor_), p.num_levels_, p.ini_fast_thr_,
                                             p.min_fast_thr_, C.c_uint(min_area)), "svgpu_orb_configure")
        self.cap = L.svgpu_orb_max_keypoints(self.ctx.handle)
        self.nc = 1 + p.num_levels_
        self.stream = torch.cuda.ExternalStream(self.ctx.stream)
        with torch.cuda.stream(self.stream):
            self.kps = torch.zeros(batch * self.cap * 28, dtype=torch.uint8, device="cuda")
            self.desc = torch.zeros(batch * self.cap * 32, dtype=torch.uint8, device="cuda")
            self.counts = torch.zeros(batch * self.nc, dtype=torch.int32, device="cuda")
        self.stream.synchronize()
        self.frames = None

    def upload(self, frames_np: np.ndarray):
        """frames_np: (B, H, W) uint8; stays resident until replaced."""
        assert frames_np.shape == (self.B, self.H, self.W) and frames_np.dtype == np.uint8
        with self.torch.cuda.stream(self.stream):
            self.frames = self.torch.from_numpy(np.ascontiguousarray(frames_np)).cuda()
        self.stream.synchronize()

    def extract(self):
        """Asynchronous on the context's stream."""
        self.ctx.check(lib().svgpu_orb_extract_batch_device(self.ctx.handle, C.c_void_p(self.frames.data_ptr()), self.B, C.c_size_t(self.W * self.H), self.W,
                                                           None, C.c_size_t(0), 0, C.c_void_p(self.kps.data_ptr()), C.c_void_p(self.desc.data_ptr()), self.cap,
                                                           C.c_void_p(self.counts.data_ptr()), None), "svgpu_orb_extract_batch_device")

    def download(self):
        """[(keypoints structured array, descriptors n x 32)] per frame (synchronises)."""
        self.ctx.synchronize()
        cnt = self.counts.cpu().numpy().reshape(self.B, self.nc)
        k = self.kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(self.B, self.cap)
        d = self.desc.cpu().numpy().reshape(self.B, self.cap, 32)
        return [(k[b, :min(cnt[b, 0], self.cap)].copy(), d[b, :min(cnt[b, 0], self.cap)].copy()) for b in range(self.B)]


def stereo_batch(left: BatchExtractor, right: BatchExtractor, focal_x_baseline: float, true_baseline: float, out=None, stream=None):
    """match::stereo::compute for every pair of the two batches (svgpu_stereo_match_batch_device); returns (x_right, depths) torch tensors
    of shape (B, cap) living in HBM.  Asynchronous on `stream` (default: the left context's); the right extraction must be complete or
    ordered before it by the caller."""
    torch = left.torch
    assert left.B == right.B and left.cap == right.cap
    if out is None:
        with torch.cuda.stream(left.stream):
            out = (torch.empty(left.B * left.cap, dtype=torch.float32, device="cuda"), torch.empty(left.B * left.cap, dtype=torch.float32, device="cuda"))
    left.ctx.check(lib().svgpu_stereo_match_batch_device(left.ctx.handle, right.ctx.handle, left.B, C.c_void_p(left.kps.data_ptr()), C.c_void_p(left.desc.data_ptr()),
                                                         C.c_void_p(left.counts.data_ptr()), C.c_void_p(right.kps.data_ptr()), C.c_void_p(right.desc.data_ptr()),
                                                         C.c_void_p(right.counts.data_ptr()), left.cap, left.nc, C.c_float(focal_x_baseline),
                                                         C.c_float(true_baseline), C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                                         None if stream is None else C.c_void_p(stream)), "svgpu_stereo_match_batch_device")
    return out
