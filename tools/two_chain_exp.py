#!/usr/bin/env python3
"""Experiment: the headline pipeline as ONE extraction chain of B frames (bench.py) against TWO chains of B/2 frames on two contexts
(each with its own matcher stream), same total frames per step.  usage: tools/two_chain_exp.py [B]"""
import ctypes as C, sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from stella_vslam_amd import feature, synthetic
from stella_vslam_amd._lib import lib
L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 640, 480
frames_np = synthetic.frame_sequence(B, W, H, seed=0x5EED)
params = feature.orb_params()
NL = params.num_levels_

class Chain:
    def __init__(self, frames, prio=0):
        self.ctx = feature.Context(0)
        self.b = len(frames)
        c = self.ctx
        c.check(L.svgpu_orb_configure(c.handle, W, H, self.b, C.c_float(params.scale_factor_), NL, params.ini_fast_thr_, params.min_fast_thr_, C.c_uint(800)), "cfg")
        self.cap = L.svgpu_orb_max_keypoints(c.handle)
        self.stream = torch.cuda.ExternalStream(c.stream)
        self.stream_b = torch.cuda.Stream()
        nc = 1 + NL
        self.nc = nc
        with torch.cuda.stream(self.stream):
            self.frames = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
            self.bufs = [dict(kps=torch.zeros(self.b * self.cap * 28, dtype=torch.uint8, device="cuda"), desc=torch.zeros(self.b * self.cap * 32, dtype=torch.uint8, device="cuda"),
                              counts=torch.zeros(self.b * nc, dtype=torch.int32, device="cuda"), matched=torch.zeros(self.b * self.cap, dtype=torch.int32, device="cuda"),
                              nmatch=torch.zeros(self.b, dtype=torch.int32, device="cuda"), ev_ext=torch.cuda.Event(), ev_match=torch.cuda.Event(), used=False) for _ in range(2)]
        self.stream.synchronize()
        self.i = 0
    def step(self):
        bf = self.bufs[self.i % 2]
        self.i += 1
        c = self.ctx
        if bf["used"]:
            self.stream.wait_event(bf["ev_match"])
        bf["used"] = True
        c.check(L.svgpu_orb_extract_batch_device(c.handle, C.c_void_p(self.frames.data_ptr()), self.b, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                                 C.c_void_p(bf["kps"].data_ptr()), C.c_void_p(bf["desc"].data_ptr()), self.cap, C.c_void_p(bf["counts"].data_ptr()), None), "extract")
        bf["ev_ext"].record(self.stream)
        self.stream_b.wait_event(bf["ev_ext"])
        c.check(L.svgpu_match_consecutive_batch_device(c.handle, self.b, C.c_void_p(bf["desc"].data_ptr()), C.c_void_p(bf["kps"].data_ptr()), C.c_void_p(bf["counts"].data_ptr()),
                                                       self.cap, self.nc, None, C.c_float(0.75), 1, C.c_void_p(bf["matched"].data_ptr()), C.c_void_p(bf["nmatch"].data_ptr()),
                                                       C.c_void_p(self.stream_b.cuda_stream)), "match")
        bf["ev_match"].record(self.stream_b)

def run(chains, steps=100):
    for _ in range(5):
        for ch in chains: ch.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ch in chains: ch.step()
    for ch in chains: ch.ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(ch.b for ch in chains) * steps / dt

one = [Chain(frames_np)]
print(f"one chain of {B}: {run(one):.0f} frames/s", flush=True)
del one
two = [Chain(frames_np[:B // 2]), Chain(frames_np[B // 2:])]
print(f"two chains of {B // 2}: {run(two):.0f} frames/s", flush=True)
del two
two = [Chain(frames_np), Chain(frames_np)]
print(f"two chains of {B}: {run(two, 50):.0f} frames/s", flush=True)
