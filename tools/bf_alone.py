#!/usr/bin/env python3
"""The brute-force matcher of the headline leg ALONE (no extraction stream beside it): one extraction of 256 frames, then `n` launches of
svgpu_match_consecutive_batch_device on its outputs -- the workload for a rocprofv3 kernel trace / PMC pass of k_bf_binsort / k_bf_mfma /
k_bf_replay without the co-running front end.  usage: tools/bf_alone.py [n]"""
import ctypes as C, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch
from stella_vslam_amd import feature, synthetic
from stella_vslam_amd._lib import lib
B, W, H = 256, 640, 480
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L = lib()
ctx = feature.Context()
prm = feature.orb_params()
NL = prm.num_levels_
ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(prm.scale_factor_), NL, prm.ini_fast_thr_, prm.min_fast_thr_, C.c_uint(800)), "configure")
cap, nc = L.svgpu_orb_max_keypoints(ctx.handle), 1 + NL
frames = torch.from_numpy(np.ascontiguousarray(synthetic.frame_sequence(B, W, H, seed=0x5EED))).cuda()
kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
counts = torch.zeros(B * nc, dtype=torch.int32, device="cuda")
matched = torch.zeros(B * cap, dtype=torch.int32, device="cuda")
nmatch = torch.zeros(B, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None, C.c_size_t(0), 0, C.c_void_p(kps.data_ptr()),
                                           C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None), "extract")
ctx.synchronize()
for _ in range(n):
    ctx.check(L.svgpu_match_consecutive_batch_device(ctx.handle, B, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()), C.c_void_p(counts.data_ptr()), cap, nc, None,
                                                     C.c_float(0.8), 1, C.c_void_p(matched.data_ptr()), C.c_void_p(nmatch.data_ptr()), None), "match")
ctx.synchronize()
print("matches per pair", nmatch.float().mean().item())
# per-kernel times of the matcher alone (HIP events through svgpu_profile_*)
for name in ("k_bf_binsort", "k_bf_topk", "k_bf_replay"):
    L.svgpu_profile_select(ctx.handle, name.encode())
    for _ in range(n):
        ctx.check(L.svgpu_match_consecutive_batch_device(ctx.handle, B, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()), C.c_void_p(counts.data_ptr()), cap, nc, None,
                                                         C.c_float(0.8), 1, C.c_void_p(matched.data_ptr()), C.c_void_p(nmatch.data_ptr()), None), "match")
    ms, cnt = C.c_double(), C.c_longlong()
    L.svgpu_profile_read(ctx.handle, C.byref(ms), C.byref(cnt))
    print(name, "us per launch of %d pairs:" % B, round(ms.value / max(cnt.value, 1) * 1000, 1))
L.svgpu_profile_select(ctx.handle, None)
