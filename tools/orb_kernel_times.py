#!/usr/bin/env python3
"""Per-kernel time of the 64-frame ORB extraction step (HIP events through svgpu_profile_*), for kernel experiments.
usage: python tools/orb_kernel_times.py [kernel ...]   (default: all front-end kernels)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stella_vslam_amd import feature, synthetic
from stella_vslam_amd._lib import lib

W, H, B = int(os.environ.get("ORB_W", "640")), int(os.environ.get("ORB_H", "480")), int(os.environ.get("ORB_B", "64"))
ctx = feature.Context(0)
L = lib()
p = feature.orb_params()
NL = p.num_levels_
ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(p.scale_factor_), NL, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(800)), "cfg")
cap = L.svgpu_orb_max_keypoints(ctx.handle)
frames = torch.from_numpy(synthetic.frame_sequence(B, W, H, seed=0x5EED)).cuda()
kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
counts = torch.zeros(B * (1 + NL), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
def step():
    ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                               C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None), "extract")
for _ in range(3):
    step()
ctx.synchronize()
names = sys.argv[1:] or ["k_resize", "k_blur", "k_fast", "k_select", "k_describe"]
out = {}
for n in names:
    L.svgpu_profile_select(ctx.handle, n.encode())
    for _ in range(10):
        step()
    ms, cnt = C.c_double(), C.c_longlong()
    L.svgpu_profile_read(ctx.handle, C.byref(ms), C.byref(cnt))
    out[n] = round(ms.value / max(cnt.value, 1) * 1000, 1)
L.svgpu_profile_select(ctx.handle, None)
print("us per launch:", out, "kp/frame", counts.view(B, 1 + NL)[:, 0].float().mean().item())
