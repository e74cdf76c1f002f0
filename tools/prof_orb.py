import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stella_vslam_amd import feature, synthetic
from stella_vslam_amd._lib import lib
W, H, B = 640, 480, 64
ctx = feature.Context(0); L = lib(); p = feature.orb_params(); NL = p.num_levels_
ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(p.scale_factor_), NL, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(800)), "cfg")
cap = L.svgpu_orb_max_keypoints(ctx.handle)
frames = torch.from_numpy(synthetic.frame_sequence(B, W, H, seed=0x5EED)).cuda()
kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
counts = torch.zeros(B * (1 + NL), dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
for _ in range(3):
    ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                               C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None), "extract")
ctx.synchronize()
