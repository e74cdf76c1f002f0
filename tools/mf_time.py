"""Mean duration of the matcher kernels over a resident batch (HIP events of svgpu_profile_select), for A/B builds of the library."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from stella_vslam_amd import synthetic  # noqa: E402
from stella_vslam_amd._lib import lib  # noqa: E402
from stella_vslam_amd.pipeline import BatchExtractor  # noqa: E402

W, H, B = 640, 480, 256
ex = BatchExtractor(W, H, B)
ex.upload(synthetic.frame_sequence(B, W, H, seed=0x5EED))
ex.extract()
ex.ctx.synchronize()
L = lib()
matched = torch.zeros(B * ex.cap, dtype=torch.int32, device="cuda")
nmatch = torch.zeros(B, dtype=torch.int32, device="cuda")


def match():
    ex.ctx.check(L.svgpu_match_consecutive_batch_device(ex.ctx.handle, B, C.c_void_p(ex.desc.data_ptr()), C.c_void_p(ex.kps.data_ptr()), C.c_void_p(ex.counts.data_ptr()),
                                                        ex.cap, ex.nc, None, C.c_float(0.9), 1, C.c_void_p(matched.data_ptr()), C.c_void_p(nmatch.data_ptr()),
                                                        C.c_void_p(ex.ctx.stream)), "match")
    ex.ctx.synchronize()


for _ in range(3):
    match()
for name in (sys.argv[1:] or ["k_bf_binsort", "k_bf_topk", "k_bf_replay"]):
    L.svgpu_profile_select(ex.ctx.handle, name.encode())
    for _ in range(10):
        match()
    ms, n = C.c_double(), C.c_longlong()
    L.svgpu_profile_read(ex.ctx.handle, C.byref(ms), C.byref(n))
    print(f"{name:14s} {ms.value / max(n.value, 1) * 1e3:8.1f} us per launch ({n.value} launches)  matches/pair {nmatch.float().mean().item():.1f}", flush=True)
L.svgpu_profile_select(ex.ctx.handle, None)
