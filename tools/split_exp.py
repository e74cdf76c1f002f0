"""Experiment: the bench.py pipeline with the extraction of a 256-frame batch split over S contexts / streams (kernels of different
sub-batches overlap).  python tools/split_exp.py [S ...]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from stella_vslam_amd import feature, synthetic  # noqa: E402
from stella_vslam_amd._lib import lib  # noqa: E402

W, H, B, STEPS = 640, 480, 256, 20
import os
NBUF = int(os.environ.get("EXP_NBUF", "2"))
L = lib()
params = feature.orb_params()
NL = params.num_levels_
frames_np = synthetic.frame_sequence(B, W, H, seed=0x5EED)


def run(S, prio_first=True):
    Bs = B // S
    import os
    ctxs = [feature.Context(0, priority=int(os.environ.get("EXP_PRIO", "1"))) for _ in range(S)]
    for c in ctxs:
        c.check(L.svgpu_orb_configure(c.handle, W, H, Bs, C.c_float(params.scale_factor_), NL, params.ini_fast_thr_, params.min_fast_thr_, C.c_uint(800)), "cfg")
    cap = L.svgpu_orb_max_keypoints(ctxs[0].handle)
    nc = 1 + NL
    streams = [torch.cuda.ExternalStream(c.stream) for c in ctxs]
    stream_b = torch.cuda.Stream(priority=int(os.environ.get("EXP_PRIO_B", "0")))
    frames = torch.from_numpy(frames_np).cuda()
    bufs = [dict(kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                 counts=torch.zeros(B * nc, dtype=torch.int32, device="cuda"), matched=torch.zeros(B * cap, dtype=torch.int32, device="cuda"),
                 nmatch=torch.zeros(B, dtype=torch.int32, device="cuda"), ev_ext=[torch.cuda.Event() for _ in range(S)], ev_match=torch.cuda.Event(), used=False)
            for _ in range(NBUF)]
    torch.cuda.synchronize()
    st = {"i": 0}

    def step():
        bf = bufs[st["i"] % NBUF]
        st["i"] += 1
        for s in range(S):
            if bf["used"]:
                streams[s].wait_event(bf["ev_match"])
            ctxs[s].check(L.svgpu_orb_extract_batch_device(ctxs[s].handle, C.c_void_p(frames.data_ptr() + s * Bs * W * H), Bs, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                                           C.c_void_p(bf["kps"].data_ptr() + s * Bs * cap * 28), C.c_void_p(bf["desc"].data_ptr() + s * Bs * cap * 32), cap,
                                                           C.c_void_p(bf["counts"].data_ptr() + s * Bs * nc * 4), None), "extract")
            bf["ev_ext"][s].record(streams[s])
        bf["used"] = True
        for s in range(S):
            stream_b.wait_event(bf["ev_ext"][s])
        ctxs[0].check(L.svgpu_match_consecutive_batch_device(ctxs[0].handle, B, C.c_void_p(bf["desc"].data_ptr()), C.c_void_p(bf["kps"].data_ptr()), C.c_void_p(bf["counts"].data_ptr()),
                                                             cap, nc, None, C.c_float(0.9), 1, C.c_void_p(bf["matched"].data_ptr()), C.c_void_p(bf["nmatch"].data_ptr()),
                                                             C.c_void_p(stream_b.cuda_stream)), "match")
        bf["ev_match"].record(stream_b)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(STEPS):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, B * STEPS / dt)
    nm = bufs[0]["nmatch"].float().mean().item()
    nk = bufs[0]["counts"].view(B, nc)[:, 0].float().mean().item()
    print(f"S={S}: {best:,.0f} frames/s  kp {nk:.1f} matches {nm:.1f}", flush=True)
    del bufs, frames
    for c in ctxs:
        c.close()
    torch.cuda.empty_cache()


for S in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    run(S)
