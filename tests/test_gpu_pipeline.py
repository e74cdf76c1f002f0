"""GPU parity of the DEVICE-RESIDENT pipeline bench.py times: a batch extracted with svgpu_orb_extract_batch_device feeds the batched
brute-force matcher without leaving HBM -- once through the ring entry point (frame t+1 against frame t, no slot copies) and once
through the two-pointer entry point on explicitly shifted copies.  Both must equal the oracle's robust::brute_force_match run on the
oracle's own extraction of every frame (keypoints / descriptors bit-exact, match lists identical)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("priority", [0, 1])
def test_batch_extract_then_ring_match_equals_oracle(priority):
    import torch
    from stella_vslam_amd import feature as F
    from stella_vslam_amd._lib import lib
    L = lib()
    B, W, H = 5, 640, 480
    frames_np = S.frame_sequence(B, W, H, seed=0x5EED + 11)
    ctx = F.Context(0, priority=priority)
    p = F.orb_params()
    NL = p.num_levels_
    ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(p.scale_factor_), NL, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(800)), "cfg")
    cap = L.svgpu_orb_max_keypoints(ctx.handle)
    nc = 1 + NL
    frames = torch.from_numpy(frames_np).cuda()
    kps = torch.zeros((B + 1) * cap * 28, dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B + 1) * cap * 32, dtype=torch.uint8, device="cuda")
    counts = torch.zeros((B + 1) * nc, dtype=torch.int32, device="cuda")
    matched = torch.full((B * cap,), -7, dtype=torch.int32, device="cuda")
    nmatch = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):  # twice: the second call reuses every workspace (selection keys reset in-kernel, blur fork / join events)
        ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                                   C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None),
                  "extract")
        ctx.check(L.svgpu_match_consecutive_batch_device(ctx.handle, B, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()),
                                                         C.c_void_p(counts.data_ptr()), cap, nc, None, C.c_float(0.8), 1,
                                                         C.c_void_p(matched.data_ptr()), C.c_void_p(nmatch.data_ptr()), None), "ring match")
    ctx.synchronize()
    ring_matched, ring_n = matched.cpu().numpy().reshape(B, cap).copy(), nmatch.cpu().numpy().copy()
    # the two-pointer form on a copy whose slot B repeats slot 0
    kps[B * cap * 28:].copy_(kps[:cap * 28])
    desc[B * cap * 32:].copy_(desc[:cap * 32])
    counts[B * nc:].copy_(counts[:nc])
    torch.cuda.synchronize()
    matched.fill_(-7)
    ctx.check(L.svgpu_match_bruteforce_batch_device(ctx.handle, B, C.c_void_p(desc.data_ptr() + cap * 32), C.c_void_p(kps.data_ptr() + cap * 28),
                                                    C.c_void_p(counts.data_ptr() + nc * 4), cap, C.c_void_p(desc.data_ptr()), C.c_void_p(kps.data_ptr()),
                                                    C.c_void_p(counts.data_ptr()), cap, nc, None, C.c_float(0.8), 1, C.c_void_p(matched.data_ptr()),
                                                    C.c_void_p(nmatch.data_ptr()), None), "pair match")
    ctx.synchronize()
    two_matched, two_n = matched.cpu().numpy().reshape(B, cap), nmatch.cpu().numpy()
    k_all = kps.cpu().numpy()[: B * cap * 28].view(O.KEYPOINT_DTYPE).reshape(B, cap)
    d_all = desc.cpu().numpy()[: B * cap * 32].reshape(B, cap, 32)
    cnt = counts.cpu().numpy()[: B * nc].reshape(B, nc)
    ref = [O.orb_extract(frames_np[t]) for t in range(B)]
    for t in range(B):
        ko, do, lv = ref[t]
        n = cnt[t, 0]
        assert n == len(ko) and np.array_equal(cnt[t, 1:], lv)
        assert np.array_equal(k_all[t, :n], ko) and np.array_equal(d_all[t, :n], do)
    # the packed-angle forms: the extractor leaves the angles as a float array beside the records (the same values), the ring matcher reads that
    angles = torch.full((B * cap,), -1.0, dtype=torch.float32, device="cuda")
    kps2, desc2, counts2 = torch.zeros_like(kps), torch.zeros_like(desc), torch.zeros_like(counts)
    matched.fill_(-7)
    ctx.check(L.svgpu_orb_extract_batch_device_angles(ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_size_t(W * H), W, None, C.c_size_t(0), 0,
                                                      C.c_void_p(kps2.data_ptr()), C.c_void_p(desc2.data_ptr()), cap, C.c_void_p(counts2.data_ptr()),
                                                      C.c_void_p(angles.data_ptr()), None), "extract + angles")
    ctx.check(L.svgpu_match_consecutive_batch_device_angles(ctx.handle, B, C.c_void_p(desc2.data_ptr()), C.c_void_p(kps2.data_ptr()), C.c_void_p(angles.data_ptr()),
                                                            C.c_void_p(counts2.data_ptr()), cap, nc, None, C.c_float(0.8), 1,
                                                            C.c_void_p(matched.data_ptr()), C.c_void_p(nmatch.data_ptr()), None), "ring match, packed angles")
    ctx.synchronize()
    packed_matched, packed_n = matched.cpu().numpy().reshape(B, cap).copy(), nmatch.cpu().numpy().copy()
    assert torch.equal(kps2[: B * cap * 28], kps[: B * cap * 28]) and torch.equal(desc2[: B * cap * 32], desc[: B * cap * 32])
    ang = angles.cpu().numpy().reshape(B, cap)
    for t in range(B):
        assert np.array_equal(ang[t, :cnt[t, 0]], k_all[t, :cnt[t, 0]]["angle"])
    for t in range(B):
        f, kf = (t + 1) % B, t
        exp = O.brute_force_match(ref[f][1], ref[f][0]["angle"], ref[kf][1], ref[kf][0]["angle"], None, 0.8, True)
        n1 = len(exp)
        assert (exp >= 0).sum() > (600 if f == kf + 1 else 50)
        assert np.array_equal(ring_matched[t, :n1], exp) and ring_n[t] == (exp >= 0).sum(), t
        assert np.array_equal(two_matched[t, :n1], exp) and two_n[t] == ring_n[t], t
        assert np.array_equal(packed_matched[t, :n1], exp) and packed_n[t] == ring_n[t], t


def test_opt_in_paths_keep_parity_in_a_fresh_process():
    """The switches read once per process -- SVGPU_FORK_BLUR (blur on the auxiliary stream beside FAST) and SVGPU_BF_VALU (the
    non-MFMA distance kernel) -- must not change a single bit: re-run the device-resident pipeline test under both."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"SVGPU_FORK_BLUR": "1"}, {"SVGPU_BF_VALU": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_pipeline.py", "-k", "ring_match and 0"],
                           cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (extra, r.stdout[-2000:], r.stderr[-1000:])
        assert "1 passed" in r.stdout, r.stdout[-500:]
