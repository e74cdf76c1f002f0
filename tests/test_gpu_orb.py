"""GPU parity: HIP ORB front end (through the C ABI) vs the CPU oracle -- bit-exact keypoints and descriptors."""
import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from stella_vslam_amd import feature
    return feature


def _assert_same(kg, dg, ko, do):
    assert len(kg) == len(ko)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(kg[f], ko[f]), f
    assert np.array_equal(dg, do)


@pytest.mark.parametrize("bands", [False, True])
@pytest.mark.parametrize("w,h,seed", [(640, 480, 0x5EED), (752, 480, 7), (1241, 376, 8), (320, 240, 9), (203, 157, 10), (1920, 1080, 11)])
def test_extract_bit_exact(F, w, h, seed, bands, monkeypatch):
    # one frame takes the per-keypoint describe kernel by default (latency); SVGPU_DESCRIBE_BANDS forces the batch kernel k_describe_bands onto it
    # (1920x1080: its bands do not fit LDS, the per-keypoint kernel serves it either way)
    if bands:
        monkeypatch.setenv("SVGPU_DESCRIBE_BANDS", "1")
    else:
        monkeypatch.delenv("SVGPU_DESCRIBE_BANDS", raising=False)
    img = S.frame(w, h, seed)
    ext = F.orb_extractor(F.orb_params())
    kg, dg = ext.extract(img)
    ko, do, counts, pyr = O.orb_extract(img, want_pyramid=True)
    assert len(ko) > 50
    # intermediate stages first (sharper failure messages)
    for l, (a, b) in enumerate(zip(ext.image_pyramid_, pyr)):
        assert np.array_equal(a, b), f"pyramid level {l}"
    for l, a in enumerate(ext.blurred_pyramid()):
        assert np.array_equal(a, O.gaussian_blur7(pyr[l])), f"blurred level {l}"
    assert np.array_equal(ext.level_counts_, counts)
    _assert_same(kg, dg, ko, do)


def test_extract_params_and_strided_input(F):
    big = S.frame(700, 500, 3)
    view = big[10:490, 30:670]  # non-contiguous rows (stride 700)
    p = F.orb_params("t", 1.3, 6, 25, 9)
    ext = F.orb_extractor(p, min_area=1000)
    kg, dg = ext.extract(view)
    ko, do, _ = O.orb_extract(np.ascontiguousarray(view), scale_factor=1.3, num_levels=6, ini_thr=25, min_thr=9, min_area=1000)
    assert len(ko) > 50
    _assert_same(kg, dg, ko, do)


def test_extract_masks(F):
    img = S.frame(960, 480, 21)
    h, w = img.shape
    mask = np.ones((h, w), np.uint8)
    mask[0:h // 4] = 0
    mask[3 * h // 4:h - 1] = 0
    ext = F.orb_extractor(F.orb_params(), min_area=1000)
    kg, dg = ext.extract(img, mask)
    ko, do, _ = O.orb_extract(img, mask=mask, min_area=1000)
    _assert_same(kg, dg, ko, do)
    assert (kg["y"] >= h // 4).all() and (kg["y"] <= 3 * h // 4).all()
    # rectangle mask (orb_extractor.cc:138-151)
    ext2 = F.orb_extractor(F.orb_params(), min_area=1000, mask_rects=[[0.0, 0.25, 0.0, 1.0], [0.75, 1.0, 0.0, 1.0]])
    kg2, dg2 = ext2.extract(img)
    ko2, do2, _ = O.orb_extract(img, mask=ext2._rectangle_mask(w, h), min_area=1000)
    _assert_same(kg2, dg2, ko2, do2)
    assert len(kg2) > 0 and (kg2["x"] > 0.25 * w - 1).all() and (kg2["x"] < 0.75 * w + 1).all()


def test_toy_rectangle_and_empty(F):
    img = np.full((600, 600), 255, np.uint8)
    img[300:, 300:] = 0
    ext = F.orb_extractor(F.orb_params(), min_area=1000)
    kg, dg = ext.extract(img)
    ko, do, _ = O.orb_extract(img, min_area=1000)
    _assert_same(kg, dg, ko, do)
    sf = ext.orb_params_.scale_factors_
    assert len(kg) > 0
    assert (np.abs(kg["x"] - 300) <= 2.0 * sf[kg["octave"]]).all() and (np.abs(kg["y"] - 300) <= 2.0 * sf[kg["octave"]]).all()
    k0, d0 = ext.extract(np.full((600, 600), 90, np.uint8))
    assert len(k0) == 0 and d0.shape == (0, 32)
    # reuse after an empty frame must still be exact (selection keys are reset in-kernel)
    kg, dg = ext.extract(img)
    _assert_same(kg, dg, ko, do)


def test_repeatable_and_two_contexts(F):
    img = S.frame(640, 480, 77)
    e1 = F.orb_extractor(F.orb_params())
    e2 = F.orb_extractor(F.orb_params())  # second context, as the stereo pair does (system.cc:427-434)
    a = e1.extract(img)
    b = e2.extract(img)
    c = e1.extract(img)
    _assert_same(a[0], a[1], b[0], b[1])
    _assert_same(a[0], a[1], c[0], c[1])


def test_batch_device_unaligned_images(F):
    """Caller-owned device images whose base / pitch are not 4-byte aligned take the gather variants of the blur and
    of the patch staging (k_blur_gather, stage_patch byte path): same bits as the aligned streaming kernels."""
    import ctypes as C
    import torch
    from stella_vslam_amd._lib import lib
    W, H, B = 322, 243, 3
    imgs = [S.frame(W, H, 20 + i) for i in range(B)]
    L = lib()
    p = F.orb_params()
    NL = p.num_levels_
    results = []
    for base_off, pitch in ((0, 324), (1, 323)):
        ctx = F.Context(0)
        ctx.check(L.svgpu_orb_configure(ctx.handle, W, H, B, C.c_float(p.scale_factor_), NL, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(800)), "cfg")
        cap = L.svgpu_orb_max_keypoints(ctx.handle)
        frame_stride = pitch * H + 5 if base_off else pitch * H
        host = np.zeros(base_off + B * frame_stride + 64, np.uint8)
        for i, im in enumerate(imgs):
            v = host[base_off + i * frame_stride: base_off + i * frame_stride + pitch * H].reshape(H, pitch)
            v[:, :W] = im
        dev = torch.from_numpy(host).cuda()
        kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        counts = torch.zeros(B * (1 + NL), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(dev.data_ptr() + base_off), B, C.c_size_t(frame_stride), pitch, None,
                                                   C.c_size_t(0), 0, C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap,
                                                   C.c_void_p(counts.data_ptr()), None), "extract")
        ctx.synchronize()
        results.append((kps.cpu().numpy().reshape(B, cap, 28), desc.cpu().numpy().reshape(B, cap, 32), counts.cpu().numpy().reshape(B, 1 + NL)))
    (ka, da, ca), (ku, du, cu) = results
    assert np.array_equal(ca, cu)
    for i in range(B):
        n = ca[i, 0]
        ko, do, cnt = O.orb_extract(imgs[i])
        assert n == len(ko) and np.array_equal(cnt, ca[i, 1:])
        assert np.array_equal(ka[i, :n], ku[i, :n]) and np.array_equal(da[i, :n], du[i, :n])
        assert np.array_equal(da[i, :n], do)


def test_noise_image_and_tiny_image(F):
    """Uniform noise (a corner candidate at most pixels: the FAST queue, the selection grid and the descriptor stage all run
    at their densest) and an image barely larger than the 19-px borders (single partial FAST cell, levels without cells)."""
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    ext = F.orb_extractor(F.orb_params())
    kg, dg = ext.extract(noise)
    ko, do, _ = O.orb_extract(noise)
    assert len(ko) > 300
    _assert_same(kg, dg, ko, do)
    tiny = S.frame(64, 52, 3)
    ext2 = F.orb_extractor(F.orb_params("tiny", 1.2, 4, 20, 7), min_area=100)
    kg, dg = ext2.extract(tiny)
    ko, do, _ = O.orb_extract(tiny, num_levels=4, min_area=100)
    _assert_same(kg, dg, ko, do)


def test_two_extractors_on_two_threads(F):
    """The stereo front end runs two extractor instances concurrently on two threads (system.cc:427-434): two contexts, two
    streams, the library calls release the GIL -- results stay bit-identical to the single-threaded ones."""
    import threading
    imgs = [S.frame(640, 480, 30), S.frame(640, 480, 31)]
    exts = [F.orb_extractor(F.orb_params()), F.orb_extractor(F.orb_params())]
    ref = [exts[i].extract(imgs[i]) for i in range(2)]
    out = [None, None]

    def work(i):
        for _ in range(10):
            out[i] = exts[i].extract(imgs[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        _assert_same(out[i][0], out[i][1], ref[i][0], ref[i][1])


def _batch_extract(F, frames, stride_pad=0, cap=None, legacy=False):
    """svgpu_orb_extract_batch_device on device-resident frames with row stride w + stride_pad; returns per-frame records, descriptors, counts."""
    import ctypes as C
    import os
    import torch
    from stella_vslam_amd._lib import lib
    B, h, w = frames.shape
    old = os.environ.pop("SVGPU_DESCRIBE_LEGACY", None)
    old_b = os.environ.pop("SVGPU_DESCRIBE_BANDS", None)
    if not legacy:
        os.environ["SVGPU_DESCRIBE_BANDS"] = "1"   # read at every launch: k_describe_bands even for a batch of three (the library would take k_describe: latency)
    if legacy:
        os.environ["SVGPU_DESCRIBE_LEGACY"] = "1"  # read by svgpu_orb_configure: the per-keypoint kernel k_describe instead of k_describe_bands
    try:
        ctx = F.Context(0)
        L, p = lib(), F.orb_params()
        ctx.check(L.svgpu_orb_configure(ctx.handle, w, h, B, C.c_float(p.scale_factor_), p.num_levels_, p.ini_fast_thr_, p.min_fast_thr_, C.c_uint(800)), "cfg")
    finally:
        os.environ.pop("SVGPU_DESCRIBE_LEGACY", None)
        if old is not None:
            os.environ["SVGPU_DESCRIBE_LEGACY"] = old
    full = L.svgpu_orb_max_keypoints(ctx.handle)
    cap = cap or full
    stride = w + stride_pad
    buf = np.zeros((B, h, stride), np.uint8)
    buf[:, :, :w] = frames
    img = torch.from_numpy(buf.reshape(-1)).cuda()
    off = 3 if stride_pad else 0  # and a base address that is not a multiple of 4
    if off:
        img = torch.cat([torch.zeros(off, dtype=torch.uint8, device="cuda"), img])
    kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    desc = torch.full((B * cap * 32,), 0xAB, dtype=torch.uint8, device="cuda")
    counts = torch.zeros(B * (1 + p.num_levels_), dtype=torch.int32, device="cuda")
    ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(img.data_ptr() + off), B, C.c_size_t(h * stride), stride, None, C.c_size_t(0), 0,
                                               C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None), "extract")
    ctx.synchronize()
    os.environ.pop("SVGPU_DESCRIBE_BANDS", None)
    if old_b is not None:
        os.environ["SVGPU_DESCRIBE_BANDS"] = old_b
    return (kps.cpu().numpy().view(O.KEYPOINT_DTYPE).reshape(B, cap), desc.cpu().numpy().reshape(B, cap, 32),
            counts.cpu().numpy().reshape(B, 1 + p.num_levels_), full)


@pytest.mark.parametrize("w,h,pad", [(640, 480, 0), (203, 157, 5), (1241, 376, 0), (331, 250, 1), (131, 100, 3)])  # (131 px: up to ten rows per staging instruction)
def test_describe_bands_equal_per_keypoint_kernel(F, w, h, pad):
    """k_describe_bands (LDS-resident bands) and k_describe (per-keypoint patches) are two routes to the same bytes: orientation, records and
    descriptors of a batch, with an unaligned strided image, and with a capacity that cuts a band's run of keypoints short."""
    frames = S.frame_sequence(3, w, h, seed=w + h)
    kb, db, cb, full = _batch_extract(F, frames, pad)
    kl, dl, cl, _ = _batch_extract(F, frames, pad, legacy=True)
    assert np.array_equal(cb, cl) and cb[:, 0].min() > 20
    for b in range(3):
        n = cb[b, 0]
        assert np.array_equal(kb[b, :n], kl[b, :n]) and np.array_equal(db[b, :n], dl[b, :n])
        ko, do, _ = O.orb_extract(frames[b])
        assert np.array_equal(db[b, :n], do) and np.array_equal(kb[b, :n]["angle"], ko["angle"])
    cap = int(cb[:, 0].min()) * 2 // 3  # fewer slots than keypoints: the first `cap` of the emission order, nothing written behind them
    kc, dc, cc, _ = _batch_extract(F, frames, pad, cap=cap)
    assert np.array_equal(cc, cb)
    for b in range(3):
        assert np.array_equal(kc[b], kb[b, :cap]) and np.array_equal(dc[b], db[b, :cap])


def test_describe_bands_equal_per_keypoint_kernel_on_random_geometries(F):
    """The band table (level pitches, rows per staging instruction, grid rows per band, wide-row staging) depends on every extractor parameter: sixteen random
    geometries -- 90..1400 x 80..700 px, scale factor 1.1..2.0, 1..9 levels, FAST thresholds, min_area, batch 1..3, row stride and base address off
    alignment -- give byte-identical records, descriptors and counts from k_describe_bands and k_describe (160 such draws were run once by hand: no mismatch)."""
    import ctypes as C
    import os
    import torch
    from stella_vslam_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(2026)
    saved = {k: os.environ.pop(k, None) for k in ("SVGPU_DESCRIBE_LEGACY", "SVGPU_DESCRIBE_BANDS")}
    try:
        total = 0
        for _ in range(16):
            w, h = int(rng.integers(90, 1400)), int(rng.integers(80, 700))
            sf, nl = float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0])), int(rng.integers(1, 10))
            ini = int(rng.integers(8, 40))
            mn, area = int(rng.integers(3, ini)), int(rng.choice([200, 800, 2000]))
            B, pad, off = int(rng.integers(1, 4)), int(rng.integers(0, 9)), int(rng.integers(0, 4))
            frames = S.frame_sequence(B, w, h, seed=int(rng.integers(1 << 12)))
            outs = []
            for env in ("SVGPU_DESCRIBE_BANDS", "SVGPU_DESCRIBE_LEGACY"):
                os.environ.pop("SVGPU_DESCRIBE_BANDS", None)
                os.environ.pop("SVGPU_DESCRIBE_LEGACY", None)
                os.environ[env] = "1"
                ctx = F.Context(0)
                ctx.check(L.svgpu_orb_configure(ctx.handle, w, h, B, C.c_float(sf), nl, ini, mn, C.c_uint(area)), "cfg")
                cap, stride = max(L.svgpu_orb_max_keypoints(ctx.handle), 1), w + pad
                buf = np.zeros((B, h, stride), np.uint8)
                buf[:, :, :w] = frames
                img = torch.cat([torch.zeros(off, dtype=torch.uint8, device="cuda"), torch.from_numpy(buf.reshape(-1)).cuda()])
                kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
                desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
                counts = torch.zeros(B * (1 + nl), dtype=torch.int32, device="cuda")
                ctx.check(L.svgpu_orb_extract_batch_device(ctx.handle, C.c_void_p(img.data_ptr() + off), B, C.c_size_t(h * stride), stride, None, C.c_size_t(0), 0,
                                                           C.c_void_p(kps.data_ptr()), C.c_void_p(desc.data_ptr()), cap, C.c_void_p(counts.data_ptr()), None), "extract")
                ctx.synchronize()
                outs.append((kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()))
            for a, b in zip(*outs):
                assert np.array_equal(a, b), (w, h, sf, nl, ini, mn, area, B, pad, off)
            total += int(outs[0][2].reshape(B, -1)[:, 0].sum())
        assert total > 5000
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_describe_bands_equal_per_keypoint_kernel_at_bench_size(F):
    """The bench workload itself (640x480 synthetic sequence, one launch of 128 frames -- the library routes batches of this size to k_describe_bands on its own):
    every record, descriptor and count equal to the per-keypoint kernel's, ~310 k keypoints."""
    frames = S.frame_sequence(128, 640, 480, seed=0x5EED)
    kb, db, cb, _ = _batch_extract(F, frames)
    kl, dl, cl, _ = _batch_extract(F, frames, legacy=True)
    assert np.array_equal(cb, cl) and cb[:, 0].sum() > 300000
    assert np.array_equal(kb, kl) and np.array_equal(db, dl)
