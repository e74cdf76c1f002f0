"""GPU parity: HIP matchers (through the C ABI) vs the CPU oracle -- identical match lists."""
import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    from stella_vslam_amd import match
    return match


@pytest.fixture(scope="module")
def ctx():
    from stella_vslam_amd import feature
    return feature.Context()


def _noisy(rng, base, flips):
    out = base.copy()
    for i in range(len(out)):
        for b in rng.choice(256, rng.integers(0, flips + 1), replace=False):
            out[i, b // 8] ^= 1 << (b % 8)
    return out


def test_hamming_known_answers_and_random(M, ctx):
    a = np.stack([np.full(32, 0b01010101, np.uint8), np.full(32, 0b01010101, np.uint8), np.full(32, 0b01100110, np.uint8)])
    b = np.stack([np.full(32, 0b01010101, np.uint8), np.full(32, 0b10101010, np.uint8), np.full(32, 0b00111100, np.uint8)])
    assert list(M.compute_descriptor_distance_32(ctx, a, b)) == [0, 256, 128]  # reference test/stella_vslam/match/base.cc:11-57
    rng = np.random.default_rng(0)
    d1 = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (333, 32), dtype=np.uint8)
    assert np.array_equal(M.hamming_matrix(ctx, d1, d2), O.hamming_matrix(d1, d2))


@pytest.mark.parametrize("seed,n1,n2,check,ratio", [(0, 150, 130, True, 0.75), (1, 2000, 2100, True, 0.75), (2, 2000, 1900, False, 0.9),
                                                    (3, 37, 300, True, 0.6), (4, 3000, 5, False, 1.0),
                                                    # lowe_ratio < 0.4: cutoff >= 128, the VALU top-K kernel instead of the MFMA one; 0.1: every pair is a candidate
                                                    (5, 900, 800, True, 0.35), (6, 500, 600, False, 0.1),
                                                    # more than 4096 queries: rows beyond the replay's register cache
                                                    (7, 5200, 4700, True, 0.8)])
def test_brute_force_matches_oracle(M, ctx, seed, n1, n2, check, ratio):
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    src = rng.integers(0, max(n1 // 2, 1), n2)  # duplicates on purpose: exercises the greedy bookkeeping
    d2 = _noisy(rng, d1[src], 45)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = ((a1[src] + rng.normal(0, 15, n2)) % 360).astype(np.float32)
    valid2 = (rng.uniform(size=n2) < 0.85).astype(np.uint8)
    pairs, out = M.robust(ratio, check, ctx).brute_force_match(d1, a1, d2, a2, valid2)
    exp = O.brute_force_match(d1, a1, d2, a2, valid2, ratio, check)
    assert (exp >= 0).sum() >= min(n1, n2) // 8
    assert np.array_equal(out, exp)


def test_brute_force_adversarial_prefix_exhaustion(M, ctx):
    """Many queries share the same few good targets: the K-prefix of late queries is fully consumed and
    the exact full-row fallback must kick in."""
    rng = np.random.default_rng(5)
    n1, n2 = 400, 300
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d1[:40] = _noisy(rng, np.repeat(d1[:1], 40, 0), 6)       # a cluster of 40 near-identical frame descriptors
    d2 = _noisy(rng, np.repeat(d1[:1], n2, 0), 10)           # every query likes exactly that cluster
    a = np.zeros(max(n1, n2), np.float32)
    pairs, out = M.robust(1.0, False, ctx).brute_force_match(d1, a[:n1], d2, a[:n2], None)
    exp = O.brute_force_match(d1, a[:n1], d2, a[:n2], None, 1.0, False)
    assert (exp >= 0).sum() >= 30
    assert np.array_equal(out, exp)


def test_brute_force_on_real_descriptors(M, ctx):
    seq = S.frame_sequence(2)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    pairs, out = M.robust(0.75, True, ctx).brute_force_match(d1, k1["angle"], d0, k0["angle"], None)
    exp = O.brute_force_match(d1, k1["angle"], d0, k0["angle"], None, 0.75, True)
    assert (exp >= 0).sum() > 1000
    assert np.array_equal(out, exp)
    assert pairs == [(int(i), int(exp[i])) for i in np.flatnonzero(exp >= 0)]


def test_brute_force_empty(M, ctx):
    d = np.zeros((0, 32), np.uint8)
    a = np.zeros(0, np.float32)
    x = np.zeros((5, 32), np.uint8)
    pairs, out = M.robust(0.75, True, ctx).brute_force_match(d, a, x, np.zeros(5, np.float32))
    assert pairs == [] and len(out) == 0
    pairs, out = M.robust(0.75, True, ctx).brute_force_match(x, np.zeros(5, np.float32), d, a)
    assert pairs == [] and (out == -1).all()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed", [0, 1])
def test_candidate_matcher_matches_oracle(M, ctx, mode, seed):
    """projection-style matching on realistic inputs: targets = keypoints of frame t+1, queries = keypoints of
    frame t 'reprojected' by the known shift, candidates from the reference's grid lookup."""
    seq = S.frame_sequence(2, seed=0x5EED + seed)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    bounds = (0.0, 640.0, 0.0, 480.0)
    off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
    sf = O.scale_tables(1.2, 8)[0]
    rng = np.random.default_rng(seed)
    cand_off, cand_idx = [0], []
    for q in range(len(k0)):
        lvl = int(k0["octave"][q])
        c = O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(k0["x"][q]) - 3.0,
                                    float(k0["y"][q]) - 1.0, 15.0 * float(sf[lvl]), max(0, lvl - 1), min(7, lvl + 1))
        cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    occupied = (rng.uniform(size=len(k1)) < 0.05).astype(np.uint8)
    q_valid = (rng.uniform(size=len(k0)) < 0.9).astype(np.uint8)
    xr_t = np.where(rng.uniform(size=len(k1)) < 0.5, k1["x"] - 20.0, -1.0).astype(np.float32)
    xr_q = (k0["x"] - 3.0 - 20.0 + rng.normal(0, 6, len(k0))).astype(np.float32)
    tol = (15.0 * sf[k0["octave"]]).astype(np.float32)
    kw = dict(t_octave=k1["octave"], q_valid=q_valid, occupied=occupied, q_angle=k0["angle"], t_angle=k1["angle"],
              q_xright=xr_q, t_xright=xr_t, q_xr_tol=tol)
    got, num = M.projection(0.8, True, ctx).match_candidates(d0, d1, cand_off, cand_idx, mode, 100, **kw)
    exp = O.match_candidates(d0, d1, cand_off, cand_idx, check_orientation=True, thr=100, lowe_ratio=0.8, mode=mode, **kw)
    assert (exp >= 0).sum() > 800
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()


@pytest.mark.parametrize("mode,thr,ratio", [(2, 50, 0.7), (3, 50, 0.95), (0, 50, 0.6)])
def test_candidate_matcher_bow_triangulation_fuse_modes(M, ctx, mode, thr, ratio):
    """bow_tree (RATIO), match_for_triangulation (TRIANGULATION) and fuse (BEST_ONLY + per-pair skip mask standing for the
    chi-square / epipolar gates the adaptor evaluates) on bucketed candidates, as FBoW node buckets would give them."""
    seq = S.frame_sequence(2, seed=0x5EED + 11)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    rng = np.random.default_rng(mode)
    # synthetic "vocabulary nodes": bucket both frames by a coarse spatial hash of the (shift-compensated) position
    b0 = ((k0["x"] - 3) // 40).astype(int) * 100 + ((k0["y"] - 1) // 40).astype(int) + 1000 * k0["octave"]
    b1 = (k1["x"] // 40).astype(int) * 100 + (k1["y"] // 40).astype(int) + 1000 * k1["octave"]
    order = np.argsort(b0, kind="stable")  # queries in node order, as the merge-join of the two std::maps visits them
    buckets = {}
    for i, b in enumerate(b1):
        buckets.setdefault(int(b), []).append(i)
    cand_off, cand_idx = [0], []
    for q in order:
        cand_idx += buckets.get(int(b0[q]), [])
        cand_off.append(len(cand_idx))
    skip = (rng.uniform(size=len(cand_idx)) < 0.1).astype(np.uint8)
    qd, qa = d0[order], k0["angle"][order]
    kw = dict(cand_skip=skip, t_octave=k1["octave"], q_angle=qa, t_angle=k1["angle"])
    got, num = M.projection(ratio, True, ctx).match_candidates(qd, d1, cand_off, cand_idx, mode, thr, **kw)
    exp = O.match_candidates(qd, d1, cand_off, cand_idx, check_orientation=True, thr=thr, lowe_ratio=ratio, mode=mode, **kw)
    assert (exp >= 0).sum() > 300
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()


@pytest.mark.parametrize("disp", [12, 31])
def test_stereo_matches_oracle(M, disp):
    """match::stereo::compute on a synthetic rectified pair (right image = scene shifted by `disp` px): sub-pixel x_right
    and depths bit-identical to the oracle, and the recovered disparity is the true one."""
    from stella_vslam_amd import feature
    big = S.frame(640 + 64, 480, 5)
    left, right = np.ascontiguousarray(big[:, 8:648]), np.ascontiguousarray(big[:, 8 + disp:648 + disp])
    el, er = feature.orb_extractor(feature.orb_params()), feature.orb_extractor(feature.orb_params())
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    fxb, tb = 458.654 * 0.11, 0.11
    xr, dp = M.stereo(el, er, kl, kr, dl, dr, fxb, tb).compute()
    pl, pr = el.image_pyramid_, er.image_pyramid_
    xo, do = O.stereo_match(kl, dl, kr, dr, pl, pr, fxb, tb)
    ok = xo >= 0
    assert ok.sum() > 800
    assert np.array_equal(xr, xo) and np.array_equal(dp, do)
    assert abs(np.median((kl["x"] - xr)[ok]) - disp) < 0.05


@pytest.mark.parametrize("seed,check", [(0, True), (1, False)])
def test_area_matcher_matches_oracle(M, ctx, seed, check):
    """area::match_in_consistent_area (the initialiser's matcher): level-0 keypoints of frame t against the keypoints of
    frame t+1 inside a window around the previous match; a closer later query takes a target from its holder."""
    seq = S.frame_sequence(2, seed=0x5EED + 11 * seed)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    bounds = (0.0, 640.0, 0.0, 480.0)
    off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
    cand_off, cand_idx = [0], []
    for q in range(len(k0)):
        if k0["octave"][q] == 0:  # match/area.cc:21-24
            c = O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(k0["x"][q]), float(k0["y"][q]), 60.0, 0, 0)
            cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    got, num = M.area(0.9, check, ctx).match_in_consistent_area(d0, k0["angle"], d1, k1["angle"], cand_off, cand_idx)
    exp = O.match_candidates(d0, d1, cand_off, cand_idx, q_angle=k0["angle"], t_angle=k1["angle"], check_orientation=check, thr=50,
                             lowe_ratio=0.9, mode=O.MODE_AREA)
    assert (exp >= 0).sum() > 100
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()


def test_area_matcher_contested_targets(M, ctx):
    """Many queries compete for few targets: the take-over bookkeeping (matched_dists_in_frm_2) decides."""
    rng = np.random.default_rng(3)
    n1, n2 = 500, 120
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    src = rng.integers(0, n2, n1)
    d1 = np.stack([_noisy(rng, d2[s:s + 1], int(rng.integers(0, 40)))[0] for s in src])
    a = np.zeros(n1, np.float32)
    cand_off = np.arange(n1 + 1, dtype=np.int32) * n2
    cand_idx = np.tile(np.arange(n2, dtype=np.int32), n1)
    got, num = M.area(0.95, False, ctx).match_in_consistent_area(d1, a, d2, a[:n2], cand_off, cand_idx)
    exp = O.match_candidates(d1, d2, cand_off, cand_idx, thr=50, lowe_ratio=0.95, mode=O.MODE_AREA)
    assert 30 < (exp >= 0).sum() <= n2
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()


@pytest.mark.parametrize("mode,seed", [(0, 0), (1, 1), (2, 2)])
def test_match_in_cells_builds_the_reference_candidate_lists(M, ctx, mode, seed):
    """Candidate lists built on the device (assign_keypoints_to_grid + get_keypoints_in_cell, data/common.cc:83-190) give the
    same matches as the oracle run on the CSR that the oracle's own grid functions produce (same order, same ties)."""
    seq = S.frame_sequence(2, seed=0x5EED + 3 * seed)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    bounds = (0.0, 640.0, 0.0, 480.0)
    off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
    sf = O.scale_tables(1.2, 8)[0]
    rng = np.random.default_rng(seed)
    q_xy = np.stack([k0["x"] - 3.0 + rng.normal(0, 2, len(k0)), k0["y"] - 1.0 + rng.normal(0, 2, len(k0))], 1).astype(np.float32)
    q_xy[::50] += 700.0                                        # some reference points far outside the image
    q_margin = (15.0 * sf[k0["octave"]]).astype(np.float32)
    q_lo = np.maximum(0, k0["octave"] - 1).astype(np.int32)
    q_hi = np.minimum(7, k0["octave"] + 1).astype(np.int32)
    q_lo[::7] = -1                                             # unbounded below / above for some queries
    q_hi[::11] = -1
    cand_off, cand_idx = [0], []
    for q in range(len(k0)):
        c = O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(q_xy[q, 0]), float(q_xy[q, 1]),
                                    float(q_margin[q]), int(q_lo[q]), int(q_hi[q]))
        cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    occupied = (rng.uniform(size=len(k1)) < 0.05).astype(np.uint8)
    q_valid = (rng.uniform(size=len(k0)) < 0.9).astype(np.uint8)
    kw = dict(q_valid=q_valid, occupied=occupied, q_angle=k0["angle"], t_angle=k1["angle"])
    t_xy = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    got, num = M.projection(0.8, True, ctx).match_in_cells(d0, q_xy, q_margin, d1, t_xy, k1["octave"], bounds, mode, 100,
                                                           q_min_level=q_lo, q_max_level=q_hi, **kw)
    exp = O.match_candidates(d0, d1, cand_off, cand_idx, check_orientation=True, thr=100, lowe_ratio=0.8, mode=mode, t_octave=k1["octave"], **kw)
    assert (exp >= 0).sum() > 800
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()
    # same call again (scratch arena already large enough: single-pass path) and an empty window everywhere
    got2, _ = M.projection(0.8, True, ctx).match_in_cells(d0, q_xy, q_margin, d1, t_xy, k1["octave"], bounds, mode, 100,
                                                          q_min_level=q_lo, q_max_level=q_hi, **kw)
    assert np.array_equal(got2, exp)
    none, n0 = M.projection(0.8, True, ctx).match_in_cells(d0, q_xy + 5000.0, q_margin, d1, t_xy, k1["octave"], bounds, mode, 100)
    assert n0 == 0 and (none == -1).all()


def test_stereo_batch_device_matches_oracle():
    """BASELINE config 4 shape: stereo pairs at the KITTI geometry (1241 x 376, ini_fast_threshold 12), both images of every pair extracted in
    device-resident batches on two contexts, match::stereo::compute for all pairs in ONE launch (+ the median filter on the device).
    x_right / depth bits equal the oracle's per pair; the oracle runs on its own extraction and its own pyramids."""
    from stella_vslam_amd import feature, pipeline
    W, H, B, disp = 1241, 376, 3, 17
    big = S.frame_sequence(B, W + 64, H, seed=0x5EED + 4)
    left = np.ascontiguousarray(big[:, :, 8:8 + W])
    right = np.ascontiguousarray(big[:, :, 8 + disp:8 + disp + W])
    p = feature.orb_params(ini_fast_thr=12)
    el, er = pipeline.BatchExtractor(W, H, B, p), pipeline.BatchExtractor(W, H, B, p)
    el.upload(left)
    er.upload(right)
    el.extract()
    er.extract()
    er.ctx.synchronize()
    fxb, tb = 718.856 * 0.537, 0.537   # KITTI 00-02: fx = 718.856, baseline 0.537 m
    xr_t, dp_t = pipeline.stereo_batch(el, er, fxb, tb)
    outl, outr = el.download(), er.download()
    xr = xr_t.cpu().numpy().reshape(B, el.cap)
    dp = dp_t.cpu().numpy().reshape(B, el.cap)
    sizes = O.level_sizes(W, H)
    for b in range(B):
        pyr = []
        for img in (left[b], right[b]):
            lv = [img]
            for l in range(1, 8):
                lv.append(O.resize_linear(lv[-1], *sizes[l]))
            pyr.append(lv)
        ko, do, _ = O.orb_extract(left[b], ini_thr=12)
        kro, dro, _ = O.orb_extract(right[b], ini_thr=12)
        assert np.array_equal(outl[b][0], ko) and np.array_equal(outl[b][1], do) and np.array_equal(outr[b][0], kro)
        xo, dpo = O.stereo_match(ko, do, kro, dro, pyr[0], pyr[1], fxb, tb)
        n = len(ko)
        assert (xo >= 0).sum() > 800
        assert np.array_equal(xr[b, :n], xo) and np.array_equal(dp[b, :n], dpo)
        assert abs(np.median((ko["x"] - xo)[xo >= 0]) - disp) < 0.1


@pytest.mark.parametrize("mode", [O.MODE_BEST_ONLY, O.MODE_RATIO_SAME_OCTAVE, O.MODE_RATIO, O.MODE_TRIANGULATION])
def test_candidate_matcher_long_and_contended_lists(M, ctx, mode):
    """The replay's fast paths against the oracle's sequential loop: lists of 1 .. 1 500 candidates (one entry per lane up to 64 --
    in-register sort --, sorted in LDS up to 1 024, walked unsorted beyond), many duplicate descriptors (ties on distance are decided by the
    scan position) and few targets for many queries (long chains of queries losing their target to an earlier one: many sweeps)."""
    rng = np.random.default_rng(100 + mode)
    nt, nq = 1800, 700
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)              # 40 look-alike groups
    td = _noisy(rng, base[rng.integers(0, 40, nt)], 6)
    qd = _noisy(rng, base[rng.integers(0, 40, nq)], 6)
    td[0:1799:7] = td[3:1799:7][: len(td[0:1799:7])]                  # exact duplicates
    lens = rng.choice([1, 2, 5, 20, 63, 64, 65, 100, 300, 1024, 1025, 1500], nq, p=[.1, .1, .1, .2, .05, .05, .05, .1, .1, .05, .05, .05])
    cand_off, cand_idx = [0], []
    for q in range(nq):
        c = rng.choice(nt, min(int(lens[q]), nt), replace=False)
        if q % 3 == 0:
            c = c % 60                                                # contention: many queries over the same 60 targets ...
            _, first = np.unique(c, return_index=True)
            c = c[np.sort(first)]                                     # ... each listed once, scan order kept
        cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    t_oct = rng.integers(0, 8, nt).astype(np.int32)
    occupied = (rng.uniform(size=nt) < 0.05).astype(np.uint8)
    kw = dict(t_octave=t_oct, occupied=occupied)
    got, num = M.projection(0.9, False, ctx).match_candidates(qd, td, cand_off, cand_idx, mode, 120, **kw)
    exp = O.match_candidates(qd, td, cand_off, cand_idx, check_orientation=False, thr=120, lowe_ratio=0.9, mode=mode, **kw)
    assert (exp >= 0).sum() > 100
    assert np.array_equal(got, exp) and num == (exp >= 0).sum()


def test_match_in_cells_capacity_guess_miss_reruns(M):
    """The cell matcher remembers the candidate capacity that sufficed for a context and enqueues the next call against it; a call whose lists
    outgrow the guess must notice (nothing written) and re-run with the exact size: a small problem first, then a much denser one on the SAME
    context, both against the oracle, then the small one again."""
    from stella_vslam_amd import feature
    own = feature.Context()
    seq = S.frame_sequence(2, seed=0x5EED + 5)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    bounds = (0.0, 640.0, 0.0, 480.0)

    def run(nq, margin):
        q_xy = np.stack([k0["x"][:nq] - 3.0, k0["y"][:nq] - 1.0], 1).astype(np.float32)
        q_margin = np.full(nq, margin, np.float32)
        lo = np.maximum(k0["octave"][:nq] - 1, 0).astype(np.int32)
        hi = np.minimum(k0["octave"][:nq] + 1, 7).astype(np.int32)
        got, num = M.projection(0.8, False, own).match_in_cells(d0[:nq], q_xy, q_margin, d1, np.stack([k1["x"], k1["y"]], 1).astype(np.float32),
                                                               k1["octave"], bounds, 0, 100, q_min_level=lo, q_max_level=hi)
        off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
        cand_off, cand_idx = [0], []
        for q in range(nq):
            c = O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(q_xy[q, 0]), float(q_xy[q, 1]), margin, int(lo[q]), int(hi[q]))
            cand_idx += c.tolist()
            cand_off.append(len(cand_idx))
        exp = O.match_candidates(d0[:nq], d1, cand_off, cand_idx, check_orientation=False, thr=100, lowe_ratio=0.8, mode=0, t_octave=k1["octave"])
        assert np.array_equal(got, exp) and num == (exp >= 0).sum()
        return len(cand_idx)

    small = run(200, 8.0)
    big = run(len(k0), 60.0)
    assert big > 20 * small + 4096  # far beyond the remembered capacity (total * 1.25 + 4096)
    run(200, 8.0)


def test_matcher_grid_one_launch_equals_four_launches(M, monkeypatch):
    """The keypoint side of the matcher grid is built by ONE single-workgroup launch (cells on LDS counters, scan, placement, per-cell
    ordering) for grids up to 4 096 cells and 8 192 keypoints, by memset + assign + scan + stable placement beyond (forced here with
    SVGPU_GRID_FOUR_LAUNCHES=1): the same cell lists in the same order, so the matcher's output -- which depends on the scan order inside
    the cells -- is identical, and equal to the oracle's."""
    from stella_vslam_amd import feature
    seq = S.frame_sequence(2, seed=0x5EED + 11)
    k0, d0, _ = O.orb_extract(seq[0])
    k1, d1, _ = O.orb_extract(seq[1])
    bounds = (0.0, 640.0, 0.0, 480.0)
    nq = len(k0)
    q_xy = np.stack([k0["x"] - 2.0, k0["y"] + 1.0], 1).astype(np.float32)
    q_margin = np.full(nq, 25.0, np.float32)
    res = []
    for four in (False, True):
        if four: monkeypatch.setenv("SVGPU_GRID_FOUR_LAUNCHES", "1")
        else: monkeypatch.delenv("SVGPU_GRID_FOUR_LAUNCHES", raising=False)
        got, num = M.projection(0.8, False, feature.Context()).match_in_cells(d0, q_xy, q_margin, d1, np.stack([k1["x"], k1["y"]], 1).astype(np.float32),
                                                                            k1["octave"], bounds, 0, 100)
        res.append((got, num))
    monkeypatch.delenv("SVGPU_GRID_FOUR_LAUNCHES", raising=False)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
    cand_off, cand_idx = [0], []
    for q in range(nq):
        c = O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(q_xy[q, 0]), float(q_xy[q, 1]), 25.0, -1, -1)
        cand_idx += c.tolist()
        cand_off.append(len(cand_idx))
    exp = O.match_candidates(d0, d1, cand_off, cand_idx, check_orientation=False, thr=100, lowe_ratio=0.8, mode=0, t_octave=k1["octave"])
    assert np.array_equal(res[0][0], exp) and res[0][1] == (exp >= 0).sum()
