"""The drop-in bar itself: the reference's optimize::local_bundle_adjuster_g2o (compiled from /root/reference into oracle/_ref/libsvref_ba.so,
g2o's optimize() played by the oracle's LM) and the PRODUCT's optimize::local_bundle_adjuster_hip (stella_vslam_amd/host/drop_in/hip_backend.cc
compiled against the very same stand-in data:: headers into oracle/_ref/libsvref_dropin.so, linked to libsvgpu.so) are handed identical
keyframe / landmark / map objects; the maps they leave behind are compared: which keyframes were written, SE3 poses within 1e-4 relative,
positions, the erased observations, the landmark refresh calls.  Below the same for the motion-only pose optimizer: the reference's
pose_optimizer_g2o.cc (libsvref_opt.so) against the product's optimize::pose_optimizer_hip compiled over the same stand-ins (libsvref_pdropin.so)."""
import ctypes as C
import os

import numpy as np
import pytest

from stella_vslam_amd import synthetic as S

pytestmark = pytest.mark.gpu
_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


@pytest.fixture(scope="module")
def libs():
    a, b = os.path.join(_DIR, "libsvref_ba.so"), os.path.join(_DIR, "libsvref_dropin.so")
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/libsvref_{ba,dropin}.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(a), C.CDLL(b)


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _both(libs, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, covis, curr, threshold, use_additional, iters=(5, 10), stop_in=0):
    ref, prod = libs
    K, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    intr = np.ascontiguousarray(sc["intr"][0])
    idx = np.zeros(E, np.int32)
    seen = np.zeros(K, np.int64)
    for e in range(E):
        idx[e] = seen[sc["obs_pose"][e]]
        seen[sc["obs_pose"][e]] += 1
    octv = np.random.default_rng(E).integers(0, 8, E).astype(np.int32)
    a = dict(kf_id=np.ascontiguousarray(kf_id, np.uint32), kf_pose=np.ascontiguousarray(sc["pose_cw"], np.float64), kf_flags=np.ascontiguousarray(kf_flags, np.uint8),
             lm_id=np.ascontiguousarray(lm_id, np.uint32), lm_pos=np.ascontiguousarray(sc["points"], np.float64), lm_erased=np.ascontiguousarray(lm_erased, np.uint8),
             obs_kf=np.ascontiguousarray(sc["obs_pose"], np.int32), obs_lm=np.ascontiguousarray(sc["obs_point"], np.int32), obs_idx=idx,
             uv=np.ascontiguousarray(sc["obs_uvr"][:, :2], np.float32), xr=np.ascontiguousarray(sc["obs_uvr"][:, 2], np.float32), oct=octv,
             covis=np.ascontiguousarray(covis, np.int32))
    head = lambda: (0, stereo, 1280, 720, _p(intr), C.c_float(1.2), 8, K, _p(a["kf_id"]), _p(a["kf_pose"]), _p(a["kf_flags"]), L, _p(a["lm_id"]), _p(a["lm_pos"]),
                    _p(a["lm_erased"]), E, _p(a["obs_kf"]), _p(a["obs_lm"]), _p(a["obs_idx"]), _p(a["uv"]), _p(a["xr"]), _p(a["oct"]), curr, len(covis),
                    _p(a["covis"]), threshold, int(use_additional), iters[0], iters[1], stop_in)

    def outs():
        return dict(kf_pose=np.zeros((K, 12)), lm_pos=np.zeros((L, 3)), n_erased=C.c_int(0), erased=np.zeros(2 * E + 2, np.int32), lm_cnt=np.zeros((L, 4), np.int32),
                    kf_set=np.zeros(K, np.int32), stop=np.zeros(1, np.uint8))
    r, g = outs(), outs()
    r.update(counts=np.zeros(3, np.int32), pose_order=np.full(K, -1, np.int32), point_order=np.full(L, -1, np.int32), edge_order=np.full(2 * E, -1, np.int32),
             iters=np.zeros(2, np.int32))
    assert ref.svref_local_ba(*head(), _p(r["counts"]), _p(r["pose_order"]), _p(r["point_order"]), _p(r["edge_order"]), _p(r["kf_pose"]), _p(r["lm_pos"]),
                              C.byref(r["n_erased"]), _p(r["erased"]), _p(r["lm_cnt"]), _p(r["kf_set"]), _p(r["iters"]), _p(r["stop"])) == 0
    g["stats"] = np.zeros(6, np.int32)
    assert prod.svref_dropin_local_ba(*head(), _p(g["kf_pose"]), _p(g["lm_pos"]), C.byref(g["n_erased"]), _p(g["erased"]), _p(g["lm_cnt"]), _p(g["kf_set"]),
                                      _p(g["stats"]), _p(g["stop"])) == 0
    return r, g


def _pairs(o):
    return {(int(o["erased"][2 * i]), int(o["erased"][2 * i + 1])) for i in range(o["n_erased"].value)}


def _compare(sc, r, g, expect_work=True):
    np.testing.assert_array_equal(r["kf_set"], g["kf_set"])          # the same keyframes written, once each
    np.testing.assert_array_equal(r["lm_cnt"], g["lm_cnt"])          # set_pos / update geometry / compute_descriptor / erase_observation calls per landmark
    assert _pairs(r) == _pairs(g)                                    # the same observations removed from the map
    assert r["stop"][0] == g["stop"][0]
    if expect_work:
        assert g["stats"][0] == 0 and [int(g["stats"][1]), int(g["stats"][2])] == r["iters"].tolist()
        assert r["kf_set"].sum() > 0 and len(_pairs(r)) > 0
    for k in range(len(r["kf_pose"])):
        if r["kf_set"][k] == 0:
            np.testing.assert_array_equal(g["kf_pose"][k], sc["pose_cw"][k])
            continue
        Tr, Tg = r["kf_pose"][k].reshape(3, 4), g["kf_pose"][k].reshape(3, 4)
        assert np.abs(Tr[:, :3] - Tg[:, :3]).max() <= 1e-4
        assert np.linalg.norm(Tr[:, 3] - Tg[:, 3]) <= 1e-4 * max(1.0, np.linalg.norm(Tr[:, 3]))
    moved = r["lm_cnt"][:, 0] > 0
    np.testing.assert_array_equal(g["lm_pos"][~moved], sc["points"][~moved])
    if not moved.any():
        return
    scale = np.maximum(1.0, np.linalg.norm(r["lm_pos"][moved], axis=1))
    assert (np.linalg.norm(r["lm_pos"][moved] - g["lm_pos"][moved], axis=1) / scale).max() <= 1e-4


@pytest.mark.parametrize("stereo", [0, 1])
@pytest.mark.parametrize("case", ["plain", "threshold_root_erased"])
def test_local_bundle_adjuster_hip_against_local_bundle_adjuster_g2o(libs, stereo, case):
    sc = S.ba_scene(num_kf=12, num_lm=600, obs_per_lm=4, num_fixed=0, seed=90 + stereo, stereo=bool(stereo))
    K, L = len(sc["pose_cw"]), len(sc["points"])
    rng = np.random.default_rng(17 + stereo)
    kf_id = 10 + 3 * np.arange(K)
    lm_id = 1000 + 7 * rng.permutation(L)
    kf_flags = np.zeros(K, np.uint8)
    lm_erased = (rng.uniform(size=L) < 0.03).astype(np.uint8)
    covis = [int(c) for c in rng.permutation(K - 1)[:7]]
    threshold = 0
    if case == "threshold_root_erased":
        covis += [-1]
        kf_flags[covis[0]] |= 1
        kf_flags[covis[1]] |= 2
        threshold = int(kf_id[sorted(covis[2:7])[0]]) + 1
    r, g = _both(libs, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, covis, K - 1, threshold, False)
    _compare(sc, r, g)


def test_a_stop_requested_before_the_call_leaves_the_map_untouched(libs):
    sc = S.ba_scene(num_kf=8, num_lm=300, obs_per_lm=4, num_fixed=0, seed=3)
    K, L = len(sc["pose_cw"]), len(sc["points"])
    r, g = _both(libs, sc, 0, 1 + np.arange(K), np.zeros(K, np.uint8), np.arange(L), np.zeros(L, np.uint8), list(range(K - 1)), K - 1, 0, False, stop_in=1)
    _compare(sc, r, g, expect_work=False)
    assert g["kf_set"].sum() == 0 and g["lm_cnt"].sum() == 0
    np.testing.assert_array_equal(g["lm_pos"], sc["points"])


def test_larger_window_with_a_null_flag(libs):
    """force_stop_flag == nullptr (global-optimisation callers) on a 40-keyframe window."""
    sc = S.ba_scene(num_kf=40, num_lm=4000, obs_per_lm=5, num_fixed=0, seed=11)
    K, L = len(sc["pose_cw"]), len(sc["points"])
    rng = np.random.default_rng(2)
    covis = [int(c) for c in rng.permutation(K - 1)[:25]]
    r, g = _both(libs, sc, 0, 100 + np.arange(K), np.zeros(K, np.uint8), rng.permutation(L), np.zeros(L, np.uint8), covis, K - 1, 0, False, stop_in=-1)
    _compare(sc, r, g)


# ---------------------------------------------------------------------------------------------------------------- pose optimizer
def _load_cases(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_cases_" + name, os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def pose_libs():
    a, b = os.path.join(_DIR, "libsvref_opt.so"), os.path.join(_DIR, "libsvref_pdropin.so")
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/libsvref_{opt,pdropin}.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(a), C.CDLL(b)


@pytest.mark.parametrize("model,stereo", [(0, 0), (0, 1), (1, 0), (2, 0), (3, 1)])
def test_pose_optimizer_hip_against_pose_optimizer_g2o(pose_libs, model, stereo):
    """optimize::pose_optimizer_g2o (the reference's, g2o's optimize() played by the oracle's LM) and optimize::pose_optimizer_hip (the product's,
    one kernel launch) on the same frame through the three overloads of optimize::pose_optimizer: the same keypoints flagged, the same count
    returned, the same number of LM iterations (give or take one at convergence), the pose within 1e-4; fewer than five live landmarks leave pose and flags untouched."""
    ref, prod = pose_libs
    T = _load_cases("test_ref_local_optimize")
    rng = np.random.default_rng(900 + 10 * model + stereo)
    for trial in range(24):
        n = int(rng.choice([3, 6, 40, 300, 1500]))
        sc = T._pose_scene(rng, model, stereo, n)
        n = sc["n"]
        tr, tn, each = [(2, 2, 10), (0, 3, 5), (2, 0, 10), (1, 1, 3)][trial % 4]
        for reset in (0, 1):
            res = []
            for fn in (ref.svref_pose_optimize, prod.svref_dropin_pose_optimize):
                pose, flags, its = np.zeros(12), np.zeros(n, np.uint8), C.c_int(0)
                fn.restype = C.c_int
                valid = fn(model, stereo, sc["cols"], sc["rows"], _p(sc["intr"]), _p(sc["pose"]), n, _p(sc["kp"]), _p(sc["octave"]),
                           None if sc["xr"] is None else _p(sc["xr"]), _p(sc["pw"]), _p(sc["state"]), C.c_float(1.2), 8, tr, tn, each, reset, trial % 3,
                           _p(pose), _p(flags), C.byref(its))
                res.append((valid, pose, flags, its.value))
            (vr, pr, fr, ir), (vg, pg, fg, ig) = res
            assert vr == vg
            np.testing.assert_array_equal(fr, fg)
            if int((sc["state"] == 1).sum()) < 5:
                np.testing.assert_array_equal(pg, sc["pose"])
                assert vg == 0
                continue
            assert abs(ir - ig) <= 1   # (at a converged pose the gain test of a further iteration is decided by rounding: 4 of 240 calls differ by one, poses 1e-10 apart)
            Rr, Rg = pr.reshape(3, 4), pg.reshape(3, 4)
            assert np.abs(Rr[:, :3] - Rg[:, :3]).max() <= 1e-4
            assert np.linalg.norm(Rr[:, 3] - Rg[:, 3]) <= 1e-4 * max(1.0, np.linalg.norm(Rr[:, 3]))


# ---------------------------------------------------------------------------------------------------------------- global bundle adjuster
def _global_both(libs, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, order, num_iter, use_huber, stop_in):
    ref, prod = libs
    B = _load_cases("test_ref_local_ba")
    r = B._run_global(ref, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, order, num_iter, use_huber, stop_in)
    K, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    intr = np.ascontiguousarray(sc["intr"][0])
    idx = np.zeros(E, np.int32)
    seen = np.zeros(K, np.int64)
    for e in range(E):
        idx[e] = seen[sc["obs_pose"][e]]
        seen[sc["obs_pose"][e]] += 1
    a = dict(kf_id=np.ascontiguousarray(kf_id, np.uint32), kf_pose=np.ascontiguousarray(sc["pose_cw"], np.float64), kf_flags=np.ascontiguousarray(kf_flags, np.uint8),
             lm_id=np.ascontiguousarray(lm_id, np.uint32), lm_pos=np.ascontiguousarray(sc["points"], np.float64), lm_erased=np.ascontiguousarray(lm_erased, np.uint8),
             obs_kf=np.ascontiguousarray(sc["obs_pose"], np.int32), obs_lm=np.ascontiguousarray(sc["obs_point"], np.int32), obs_idx=idx,
             uv=np.ascontiguousarray(sc["obs_uvr"][:, :2], np.float32), xr=np.ascontiguousarray(sc["obs_uvr"][:, 2], np.float32), oct=r["octave"],
             order=np.ascontiguousarray(order, np.int32))
    g = dict(kf_pose=np.zeros((K, 12)), lm_pos=np.zeros((L, 3)), kf_opt=np.zeros(K, np.uint8), lm_opt=np.zeros(L, np.uint8), iters=np.zeros(2, np.int32),
             stop=np.zeros(1, np.uint8))
    prod.svref_dropin_global_ba.restype = C.c_int
    g["ok"] = prod.svref_dropin_global_ba(0, stereo, 1280, 720, _p(intr), C.c_float(1.2), 8, K, _p(a["kf_id"]), _p(a["kf_pose"]), _p(a["kf_flags"]), L, _p(a["lm_id"]),
                                          _p(a["lm_pos"]), _p(a["lm_erased"]), E, _p(a["obs_kf"]), _p(a["obs_lm"]), _p(a["obs_idx"]), _p(a["uv"]), _p(a["xr"]),
                                          _p(a["oct"]), len(order), _p(a["order"]), num_iter, int(use_huber), stop_in, _p(g["kf_pose"]), _p(g["lm_pos"]),
                                          _p(g["kf_opt"]), _p(g["lm_opt"]), _p(g["iters"]), _p(g["stop"]))
    return r, g


@pytest.mark.parametrize("stereo,use_huber,num_iter,messy", [(0, True, 10, False), (1, False, 10, False), (0, True, 10, True), (0, True, 30, False)])
def test_global_bundle_adjuster_hip_against_global_bundle_adjuster(libs, stereo, use_huber, num_iter, messy):
    """optimize::global_bundle_adjuster::optimize (the reference's, compiled from global_bundle_adjuster.cc) and
    optimize::global_bundle_adjuster_hip::optimize (the product's) on identical keyframes: the same return value, the same ids in the
    optimised sets, the result maps within 1e-4, the same iteration count, the caller's flag written by the gain rule alike."""
    sc = S.ba_scene(num_kf=14, num_lm=700, obs_per_lm=4, num_fixed=0, seed=70 + stereo + num_iter, stereo=bool(stereo))
    K, L = len(sc["pose_cw"]), len(sc["points"])
    rng = np.random.default_rng(5 + num_iter)
    kf_id, lm_id = 3 + 2 * np.arange(K), 500 + rng.permutation(L)
    kf_flags, lm_erased = np.zeros(K, np.uint8), np.zeros(L, np.uint8)
    kf_flags[4] |= 2
    if messy:
        kf_flags[7] |= 1
        lm_erased[rng.uniform(size=L) < 0.05] = 1
    r, g = _global_both(libs, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, rng.permutation(K), num_iter, use_huber, 0)
    assert r["ok"] == 1 and g["ok"] == 1
    np.testing.assert_array_equal(r["kf_opt"], g["kf_opt"])
    np.testing.assert_array_equal(r["lm_opt"], g["lm_opt"])
    assert r["iters"][0] == g["iters"][0] and r["stop"][0] == g["stop"][0]
    if num_iter == 30:
        assert g["stop"][0] == 1 and g["iters"][0] < 30 and g["iters"][1] == 1   # the gain rule stopped it, through the caller's flag
    for k in range(K):
        Tr, Tg = r["kf_pose"][k].reshape(3, 4), g["kf_pose"][k].reshape(3, 4)
        assert np.abs(Tr[:, :3] - Tg[:, :3]).max() <= 1e-4
        assert np.linalg.norm(Tr[:, 3] - Tg[:, 3]) <= 1e-4 * max(1.0, np.linalg.norm(Tr[:, 3]))
    scale = np.maximum(1.0, np.linalg.norm(r["lm_pos"], axis=1))
    assert (np.linalg.norm(r["lm_pos"] - g["lm_pos"], axis=1) / scale).max() <= 1e-4
    np.testing.assert_array_equal(g["lm_pos"][g["lm_opt"] == 0], sc["points"][g["lm_opt"] == 0])


def test_global_bundle_adjuster_hip_discards_a_run_the_caller_stopped(libs):
    sc = S.ba_scene(num_kf=8, num_lm=300, obs_per_lm=4, num_fixed=0, seed=61)
    K, L = len(sc["pose_cw"]), len(sc["points"])
    flags = np.zeros(K, np.uint8)
    flags[0] = 2
    r, g = _global_both(libs, sc, 0, 1 + np.arange(K), flags, 1 + np.arange(L), np.zeros(L, np.uint8), np.arange(K), 10, True, 1)
    assert r["ok"] == 0 and g["ok"] == 0 and g["kf_opt"].sum() == 0 and g["lm_opt"].sum() == 0 and g["stop"][0] == 1


# ---------------------------------------------------------------------------------------------------------------- extractor
class _Renamed:
    """The extractor fixture of the product library under the name the reference-side helper calls."""

    def __init__(self, lib):
        self.svref_orb_extract = lib.svref_dropin_orb_extract


@pytest.fixture(scope="module")
def extract_libs():
    a, b = os.path.join(_DIR, "libsvref.so"), os.path.join(_DIR, "libsvref_xdropin.so")
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/libsvref{,_xdropin}.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(a), _Renamed(C.CDLL(b))


@pytest.mark.parametrize("w,h,seed,kw", [(640, 480, 1, {}), (641, 479, 3, {}), (1241, 376, 4, dict(ini_thr=12)), (200, 150, 5, dict(min_area=100)),
                                         (640, 480, 6, dict(scale_factor=1.5, num_levels=4)), (640, 480, 8, dict(ini_thr=40, min_thr=30))])
def test_orb_extractor_adaptor_against_the_reference_extractor(extract_libs, w, h, seed, kw):
    """feature::orb_extractor::extract of the reference (feature/orb_extractor.cc, orb_impl.cc compiled where they lie; the OpenCV primitives
    behind the stand-in headers are the oracle's) and the product's adaptor class (host/orb_extractor.cpp in its OpenCV mode over the same
    stand-in cv::Mat / cv::KeyPoint / _InputArray / _OutputArray) on the same image: the seven keypoint fields and the descriptors bit for
    bit, image_pyramid_ level by level."""
    ref, prod = extract_libs
    X = _load_cases("test_ref_local")
    img = S.frame_sequence(1, w, h, seed=seed)[0]
    rk, rd, rp = X._ref_extract(ref, img, **kw)
    gk, gd, gp = X._ref_extract(prod, img, **kw)
    assert len(rk) > 0
    np.testing.assert_array_equal(rk.view(np.uint32), gk.view(np.uint32))
    np.testing.assert_array_equal(rd, gd)
    for a, b in zip(rp, gp):
        np.testing.assert_array_equal(a, b)


def test_orb_extractor_adaptor_with_masks_against_the_reference_extractor(extract_libs):
    ref, prod = extract_libs
    X = _load_cases("test_ref_local")
    img = S.frame_sequence(1, 640, 480, seed=9)[0]
    yy, xx = np.mgrid[0:480, 0:640]
    mask = np.ones((480, 640), np.uint8)
    mask[(xx - 320) ** 2 + (yy - 200) ** 2 <= 120 ** 2] = 0
    mask[400:, :] = 0
    rects = [[0.0, 1.0, 0.0, 0.2], [0.3, 0.55, 0.45, 0.75], [0.8, 1.0, 0.0, 1.0], [0.1, 0.1203125, 0.5, 0.503125]]  # create_rectangle_mask on both sides
    for kw in (dict(mask=mask), dict(rects=rects)):
        rk, rd, _ = X._ref_extract(ref, img, **kw)
        gk, gd, _ = X._ref_extract(prod, img, **kw)
        assert len(rk) > 0
        np.testing.assert_array_equal(rk.view(np.uint32), gk.view(np.uint32))
        np.testing.assert_array_equal(rd, gd)


@pytest.mark.parametrize("disp,noise", [(12, 0), (31, 2)])
def test_stereo_adaptor_against_the_reference_stereo_matcher(disp, noise):
    """A stereo frame through the product's classes (two feature::orb_extractor adaptors, match::stereo on their device-resident pyramids), then
    the REFERENCE's match::stereo (match/stereo.cc compiled where it lies) on exactly what the two extractors put out -- keypoints, descriptors,
    image_pyramid_: x_right and depths bit for bit."""
    a, b = os.path.join(_DIR, "libsvref.so"), os.path.join(_DIR, "libsvref_xdropin.so")
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/libsvref{,_xdropin}.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    ref, prod = C.CDLL(a), C.CDLL(b)
    from oracle import oracle as O
    big = S.frame(640 + 64, 480, 5)
    left, right = np.ascontiguousarray(big[:, 8:648]), np.ascontiguousarray(big[:, 8 + disp:648 + disp])
    if noise:
        rng = np.random.default_rng(disp)
        right = np.clip(right.astype(np.int16) + rng.integers(-noise, noise + 1, right.shape), 0, 255).astype(np.uint8)
    w, h, L, cap = 640, 480, 8, 4000
    sizes = O.level_sizes(w, h, 1.2, L)
    kl, kr = np.zeros((cap, 7), np.float32), np.zeros((cap, 7), np.float32)
    dl, dr = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    pyl, pyr = np.zeros(sum(x * y for x, y in sizes), np.uint8), np.zeros(sum(x * y for x, y in sizes), np.uint8)
    gxr, gdp = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nr = C.c_int(0)
    fxb, bl = 458.654 * 0.11, 0.11
    prod.svref_dropin_stereo.restype = C.c_int
    nl = prod.svref_dropin_stereo(_p(left), _p(right), w, h, C.c_float(1.2), L, 20, 7, C.c_float(fxb), C.c_float(bl), cap, _p(kl), _p(dl), _p(kr), _p(dr),
                                  C.byref(nr), _p(pyl), _p(pyr), _p(gxr), _p(gdp))
    assert nl > 1500 and nr.value > 1500
    levels_l, levels_r, off = [], [], 0
    for (lw_, lh_) in sizes:
        levels_l.append(np.ascontiguousarray(pyl[off:off + lw_ * lh_].reshape(lh_, lw_)))
        levels_r.append(np.ascontiguousarray(pyr[off:off + lw_ * lh_].reshape(lh_, lw_)))
        off += lw_ * lh_
    np.testing.assert_array_equal(levels_l[0], left)   # image_pyramid_[0] is the caller's image
    PL = (C.c_void_p * L)(*[x.ctypes.data for x in levels_l])
    PR = (C.c_void_p * L)(*[x.ctypes.data for x in levels_r])
    lw = np.array([x.shape[1] for x in levels_l], np.int32)
    lh = np.array([x.shape[0] for x in levels_l], np.int32)
    ls = np.array([x.strides[0] for x in levels_l], np.int32)
    kl28, kr28 = np.ascontiguousarray(kl[:nl]), np.ascontiguousarray(kr[:nr.value])   # 7 x 4 bytes per keypoint: the record svref_stereo_compute takes
    dl_, dr_ = np.ascontiguousarray(dl[:nl]), np.ascontiguousarray(dr[:nr.value])
    rxr, rdp = np.zeros(nl, np.float32), np.zeros(nl, np.float32)
    ref.svref_stereo_compute.restype = None
    ref.svref_stereo_compute(_p(kl28), _p(dl_), nl, _p(kr28), _p(dr_), nr.value, PL, PR, _p(lw), _p(lh), _p(ls), _p(ls), C.c_float(1.2), L, C.c_float(fxb),
                             C.c_float(bl), _p(rxr), _p(rdp))
    assert (rxr >= 0).sum() > 500
    np.testing.assert_array_equal(rxr.view(np.uint32), gxr[:nl].view(np.uint32))
    np.testing.assert_array_equal(rdp.view(np.uint32), gdp[:nl].view(np.uint32))


# ------------------------------------------------------------------------------------------------ the per-frame tracker
@pytest.fixture(scope="module")
def tracker_libs():
    a, b = os.path.join(_DIR, "libsvref_trk.so"), os.path.join(_DIR, "libsvref_tdropin.so")
    if not (os.path.exists(a) and os.path.exists(b)):
        pytest.skip("oracle/_ref/libsvref_{trk,tdropin}.so absent: built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(a), C.CDLL(b)


def _track_both(libs, sc, stereo, thr, margin, seed=1, run_local=1, override=None, local=None, margin_local=5.0, lm_flags=None, guess_off=(0.25, 0.01)):
    """svref_track_frame of both libraries on the same arrays -> (reference outputs, product outputs)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("match_problems", os.path.join(os.path.dirname(os.path.abspath(__file__)), "match_problems.py"))
    MP = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(MP)
    ref, prod = libs
    cam = MP.make_cams(sc, "oracle")
    L, (last, cur), T = sc["landmarks"], sc["views"], sc["tables"]
    n_lm = len(L["pos_w"])
    rng = np.random.default_rng(100 + seed)
    has_obs, erased = np.ones(n_lm, np.uint8), np.zeros(n_lm, np.uint8)
    if lm_flags is None:
        has_obs[rng.uniform(size=n_lm) < 0.15] = 0
        erased[rng.uniform(size=n_lm) < 0.03] = 1
    a = dict(id=(3 * np.arange(n_lm) + 5).astype(np.uint32), pos=np.ascontiguousarray(L["pos_w"], np.float64), nrm=np.ascontiguousarray(L["mean_normal"], np.float64),
             mn=np.ascontiguousarray(L["min_valid_dist"], np.float32), mx=np.ascontiguousarray(L["max_valid_dist"], np.float32), desc=np.ascontiguousarray(L["desc"], np.uint8),
             ho=has_obs, er=erased, lo=np.ascontiguousarray(last["octave"], np.int32), la=np.ascontiguousarray(last["angle"], np.float32),
             ll=np.ascontiguousarray(last["lm"], np.int32), cd=np.ascontiguousarray(cur["desc"], np.uint8), cxy=np.ascontiguousarray(cur["xy"], np.float32),
             co=np.ascontiguousarray(cur["octave"], np.int32), ca=np.ascontiguousarray(cur["angle"], np.float32),
             cxr=np.ascontiguousarray(cur["x_right"], np.float32) if stereo else None, cb=np.ascontiguousarray(MP._bearings(sc, cur["xy"]), np.float64))
    pose_last = np.hstack([last["rot_cw"], last["trans_cw"][:, None]]).astype(np.float64)
    R, t = MP._perturb(cur["rot_cw"], cur["trans_cw"], np.random.default_rng(seed), *guess_off)
    G, Lw = np.eye(4), np.eye(4)
    G[:3, :3], G[:3, 3] = R, t
    Lw[:3, :] = pose_last
    velocity = np.ascontiguousarray(G @ np.linalg.inv(Lw), np.float64)
    local_idx = np.ascontiguousarray(np.arange(n_lm) if local is None else local, np.int32)
    ov = None if override is None else np.ascontiguousarray(override, np.float64)
    sf = np.asarray(T["scale_factors"], np.float32)
    n_last, n_cur = len(a["lo"]), len(a["cd"])
    outs = []
    for lib in (ref, prod):
        o = dict(ret=np.full(2, -9, np.int32), lm1=np.full(n_cur, -9, np.int32), p1=np.zeros(12), lm2=np.full(n_cur, -9, np.int32), p2=np.zeros(12),
                 nobs=np.zeros(n_lm, np.int32))
        rc = lib.svref_track_frame(C.byref(cam), int(not stereo), C.c_float(0.11), C.c_float(float(sf[1] / sf[0])), len(sf), 64, 48, n_lm, _p(a["id"]), _p(a["pos"]),
                                   _p(a["nrm"]), _p(a["mn"]), _p(a["mx"]), _p(a["desc"]), _p(a["ho"]), _p(a["er"]), n_last, _p(a["lo"]), _p(a["la"]), _p(a["ll"]),
                                   _p(np.ascontiguousarray(pose_last.reshape(12))), n_cur, _p(a["cd"]), _p(a["cxy"]), _p(a["co"]), _p(a["ca"]),
                                   _p(a["cxr"]) if stereo else None, _p(a["cb"]), _p(velocity), int(thr), C.c_float(margin), int(run_local),
                                   _p(ov) if ov is not None else None, len(local_idx), _p(local_idx), 0, C.c_float(margin_local), C.c_float(0.8), _p(o["ret"]),
                                   _p(o["lm1"]), _p(o["p1"]), _p(o["lm2"]), _p(o["p2"]), _p(o["nobs"]))
        assert rc == 0
        outs.append(o)
    return outs[0], outs[1], G


def _rel_pose(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("stereo", [False, True])
def test_chain_vs_reference_tracker(tracker_libs, stereo):
    """hip::tracked_frame_chain::motion_based_track / track_local_map beside the reference's OWN module/frame_tracker.cc (+ match/projection.cc,
    optimize/pose_optimizer_g2o.cc; search_local_landmarks / optimize_current_frame_with_local_map restated in the fixture) on identical
    data::frame / data::landmark objects: the same return values, the same landmark per keypoint of curr_frm after each half (matches AND
    outlier erasures), the same num_observable marks, poses within 1e-4 -- on the plain path, the 2 x margin retry (frame_tracker.cc:32-36),
    both num_matches_thr failures (:38-41, :53-56), the local-map half behind a failed first half (the frame's pose set by a fallback tracker)
    and the "no projection candidate" return (tracking_module.cc:596-599)."""
    sc = S.map_scene(seed=11 if stereo else 7, stereo=stereo)
    margin = 10.0 if stereo else 20.0   # frame_tracker's margin_ (tracking_module.cc:37-38: 10 for stereo / RGB-D, 20 for monocular)

    def same(r, g, halves=2):
        assert np.array_equal(r["ret"], g["ret"]), (r["ret"], g["ret"])
        assert np.array_equal(r["lm1"], g["lm1"])
        assert _rel_pose(g["p1"], r["p1"]) < 1e-4
        if halves == 2:
            assert np.array_equal(r["lm2"], g["lm2"])
            assert _rel_pose(g["p2"], r["p2"]) < 1e-4
            assert np.array_equal(r["nobs"], g["nobs"])

    # 1. the plain path: both halves succeed
    r, g, G = _track_both(tracker_libs, sc, stereo, 20, margin)
    assert list(r["ret"]) == [1, 1]
    n1, n2 = int((r["lm1"] >= 0).sum()), int((r["lm2"] >= 0).sum())
    assert n1 > 200 and n2 > n1 + 50 and (r["nobs"] > 1).sum() > 300
    same(r, g)
    # 2. the retry: a margin so small that the first search stays under the threshold and the doubled one passes it
    tiny = 1.0
    r_a, _, _ = _track_both(tracker_libs, sc, stereo, 0, tiny, run_local=0, guess_off=(0.6, 0.02))
    r_b, _, _ = _track_both(tracker_libs, sc, stereo, 0, 2 * tiny, run_local=0, guess_off=(0.6, 0.02))
    ma, mb = int((r_a["lm1"] >= 0).sum()), int((r_b["lm1"] >= 0).sum())    # inliers of the single-margin and the double-margin searches (thr 0: no retry)
    assert mb > ma + 20, (ma, mb)
    thr = (ma + mb) // 2 + 1   # more than the first search can match (matches >= inliers; checked below through the outcome), fewer than the second's inliers
    r, g, _ = _track_both(tracker_libs, sc, stereo, thr, tiny, guess_off=(0.6, 0.02))
    same(r, g)
    if r["ret"][0] == 1:       # the doubled margin was what passed: the frame holds the second search's matches
        assert int((r["lm1"] >= 0).sum()) >= thr > ma
    # 3. num_matches_thr beyond both searches (:38-41): false, the frame keeps the motion-model pose and the second search's matches
    r, g, G = _track_both(tracker_libs, sc, stereo, 100000, margin, run_local=0)
    assert r["ret"][0] == 0 and (r["lm1"] >= 0).sum() > 200
    assert np.array_equal(r["p1"], G[:3, :].reshape(12)) and np.array_equal(g["p1"], r["p1"])
    same(r, g, halves=1)
    # 4. enough matches, too few inliers (:53-56): false AFTER the optimisation and the outlier erasures
    r0, _, _ = _track_both(tracker_libs, sc, stereo, 0, margin, run_local=0)
    inl = int((r0["lm1"] >= 0).sum())
    r, g, _ = _track_both(tracker_libs, sc, stereo, inl + 1, margin, run_local=0)
    assert r["ret"][0] == 0 and int((r["lm1"] >= 0).sum()) == inl and not np.array_equal(r["p1"], G[:3, :].reshape(12))
    same(r, g, halves=1)
    # 5. the local-map half behind a FAILED first half, from the pose a fallback tracker left (not the device's stale one)
    cur = sc["views"][1]
    fb = np.hstack([cur["rot_cw"], (cur["trans_cw"] + np.array([0.01, -0.008, 0.004]))[:, None]]).reshape(12)
    r, g, _ = _track_both(tracker_libs, sc, stereo, 100000, margin, override=fb)
    assert list(r["ret"]) == [0, 1] and (r["lm2"] >= 0).sum() > 300
    same(r, g)
    # 6. no projection candidate (:596-599): every local landmark is already held by the frame
    r1, _, _ = _track_both(tracker_libs, sc, stereo, 20, margin, run_local=0)
    held = np.unique((r1["lm1"][r1["lm1"] >= 0] - 5) // 3).astype(np.int32)
    r, g, _ = _track_both(tracker_libs, sc, stereo, 20, margin, local=held)
    assert list(r["ret"]) == [1, 0]
    same(r, g)
