"""Pins the oracle's local bundle adjustment (oracle/ba_oracle.c orc_local_ba: the two-stage schedule, the chi-square / depth gate, the
kernel removal, the outlier list) and the flat-problem convention of the adaptors against the REFERENCE's own
optimize/local_bundle_adjuster_g2o.cc, compiled where it lies into oracle/_ref/libsvref_ba.so over stand-in data:: headers and a g2o stand-in
whose SparseOptimizer::optimize() is played by the oracle's Levenberg-Marquardt on the graph the reference built.  What is the reference's
own code here: the gather of local / fixed keyframes and local landmarks (covisibilities, spanning root, erased keyframes, the map's fixed-id
threshold, the two extra fixed keyframes of a monocular map), vertex and edge construction, the schedule, the gate, the write-back."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from stella_vslam_amd import synthetic as S

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref_ba.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref_ba.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _run(ref, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, covis, curr, threshold, use_additional, iters=(5, 10), stop_in=-1):
    K, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    intr = np.ascontiguousarray(sc["intr"][0])
    # keypoint index of an observation inside its keyframe: running count per keyframe
    idx = np.zeros(E, np.int32)
    seen = np.zeros(K, np.int64)
    for e in range(E):
        idx[e] = seen[sc["obs_pose"][e]]
        seen[sc["obs_pose"][e]] += 1
    rng = np.random.default_rng(E)
    octv = rng.integers(0, 8, E).astype(np.int32)
    uv = np.ascontiguousarray(sc["obs_uvr"][:, :2], np.float32)
    xr = np.ascontiguousarray(sc["obs_uvr"][:, 2], np.float32)
    a = dict(kf_id=np.ascontiguousarray(kf_id, np.uint32), kf_pose=np.ascontiguousarray(sc["pose_cw"], np.float64), kf_flags=np.ascontiguousarray(kf_flags, np.uint8),
             lm_id=np.ascontiguousarray(lm_id, np.uint32), lm_pos=np.ascontiguousarray(sc["points"], np.float64), lm_erased=np.ascontiguousarray(lm_erased, np.uint8),
             obs_kf=np.ascontiguousarray(sc["obs_pose"], np.int32), obs_lm=np.ascontiguousarray(sc["obs_point"], np.int32), obs_idx=idx, uv=uv, xr=xr, oct=octv,
             covis=np.ascontiguousarray(covis, np.int32))
    out = dict(counts=np.zeros(3, np.int32), pose_order=np.full(K, -1, np.int32), point_order=np.full(L, -1, np.int32), edge_order=np.full(2 * E, -1, np.int32),
               kf_pose=np.zeros((K, 12)), lm_pos=np.zeros((L, 3)), n_erased=C.c_int(0), erased=np.zeros(2 * E + 2, np.int32), lm_cnt=np.zeros((L, 4), np.int32),
               kf_set=np.zeros(K, np.int32), iters=np.zeros(2, np.int32), stop=np.zeros(1, np.uint8))
    rc = ref.svref_local_ba(0, stereo, 1280, 720, _p(intr), C.c_float(1.2), 8, K, _p(a["kf_id"]), _p(a["kf_pose"]), _p(a["kf_flags"]), L, _p(a["lm_id"]),
                            _p(a["lm_pos"]), _p(a["lm_erased"]), E, _p(a["obs_kf"]), _p(a["obs_lm"]), _p(a["obs_idx"]), _p(a["uv"]), _p(a["xr"]), _p(a["oct"]),
                            curr, len(covis), _p(a["covis"]), threshold, int(use_additional), iters[0], iters[1], stop_in, _p(out["counts"]),
                            _p(out["pose_order"]), _p(out["point_order"]), _p(out["edge_order"]), _p(out["kf_pose"]), _p(out["lm_pos"]), C.byref(out["n_erased"]),
                            _p(out["erased"]), _p(out["lm_cnt"]), _p(out["kf_set"]), _p(out["iters"]), _p(out["stop"]))
    assert rc == 0
    out["octave"] = octv
    return out


def _expected_sets(sc, kf_id, kf_flags, lm_erased, covis, curr, threshold, use_additional, stereo):
    """The gather rules of local_bundle_adjuster_g2o.cc:38-147, restated on index sets."""
    local = {curr}
    for c in covis:
        if c < 0 or kf_flags[c] & 1 or kf_flags[c] & 2 or kf_id[c] < threshold:
            continue
        local.add(int(c))
    lms = set()
    for e in range(len(sc["obs_pose"])):
        if int(sc["obs_pose"][e]) in local and not lm_erased[sc["obs_point"][e]]:
            lms.add(int(sc["obs_point"][e]))
    fixed = set()
    for e in range(len(sc["obs_pose"])):
        k = int(sc["obs_pose"][e])
        if int(sc["obs_point"][e]) in lms and k not in local and not (kf_flags[k] & 1):
            fixed.add(k)
    return local, fixed, lms


@pytest.mark.parametrize("stereo", [0, 1])
@pytest.mark.parametrize("case", ["plain", "threshold_root_erased", "stopped_before"])
def test_local_ba_against_the_reference(ref, stereo, case):
    sc = S.ba_scene(num_kf=12, num_lm=600, obs_per_lm=4, num_fixed=0, seed=90 + stereo, stereo=bool(stereo))
    K, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    rng = np.random.default_rng(17 + stereo)
    kf_id = 10 + 3 * np.arange(K)
    lm_id = 1000 + 7 * rng.permutation(L)
    kf_flags = np.zeros(K, np.uint8)
    lm_erased = (rng.uniform(size=L) < 0.03).astype(np.uint8)
    curr = K - 1
    covis = [int(c) for c in rng.permutation(K - 1)[:7]]
    threshold = 0
    if case == "threshold_root_erased":
        covis += [-1]
        kf_flags[covis[0]] |= 1      # a covisibility that will be erased
        kf_flags[covis[1]] |= 2      # the spanning root stays out of the local set
        threshold = int(kf_id[sorted(covis[2:7])[0]]) + 1   # the lowest-id remaining covisibility falls under the map's fixed threshold
    stop_in = 1 if case == "stopped_before" else 0
    out = _run(ref, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, covis, curr, threshold, False, stop_in=stop_in)
    if case == "stopped_before":   # :308-310: nothing is touched
        assert out["counts"].tolist() == [0, 0, 0] and out["kf_set"].sum() == 0 and out["n_erased"].value == 0
        np.testing.assert_array_equal(out["kf_pose"], sc["pose_cw"])
        np.testing.assert_array_equal(out["lm_pos"], sc["points"])
        return
    P, Lc, Ec = out["counts"]
    local, fixed, lms = _expected_sets(sc, kf_id, kf_flags, lm_erased, covis, curr, threshold, False, stereo)
    po = out["pose_order"][:P]
    got_local = {int(x) for x in po if not (x >> 30) & 1}
    got_fixed = {int(x & ~(1 << 30)) for x in po if (x >> 30) & 1}
    assert got_local == local and got_fixed == fixed
    assert {int(x) for x in out["point_order"][:Lc]} == lms
    # the flat problem in the order the reference built its graph, through the oracle's own local BA
    pidx = [int(x & ~(1 << 30)) for x in po]
    lidx = [int(x) for x in out["point_order"][:Lc]]
    pos_of_kf = {k: i for i, k in enumerate(pidx)}
    pos_of_lm = {l: i for i, l in enumerate(lidx)}
    key = {(int(sc["obs_pose"][e]), int(sc["obs_point"][e])): e for e in range(E)}
    eo = out["edge_order"][:2 * Ec].reshape(-1, 2)
    src = np.array([key[(int(k), int(l))] for k, l in eo])
    inv_sigma = O.scale_tables(1.2, 8)[3]
    flat = dict(pose_cw=sc["pose_cw"][pidx], pose_fixed=np.array([(x >> 30) & 1 for x in po], np.uint8), points=sc["points"][lidx],
                obs_pose=np.array([pos_of_kf[int(k)] for k in eo[:, 0]], np.int32), obs_point=np.array([pos_of_lm[int(l)] for l in eo[:, 1]], np.int32),
                obs_uvr=sc["obs_uvr"][src], obs_inv_sigma_sq=np.array([inv_sigma[o] for o in out["octave"][src]], np.float32),
                obs_huber=np.full(Ec, np.sqrt(np.float32(7.81473)) if stereo else np.sqrt(np.float32(5.99146)), np.float32),
                intr=np.tile(sc["intr"][0], (P, 1)))
    orc = O.local_ba(flat, iters1=5, iters2=10)
    assert [int(orc["stats"][2]), int(orc["stats"][3])] == out["iters"].tolist()
    for i, k in enumerate(pidx):
        if (po[i] >> 30) & 1:
            np.testing.assert_array_equal(out["kf_pose"][k], sc["pose_cw"][k])   # fixed keyframes are not written
            assert out["kf_set"][k] == 0
        else:
            np.testing.assert_allclose(out["kf_pose"][k], orc["pose_cw"][i], rtol=0, atol=1e-15)
            assert out["kf_set"][k] == 1
    np.testing.assert_array_equal(out["lm_pos"][lidx], orc["points"])
    erased = {(int(out["erased"][2 * i]), int(out["erased"][2 * i + 1])) for i in range(out["n_erased"].value)}
    assert erased == {(int(eo[e, 0]), int(eo[e, 1])) for e in np.flatnonzero(orc["outlier"])}
    assert len(erased) > 0
    for l in range(L):
        n_out = sum(1 for (_, ll) in erased if ll == l)
        if l in lms:   # set_pos_in_world + one geometry refresh, plus descriptor + geometry per erased observation (:381-385, :402-403)
            assert out["lm_cnt"][l].tolist() == [1, 1 + n_out, n_out, n_out]
        else:
            assert out["lm_cnt"][l].tolist() == [0, 0, 0, 0]


def test_two_additional_fixed_keyframes_for_a_monocular_map(ref):
    """use_additional_keyframes_for_monocular (:135-147): with fewer than two fixed keyframes the first local ones (unordered_map begin())
    become fixed -- which ones is unspecified, how many is not."""
    sc = S.ba_scene(num_kf=6, num_lm=300, obs_per_lm=4, num_fixed=0, seed=97)
    K = len(sc["pose_cw"])
    kf_id, lm_id = 5 + np.arange(K), 100 + np.arange(len(sc["points"]))
    for use in (False, True):
        out = _run(ref, sc, 0, kf_id, np.zeros(K, np.uint8), lm_id, np.zeros(len(sc["points"]), np.uint8), list(range(K - 1)), K - 1, 0, use)
        po = out["pose_order"][:out["counts"][0]]
        assert sum(1 for x in po if (x >> 30) & 1) == (2 if use else 0) and len(po) == K


def _run_global(ref, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, order, num_iter, use_huber, stop_in):
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
             order=np.ascontiguousarray(order, np.int32))
    out = dict(counts=np.zeros(3, np.int32), pose_order=np.full(K, -1, np.int32), point_order=np.full(L, -1, np.int32), edge_order=np.full(2 * E, -1, np.int32),
               kf_pose=np.zeros((K, 12)), lm_pos=np.zeros((L, 3)), kf_opt=np.zeros(K, np.uint8), lm_opt=np.zeros(L, np.uint8), iters=np.zeros(2, np.int32),
               stop=np.zeros(1, np.uint8), octave=octv)
    ref.svref_global_ba.restype = C.c_int
    out["ok"] = ref.svref_global_ba(0, stereo, 1280, 720, _p(intr), C.c_float(1.2), 8, K, _p(a["kf_id"]), _p(a["kf_pose"]), _p(a["kf_flags"]), L, _p(a["lm_id"]),
                                    _p(a["lm_pos"]), _p(a["lm_erased"]), E, _p(a["obs_kf"]), _p(a["obs_lm"]), _p(a["obs_idx"]), _p(a["uv"]), _p(a["xr"]), _p(a["oct"]),
                                    len(order), _p(a["order"]), num_iter, int(use_huber), stop_in, _p(out["counts"]), _p(out["pose_order"]), _p(out["point_order"]),
                                    _p(out["edge_order"]), _p(out["kf_pose"]), _p(out["lm_pos"]), _p(out["kf_opt"]), _p(out["lm_opt"]), _p(out["iters"]), _p(out["stop"]))
    return out


@pytest.mark.parametrize("stereo,use_huber,num_iter,messy", [(0, True, 10, False), (1, False, 10, False), (0, True, 10, True), (0, True, 30, False)])
def test_global_ba_against_the_reference(ref, stereo, use_huber, num_iter, messy):
    """optimize/global_bundle_adjuster.cc compiled from the reference: landmarks gathered from the keyframes first seen first, the
    spanning root fixed, erased keyframes / landmarks left out, a landmark without an edge removed again (is_optimized_lm), the Huber
    kernels optional, ONE optimize(num_iter), the result maps -- against orc_local_ba with no second stage on the same flat problem."""
    sc = S.ba_scene(num_kf=14, num_lm=700, obs_per_lm=4, num_fixed=0, seed=70 + stereo + num_iter, stereo=bool(stereo))
    K, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    rng = np.random.default_rng(5 + num_iter)
    kf_id, lm_id = 3 + 2 * np.arange(K), 500 + rng.permutation(L)
    kf_flags, lm_erased = np.zeros(K, np.uint8), np.zeros(L, np.uint8)
    kf_flags[4] |= 2   # the spanning root
    if messy:
        kf_flags[7] |= 1
        lm_erased[rng.uniform(size=L) < 0.05] = 1
    order = rng.permutation(K)
    out = _run_global(ref, sc, stereo, kf_id, kf_flags, lm_id, lm_erased, order, num_iter, use_huber, 0)
    P, Lc, Ec = out["counts"]
    po = out["pose_order"][:P]
    pidx = [int(x & ~(1 << 30)) for x in po]
    assert pidx == [int(k) for k in order if not kf_flags[k] & 1]              # keyframes in the caller's order, erased ones skipped
    assert [int((x >> 30) & 1) for x in po] == [int(kf_flags[k] >> 1) for k in pidx]   # only the spanning root is fixed
    # landmarks: first seen first over the keyframes' keypoint lists, erased ones and those left without an edge dropped
    first_seen, seen = [], set()
    per_kf = {k: [] for k in range(K)}
    for e in range(E):
        per_kf[int(sc["obs_pose"][e])].append(int(sc["obs_point"][e]))
    for k in order:
        for l in per_kf[int(k)]:
            if l not in seen and not lm_erased[l]:
                seen.add(l)
                first_seen.append(l)
    with_edge = {int(sc["obs_point"][e]) for e in range(E) if not kf_flags[sc["obs_pose"][e]] & 1}
    lidx = [int(x) for x in out["point_order"][:Lc]]
    assert lidx == [l for l in first_seen if l in with_edge]
    pos_of_kf, pos_of_lm = {k: i for i, k in enumerate(pidx)}, {l: i for i, l in enumerate(lidx)}
    key = {(int(sc["obs_pose"][e]), int(sc["obs_point"][e])): e for e in range(E)}
    eo = out["edge_order"][:2 * Ec].reshape(-1, 2)
    src = np.array([key[(int(k), int(l))] for k, l in eo])
    inv_sigma = O.scale_tables(1.2, 8)[3]
    hub = (np.sqrt(np.float32(7.81473)) if stereo else np.sqrt(np.float32(5.99146))) if use_huber else np.float32(0)
    flat = dict(pose_cw=sc["pose_cw"][pidx], pose_fixed=np.array([(x >> 30) & 1 for x in po], np.uint8), points=sc["points"][lidx],
                obs_pose=np.array([pos_of_kf[int(k)] for k in eo[:, 0]], np.int32), obs_point=np.array([pos_of_lm[int(l)] for l in eo[:, 1]], np.int32),
                obs_uvr=sc["obs_uvr"][src], obs_inv_sigma_sq=np.array([inv_sigma[o] for o in out["octave"][src]], np.float32),
                obs_huber=np.full(Ec, hub, np.float32), intr=np.tile(sc["intr"][0], (P, 1)))
    flag = np.zeros(1, np.uint8)
    orc = O.local_ba(flat, iters1=num_iter, iters2=0, stop=flag)
    assert out["ok"] == 1 and int(orc["stats"][2]) == out["iters"][0]
    assert out["stop"][0] == flag[0]                       # the gain rule writes through the caller's flag; the result is kept all the same
    if num_iter == 30:
        assert flag[0] == 1 and out["iters"][0] < 30
    for i, k in enumerate(pidx):
        np.testing.assert_allclose(out["kf_pose"][k], orc["pose_cw"][i], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(out["lm_pos"][lidx], orc["points"])
    assert out["kf_opt"].tolist() == [0 if kf_flags[k] & 1 else 1 for k in range(K)]
    assert out["lm_opt"].tolist() == [1 if l in pos_of_lm else 0 for l in range(L)]


def test_global_ba_stopped_by_the_caller_is_discarded(ref):
    """global_bundle_adjuster.cc:341-343: force_stop_flag raised by the caller (not by the gain rule) -> false, nothing reported."""
    sc = S.ba_scene(num_kf=8, num_lm=300, obs_per_lm=4, num_fixed=0, seed=61)
    K, L = len(sc["pose_cw"]), len(sc["points"])
    flags = np.zeros(K, np.uint8)
    flags[0] = 2
    out = _run_global(ref, sc, 0, 1 + np.arange(K), flags, 1 + np.arange(L), np.zeros(L, np.uint8), np.arange(K), 10, True, 1)
    assert out["ok"] == 0 and out["iters"][0] == 0 and out["kf_opt"].sum() == 0 and out["lm_opt"].sum() == 0
