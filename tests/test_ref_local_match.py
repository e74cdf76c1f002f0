"""Pins the per-method matcher oracles (oracle/match2_oracle.c, match_oracle.c) against the REFERENCE's own matcher code: match/robust.cc,
bow_tree.cc, projection.cc, fuse.cc and area.cc are compiled where they lie over stand-in Eigen / data:: / camera:: headers
(oracle/ref_local/shim; the grid lookup and the camera reprojection behind them are the oracle's) into oracle/_ref/libsvref.so, the
fixtures of oracle/ref_local/ref_match_exports.cc build the object graph each method takes, and the match lists must be identical.
Where the reference writes into a shared structure (frame landmarks), the oracle's per-query list is replayed into the same structure."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import match_problems as MP

_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libsvref.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(_SO):
        pytest.skip("oracle/_ref/libsvref.so absent: it is built from /root/reference by `make -C oracle/ref_local` (build container only)")
    return C.CDLL(_SO)


@pytest.fixture(scope="module")
def sc():
    return MP.scene(seed=7)


@pytest.fixture(scope="module")
def sc_stereo():
    return MP.scene(seed=9, stereo=True)


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _c(a, t):
    return None if a is None else np.ascontiguousarray(a, t)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def _replay(out, n_targets, occupied=None):
    """per-query target list -> who holds every target at the end (later queries overwrite; -2 = the initial occupant)"""
    holder = np.full(n_targets, -1, np.int32)
    if occupied is not None:
        holder[np.asarray(occupied) != 0] = -2
    for q, t in enumerate(out):
        if t >= 0:
            holder[t] = q
    return holder


@pytest.mark.parametrize("ratio,ori", [(0.8, True), (0.75, False), (0.6, True)])
def test_brute_force_match(ref, sc, ratio, ori):
    v1, v2 = sc["views"]
    valid2 = (v2["lm"] >= 0).astype(np.uint8)
    for valid in (None, valid2):
        exp = O.brute_force_match(v1["desc"], v1["angle"], v2["desc"], v2["angle"], valid, ratio, ori)
        d1, d2, a1, a2 = _c(v1["desc"], np.uint8), _c(v2["desc"], np.uint8), _c(v1["angle"], np.float32), _c(v2["angle"], np.float32)
        out = np.full(len(d1), -1, np.int32)
        num = ref.svref_brute_force_match(_p(d1), _p(a1), len(d1), _p(d2), _p(a2), _p(valid), len(d2), C.c_float(ratio), int(ori), _p(out))
        assert num == (exp >= 0).sum() > 100 and np.array_equal(out, exp)


@pytest.mark.parametrize("keyframes", [0, 1])
def test_robust_wrappers(ref, sc, keyframes):
    """robust::match_frame_and_keyframe (match/robust.cc:194-230) and robust::match_keyframes (:148-192): brute force, then the essential-matrix
    RANSAC, then the landmark assignment.  The RANSAC is the scripted stand-in of oracle/ref_local/shim_data (inlier iff (7 i1 + 3 i2) % mod != 0),
    so what is pinned is everything around it: the match list handed to the solver, its arguments (1000 iterations + recompute / 50 without,
    the seed flag), the early return for an invalid solution, the landmarks written."""
    v1, v2 = sc["views"]
    valid2 = (v2["lm"] >= 0).astype(np.uint8)
    d1, d2, a1, a2 = _c(v1["desc"], np.uint8), _c(v2["desc"], np.uint8), _c(v1["angle"], np.float32), _c(v2["angle"], np.float32)
    ratio, ori = 0.8, True
    bf = O.brute_force_match(d1, a1, d2, a2, valid2, ratio, ori)
    n_bf = int((bf >= 0).sum())
    assert n_bf > 100
    for validate, fixed_seed, s_valid, s_mod in ((1, 1, 1, 3), (1, 0, 1, 1), (1, 1, 0, 3), (0, 0, 1, 3)):
        if not keyframes and not validate:
            continue  # match_frame_and_keyframe always validates
        out = np.full(len(d1), -7, np.int32)
        rec = np.zeros(8, np.int32)
        num = ref.svref_robust_match_wrapped(keyframes, _p(d1), _p(a1), len(d1), _p(d2), _p(a2), _p(valid2), len(d2), C.c_float(ratio), int(ori), validate,
                                             fixed_seed, s_valid, s_mod, _p(out), _p(rec))
        exp = np.full(len(d1), -1, np.int32)
        if validate:
            assert list(rec[:7]) == [1, 50 if keyframes else 1000, 0 if keyframes else 1, fixed_seed, n_bf, len(d1), len(d2)]
            if s_valid:
                i1 = np.nonzero(bf >= 0)[0]
                keep = np.ones(len(i1), bool) if s_mod <= 1 else ((7 * i1 + 3 * bf[i1]) % s_mod != 0)
                exp[i1[keep]] = bf[i1[keep]]  # the landmark at keyframe keypoint j carries id j
        else:
            assert rec[0] == 0
            exp = bf.copy()
        assert num == int((exp >= 0).sum()) and np.array_equal(out, exp), (keyframes, validate, s_valid, s_mod)
        if validate and s_valid:
            assert num > 50


@pytest.mark.parametrize("stereo", [False, True])
@pytest.mark.parametrize("with_nodes", [False, True])
def test_match_for_triangulation(ref, sc, sc_stereo, stereo, with_nodes):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    v1, v2 = s["views"]
    kw = MP.triangulation(s, lambda R, t, c: O.reproject_to_bearing(cam, R, t, c), with_nodes=with_nodes)
    for ratio, ori in ((0.8, True), (0.6, False)):
        exp, num = O.match_for_triangulation(ratio, ori, **kw)
        n1, n2 = len(kw["desc1"]), len(kw["desc2"])
        out = np.full(n1, -1, np.int32)
        sf = np.asarray(kw["scale_factors"], np.float32)
        args = [C.byref(cam), _p(_f64(v1["rot_cw"])), _p(_f64(v1["trans_cw"])), _p(_f64(v2["rot_cw"])), _p(_f64(v2["trans_cw"]))]
        a = dict(d1=_c(kw["desc1"], np.uint8), a1=_c(kw["angle1"], np.float32), o1=_c(kw["octave1"], np.int32), b1=_f64(kw["bearings1"]),
                 h1=_c(kw["has_lm1"], np.uint8), x1=_c(kw.get("xright1"), np.float32), d2=_c(kw["desc2"], np.uint8), a2=_c(kw["angle2"], np.float32),
                 b2=_f64(kw["bearings2"]), h2=_c(kw["has_lm2"], np.uint8), x2=_c(kw.get("xright2"), np.float32), nd1=_c(kw.get("node1"), np.int32),
                 nd2=_c(kw.get("node2"), np.int32), E=_f64(kw["E_12"]))
        got = ref.svref_match_for_triangulation(*args, _p(a["d1"]), _p(a["a1"]), _p(a["o1"]), _p(a["b1"]), _p(a["h1"]), _p(a["x1"]), n1, _p(a["d2"]), _p(a["a2"]),
                                                _p(a["b2"]), _p(a["h2"]), _p(a["x2"]), n2, _p(a["nd1"]), _p(a["nd2"]), _p(a["E"]), C.c_float(float(sf[1] / sf[0])),
                                                len(sf), C.c_float(kw["residual_rad_thr"]), C.c_float(ratio), int(ori), _p(out))
        assert got == num > 20 and np.array_equal(out, exp)


@pytest.mark.parametrize("keyframes", [False, True])
def test_bow_match(ref, sc, keyframes):
    kw = MP.bow(sc, keyframes=keyframes)
    kw.pop("occupied2", None)  # an extension of the oracle's signature; the reference starts with no frame keypoint taken
    for ratio, ori in ((0.75, True), (0.9, False)):
        exp, num = O.bow_match(ratio, ori, **kw)
        d1, d2 = _c(kw["desc1"], np.uint8), _c(kw["desc2"], np.uint8)
        a1, a2 = _c(kw["angle1"], np.float32), _c(kw["angle2"], np.float32)
        v1, v2 = _c(kw["valid1"], np.uint8), _c(kw.get("valid2"), np.uint8)
        n1, n2 = _c(kw["node1"], np.int32), _c(kw["node2"], np.int32)
        out = np.full(len(d1), -1, np.int32)
        got = ref.svref_bow_match(_p(d1), _p(a1), _p(v1), _p(n1), len(d1), _p(d2), _p(a2), _p(v2), _p(n2), len(d2), int(keyframes), C.c_float(ratio), int(ori),
                                  _p(out))
        assert got == num > 100 and np.array_equal(out, exp)


@pytest.mark.parametrize("stereo", [False, True])
def test_match_current_and_last_frames(ref, sc, sc_stereo, stereo):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    kw = MP.current_and_last(s)
    sf = np.asarray(kw["scale_factors"], np.float32)
    for ori in (True, False):
        exp, num = O.match_current_and_last_frames(ori, cam, **kw)
        a = dict(pw=_f64(kw["pos_w"]), valid=_c(kw["valid"], np.uint8), ld=_c(kw["lm_desc"], np.uint8), ol=_c(kw["octave_last"], np.int32),
                 al=_c(kw["angle_last"], np.float32), ho=_c(kw["lm_has_observation"], np.uint8), td=_c(kw["tdesc"], np.uint8), xy=_c(kw["t_xy"], np.float32),
                 to=_c(kw["t_octave"], np.int32), ta=_c(kw["t_angle"], np.float32), occ=_c(kw["occupied"], np.uint8), xr=_c(kw.get("t_xright"), np.float32))
        nt = len(a["td"])
        holder = np.full(nt, -9, np.int32)
        got = ref.svref_match_current_and_last_frames(
            C.byref(cam), _p(_f64(kw["rot_cw"])), _p(_f64(kw["trans_cw"])), _p(_f64(kw["rot_lw"])), _p(_f64(kw["trans_lw"])), int(kw["is_monocular"]),
            C.c_float(kw["true_baseline"]), len(a["pw"]), _p(a["pw"]), _p(a["valid"]), _p(a["ld"]), _p(a["ol"]), _p(a["al"]), _p(a["ho"]),
            C.c_float(float(sf[1] / sf[0])), len(sf), C.c_float(kw["margin"]), _p(a["td"]), _p(a["xy"]), _p(a["to"]), _p(a["ta"]), nt, _p(a["occ"]), _p(a["xr"]),
            64, 48, int(ori), _p(holder))
        assert got == num > 300
        assert np.array_equal(holder, _replay(exp, nt, kw["occupied"]))


def test_match_frame_and_keyframe_projection(ref, sc):
    cam = MP.make_cams(sc, "oracle")
    kw = MP.frame_and_keyframe(sc)
    sf = np.asarray(kw["scale_factors"], np.float32)
    for ori in (True, False):
        exp, num = O.match_frame_and_keyframe_projection(ori, cam, **kw)
        a = dict(pw=_f64(kw["pos_w"]), valid=_c(kw["valid"], np.uint8), mn=_c(kw["min_valid_dist"], np.float32), mx=_c(kw["max_valid_dist"], np.float32),
                 ld=_c(kw["lm_desc"], np.uint8), ak=_c(kw["angle_kf"], np.float32), td=_c(kw["tdesc"], np.uint8), xy=_c(kw["t_xy"], np.float32),
                 to=_c(kw["t_octave"], np.int32), ta=_c(kw["t_angle"], np.float32), occ=_c(kw["occupied"], np.uint8))
        nt = len(a["td"])
        holder = np.full(nt, -9, np.int32)
        got = ref.svref_match_frame_and_keyframe_projection(
            C.byref(cam), _p(_f64(kw["rot_cw"])), _p(_f64(kw["trans_cw"])), len(a["pw"]), _p(a["pw"]), _p(a["valid"]), _p(a["mn"]), _p(a["mx"]), _p(a["ld"]),
            _p(a["ak"]), C.c_float(float(sf[1] / sf[0])), len(sf), C.c_float(kw["margin"]), int(kw["hamm_dist_thr"]), _p(a["td"]), _p(a["xy"]), _p(a["to"]),
            _p(a["ta"]), nt, _p(a["occ"]), 64, 48, int(ori), _p(holder))
        assert got == num > 300
        assert np.array_equal(holder, _replay(exp, nt, kw["occupied"]))


def _sf(kw):
    sf = np.asarray(kw["scale_factors"], np.float32)
    return C.c_float(float(sf[1] / sf[0])), len(sf)


def test_match_by_sim3_transform(ref, sc):
    cam = MP.make_cams(sc, "oracle")
    kw = MP.by_sim3(sc)
    exp, num = O.match_by_sim3_transform(cam, **kw)
    a = dict(S=_f64(kw["sim3_cw"]), pw=_f64(kw["pos_w"]), valid=_c(kw["valid"], np.uint8), mn=_c(kw["min_valid_dist"], np.float32),
             mx=_c(kw["max_valid_dist"], np.float32), nrm=_f64(kw["mean_normal"]), ld=_c(kw["lm_desc"], np.uint8), td=_c(kw["tdesc"], np.uint8),
             xy=_c(kw["t_xy"], np.float32), to=_c(kw["t_octave"], np.int32), occ=_c(kw["occupied"], np.uint8))
    nt = len(a["td"])
    holder = np.full(nt, -9, np.int32)
    got = ref.svref_match_by_sim3_transform(C.byref(cam), _p(a["S"]), len(a["pw"]), _p(a["pw"]), _p(a["valid"]), _p(a["mn"]), _p(a["mx"]), _p(a["nrm"]), _p(a["ld"]),
                                            *_sf(kw), C.c_float(kw["margin"]), _p(a["td"]), _p(a["xy"]), _p(a["to"]), nt, _p(a["occ"]), 64, 48, _p(holder))
    assert got == num > 300
    assert np.array_equal(holder, _replay(exp, nt, kw["occupied"]))


def test_match_keyframes_mutually(ref, sc):
    cam = MP.make_cams(sc, "oracle")
    kw = MP.mutually(sc)
    m21, m12, mut, num = O.match_keyframes_mutually(cam, cam, **kw)
    k1, k2 = kw["kf1"], kw["kf2"]

    def side(k):
        return [_f64(k["pos_w"]), _c(k["valid"], np.uint8), _c(k["min_valid_dist"], np.float32), _c(k["max_valid_dist"], np.float32), _c(k["lm_desc"], np.uint8),
                _c(k["desc"], np.uint8), _c(k["xy"], np.float32), _c(k["octave"], np.int32)]
    s1, s2 = side(k1), side(k2)
    out = np.full(len(s1[0]), -9, np.int32)
    got = ref.svref_match_keyframes_mutually(C.byref(cam), _p(_f64(kw["rot_1w"])), _p(_f64(kw["trans_1w"])), _p(_f64(kw["rot_2w"])), _p(_f64(kw["trans_2w"])),
                                             C.c_float(kw["s_12"]), _p(_f64(kw["rot_12"])), _p(_f64(kw["trans_12"])), len(s1[0]), *[_p(x) for x in s1], len(s2[0]),
                                             *[_p(x) for x in s2], *_sf(kw), C.c_float(kw["margin"]), 64, 48, _p(out))
    assert got == num > 200
    assert np.array_equal(out, mut)


@pytest.mark.parametrize("reproj", [False, True])
@pytest.mark.parametrize("stereo", [False, True])
def test_fuse_detect_duplication(ref, sc, sc_stereo, stereo, reproj):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    kw = MP.fuse(s, do_reprojection_matching=reproj)
    exp, num = O.fuse_detect_duplication(cam, **kw)
    a = dict(pw=_f64(kw["pos_w"]), valid=_c(kw["valid"], np.uint8), mn=_c(kw["min_valid_dist"], np.float32), mx=_c(kw["max_valid_dist"], np.float32),
             nrm=_f64(kw["mean_normal"]), ld=_c(kw["lm_desc"], np.uint8), td=_c(kw["tdesc"], np.uint8), xy=_c(kw["t_xy"], np.float32),
             to=_c(kw["t_octave"], np.int32), xr=_c(kw.get("t_xright"), np.float32))
    out = np.full(len(a["pw"]), -9, np.int32)
    got = ref.svref_fuse_detect_duplication(C.byref(cam), _p(_f64(kw["rot_cw"])), _p(_f64(kw["trans_cw"])), len(a["pw"]), _p(a["pw"]), _p(a["valid"]), _p(a["mn"]),
                                            _p(a["mx"]), _p(a["nrm"]), _p(a["ld"]), *_sf(kw), C.c_float(kw["margin"]), int(reproj), _p(a["td"]), _p(a["xy"]),
                                            _p(a["to"]), _p(a["xr"]), len(a["td"]), 64, 48, _p(out))
    assert got == num > 300
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("stereo", [False, True])
def test_match_frame_and_landmarks(ref, sc, sc_stereo, stereo):
    """projection::match_frame_and_landmarks on the maps frame::can_observe fills: the oracle side is can_observe + the grid lookup + the
    ratio-within-octave candidate matcher (the composition the device entry point svgpu_match_frame_and_landmarks is tested against)."""
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    frm = s["views"][1]
    L, T = s["landmarks"], s["tables"]
    rng = np.random.default_rng(4)
    R, t = MP._perturb(frm["rot_cw"], frm["trans_cw"], rng)
    n = len(L["pos_w"])
    vis, rp, xr, lv = O.can_observe(cam, R, t, L["pos_w"], L["mean_normal"], L["min_valid_dist"], L["max_valid_dist"], 0.5, T["num_levels"], float(T["log_scale_factor"]))
    vis = (vis.astype(bool) & (rng.uniform(0, 1, n) < 0.9)).astype(np.uint8)
    occupied = (rng.uniform(0, 1, len(frm["xy"])) < 0.1).astype(np.uint8)
    margin, ratio = 5.0, 0.8
    kx, ky = np.ascontiguousarray(frm["xy"][:, 0]), np.ascontiguousarray(frm["xy"][:, 1])
    bounds = (cam.min_x, cam.max_x, cam.min_y, cam.max_y)
    off_g, items = O.assign_keypoints_to_grid(kx, ky, bounds)
    lvq = np.where(vis == 1, lv, 0)
    qm = (np.float32(margin) * T["scale_factors"][lvq]).astype(np.float32)
    cand_off, cand = [0], []
    for q in range(n):
        if vis[q]:
            cand += O.get_keypoints_in_cell(kx, ky, frm["octave"], off_g, items, bounds, float(np.float32(rp[q, 0])), float(np.float32(rp[q, 1])), float(qm[q]),
                                            max(0, int(lv[q]) - 1), min(T["num_levels"] - 1, int(lv[q]) + 1)).tolist()
        cand_off.append(len(cand))
    kwx = dict(q_xright=xr, t_xright=frm["x_right"], q_xr_tol=qm) if stereo else {}
    exp = O.match_candidates(L["desc"], frm["desc"], cand_off, cand, check_orientation=False, thr=100, lowe_ratio=ratio, mode=O.MODE_RATIO_SAME_OCTAVE,
                             t_octave=frm["octave"], q_valid=vis, occupied=occupied, **kwx)
    a = dict(vis=_c(vis, np.uint8), rp=_f64(rp), xr=_c(xr, np.float32), lv=_c(lv, np.int32), ld=_c(L["desc"], np.uint8), td=_c(frm["desc"], np.uint8),
             xy=_c(frm["xy"], np.float32), to=_c(frm["octave"], np.int32), txr=_c(frm["x_right"], np.float32) if stereo else None, occ=occupied)
    nt = len(a["td"])
    holder = np.full(nt, -9, np.int32)
    got = ref.svref_match_frame_and_landmarks(C.byref(cam), int(not stereo), n, _p(a["vis"]), _p(a["rp"]), _p(a["xr"]), _p(a["lv"]), _p(a["ld"]),
                                              C.c_float(1.2), int(T["num_levels"]), C.c_float(margin), C.c_float(ratio), _p(a["td"]), _p(a["xy"]), _p(a["to"]),
                                              _p(a["txr"]), nt, _p(a["occ"]), 64, 48, _p(holder))
    assert got == (exp >= 0).sum() > 300
    assert np.array_equal(holder, _replay(exp, nt, occupied))
    # the literal per-method oracle (oracle/match2_oracle.c orc_match_frame_and_landmarks, what bench.py's CPU tracked-frame chain times)
    lit, lnum = O.match_frame_and_landmarks(cam, vis, rp, xr, lv, L["desc"], T["scale_factors"], margin, ratio, frm["desc"], frm["xy"], frm["octave"],
                                            occupied=occupied, t_xright=frm["x_right"] if stereo else None)
    assert lnum == got and np.array_equal(lit, exp)


@pytest.mark.parametrize("check", [True, False])
def test_match_in_consistent_area(ref, sc, check):
    """area::match_in_consistent_area (the initializer's matcher): the oracle side is the grid lookup at level 0 + the sequential area mode.
    Many frame-1 keypoints compete for the same frame-2 keypoints, so later, closer queries take targets back from their holders."""
    cam = MP.make_cams(sc, "oracle")
    rng = np.random.default_rng(6)
    n1, n2 = 1500, 900
    xy2 = np.stack([rng.uniform(20, sc["width"] - 20, n2), rng.uniform(20, sc["height"] - 20, n2)], 1).astype(np.float32)
    o2 = (rng.uniform(0, 1, n2) < 0.15).astype(np.int32)   # most keypoints on level 0
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    a2 = rng.uniform(0, 360, n2).astype(np.float32)
    src = rng.integers(0, n2 // 2, n1)
    d1 = d2[src].copy()
    for i in range(n1):
        for b in rng.choice(256, int(rng.integers(0, 41)), replace=False):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    a1 = ((a2[src] + rng.normal(0, 12, n1)) % 360).astype(np.float32)
    o1 = (rng.uniform(0, 1, n1) < 0.15).astype(np.int32)
    prev = (xy2[src] + rng.normal(0, 6.0, (n1, 2))).astype(np.float32)
    margin, ratio = 30, 0.9
    kx, ky = np.ascontiguousarray(xy2[:, 0]), np.ascontiguousarray(xy2[:, 1])
    bounds = (cam.min_x, cam.max_x, cam.min_y, cam.max_y)
    off_g, items = O.assign_keypoints_to_grid(kx, ky, bounds)
    cand_off, cand = [0], []
    for q in range(n1):
        if o1[q] == 0:
            cand += O.get_keypoints_in_cell(kx, ky, o2, off_g, items, bounds, float(prev[q, 0]), float(prev[q, 1]), float(margin), 0, 0).tolist()
        cand_off.append(len(cand))
    exp = O.match_candidates(d1, d2, cand_off, cand, q_angle=a1, t_angle=a2, check_orientation=check, thr=50, lowe_ratio=ratio, mode=O.MODE_AREA)
    work = prev.copy()
    out = np.full(n1, -9, np.int32)
    got = ref.svref_match_in_consistent_area(C.byref(cam), _p(d1), _p(o1), _p(a1), n1, _p(work), _p(d2), _p(xy2), _p(o2), _p(a2), n2, margin, C.c_float(ratio),
                                             int(check), 64, 48, _p(out))
    assert got == (exp >= 0).sum() > 150
    assert np.array_equal(out, exp)
    moved = exp >= 0
    assert np.array_equal(work[moved], xy2[exp[moved]]) and np.array_equal(work[~moved], prev[~moved])


@pytest.mark.parametrize("disp,noise", [(12, 0), (31, 2)])
def test_stereo_compute(ref, disp, noise):
    """match::stereo::compute (match/stereo.cc): row bins, level / disparity gates, Hamming best, the 11 x 11 L1 patch slide, the parabola and the
    2 x median rejection -- the reference's own code on two extractor outputs (over the typed stand-in cv::Mat) against the oracle, bit for bit."""
    from stella_vslam_amd import synthetic as S
    big = S.frame(640 + 64, 480, 5)
    left, right = np.ascontiguousarray(big[:, 8:648]), np.ascontiguousarray(big[:, 8 + disp:648 + disp])
    if noise:
        rng = np.random.default_rng(disp)
        right = np.clip(right.astype(np.int16) + rng.integers(-noise, noise + 1, right.shape), 0, 255).astype(np.uint8)
    kl, dl, _, pl = O.orb_extract(left, want_pyramid=True)
    kr, dr, _, pr = O.orb_extract(right, want_pyramid=True)
    fxb, bl = 458.654 * 0.11, 0.11
    exr, edp = O.stereo_match(kl, dl, kr, dr, pl, pr, fxb, bl)
    L = len(pl)
    pl = [np.ascontiguousarray(a) for a in pl]
    pr = [np.ascontiguousarray(a) for a in pr]
    PL = (C.c_void_p * L)(*[a.ctypes.data for a in pl])
    PR = (C.c_void_p * L)(*[a.ctypes.data for a in pr])
    lw = np.array([a.shape[1] for a in pl], np.int32)
    lh = np.array([a.shape[0] for a in pl], np.int32)
    lsl = np.array([a.strides[0] for a in pl], np.int32)
    lsr = np.array([a.strides[0] for a in pr], np.int32)
    kl, kr = np.ascontiguousarray(kl), np.ascontiguousarray(kr)
    dl, dr = np.ascontiguousarray(dl, np.uint8), np.ascontiguousarray(dr, np.uint8)
    xr, dp = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
    ref.svref_stereo_compute.restype = None
    ref.svref_stereo_compute(_p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), PL, PR, _p(lw), _p(lh), _p(lsl), _p(lsr), C.c_float(1.2), L, C.c_float(fxb),
                             C.c_float(bl), _p(xr), _p(dp))
    assert (exr >= 0).sum() > 500
    assert np.array_equal(xr.view(np.uint32), exr.view(np.uint32)) and np.array_equal(dp.view(np.uint32), edp.view(np.uint32))
