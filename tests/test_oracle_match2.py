"""The per-method matcher oracles (oracle/match2_oracle.c: literal restatements of match/robust.cc, bow_tree.cc, projection.cc,
fuse.cc) against an INDEPENDENT composition of the generic, already pinned pieces -- numpy reprojection, the grid lookup of
data/common.cc (oracle get_keypoints_in_cell, pinned on the reference's own cell-index vectors), and the generic candidate-list
scan (oracle match_candidates, pinned against pure-Python loops in test_oracle_match.py).  Two different decompositions of the
same reference text have to produce the same match lists."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import match_problems as MP


@pytest.fixture(scope="module")
def sc():
    return MP.scene(seed=7)


@pytest.fixture(scope="module")
def sc_stereo():
    return MP.scene(seed=9, stereo=True)


def _project(cam, R, t, pw):
    pc = pw @ np.asarray(R).T + t
    z = pc[:, 2]
    ok = z > 0
    zi = 1.0 / np.where(ok, z, 1.0)
    u, v = cam.fx * pc[:, 0] * zi + cam.cx, cam.fy * pc[:, 1] * zi + cam.cy
    xr = (u - cam.focal_x_baseline * zi).astype(np.float32)
    ok &= (cam.min_x < u) & (u < cam.max_x) & (cam.min_y < v) & (v < cam.max_y)
    return ok, np.stack([u, v], 1), xr


def _pred_level(max_valid, dist, T):
    ratio = (max_valid / np.float32(dist)).astype(np.float32)
    lvl = np.ceil(np.log(ratio).astype(np.float32) / np.float32(T["log_scale_factor"])).astype(int)
    return np.clip(lvl, 0, T["num_levels"] - 1)


def _csr_from_cells(cam, xy, octave, q_ok, q_xy, q_margin, q_lo, q_hi):
    bounds = (cam.min_x, cam.max_x, cam.min_y, cam.max_y)
    kx, ky = np.ascontiguousarray(xy[:, 0]), np.ascontiguousarray(xy[:, 1])
    off, items = O.assign_keypoints_to_grid(kx, ky, bounds)
    cand_off, cand = [0], []
    for q in range(len(q_ok)):
        if q_ok[q]:
            cand.extend(O.get_keypoints_in_cell(kx, ky, octave, off, items, bounds, np.float32(q_xy[q, 0]), np.float32(q_xy[q, 1]), np.float32(q_margin[q]),
                                                int(q_lo[q]), int(q_hi[q])).tolist())
        cand_off.append(len(cand))
    return np.array(cand_off, np.int32), np.array(cand, np.int32)


@pytest.mark.parametrize("stereo", [False, True])
def test_current_and_last_frames(sc, sc_stereo, stereo):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    kw = MP.current_and_last(s)
    for check_ori in (True, False):
        got, num = O.match_current_and_last_frames(check_ori, cam, **kw)
        assert num == (got >= 0).sum() > 300
        # independent composition; lm_has_observation = all ones here (the non-blocking case is exercised against the device path)
        kw1 = dict(kw, lm_has_observation=None)
        got1, _ = O.match_current_and_last_frames(check_ori, cam, **kw1)
        T = s["tables"]
        ok, uv, xr = _project(cam, kw["rot_cw"], kw["trans_cw"], kw["pos_w"])
        ok &= kw["valid"] > 0
        lv = kw["octave_last"]
        fwd = bwd = False
        if stereo:
            twc = -np.asarray(kw["rot_cw"]).T @ kw["trans_cw"]
            tlc = np.asarray(kw["rot_lw"]) @ twc + kw["trans_lw"]
            fwd, bwd = tlc[2] > kw["true_baseline"], -tlc[2] > kw["true_baseline"]
        lo = lv if fwd else np.maximum(0, lv - 1)
        hi = lv if (bwd and not fwd) else np.minimum(T["num_levels"] - 1, lv + 1)
        qm = (np.float32(kw["margin"]) * T["scale_factors"][lv]).astype(np.float32)
        off, cand = _csr_from_cells(cam, kw["t_xy"], kw["t_octave"], ok, uv, qm, lo, hi)
        ref = O.match_candidates(kw["lm_desc"], kw["tdesc"], off, cand, q_valid=ok.astype(np.uint8), occupied=kw["occupied"], q_angle=kw["angle_last"],
                                    t_angle=kw["t_angle"], check_orientation=check_ori, q_xright=xr if stereo else None,
                                    t_xright=kw["t_xright"], q_xr_tol=qm if stereo else None, thr=100, mode=0)
        assert np.array_equal(got1, ref)
    # a landmark without observation leaves its keypoint open: some keypoint must be claimed twice
    got, _ = O.match_current_and_last_frames(True, cam, **kw)
    taken = got[got >= 0]
    assert len(np.unique(taken)) <= len(taken)


@pytest.mark.parametrize("reproj", [False, True])
@pytest.mark.parametrize("stereo", [False, True])
def test_fuse_detect_duplication(sc, sc_stereo, stereo, reproj):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    kw = MP.fuse(s, do_reprojection_matching=reproj)
    got, num = O.fuse_detect_duplication(cam, **kw)
    assert num == (got >= 0).sum() > 300
    assert len(np.unique(got[got >= 0])) == num   # already_matched_idx_in_keyfrm: a keypoint fuses once
    T = s["tables"]
    R, t = np.asarray(kw["rot_cw"]), np.asarray(kw["trans_cw"])
    ok, uv, xr = _project(cam, R, t, kw["pos_w"])
    ok &= kw["valid"] > 0
    v = kw["pos_w"] - (-R.T @ t)
    dist = np.linalg.norm(v, axis=1)
    ok &= ~((dist < (1.0 / 1.3) * kw["min_valid_dist"].astype(np.float64)) | (1.3 * kw["max_valid_dist"].astype(np.float64) < dist))
    ok &= ~((v * kw["mean_normal"]).sum(1) < 0.5 * dist)
    lv = _pred_level(kw["max_valid_dist"], dist, T)
    qm = (np.float32(kw["margin"]) * T["scale_factors"][lv]).astype(np.float32)
    off, cand = _csr_from_cells(cam, kw["t_xy"], kw["t_octave"], ok, uv, qm, np.maximum(0, lv - 1), np.minimum(T["num_levels"] - 1, lv + 1))
    skip = np.zeros(len(cand), np.uint8)
    if reproj:
        for q in range(len(ok)):
            for c in range(off[q], off[q + 1]):
                k = cand[c]
                ex, ey = uv[q, 0] - float(kw["t_xy"][k, 0]), uv[q, 1] - float(kw["t_xy"][k, 1])
                w = float(T["inv_level_sigma_sq"][kw["t_octave"][k]])
                if stereo and kw["t_xright"][k] >= 0:
                    exr = np.float32(xr[q]) - np.float32(kw["t_xright"][k])
                    skip[c] = float(np.float32(7.81473)) < (ex * ex + ey * ey + float(np.float32(exr * exr))) * w
                else:
                    skip[c] = float(np.float32(5.99146)) < (ex * ex + ey * ey) * w
    ref = O.match_candidates(kw["lm_desc"], kw["tdesc"], off, cand, cand_skip=skip, q_valid=ok.astype(np.uint8), thr=50, mode=0, check_orientation=False)
    assert np.array_equal(got, ref)


def _bucket_csr(node1, node2, valid1, order_only=False):
    """(query order, CSR) of the bow merge-join: queries in (node, index) order, candidates = same-node keypoints in index order."""
    order = [i for i in np.lexsort((np.arange(len(node1)), node1)) if node1[i] >= 0]
    buckets = {}
    for j, nd in enumerate(node2):
        if nd >= 0:
            buckets.setdefault(int(nd), []).append(j)
    order = [i for i in order if int(node1[i]) in buckets]
    off, cand = [0], []
    for i in order:
        if valid1 is None or valid1[i]:
            cand.extend(buckets[int(node1[i])])
        off.append(len(cand))
    return np.array(order), np.array(off, np.int32), np.array(cand, np.int32)


@pytest.mark.parametrize("keyframes", [False, True])
def test_bow_match(sc, keyframes):
    kw = MP.bow(sc, keyframes=keyframes)
    for check_ori in (True, False):
        got, num = O.bow_match(0.75, check_ori, **kw)
        assert num == (got >= 0).sum() > 200
        order, off, cand = _bucket_csr(kw["node1"], kw["node2"], kw["valid1"])
        skip = None
        if keyframes:
            skip = (kw["valid2"][cand] == 0).astype(np.uint8)
        ref_rows = O.match_candidates(kw["desc1"][order], kw["desc2"], off, cand, cand_skip=skip, occupied=kw.get("occupied2"),
                                         q_angle=kw["angle1"][order], t_angle=kw["angle2"], check_orientation=check_ori, thr=50, mode=2, lowe_ratio=0.75)
        ref = np.full(len(got), -1, np.int32)
        ref[order] = ref_rows
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("with_nodes", [False, True])
@pytest.mark.parametrize("stereo", [False, True])
def test_match_for_triangulation(sc, sc_stereo, stereo, with_nodes):
    s = sc_stereo if stereo else sc
    cam = MP.make_cams(s, "oracle")
    kw = MP.triangulation(s, lambda R, t, c: O.reproject_to_bearing(cam, R, t, c), with_nodes=with_nodes)
    got, num = O.match_for_triangulation(0.8, True, **kw)
    assert num == (got >= 0).sum() > (30 if with_nodes else 100)
    assert len(np.unique(got[got >= 0])) == num
    # independent composition: the static pair gates evaluated here, the dynamic rule by the generic TRIANGULATION scan
    n1, n2 = len(kw["desc1"]), len(kw["desc2"])
    if with_nodes:
        order, off, cand = _bucket_csr(kw["node1"], kw["node2"], 1 - kw["has_lm1"])
    else:
        order = np.arange(n1)
        off = np.concatenate([[0], np.cumsum(np.where(kw["has_lm1"] == 0, n2, 0))]).astype(np.int32)
        cand = np.tile(np.arange(n2, dtype=np.int32), int((kw["has_lm1"] == 0).sum()))
    rows = np.repeat(np.arange(len(order)), np.diff(off))
    q = order[rows]
    b1, b2 = kw["bearings1"][q], kw["bearings2"][cand]
    epl = b2 @ np.asarray(kw["E_12"]).T
    cosr = np.clip((epl * b1).sum(1) / np.linalg.norm(epl, axis=1), -1.0, 1.0)
    thr = (kw["scale_factors"][kw["octave1"][q]] * np.float32(kw["residual_rad_thr"])).astype(np.float64)
    resid = np.abs(np.pi / 2.0 - np.arccos(cosr))
    near = np.abs(resid - thr) < 1e-9   # libm acos vs numpy arccos may differ in the last ulp: none may sit on the threshold
    assert not near.any()
    skip = ~(resid < thr)
    skip |= kw["has_lm2"][cand] > 0
    if kw["valid_epipole"]:
        st1 = np.zeros(len(q), bool) if kw["xright1"] is None else kw["xright1"][q] >= 0
        st2 = np.zeros(len(q), bool) if kw["xright2"] is None else kw["xright2"][cand] >= 0
        skip |= (~st1 & ~st2) & (0.99862953475 < b2 @ np.asarray(kw["epipole_in_2"]))
    ref_rows = O.match_candidates(kw["desc1"][order], kw["desc2"], off, cand, cand_skip=skip.astype(np.uint8), q_angle=kw["angle1"][order],
                                     t_angle=kw["angle2"], check_orientation=True, thr=50, mode=3, lowe_ratio=0.8)
    ref = np.full(n1, -1, np.int32)
    ref[order] = ref_rows
    assert np.array_equal(got, ref)


def test_projection_variants_run_and_respect_their_gates(sc):
    """match_frame_and_keyframe / match_by_Sim3_transform / match_keyframes_mutually share the reprojection + grid + best-only core of
    the two methods composed above; here: structural properties of their own outputs."""
    cam = MP.make_cams(sc, "oracle")
    kw = MP.frame_and_keyframe(sc)
    got, num = O.match_frame_and_keyframe_projection(True, cam, **kw)
    assert num == (got >= 0).sum() > 300 and len(np.unique(got[got >= 0])) == num
    assert not kw["occupied"][got[got >= 0]].any() and kw["valid"][got >= 0].all()
    strict, _ = O.match_frame_and_keyframe_projection(True, cam, **dict(kw, hamm_dist_thr=20))
    assert 0 < (strict >= 0).sum() < num
    kw = MP.by_sim3(sc)
    got, num = O.match_by_sim3_transform(cam, **kw)
    assert num == (got >= 0).sum() > 300 and len(np.unique(got[got >= 0])) == num and not kw["occupied"][got[got >= 0]].any()
    same, _ = O.match_by_sim3_transform(cam, **dict(kw, sim3_cw=kw["sim3_cw"] * np.array([[2.0], [2.0], [2.0], [1.0]])))
    assert np.array_equal(same, got)   # the Sim3 scale drops out of the SE3 it is converted to
    kw = MP.mutually(sc)
    m21, m12, mut, num = O.match_keyframes_mutually(cam, cam, **kw)
    assert num == (mut >= 0).sum() > 200
    sel = np.flatnonzero(mut >= 0)
    assert np.array_equal(m12[mut[sel]], sel) and np.array_equal(m21[sel], mut[sel])
    assert (m21 >= 0).sum() >= num and (m12 >= 0).sum() >= num
