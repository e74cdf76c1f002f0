"""The tracked-frame chain (svgpu_map_* / svgpu_track_*): a device-resident landmark table addressed by landmark id,
and the two halves of tracking_module's per-frame work as one submission each.  Parity is asserted three ways:
  * against the literal CPU oracle of every step (match_current_and_last_frames, pose_optimize, can_observe, match_frame_and_landmarks)
    chained exactly as module/frame_tracker.cc:22-60 and tracking_module.cc:533-608 chain them;
  * against the product's own separate entry points fed from HOST-FLATTENED landmark arrays (the path the chain replaces): identical
    match lists, outlier flags and poses, bit for bit;
  * with the fused extraction: keypoints / descriptors / undistorted keypoints / bearings equal the stand-alone extractor + frame
    observation, and the chain's result equals the chain run on the adopted frame.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import match_problems as MP

pytestmark = pytest.mark.gpu

TOL = 1e-4
CHI2 = np.float32(np.sqrt(np.float32(5.99146))), np.float32(np.sqrt(np.float32(7.81473)))


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


@pytest.fixture(scope="module")
def ctx():
    from stella_vslam_amd import feature
    return feature.Context()


def _records(view):
    from stella_vslam_amd.feature import KEYPOINT_DTYPE as KP_DTYPE
    k = np.zeros(len(view["xy"]), KP_DTYPE)
    k["x"], k["y"] = view["xy"][:, 0], view["xy"][:, 1]
    k["octave"], k["angle"] = view["octave"], view["angle"]
    return k


def _pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(3, 3), np.asarray(t, np.float64).reshape(3, 1)], 1).reshape(12)


FISHEYE = dict(fx=190.978, fy=190.973, cx=254.93, cy=256.90, k=(0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367), cols=512, rows=512)  # TUM-VI-like


def _other_cams(sc, model, ctx):
    from stella_vslam_amd import camera
    if model == "equirectangular":
        return (camera.equirectangular("theta", "Gray", sc["width"], sc["height"], 30.0, ctx=ctx), O.make_camera(O.CAM_EQUIRECTANGULAR, sc["width"], sc["height"]))
    F = FISHEYE
    return (camera.fisheye("f", "Monocular", "Gray", F["cols"], F["rows"], 30.0, F["fx"], F["fy"], F["cx"], F["cy"], *F["k"], ctx=ctx),
            O.make_camera(O.CAM_FISHEYE, F["cols"], F["rows"], F["fx"], F["fy"], F["cx"], F["cy"], F["k"]))


def _reproject_scene(sc, model, seed):
    """map_scene's two views seen through another camera model: the keypoints of the landmarks move to that model's projection of the
    landmark (+ the same pixel noise), the clutter is spread over its image.  The projection is the ORACLE's own (can_observe's reprojection
    with every gate opened), so the scene is consistent with the checker by construction; keypoints are stored UNDISTORTED, as frames hold them."""
    rng = np.random.default_rng(1000 + seed)
    if model == "equirectangular":
        sc["width"], sc["height"] = 1920, 960
    else:
        sc["width"], sc["height"] = FISHEYE["cols"], FISHEYE["rows"]
    ocam = (O.make_camera(O.CAM_EQUIRECTANGULAR, 1920, 960) if model == "equirectangular" else
            O.make_camera(O.CAM_FISHEYE, FISHEYE["cols"], FISHEYE["rows"], FISHEYE["fx"], FISHEYE["fy"], FISHEYE["cx"], FISHEYE["cy"], FISHEYE["k"]))
    L = sc["landmarks"]
    for v in sc["views"]:
        has = v["lm"] >= 0
        idx = np.where(has, v["lm"], 0)
        n = len(idx)
        # every gate of can_observe opened: normals towards the camera, an unbounded distance range
        normal = L["pos_w"][idx] - v["center"]
        normal /= np.linalg.norm(normal, axis=1, keepdims=True)
        vis, rp, _, _ = O.can_observe(ocam, v["rot_cw"], v["trans_cw"], L["pos_w"][idx], normal, np.full(n, 1e-3, np.float32), np.full(n, 1e9, np.float32), -1.0, 8,
                                      float(sc["tables"]["log_scale_factor"]))
        ok = has & (vis == 1)
        noise = rng.normal(0, 1.0, (n, 2)) * sc["tables"]["scale_factors"][v["octave"]][:, None]
        xy = np.where(ok[:, None], rp + noise, np.stack([rng.uniform(ocam.min_x + 2, ocam.max_x - 2, n), rng.uniform(ocam.min_y + 2, ocam.max_y - 2, n)], 1))
        v["xy"] = np.ascontiguousarray(xy, np.float32)
        v["lm"] = np.where(ok, v["lm"], -1)


class _World:
    """map_scene + a landmark table whose ids are NOT the scene's indices (id = 3 * index + 5), with a few landmarks erased, a few
    without descriptor and a few without observations -- the three flags the chain's gates read."""

    def __init__(self, ctx, seed, stereo, n_lm=1500, n_extra=600, model="perspective"):
        from stella_vslam_amd import data, tracking
        self.sc = sc = MP.scene(seed=seed, stereo=stereo, n_lm=n_lm, n_extra=n_extra)
        self.stereo = stereo
        self.model = model
        if model != "perspective":
            _reproject_scene(sc, model, seed)
        L = sc["landmarks"]
        n = len(L["pos_w"])
        rng = np.random.default_rng(100 + seed)
        self.ids = (3 * np.arange(n) + 5).astype(np.int32)
        flags = np.full(n, tracking.LM_PRESENT | tracking.LM_HAS_OBSERVATION | tracking.LM_HAS_DESCRIPTOR, np.uint32)
        flags[rng.uniform(size=n) < 0.03] = 0                                          # will_be_erased
        nodesc = rng.uniform(size=n) < 0.03
        flags[nodesc] &= ~np.uint32(tracking.LM_HAS_DESCRIPTOR)
        noobs = rng.uniform(size=n) < 0.15
        flags[noobs] &= ~np.uint32(tracking.LM_HAS_OBSERVATION)
        self.flags = flags
        self.rec = tracking.landmark_records(L["pos_w"], L["mean_normal"], L["min_valid_dist"], L["max_valid_dist"], L["desc"], flags)
        self.table = tracking.landmark_table(ctx).upsert(self.ids, self.rec)
        self.cam, self.ocam = (MP.make_cams(sc, "svgpu"), MP.make_cams(sc, "oracle")) if model == "perspective" else _other_cams(sc, model, ctx)
        last, cur = sc["views"]
        self.last, self.cur = last, cur
        self.rf_last = data.resident_frame(ctx).upload(self.cam, _records(last), last["desc"], last["x_right"] if stereo else None)
        self.rf_cur = data.resident_frame(ctx).upload(self.cam, _records(cur), cur["desc"], cur["x_right"] if stereo else None)
        self.last_ids = np.where(last["lm"] >= 0, 3 * last["lm"] + 5, -1).astype(np.int32)
        self.last_ids[rng.uniform(size=len(self.last_ids)) < 0.02] = 10 ** 6             # an id the table has never seen
        T = sc["tables"]
        self.T = T
        self.tracker = tracking.tracker(ctx, self.table, self.cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"], is_monocular=not stereo,
                                        true_baseline=0.11)
        R, t = MP._perturb(cur["rot_cw"], cur["trans_cw"], np.random.default_rng(seed), 0.25, 0.01)
        self.guess = _pose12(R, t)
        self.pose_last = _pose12(last["rot_cw"], last["trans_cw"])
        fx, fy, cx, cy, fxb = sc["K"]
        self.intr = np.array([fx, fy, cx, cy, fxb], np.float64)
        if model == "equirectangular":   # the row that selects the equirectangular edges (include/svgpu.h)
            self.intr = np.array([0.0, 0.0, sc["width"], sc["height"], 0.0])
        elif model == "fisheye":          # pinhole on the undistorted keypoints, with the fisheye camera's own intrinsics
            self.intr = np.array([FISHEYE["fx"], FISHEYE["fy"], FISHEYE["cx"], FISHEYE["cy"], 0.0])

    # ---- the table as the oracle / the flattened path sees it
    def lookup(self, ids):
        """per id: (index into the scene's landmarks or 0, flags or 0)"""
        ids = np.asarray(ids, np.int64)
        idx = (ids - 5) // 3
        known = (ids >= 5) & ((ids - 5) % 3 == 0) & (idx < len(self.flags))
        idx = np.where(known, idx, 0)
        return idx, np.where(known, self.flags[idx], 0).astype(np.uint32)

    def pose_problem(self, cur_lm):
        """pose_optimizer_g2o.cc:81-107: one edge per keypoint whose landmark exists, in keypoint order"""
        from stella_vslam_amd import tracking
        idx, fl = self.lookup(cur_lm)
        keep = np.flatnonzero((cur_lm >= 0) & ((fl & tracking.LM_PRESENT) != 0))
        v = self.cur
        uvr = np.stack([v["xy"][keep, 0], v["xy"][keep, 1], v["x_right"][keep] if self.stereo else np.full(len(keep), -1.0, np.float32)], 1).astype(np.float32)
        return keep, dict(pos_w=self.sc["landmarks"]["pos_w"][idx[keep]], uvr=uvr, inv_sigma_sq=self.T["inv_level_sigma_sq"][v["octave"][keep]].astype(np.float32),
                          huber=np.full(len(keep), CHI2[1 if self.stereo else 0], np.float32))


def _apply(cur_lm, match, q_ids):
    """frm.add_landmark(lm, idx) in increasing query order (a later query overwrites)"""
    out = cur_lm.copy()
    for q in np.flatnonzero(match >= 0):
        out[match[q]] = q_ids[q]
    return out


def _oracle_motion(W, margin, check_orientation=True):
    from stella_vslam_amd import tracking
    L, last, cur = W.sc["landmarks"], W.last, W.cur
    idx, fl = W.lookup(W.last_ids)
    valid = ((W.last_ids >= 0) & ((fl & tracking.LM_PRESENT) != 0) & ((fl & tracking.LM_HAS_DESCRIPTOR) != 0)).astype(np.uint8)
    has_obs = ((fl & tracking.LM_HAS_OBSERVATION) != 0).astype(np.uint8)
    G = W.guess.reshape(3, 4)
    PL = W.pose_last.reshape(3, 4)
    kw = dict(rot_cw=G[:, :3], trans_cw=G[:, 3], rot_lw=PL[:, :3], trans_lw=PL[:, 3], pos_w=L["pos_w"][idx], valid=valid, lm_desc=np.ascontiguousarray(L["desc"][idx]),
              octave_last=last["octave"], angle_last=last["angle"], scale_factors=W.T["scale_factors"], margin=margin, tdesc=cur["desc"], t_xy=cur["xy"],
              t_octave=cur["octave"], t_angle=cur["angle"], occupied=None, t_xright=cur["x_right"] if W.stereo else None, lm_has_observation=has_obs,
              is_monocular=not W.stereo, true_baseline=0.11)
    m, num = O.match_current_and_last_frames(check_orientation, W.ocam, **kw)
    cur_lm = _apply(np.full(len(cur["xy"]), -1, np.int32), m, W.last_ids)
    keep, pr = W.pose_problem(cur_lm)
    nv, pose, outl, st = O.pose_optimize(W.guess, pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], W.intr)
    outlier = np.zeros(len(cur["xy"]), np.uint8)
    outlier[keep] = outl
    return dict(kw=kw, match=m, num=num, cur_lm=cur_lm, pose=pose, outlier=outlier, nv=nv, iters=int(st[0]), n_obs=len(keep), pr=pr, keep=keep)


@pytest.mark.parametrize("stereo,margin,seed", [(False, 20.0, 3), (True, 15.0, 4)])
def test_motion_chain_equals_oracle_and_flattened_path(ctx, stereo, margin, seed):
    from stella_vslam_amd import match, optimize
    W = _World(ctx, seed, stereo)
    exp = _oracle_motion(W, margin)
    got = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, margin)
    r = got["result"]
    assert exp["num"] > 300 and r["num_matches"] == exp["num"]
    assert np.array_equal(got["match_last"], exp["match"])
    assert r["num_observations"] == exp["n_obs"] and r["num_valid"] == exp["nv"] and r["lm_iterations"] == exp["iters"]
    assert np.array_equal(got["outlier"], exp["outlier"]) and 0 < exp["outlier"].sum() < exp["n_obs"] // 2
    assert _rel(r["pose_cw"], exp["pose"]) < TOL
    # the path the chain replaces: host-flattened landmark arrays through the separate entry points -- bit-identical
    flat, fnum = match.projection_flat(0.9, True, ctx).match_current_and_last_frames(W.cam, **exp["kw"])
    assert fnum == exp["num"] and np.array_equal(flat, got["match_last"])
    pr = exp["pr"]
    nv, pose, outl, iters = optimize.pose_optimizer(ctx=ctx).optimize_flat(W.guess, pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], W.intr)
    assert nv == r["num_valid"] and iters == r["lm_iterations"] and np.array_equal(pose, r["pose_cw"])
    assert np.array_equal(outl, got["outlier"][exp["keep"]])
    assert W.tracker.counters() == (4, 1)   # the ids' upload, lists, replay, optimisation -- and ONE synchronisation
    # twice the margin (frame_tracker.cc:36-40) through the same tracker: more candidates, same agreement
    exp2 = _oracle_motion(W, 2 * margin)
    got2 = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 2 * margin)
    assert np.array_equal(got2["match_last"], exp2["match"]) and got2["result"]["num_candidates"] >= r["num_candidates"]
    assert np.array_equal(got2["outlier"], exp2["outlier"]) and _rel(got2["result"]["pose_cw"], exp2["pose"]) < TOL


def _oracle_local(W, cur_lm, local_ids, pose, margin, lowe):
    from stella_vslam_amd import tracking
    L, cur = W.sc["landmarks"], W.cur
    idx, fl = W.lookup(local_ids)
    offered = (local_ids >= 0) & ((fl & tracking.LM_PRESENT) != 0)
    P = pose.reshape(3, 4)
    vis, rp, xr, lv = O.can_observe(W.ocam, P[:, :3], P[:, 3], L["pos_w"][idx], L["mean_normal"][idx], L["min_valid_dist"][idx], L["max_valid_dist"][idx], 0.5, 8,
                                    float(W.T["log_scale_factor"]))
    vis = (vis.astype(bool) & offered).astype(np.uint8)
    cidx, cfl = W.lookup(cur_lm)
    occupied = ((cur_lm >= 0) & ((cfl & tracking.LM_HAS_OBSERVATION) != 0)).astype(np.uint8)
    q_valid = (vis.astype(bool) & ((fl & tracking.LM_HAS_DESCRIPTOR) != 0)).astype(np.uint8)
    m, num = O.match_frame_and_landmarks(W.ocam, q_valid, rp, xr, lv, np.ascontiguousarray(L["desc"][idx]), W.T["scale_factors"], margin, lowe, cur["desc"], cur["xy"],
                                         cur["octave"], occupied=occupied, t_xright=cur["x_right"] if W.stereo else None,
                                         lm_has_observation=((fl & tracking.LM_HAS_OBSERVATION) != 0).astype(np.uint8))
    new_lm = _apply(cur_lm, m, local_ids)
    keep, pr = W.pose_problem(new_lm)
    nv, pose_o, outl, st = O.pose_optimize(pose, pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], W.intr)
    outlier = np.zeros(len(cur["xy"]), np.uint8)
    outlier[keep] = outl
    return dict(vis=vis, rp=rp, xr=xr, lv=lv, match=m, num=num, pose=pose_o, outlier=outlier, nv=nv, iters=int(st[0]), n_obs=len(keep), occupied=occupied,
                q_valid=q_valid, idx=idx, fl=fl, pr=pr, keep=keep)


@pytest.mark.parametrize("stereo,margin,seed,pose_on_device", [(False, 5.0, 3, True), (True, 5.0, 4, False), (False, 15.0, 5, True)])
def test_local_map_chain_equals_oracle_and_flattened_path(ctx, stereo, margin, seed, pose_on_device):
    from stella_vslam_amd import match, optimize
    W = _World(ctx, seed, stereo)
    first = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 20.0)
    exp1 = _oracle_motion(W, 20.0)
    assert np.array_equal(first["match_last"], exp1["match"])
    # discard_outliers (frame_tracker.cc:88-110), then the local map: every landmark the frame does not hold, shuffled, a few withheld
    cur_lm = np.where(first["outlier"] == 1, -1, exp1["cur_lm"]).astype(np.int32)
    rng = np.random.default_rng(seed)
    held = set(cur_lm[cur_lm >= 0].tolist())
    local_ids = np.array([i for i in W.ids[rng.permutation(len(W.ids))] if int(i) not in held], np.int32)
    local_ids[rng.uniform(size=len(local_ids)) < 0.05] = -1
    pose1 = first["result"]["pose_cw"]
    exp = _oracle_local(W, cur_lm, local_ids, pose1, margin, 0.8)
    got = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, margin, 0.8, 0.5, pose_cw=None if pose_on_device else pose1)
    r = got["result"]
    assert np.array_equal(got["visible"], exp["vis"]) and exp["vis"].sum() > 100
    assert exp["num"] > 50 and r["num_matches"] == exp["num"] and np.array_equal(got["match_local"], exp["match"])
    assert r["num_observations"] == exp["n_obs"] and r["num_valid"] == exp["nv"] and r["lm_iterations"] == exp["iters"]
    assert np.array_equal(got["outlier"], exp["outlier"])
    assert _rel(r["pose_cw"], exp["pose"]) < TOL
    rp, xr, lv = W.tracker.local_map_observability()
    v = exp["vis"] == 1
    assert np.array_equal(rp[v], exp["rp"][v]) and np.array_equal(xr[v], exp["xr"][v]) and np.array_equal(lv[v], exp["lv"][v])
    # the path the chain replaces: reproject the flattened landmarks, then the cell matcher on the reprojections, then the optimiser
    L, cur = W.sc["landmarks"], W.cur
    P = pose1.reshape(3, 4)
    idx = exp["idx"]
    skip = (1 - ((local_ids >= 0) & ((exp["fl"] & 1) != 0))).astype(np.uint8)
    vis2, rp2, xr2, lv2 = W.cam.reproject_landmarks(P[:, :3], P[:, 3], L["pos_w"][idx], L["mean_normal"][idx], L["min_valid_dist"][idx], L["max_valid_dist"][idx], 0.5, 8,
                                                    float(W.T["log_scale_factor"]), skip=skip)
    assert np.array_equal(vis2, got["visible"])
    lvq = np.where(vis2 == 1, lv2, 0)
    sf = W.T["scale_factors"]
    qm = (np.float32(margin) * sf[lvq]).astype(np.float32)
    kw = dict(q_xright=xr2, t_xright=cur["x_right"], q_xr_tol=qm) if stereo else {}
    two, tnum = match.projection(0.8, False, ctx).match_in_cells(np.ascontiguousarray(L["desc"][idx]), rp2.astype(np.float32), qm, cur["desc"], cur["xy"], cur["octave"],
                                                                 W.cam.img_bounds_.as_tuple(), match.MATCH_RATIO_SAME_OCTAVE, 100, q_min_level=np.maximum(0, lvq - 1),
                                                                 q_max_level=np.minimum(7, lvq + 1), q_valid=exp["q_valid"], occupied=exp["occupied"],
                                                                 q_blocks=((exp["fl"] & 2) != 0).astype(np.uint8), **kw)
    assert tnum == r["num_matches"] and np.array_equal(two, got["match_local"])
    pr = exp["pr"]
    nv, pose, outl, iters = optimize.pose_optimizer(ctx=ctx).optimize_flat(pose1, pr["pos_w"], pr["uvr"], pr["inv_sigma_sq"], pr["huber"], W.intr)
    assert nv == r["num_valid"] and iters == r["lm_iterations"] and np.array_equal(pose, r["pose_cw"]) and np.array_equal(outl, got["outlier"][exp["keep"]])


@pytest.mark.parametrize("model,seed", [("equirectangular", 11), ("fisheye", 12)])
def test_chain_on_the_other_camera_models(ctx, model, seed):
    """Both halves of the chain through the equirectangular and the fisheye model (camera/equirectangular.cc, camera/fisheye.cc;
    equirectangular_pose_opt_edge.h for the optimiser's edges): the same device functions as the per-call kernels, but k_track_cand's and
    k_pose_opt<EQ, TRK>'s own instantiations.  Matches and verdicts bit-identical to the oracle, poses to the tolerance of the optimiser tests."""
    W = _World(ctx, seed, False, model=model)
    assert (W.last["lm"] >= 0).sum() > 400 and (W.cur["lm"] >= 0).sum() > 400
    margin = 20.0
    exp = _oracle_motion(W, margin)
    got = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, margin)
    r = got["result"]
    assert exp["num"] > 200 and r["num_matches"] == exp["num"] and np.array_equal(got["match_last"], exp["match"])
    assert r["num_observations"] == exp["n_obs"] and r["num_valid"] == exp["nv"] and r["lm_iterations"] == exp["iters"]
    assert np.array_equal(got["outlier"], exp["outlier"]) and _rel(r["pose_cw"], exp["pose"]) < TOL
    cur_lm = np.where(got["outlier"] == 1, -1, exp["cur_lm"]).astype(np.int32)
    rng = np.random.default_rng(seed)
    held = set(cur_lm[cur_lm >= 0].tolist())
    local_ids = np.array([i for i in W.ids[rng.permutation(len(W.ids))] if int(i) not in held], np.int32)
    pose1 = r["pose_cw"]
    exp2 = _oracle_local(W, cur_lm, local_ids, pose1, 5.0, 0.8)
    got2 = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 5.0, 0.8, 0.5)
    r2 = got2["result"]
    assert np.array_equal(got2["visible"], exp2["vis"]) and exp2["vis"].sum() > 50
    assert exp2["num"] > 20 and r2["num_matches"] == exp2["num"] and np.array_equal(got2["match_local"], exp2["match"])
    assert r2["num_valid"] == exp2["nv"] and r2["lm_iterations"] == exp2["iters"] and np.array_equal(got2["outlier"], exp2["outlier"])
    assert _rel(r2["pose_cw"], exp2["pose"]) < TOL


def test_table_written_by_the_mapping_thread_while_the_tracking_thread_reads(ctx):
    """The mapping thread refreshes landmarks (BA write-backs, new landmarks: the table GROWS, i.e. moves) on its own context while the
    tracking thread runs the chain on another: svgpu_map orders the two streams (ev_write / ev_read, a growth waits for the readers).
    200 tracked frames against 200 concurrent upserts with a reallocation every tenth: every frame returns the bits of the quiet run."""
    import threading
    from stella_vslam_amd import feature, tracking
    W = _World(ctx, 3, False)
    quiet = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 20.0)
    cur_lm = np.where(quiet["outlier"] == 1, -1, _oracle_motion(W, 20.0)["cur_lm"]).astype(np.int32)
    local_ids = np.array([i for i in W.ids if int(i) not in set(cur_lm[cur_lm >= 0].tolist())], np.int32)
    quiet2 = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 5.0, 0.8, 0.5)
    ctx2 = feature.Context()
    stop, err = threading.Event(), []
    far = tracking.landmark_records(np.full((1, 3), 1e6), np.array([[0.0, 0.0, 1.0]]), np.array([1.0], np.float32), np.array([2.0], np.float32), np.zeros((1, 32), np.uint8))

    def writer():
        try:
            k, cap = 0, W.table.capacity
            while not stop.is_set():
                W.table.upsert(W.ids, W.rec, ctx=ctx2)                       # the same records again: a write-back that changes nothing
                k += 1
                if k % 10 == 0:                                              # a new landmark far behind the cameras, beyond the capacity: the table moves
                    cap = max(cap, W.table.capacity) * 2
                    if cap < (1 << 22):
                        W.table.upsert(np.array([cap + 7], np.uint32), far, ctx=ctx2)
        except Exception as e:  # pragma: no cover
            err.append(e)

    th = threading.Thread(target=writer)
    th.start()
    try:
        for _ in range(200):
            a = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 20.0)
            assert np.array_equal(a["match_last"], quiet["match_last"]) and np.array_equal(a["outlier"], quiet["outlier"])
            assert np.array_equal(a["result"]["pose_cw"], quiet["result"]["pose_cw"])
            b = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 5.0, 0.8, 0.5)
            assert np.array_equal(b["match_local"], quiet2["match_local"]) and np.array_equal(b["visible"], quiet2["visible"])
            assert np.array_equal(b["result"]["pose_cw"], quiet2["result"]["pose_cw"])
    finally:
        stop.set()
        th.join(timeout=60)
    assert not err, err
    assert W.table.capacity > len(W.ids) * 3     # it did move while the frames ran


def test_two_readers_on_two_streams_and_a_growing_writer(ctx):
    """Two trackers on two contexts (= two streams: the two cameras of a rig, or a relocaliser beside the tracker) read the table while a third
    context keeps writing and GROWING it.  The table keeps one read event per read a writer has not yet waited for (round 4 kept ONE event: the
    later reader overwrote the earlier one's record and a growth could free the table under it).  Every frame of both readers returns the
    bits of the quiet run; a download on the second reader's context joins the reads."""
    import threading
    from stella_vslam_amd import data, feature, tracking
    W = _World(ctx, 3, False)
    quiet = W.tracker.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 20.0)
    cur_lm = np.where(quiet["outlier"] == 1, -1, _oracle_motion(W, 20.0)["cur_lm"]).astype(np.int32)
    local_ids = np.array([i for i in W.ids if int(i) not in set(cur_lm[cur_lm >= 0].tolist())], np.int32)
    quiet2 = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 5.0, 0.8, 0.5)
    ctx_r2, ctx_w = feature.Context(), feature.Context()
    T = W.T
    tracker2 = tracking.tracker(ctx_r2, W.table, W.cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"], is_monocular=True, true_baseline=0.11)
    rf_last2 = data.resident_frame(ctx_r2).upload(W.cam, _records(W.last), W.last["desc"], None)
    rf_cur2 = data.resident_frame(ctx_r2).upload(W.cam, _records(W.cur), W.cur["desc"], None)
    rec_quiet = W.table.download(W.ids[:64], ctx=ctx_r2)
    stop, err = threading.Event(), []
    far = tracking.landmark_records(np.full((1, 3), 1e6), np.array([[0.0, 0.0, 1.0]]), np.array([1.0], np.float32), np.array([2.0], np.float32), np.zeros((1, 32), np.uint8))

    def writer():
        try:
            k, cap = 0, W.table.capacity
            while not stop.is_set():
                W.table.upsert(W.ids, W.rec, ctx=ctx_w)
                k += 1
                if k % 5 == 0:
                    cap = max(cap, W.table.capacity) * 2
                    if cap < (1 << 22):
                        W.table.upsert(np.array([cap + 11], np.uint32), far, ctx=ctx_w)
        except Exception as e:  # pragma: no cover
            err.append(e)

    def reader(trk, rf_cur, rf_last, with_download):
        try:
            for i in range(150):
                a = trk.track_motion(rf_cur, rf_last, W.last_ids, W.guess, W.pose_last, 20.0)
                assert np.array_equal(a["match_last"], quiet["match_last"]) and np.array_equal(a["outlier"], quiet["outlier"])
                assert np.array_equal(a["result"]["pose_cw"], quiet["result"]["pose_cw"])
                b = trk.track_local_map(rf_cur, cur_lm, local_ids, 5.0, 0.8, 0.5)
                assert np.array_equal(b["match_local"], quiet2["match_local"]) and np.array_equal(b["visible"], quiet2["visible"])
                assert np.array_equal(b["result"]["pose_cw"], quiet2["result"]["pose_cw"])
                if with_download and i % 10 == 0:
                    assert W.table.download(W.ids[:64], ctx=ctx_r2).tobytes() == rec_quiet.tobytes()
        except Exception as e:  # pragma: no cover
            err.append(e)
            stop.set()

    tw = threading.Thread(target=writer)
    t2 = threading.Thread(target=reader, args=(tracker2, rf_cur2, rf_last2, True))
    tw.start()
    t2.start()
    try:
        reader(W.tracker, W.rf_cur, W.rf_last, False)
    finally:
        t2.join(timeout=120)
        stop.set()
        tw.join(timeout=60)
    assert not err, err
    assert W.table.capacity > len(W.ids) * 3     # it did move while the frames ran


def test_candidate_lists_beyond_the_first_capacity(ctx):
    """more list entries than the tracker's initial capacity (65 536): the chain notices on the device, the host grows and re-runs"""
    W = _World(ctx, 8, False, n_lm=6000, n_extra=2000)
    cur_lm = np.full(len(W.cur["xy"]), -1, np.int32)
    local_ids = W.ids.copy()
    pose = _pose12(W.cur["rot_cw"], W.cur["trans_cw"])
    got = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 40.0, 0.8, 0.5, pose_cw=pose)
    exp = _oracle_local(W, cur_lm, local_ids, pose, 40.0, 0.8)
    assert got["result"]["num_candidates"] > 65536
    assert np.array_equal(got["match_local"], exp["match"]) and np.array_equal(got["outlier"], exp["outlier"]) and _rel(got["result"]["pose_cw"], exp["pose"]) < TOL
    # (6 000 queries leave the replay ONE staged entry per list in LDS: lists longer than their staged head are walked on in global memory --
    #  the round-3 kernel stopped at the head's padding; the cell matcher of the flattened path shares that kernel)
    from stella_vslam_amd import match
    L, cur = W.sc["landmarks"], W.cur
    lvq = np.where(exp["vis"] == 1, exp["lv"], 0)
    qm = (np.float32(40.0) * W.T["scale_factors"][lvq]).astype(np.float32)
    two, _ = match.projection(0.8, False, ctx).match_in_cells(np.ascontiguousarray(L["desc"][exp["idx"]]), exp["rp"].astype(np.float32), qm, cur["desc"], cur["xy"],
                                                              cur["octave"], W.cam.img_bounds_.as_tuple(), match.MATCH_RATIO_SAME_OCTAVE, 100,
                                                              q_min_level=np.maximum(0, lvq - 1), q_max_level=np.minimum(7, lvq + 1), q_valid=exp["q_valid"],
                                                              occupied=exp["occupied"], q_blocks=((exp["fl"] & 2) != 0).astype(np.uint8))
    assert np.array_equal(two, exp["match"])
    _, syncs = W.tracker.counters()
    assert syncs == 2   # the attempt that overflowed + the one that fitted
    again = W.tracker.track_local_map(W.rf_cur, cur_lm, local_ids, 40.0, 0.8, 0.5, pose_cw=pose)
    assert np.array_equal(again["match_local"], got["match_local"]) and W.tracker.counters()[1] == 3


def test_landmark_table_upsert_erase_download(ctx):
    from stella_vslam_amd import tracking
    rng = np.random.default_rng(0)
    t = tracking.landmark_table(ctx)
    n = 700
    ids = rng.choice(5000, n, replace=False).astype(np.uint32)
    rec = tracking.landmark_records(rng.normal(size=(n, 3)), rng.normal(size=(n, 3)), rng.uniform(1, 2, n), rng.uniform(3, 9, n),
                                    rng.integers(0, 256, (n, 32), dtype=np.uint8), rng.integers(1, 8, n).astype(np.uint32))
    t.upsert(ids, rec)
    assert t.capacity > ids.max()
    back = t.download(ids)
    assert back.tobytes() == rec.tobytes()
    # ids the table never saw (inside and beyond its capacity) read as absent
    unseen = np.setdiff1d(np.arange(5000, dtype=np.uint32), ids)[:50]
    assert not t.download(np.concatenate([unseen, [10 ** 7]]).astype(np.uint32))["flags"].any()
    # a repeated id in one call: the last record wins; growth keeps what is there
    r2 = rec[:3].copy()
    r2["pos_w"] += 1.0
    t.upsert(np.array([ids[0], ids[0], 250000], np.uint32), np.array([rec[1], r2[0], r2[2]]))
    assert t.capacity > 250000
    got = t.download(np.array([ids[0], 250000, ids[5]], np.uint32))
    assert got[0].tobytes() == r2[0].tobytes() and got[1].tobytes() == r2[2].tobytes() and got[2].tobytes() == rec[5].tobytes()
    t.erase(ids[10:20])
    fl = t.download(ids)["flags"]
    assert not fl[10:20].any() and np.array_equal(np.delete(fl, np.arange(10, 20)), np.delete(rec["flags"], np.arange(10, 20)))
    t.erase(np.array([10 ** 8], np.uint32))   # beyond the table: ignored


def test_fused_extraction_equals_the_separate_steps():
    """svgpu_track_motion with an image: extraction, undistortion, bearings, grid, matcher and optimiser in ONE submission.  The observation
    equals the stand-alone extractor + frame observation, the tracking result equals the chain on the adopted frame."""
    from stella_vslam_amd import camera, data, feature, synthetic, tracking
    imgs = synthetic.frame_sequence(2, 640, 480, seed=11)
    ext = feature.orb_extractor(feature.orb_params())
    ctx = ext.ctx
    dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    fx = fy = 458.654
    cx, cy = 320.0, 240.0
    cam = camera.perspective("t", "Monocular", "Gray", 640, 480, 30.0, fx, fy, cx, cy, *dist, ctx=ctx)
    T = synthetic.orb_tables(1.2, 8)
    # the last frame: extracted the usual way; its keypoints back-projected onto a plane at depth Z are the map
    k0, d0 = ext.extract(imgs[0])
    rf_last = data.resident_frame(ctx)
    und0, brg0 = rf_last.adopt_extraction(cam, 64, 48)
    Z = 5.0
    pos = np.stack([(und0["x"] - cx) / fx * Z, (und0["y"] - cy) / fy * Z, np.full(len(und0), Z)], 1).astype(np.float64)
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)   # mean viewing direction: from the camera towards the point (landmark.cc:290-300)
    dist0 = np.linalg.norm(pos, axis=1)
    maxd = (dist0 * T["scale_factors"][und0["octave"]]).astype(np.float32)
    mind = (maxd * T["inv_scale_factors"][7]).astype(np.float32)
    ids = (np.arange(len(und0)) * 2 + 1).astype(np.int32)
    table = tracking.landmark_table(ctx).upsert(ids, tracking.landmark_records(pos, nrm, mind, maxd, d0))
    trk = tracking.tracker(ctx, table, cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"])
    pose_last = _pose12(np.eye(3), np.zeros(3))
    guess = _pose12(np.eye(3), np.array([-3.0 * Z / fx + 0.004, -1.0 * Z / fy - 0.003, 0.002]))   # the sequence shifts by (3, 1) px per frame
    rf_cur = data.resident_frame(ctx)
    got = trk.track_motion(rf_cur, rf_last, ids, guess, pose_last, 20.0, img=imgs[1])
    # the observation
    k1, d1 = ext.extract(imgs[1])
    assert len(k1) == got["result"]["n_keypoints"] > 1500 and rf_cur.size == len(k1)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got["keypts"][f], k1[f]), f
    assert np.array_equal(got["descriptors"], d1)
    obs = data.frame_observation(cam, k1, d1)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got["undist_keypts"][f], obs.undist_keypts_[f]), f
    assert np.array_equal(got["bearings"], obs.bearings_)
    # the tracking result: the same chain on a frame adopted from the stand-alone extraction
    rf_ref = data.resident_frame(ctx)
    rf_ref.adopt_extraction(cam, 64, 48)
    ref = trk.track_motion(rf_ref, rf_last, ids, guess, pose_last, 20.0)
    assert got["result"]["num_matches"] == ref["result"]["num_matches"] > 500
    assert np.array_equal(got["match_last"], ref["match_last"]) and np.array_equal(got["outlier"], ref["outlier"])
    assert np.array_equal(got["result"]["pose_cw"], ref["result"]["pose_cw"]) and got["result"]["num_valid"] == ref["result"]["num_valid"] > 400
    # ... and the fused frame serves the second half like any resident frame
    cur_lm = np.full(len(k1), -1, np.int32)
    m = got["match_last"]
    cur_lm[m[m >= 0]] = ids[m >= 0]
    cur_lm[got["outlier"] == 1] = -1
    held = set(cur_lm[cur_lm >= 0].tolist())
    local = np.array([i for i in ids if int(i) not in held], np.int32)
    a = trk.track_local_map(rf_cur, cur_lm, local, 5.0, 0.8, 0.5)
    b = trk.track_local_map(rf_ref, cur_lm, local, 5.0, 0.8, 0.5, pose_cw=got["result"]["pose_cw"])
    assert np.array_equal(a["match_local"], b["match_local"]) and np.array_equal(a["result"]["pose_cw"], b["result"]["pose_cw"])
    assert a["visible"].sum() > 300 and a["result"]["num_matches"] > 20 and a["result"]["num_valid"] > 0.95 * got["result"]["num_valid"]


def test_fused_stereo_extraction_equals_the_separate_steps():
    """svgpu_track_motion_stereo: both extractions (two contexts, two streams), match::stereo::compute, the left observation, matcher and
    optimiser in ONE submission (system.cc:406-447 + frame_tracker.cc:22-60).  stereo_x_right_ / depths_ equal the stand-alone extractors +
    match::stereo on their pyramids bit for bit, the observation equals the stand-alone one, and the tracking result (stereo gates of
    projection.cc:179-181, stereo edges of the pose optimizer) equals the chain on a frame adopted from the separate steps."""
    from stella_vslam_amd import camera, data, feature, match, synthetic, tracking
    Wk, Hk, disp = 752, 480, 14
    big = synthetic.frame_sequence(2, Wk + 64, Hk, seed=23)
    left = [np.ascontiguousarray(b[:, 8:8 + Wk]) for b in big]
    right = [np.ascontiguousarray(b[:, 8 + disp:8 + disp + Wk]) for b in big]
    ext_l, ext_r = feature.orb_extractor(feature.orb_params()), feature.orb_extractor(feature.orb_params())
    ctx = ext_l.ctx
    fx = fy = 458.654
    cx, cy, bl = 367.215, 248.375, 0.11
    fxb = fx * bl
    cam = camera.perspective("t", "Stereo", "Gray", Wk, Hk, 30.0, fx, fy, cx, cy, 0, 0, 0, 0, 0, focal_x_baseline=fxb, ctx=ctx)
    T = synthetic.orb_tables(1.2, 8)
    # the last frame, the usual way: both extractions, stereo matcher, adopted observation; its keypoints with a stereo depth are the map
    k0, d0 = ext_l.extract(left[0])
    k0r, d0r = ext_r.extract(right[0])
    xr0, dp0 = match.stereo(ext_l, ext_r, k0, k0r, d0, d0r, fxb, bl).compute()
    rf_last = data.resident_frame(ctx)
    und0, _ = rf_last.adopt_extraction(cam, 64, 48)
    rf_last.set_stereo(xr0)
    assert (xr0 >= 0).sum() > 800
    Z = np.where(dp0 > 0, dp0, 5.0).astype(np.float64)
    pos = np.stack([(und0["x"] - cx) / fx * Z, (und0["y"] - cy) / fy * Z, Z], 1)
    dist0 = np.linalg.norm(pos, axis=1)
    nrm = pos / dist0[:, None]
    maxd = (dist0 * T["scale_factors"][und0["octave"]]).astype(np.float32)
    mind = (maxd * T["inv_scale_factors"][7]).astype(np.float32)
    ids = np.where(dp0 > 0, np.arange(len(und0)) * 2 + 1, -1).astype(np.int32)
    have = ids >= 0
    table = tracking.landmark_table(ctx).upsert(ids[have], tracking.landmark_records(pos[have], nrm[have], mind[have], maxd[have], d0[have]))
    trk = tracking.tracker(ctx, table, cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"], is_monocular=False, true_baseline=bl)
    pose_last = _pose12(np.eye(3), np.zeros(3))
    zbar = float(np.median(Z[have]))
    guess = _pose12(np.eye(3), np.array([-3.0 * zbar / fx, -1.0 * zbar / fy, 0.0]))   # the sequence shifts by (3, 1) px per frame: exact for the median depth only
    rf_cur = data.resident_frame(ctx)
    got = trk.track_motion_stereo(rf_cur, rf_last, ids, guess, pose_last, 15.0, left[1], right[1], ext_r.ctx)
    # the stereo observation against the separate steps
    k1, d1 = ext_l.extract(left[1])
    k1r, d1r = ext_r.extract(right[1])
    xr1, dp1 = match.stereo(ext_l, ext_r, k1, k1r, d1, d1r, fxb, bl).compute()
    assert len(k1) == got["result"]["n_keypoints"] > 1500 and rf_cur.size == len(k1)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got["keypts"][f], k1[f]), f
    assert np.array_equal(got["descriptors"], d1)
    assert (xr1 >= 0).sum() > 800
    assert np.array_equal(got["stereo_x_right"].view(np.uint32), xr1.view(np.uint32))
    assert np.array_equal(got["depths"].view(np.uint32), dp1.view(np.uint32))
    obs = data.frame_observation(cam, k1, d1)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(got["undist_keypts"][f], obs.undist_keypts_[f]), f
    assert np.array_equal(got["bearings"], obs.bearings_)
    # the tracking result: the same chain on a frame adopted from the stand-alone extraction + stereo matcher
    rf_ref = data.resident_frame(ctx)
    rf_ref.adopt_extraction(cam, 64, 48)   # (of ext_l's last extraction = left[1])
    rf_ref.set_stereo(xr1)
    ref = trk.track_motion(rf_ref, rf_last, ids, guess, pose_last, 15.0)
    assert got["result"]["num_matches"] == ref["result"]["num_matches"] > 300
    assert np.array_equal(got["match_last"], ref["match_last"]) and np.array_equal(got["outlier"], ref["outlier"])
    assert np.array_equal(got["result"]["pose_cw"], ref["result"]["pose_cw"]) and got["result"]["num_valid"] == ref["result"]["num_valid"] > 200
    # ... a monocular tracker refuses the stereo entry point, and so does a right context that is the left one
    trk_mono = tracking.tracker(ctx, table, cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"], is_monocular=True)
    with pytest.raises(RuntimeError):
        trk_mono.track_motion_stereo(data.resident_frame(ctx), rf_last, ids, guess, pose_last, 15.0, left[1], right[1], ext_r.ctx)
    with pytest.raises(RuntimeError):
        trk.track_motion_stereo(data.resident_frame(ctx), rf_last, ids, guess, pose_last, 15.0, left[1], right[1], ctx)


def test_fused_rgbd_frame_equals_the_separate_steps():
    """svgpu_track_motion_rgbd: extraction, undistortion, the depth sampled at the distorted keypoint (system.cc:492-510), bearings, grid, matcher and
    optimiser in one submission.  stereo_x_right_ / depths_ equal the reference's arithmetic (float coordinates truncated, float - double / float)
    bit for bit; the tracking result equals the chain on a frame adopted from the separate steps."""
    from stella_vslam_amd import camera, data, feature, synthetic, tracking
    Wk, Hk = 640, 480
    imgs = synthetic.frame_sequence(2, Wk, Hk, seed=31)
    ext = feature.orb_extractor(feature.orb_params())
    ctx = ext.ctx
    fx = fy = 517.3
    cx, cy, fxb = 318.6, 255.3, 40.0
    dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)     # TUM fr1-like distortion
    cam = camera.perspective("t", "RGBD", "Gray", Wk, Hk, 30.0, fx, fy, cx, cy, *dist, focal_x_baseline=fxb, ctx=ctx)
    T = synthetic.orb_tables(1.2, 8)
    yy, xx = np.mgrid[0:Hk, 0:Wk]
    depth = (4.0 + 1.5 * np.sin(xx / 70.0) + 0.8 * np.cos(yy / 55.0)).astype(np.float32)
    depth[(xx // 40 + yy // 40) % 7 == 0] = 0.0            # holes of the sensor
    depth[100:140, 300:360] = -1.0

    def stereo_of(k, und):
        d = depth[k["y"].astype(np.int32), k["x"].astype(np.int32)]      # at<float>(y, x): float -> int truncation
        ok = ~(d <= 0)
        dd = np.where(ok, d, np.float32(1)).astype(np.float64)
        xr = np.where(ok, (und["x"].astype(np.float64) - fxb / dd).astype(np.float32), np.float32(-1))
        return xr.astype(np.float32), np.where(ok, d, np.float32(-1)).astype(np.float32)

    k0, d0 = ext.extract(imgs[0])
    rf_last = data.resident_frame(ctx)
    und0, _ = rf_last.adopt_extraction(cam, 64, 48)
    xr0, dp0 = stereo_of(k0, und0)
    rf_last.set_stereo(xr0)
    Z = np.where(dp0 > 0, dp0, 4.0).astype(np.float64)
    pos = np.stack([(und0["x"] - cx) / fx * Z, (und0["y"] - cy) / fy * Z, Z], 1)
    dist0 = np.linalg.norm(pos, axis=1)
    nrm = pos / dist0[:, None]
    maxd = (dist0 * T["scale_factors"][und0["octave"]]).astype(np.float32)
    mind = (maxd * T["inv_scale_factors"][7]).astype(np.float32)
    ids = np.where(dp0 > 0, np.arange(len(und0)) * 2 + 1, -1).astype(np.int32)
    have = ids >= 0
    table = tracking.landmark_table(ctx).upsert(ids[have], tracking.landmark_records(pos[have], nrm[have], mind[have], maxd[have], d0[have]))
    trk = tracking.tracker(ctx, table, cam, T["scale_factors"], T["inv_level_sigma_sq"], T["log_scale_factor"], is_monocular=False, true_baseline=fxb / fx)
    pose_last = _pose12(np.eye(3), np.zeros(3))
    zbar = float(np.median(Z[have]))
    guess = _pose12(np.eye(3), np.array([-3.0 * zbar / fx, -1.0 * zbar / fy, 0.0]))
    rf_cur = data.resident_frame(ctx)
    got = trk.track_motion_rgbd(rf_cur, rf_last, ids, guess, pose_last, 15.0, imgs[1], depth)
    k1, d1 = ext.extract(imgs[1])
    assert len(k1) == got["result"]["n_keypoints"] > 1500
    obs = data.frame_observation(cam, k1, d1)
    for f in ("x", "y", "octave", "angle"):
        assert np.array_equal(got["undist_keypts"][f], obs.undist_keypts_[f]), f
    xr1, dp1 = stereo_of(k1, obs.undist_keypts_)
    assert (xr1 >= 0).sum() > 1000 and (dp1 < 0).sum() > 100
    assert np.array_equal(got["stereo_x_right"].view(np.uint32), xr1.view(np.uint32))
    assert np.array_equal(got["depths"].view(np.uint32), dp1.view(np.uint32))
    rf_ref = data.resident_frame(ctx)
    rf_ref.adopt_extraction(cam, 64, 48)
    rf_ref.set_stereo(xr1)
    ref = trk.track_motion(rf_ref, rf_last, ids, guess, pose_last, 15.0)
    assert got["result"]["num_matches"] == ref["result"]["num_matches"] > 300
    assert np.array_equal(got["match_last"], ref["match_last"]) and np.array_equal(got["outlier"], ref["outlier"])
    assert np.array_equal(got["result"]["pose_cw"], ref["result"]["pose_cw"])


def test_track_entry_points_reject_bad_arguments(ctx):
    from stella_vslam_amd import tracking
    from stella_vslam_amd._lib import SvgpuError
    W = _World(ctx, 3, False, n_lm=300, n_extra=100)
    with pytest.raises(SvgpuError):   # the same frame on both sides
        W.tracker.track_motion(W.rf_cur, W.rf_cur, np.full(W.rf_cur.size, -1, np.int32), W.guess, W.pose_last, 10.0)
    with pytest.raises(SvgpuError):   # fused extraction on a context that was never configured for ORB
        from stella_vslam_amd import feature
        c2 = feature.Context()
        t2 = tracking.tracker(c2, W.table, W.cam, W.T["scale_factors"], W.T["inv_level_sigma_sq"], W.T["log_scale_factor"])
        t2.track_motion(W.rf_cur, W.rf_last, W.last_ids, W.guess, W.pose_last, 10.0, img=np.zeros((480, 752), np.uint8))
    with pytest.raises(SvgpuError):   # no pose on the device yet
        t3 = tracking.tracker(ctx, W.table, W.cam, W.T["scale_factors"], W.T["inv_level_sigma_sq"], W.T["log_scale_factor"])
        t3.track_local_map(W.rf_cur, np.full(W.rf_cur.size, -1, np.int32), W.ids, 5.0)
    # empty inputs are fine: no landmarks offered, nothing held
    r = W.tracker.track_local_map(W.rf_cur, np.full(W.rf_cur.size, -1, np.int32), np.zeros(0, np.int32), 5.0, pose_cw=W.guess)
    assert r["result"]["num_matches"] == 0 and r["result"]["num_valid"] == 0 and np.array_equal(r["result"]["pose_cw"], W.guess)
