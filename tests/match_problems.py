"""Flat problems for the function-specific matchers, built from stella_vslam_amd.synthetic.map_scene.
Shared by the oracle tests (CPU) and the device parity tests (GPU): every builder returns the keyword arguments that BOTH the
oracle wrapper (oracle/oracle.py) and the C-ABI mirror (stella_vslam_amd/match.py) accept, except for `cam`, which each side
supplies in its own type (`make_cams`)."""
import numpy as np

from stella_vslam_amd import synthetic as S


def scene(seed=7, stereo=False, **kw):
    return S.map_scene(seed=seed, stereo=stereo, **kw)


def make_cams(sc, kind):
    """kind 'oracle' -> oracle.Camera, 'svgpu' -> stella_vslam_amd.camera.perspective (image bounds computed by each side)."""
    fx, fy, cx, cy, fxb = sc["K"]
    if kind == "oracle":
        from oracle import oracle as O
        return O.make_camera(O.CAM_PERSPECTIVE, sc["width"], sc["height"], fx, fy, cx, cy, (0, 0, 0, 0, 0), fxb)
    from stella_vslam_amd import camera
    return camera.perspective("t", "Stereo" if fxb else "Monocular", "Gray", sc["width"], sc["height"], 30.0, fx, fy, cx, cy, 0, 0, 0, 0, 0,
                              focal_x_baseline=fxb)


def _bearings(sc, xy):
    fx, fy, cx, cy, _ = sc["K"]
    x, y = (xy[:, 0].astype(np.float64) - cx) / fx, (xy[:, 1].astype(np.float64) - cy) / fy
    l2 = np.sqrt(x * x + y * y + 1.0)
    return np.stack([x / l2, y / l2, 1.0 / l2], 1)


def _perturb(R, t, rng, rot_deg=0.3, trans=0.01):
    w = rng.normal(0, 1, 3)
    w *= np.deg2rad(rot_deg) / np.linalg.norm(w)
    dR = S._rodrigues(w)
    return dR @ R, dR @ t + rng.normal(0, trans, 3)


def _lm_of(sc, view, fill=0.0):
    """Per keypoint of `view`: landmark position / descriptor / distance range (zeros where the keypoint has no landmark)."""
    L = sc["landmarks"]
    lm = view["lm"]
    has = lm >= 0
    idx = np.where(has, lm, 0)
    pos = np.where(has[:, None], L["pos_w"][idx], fill)
    return dict(has=has, pos_w=pos, desc=np.ascontiguousarray(L["desc"][idx]), min_valid_dist=L["min_valid_dist"][idx], max_valid_dist=L["max_valid_dist"][idx],
                mean_normal=L["mean_normal"][idx])


def current_and_last(sc, seed=1, margin=15.0, dup=80):
    """`dup` keypoints of the last frame are repeated at the end (two keypoints of the last frame carrying the same landmark data): with
    landmarks that have no observation (lm_has_observation = 0) both copies claim the same keypoint of the current frame."""
    rng = np.random.default_rng(seed)
    last, cur = sc["views"]
    T = sc["tables"]
    lm = _lm_of(sc, last)
    if dup:
        rep = rng.choice(np.flatnonzero(lm["has"]), dup, replace=False)
        ext = lambda a: np.concatenate([a, a[rep]])
        lm = {k: ext(v) for k, v in lm.items()}
        last = dict(last, octave=ext(last["octave"]), angle=ext(last["angle"]))
    R, t = _perturb(cur["rot_cw"], cur["trans_cw"], rng)
    stereo = sc["K"][4] != 0
    return dict(rot_cw=R, trans_cw=t, rot_lw=last["rot_cw"], trans_lw=last["trans_cw"], pos_w=lm["pos_w"],
                valid=(lm["has"] & (rng.uniform(0, 1, len(lm["has"])) < 0.95)).astype(np.uint8), lm_desc=lm["desc"], octave_last=last["octave"],
                angle_last=last["angle"], scale_factors=T["scale_factors"], margin=margin, tdesc=cur["desc"], t_xy=cur["xy"], t_octave=cur["octave"],
                t_angle=cur["angle"], occupied=(rng.uniform(0, 1, len(cur["xy"])) < 0.05).astype(np.uint8), t_xright=cur["x_right"] if stereo else None,
                lm_has_observation=(rng.uniform(0, 1, len(lm["has"])) < 0.85).astype(np.uint8), is_monocular=not stereo, true_baseline=0.11)


def frame_and_keyframe(sc, seed=2, margin=10.0, thr=100):
    rng = np.random.default_rng(seed)
    kf, frm = sc["views"]
    T = sc["tables"]
    lm = _lm_of(sc, kf)
    R, t = _perturb(frm["rot_cw"], frm["trans_cw"], rng)
    return dict(rot_cw=R, trans_cw=t, pos_w=lm["pos_w"], valid=(lm["has"] & (rng.uniform(0, 1, len(lm["has"])) < 0.9)).astype(np.uint8),
                min_valid_dist=lm["min_valid_dist"], max_valid_dist=lm["max_valid_dist"], lm_desc=lm["desc"], angle_kf=kf["angle"],
                scale_factors=T["scale_factors"], log_scale_factor=T["log_scale_factor"], margin=margin, hamm_dist_thr=thr, tdesc=frm["desc"],
                t_xy=frm["xy"], t_octave=frm["octave"], t_angle=frm["angle"], occupied=(rng.uniform(0, 1, len(frm["xy"])) < 0.1).astype(np.uint8))


def by_sim3(sc, seed=3, margin=7.5, scale=1.3):
    rng = np.random.default_rng(seed)
    kf = sc["views"][1]
    T, L = sc["tables"], sc["landmarks"]
    R, t = _perturb(kf["rot_cw"], kf["trans_cw"], rng, 0.2, 0.005)
    sim3 = np.eye(4)
    sim3[:3, :3] = scale * R
    sim3[:3, 3] = scale * t
    n = len(L["pos_w"])
    return dict(sim3_cw=sim3, pos_w=L["pos_w"], valid=(rng.uniform(0, 1, n) < 0.9).astype(np.uint8), min_valid_dist=L["min_valid_dist"],
                max_valid_dist=L["max_valid_dist"], mean_normal=L["mean_normal"], lm_desc=L["desc"], scale_factors=T["scale_factors"],
                log_scale_factor=T["log_scale_factor"], margin=margin, tdesc=kf["desc"], t_xy=kf["xy"], t_octave=kf["octave"],
                occupied=(rng.uniform(0, 1, len(kf["xy"])) < 0.1).astype(np.uint8))


def mutually(sc, seed=4, margin=7.5, s_12=1.02):
    rng = np.random.default_rng(seed)
    v1, v2 = sc["views"]
    T = sc["tables"]
    R12 = v1["rot_cw"] @ v2["rot_cw"].T
    t12 = v1["trans_cw"] - R12 @ v2["trans_cw"]

    def side(v):
        lm = _lm_of(sc, v)
        return dict(pos_w=lm["pos_w"], valid=(lm["has"] & (rng.uniform(0, 1, len(lm["has"])) < 0.9)).astype(np.uint8), min_valid_dist=lm["min_valid_dist"],
                    max_valid_dist=lm["max_valid_dist"], lm_desc=lm["desc"], desc=v["desc"], xy=v["xy"], octave=v["octave"])
    return dict(rot_1w=v1["rot_cw"], trans_1w=v1["trans_cw"], rot_2w=v2["rot_cw"], trans_2w=v2["trans_cw"], s_12=s_12, rot_12=R12, trans_12=t12,
                kf1=side(v1), kf2=side(v2), scale_factors=T["scale_factors"], log_scale_factor=T["log_scale_factor"], margin=margin)


def fuse(sc, seed=5, margin=3.0, do_reprojection_matching=True):
    rng = np.random.default_rng(seed)
    kf = sc["views"][1]
    T, L = sc["tables"], sc["landmarks"]
    n = len(L["pos_w"])
    stereo = sc["K"][4] != 0
    return dict(rot_cw=kf["rot_cw"], trans_cw=kf["trans_cw"], pos_w=L["pos_w"], valid=(rng.uniform(0, 1, n) < 0.9).astype(np.uint8),
                min_valid_dist=L["min_valid_dist"], max_valid_dist=L["max_valid_dist"], mean_normal=L["mean_normal"], lm_desc=L["desc"],
                scale_factors=T["scale_factors"], inv_level_sigma_sq=T["inv_level_sigma_sq"], log_scale_factor=T["log_scale_factor"], margin=margin,
                tdesc=kf["desc"], t_xy=kf["xy"], t_octave=kf["octave"], t_xright=kf["x_right"] if stereo else None,
                do_reprojection_matching=do_reprojection_matching)


def _nodes(sc, view, rng, n_nodes=80):
    """bow_feat_vec_ membership: keypoints of one landmark share a node most of the time, clutter gets random nodes, a few get none."""
    lm = view["lm"]
    node = np.where(lm >= 0, (lm * 2654435761 % 1000003) % n_nodes, rng.integers(0, n_nodes, len(lm))).astype(np.int32)
    moved = rng.uniform(0, 1, len(lm)) < 0.1
    node[moved] = rng.integers(0, n_nodes + 20, moved.sum())   # includes node ids the other side never has
    node[rng.uniform(0, 1, len(lm)) < 0.02] = -1
    return node


def triangulation(sc, cams_epipole, seed=6, with_nodes=False, residual_rad_thr=0.01, frac_with_lm=0.4):
    """cams_epipole: callable(rot_2w, trans_2w, centre_1) -> (bearing, valid), i.e. the side's own reproject_to_bearing."""
    rng = np.random.default_rng(seed)
    v1, v2 = sc["views"]
    T = sc["tables"]
    R12 = v1["rot_cw"] @ v2["rot_cw"].T
    t12 = v1["trans_cw"] - R12 @ v2["trans_cw"]
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    E12 = tx @ R12
    epi, valid = cams_epipole(v2["rot_cw"], v2["trans_cw"], v1["center"])
    stereo = sc["K"][4] != 0
    kw = dict(desc1=v1["desc"], angle1=v1["angle"], octave1=v1["octave"], bearings1=_bearings(sc, v1["xy"]),
              has_lm1=(rng.uniform(0, 1, len(v1["xy"])) < frac_with_lm).astype(np.uint8), desc2=v2["desc"], angle2=v2["angle"],
              bearings2=_bearings(sc, v2["xy"]), has_lm2=(rng.uniform(0, 1, len(v2["xy"])) < frac_with_lm).astype(np.uint8), E_12=E12, epipole_in_2=epi,
              valid_epipole=valid, scale_factors=T["scale_factors"], residual_rad_thr=residual_rad_thr,
              xright1=v1["x_right"] if stereo else None, xright2=v2["x_right"] if stereo else None)
    if with_nodes:
        kw["node1"], kw["node2"] = _nodes(sc, v1, rng), _nodes(sc, v2, rng)
    return kw


def bow(sc, seed=8, keyframes=False):
    rng = np.random.default_rng(seed)
    v1, v2 = sc["views"]
    kw = dict(desc1=v1["desc"], angle1=v1["angle"], valid1=((v1["lm"] >= 0) & (rng.uniform(0, 1, len(v1["lm"])) < 0.95)).astype(np.uint8),
              node1=_nodes(sc, v1, rng), desc2=v2["desc"], angle2=v2["angle"], node2=_nodes(sc, v2, rng))
    if keyframes:
        kw["valid2"] = ((v2["lm"] >= 0) & (rng.uniform(0, 1, len(v2["lm"])) < 0.95)).astype(np.uint8)
    else:
        kw["occupied2"] = (rng.uniform(0, 1, len(v2["lm"])) < 0.03).astype(np.uint8)
    return kw
