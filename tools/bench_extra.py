#!/usr/bin/env python3
"""Latency of the secondary entry points (host-buffer C ABI, one call at a time, as the reference's per-frame code would use
them) next to the CPU oracle on the same inputs.  Prints one JSON object; `profiles/r01_extra_bench.json` keeps a run."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from stella_vslam_amd import camera as CAM, data as D, feature, match as M, optimize, synthetic as S


def timeit(f, n, warm=2):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    return (time.perf_counter() - t0) / n * 1e3, r


out = {}
ctx = feature.Context(0)
seq = S.frame_sequence(2)
# ---- single-frame extraction through the host-buffer API (H2D image, D2H keypoints + descriptors included)
ext = feature.orb_extractor(feature.orb_params())
g, (k0, d0) = timeit(lambda: ext.extract(seq[0]), 50)
c, _ = timeit(lambda: O.orb_extract(seq[0]), 5, 1)
out["orb_extract_single_frame_ms"] = {"gpu": round(g, 3), "cpu_oracle": round(c, 3), "keypoints": int(len(k0))}
k1, d1 = ext.extract(seq[1])
# ---- brute force, one pair, host buffers
g, _ = timeit(lambda: M.robust(0.8, True, ctx).brute_force_match(d1, k1["angle"], d0, k0["angle"], None), 50)
c, _ = timeit(lambda: O.brute_force_match(d1, k1["angle"], d0, k0["angle"], None, 0.8, True), 5, 1)
out["brute_force_match_single_pair_ms"] = {"gpu": round(g, 3), "cpu_oracle": round(c, 3)}
# ---- projection-style candidate matching with the lists built on the device vs host-built lists + oracle
bounds = (0.0, 640.0, 0.0, 480.0)
sf = O.scale_tables(1.2, 8)[0]
q_xy = np.stack([k0["x"] - 3.0, k0["y"] - 1.0], 1).astype(np.float32)
q_margin = (15.0 * sf[k0["octave"]]).astype(np.float32)
q_lo, q_hi = np.maximum(0, k0["octave"] - 1).astype(np.int32), np.minimum(7, k0["octave"] + 1).astype(np.int32)
t_xy = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
proj = M.projection(0.8, True, ctx)
g, (got, num) = timeit(lambda: proj.match_in_cells(d0, q_xy, q_margin, d1, t_xy, k1["octave"], bounds, 1, 100, q_min_level=q_lo, q_max_level=q_hi,
                                                   q_angle=k0["angle"], t_angle=k1["angle"]), 50)


def cpu_projection():
    off_g, items = O.assign_keypoints_to_grid(k1["x"], k1["y"], bounds)
    cand_off, cand_idx = [0], []
    for q in range(len(k0)):
        cand_idx += O.get_keypoints_in_cell(k1["x"], k1["y"], k1["octave"], off_g, items, bounds, float(q_xy[q, 0]), float(q_xy[q, 1]),
                                            float(q_margin[q]), int(q_lo[q]), int(q_hi[q])).tolist()
        cand_off.append(len(cand_idx))
    return O.match_candidates(d0, d1, cand_off, cand_idx, check_orientation=True, thr=100, lowe_ratio=0.8, mode=1, t_octave=k1["octave"],
                              q_angle=k0["angle"], t_angle=k1["angle"])


c, exp = timeit(cpu_projection, 3, 1)
assert np.array_equal(got, exp)
out["projection_match_in_cells_ms"] = {"gpu": round(g, 3), "cpu_oracle_incl_python_list_building": round(c, 3), "queries": int(len(k0)),
                                       "matches": int(num)}
# ---- frame observation (undistort + bearings + grid) and the fused can_observe + match_frame_and_landmarks pass
EU = dict(fx=458.654, fy=457.296, cx=320.0, cy=240.0, k=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0))
cam = CAM.perspective("euroc-like", "Monocular", "Gray", 640, 480, 20.0, EU["fx"], EU["fy"], EU["cx"], EU["cy"], *EU["k"], ctx=ctx)
ocam = O.make_camera(O.CAM_PERSPECTIVE, 640, 480, EU["fx"], EU["fy"], EU["cx"], EU["cy"], EU["k"])
g, obs = timeit(lambda: D.frame_observation(cam, k1, d1), 100)


def cpu_observation():
    und = O.undistort_keypoints(ocam, np.stack([k1["x"], k1["y"]], 1))
    return und, O.keypoints_to_bearings(ocam, und), O.assign_keypoints_to_grid(und[:, 0], und[:, 1], (ocam.min_x, ocam.max_x, ocam.min_y, ocam.max_y))


c, (und_c, brg_c, _) = timeit(cpu_observation, 20, 1)
assert np.array_equal(np.stack([obs.undist_keypts_["x"], obs.undist_keypts_["y"]], 1), und_c) and np.array_equal(obs.bearings_, brg_c)
out["frame_observation_ms"] = {"gpu": round(g, 3), "cpu_oracle": round(c, 3), "keypoints": int(len(k1))}
rng = np.random.default_rng(0)
nl = 4000                                     # a typical local map seen from this frame
pick = rng.integers(0, len(k1), nl)
depth = rng.uniform(2, 9, nl)
pw = obs.bearings_[pick] / obs.bearings_[pick, 2:3] * depth[:, None] + rng.normal(0, 0.01, (nl, 3))
nv = pw / np.linalg.norm(pw, axis=1, keepdims=True)
mx = (np.linalg.norm(pw, axis=1) * rng.uniform(0.85, 1.15, nl) * sf[k1["octave"][pick]]).astype(np.float32)
mn = (mx / sf[7]).astype(np.float32)
lmd = d1[pick].copy()
lmd[:, rng.integers(0, 32)] ^= 0x55
lsf = float(np.log(np.float32(1.2)))
g, r = timeit(lambda: proj.match_frame_and_landmarks(cam, np.eye(3), np.zeros(3), pw, nv, mn, mx, lmd, obs, sf, lsf, margin=5.0), 50)


def cpu_track():
    vis, rp, xr, lv = O.can_observe(ocam, np.eye(3), np.zeros(3), pw, nv, mn, mx, 0.5, 8, lsf)
    u = obs.undist_keypts_
    b = (ocam.min_x, ocam.max_x, ocam.min_y, ocam.max_y)
    off_g, items = O.assign_keypoints_to_grid(u["x"], u["y"], b)
    lvq = np.where(vis == 1, lv, 0)
    qm = (np.float32(5.0) * sf[lvq]).astype(np.float32)
    cand_off, cand_idx = [0], []
    for q in range(nl):
        if vis[q]:
            cand_idx += O.get_keypoints_in_cell(u["x"], u["y"], u["octave"], off_g, items, b, float(np.float32(rp[q, 0])), float(np.float32(rp[q, 1])),
                                                float(qm[q]), max(0, int(lv[q]) - 1), min(7, int(lv[q]) + 1)).tolist()
        cand_off.append(len(cand_idx))
    return O.match_candidates(lmd, obs.descriptors_, cand_off, cand_idx, check_orientation=False, thr=100, lowe_ratio=0.8, mode=1,
                              t_octave=u["octave"], q_valid=vis)


c, exp = timeit(cpu_track, 3, 1)
assert np.array_equal(r[0], exp)
out["can_observe_plus_match_frame_and_landmarks_ms"] = {"gpu": round(g, 3), "cpu_oracle_incl_python_list_building": round(c, 3), "landmarks": nl,
                                                        "visible": int(r[2].sum()), "matches": int(r[1])}
# ---- stereo
big = S.frame(640 + 64, 480, 5)
left, right = np.ascontiguousarray(big[:, 8:648]), np.ascontiguousarray(big[:, 8 + 20:648 + 20])
el, er = feature.orb_extractor(feature.orb_params()), feature.orb_extractor(feature.orb_params())
kl, dl = el.extract(left)
kr, dr = er.extract(right)
g, _ = timeit(lambda: M.stereo(el, er, kl, kr, dl, dr, 458.654 * 0.11, 0.11).compute(), 30)
pl, pr = el.image_pyramid_, er.image_pyramid_
c, _ = timeit(lambda: O.stereo_match(kl, dl, kr, dr, pl, pr, 458.654 * 0.11, 0.11), 3, 1)
out["stereo_compute_ms"] = {"gpu": round(g, 3), "cpu_oracle": round(c, 3), "left_keypoints": int(len(kl))}
# ---- pose optimizer (motion-only BA), ~1200 observations
sc = S.ba_scene(num_kf=3, num_lm=1200, obs_per_lm=3, num_fixed=0, seed=4, outlier_frac=0.1)
sel = sc["obs_pose"] == 1
pr_ = dict(pose_cw=sc["pose_cw"][1], pos_w=sc["points_gt"][sc["obs_point"][sel]], uvr=sc["obs_uvr"][sel], w=sc["obs_inv_sigma_sq"][sel],
           h=sc["obs_huber"][sel], intr=sc["intr"][1])
po = optimize.pose_optimizer(ctx=ctx)
g, _ = timeit(lambda: po.optimize_flat(pr_["pose_cw"], pr_["pos_w"], pr_["uvr"], pr_["w"], pr_["h"], pr_["intr"]), 100)
c, _ = timeit(lambda: O.pose_optimize(pr_["pose_cw"], pr_["pos_w"], pr_["uvr"], pr_["w"], pr_["h"], pr_["intr"]), 10, 1)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")  # the runtime libsvgpu is linked against (torch would bring a second copy)
dv = []
for k, t in (("pos_w", np.float64), ("uvr", np.float32), ("w", np.float32), ("h", np.float32)):
    a = np.ascontiguousarray(pr_[k], t)
    ptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(a.nbytes)) == 0
    assert hip.hipMemcpy(ptr, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 1) == 0
    dv.append(ptr.value)
gd, _ = timeit(lambda: po.optimize_device(pr_["pose_cw"], int(sel.sum()), *dv, pr_["intr"]), 100)
out["pose_optimizer_ms"] = {"gpu": round(g, 3), "gpu_device_resident_inputs": round(gd, 3), "cpu_oracle": round(c, 3), "observations": int(sel.sum())}
# ---- local BA config 3 and the global-BA sized problem (rocSOLVER path)
ba = optimize.local_bundle_adjuster(ctx=ctx)
sc3 = S.ba_scene()
g, r = timeit(lambda: ba.optimize_flat(sc3), 10)
out["local_ba_config3_ms"] = {"gpu": round(g, 3), "lm_iterations": int(r["stats"]["iters_stage1"] + r["stats"]["iters_stage2"])}
sc5 = S.ba_scene(num_kf=100, num_lm=40000, obs_per_lm=6, num_fixed=1, seed=9, loop=True)
g, r = timeit(lambda: ba.optimize_global_flat(sc5, num_iter=10), 3, 1)
out["global_ba_100kf_40k_landmarks_ms"] = {"gpu": round(g, 3), "observations": int(len(sc5["obs_pose"])), "lm_iterations": int(r["stats"]["iters_stage1"])}
print(json.dumps(out))
