#!/usr/bin/env python3
"""Packs the .npy files written by dump_fixtures into tests/golden/ref_*.npz (see README.md).
usage: pack_npz.py <fixtures dir> <tests/golden dir>"""
import glob
import os
import sys

import numpy as np

src, dst = sys.argv[1], sys.argv[2]
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
for tag in ("640x480", "1241x376"):
    out = {}
    for t in (0, 1):
        base = os.path.join(src, f"ref_orb_{tag}_f{t}")
        if not os.path.exists(base + "_desc.npy"):
            continue
        f5, i2 = np.load(base + "_kp_f32.npy"), np.load(base + "_kp_i32.npy")
        kp = np.zeros(len(f5), KEYPOINT_DTYPE)
        for j, name in enumerate(("x", "y", "size", "angle", "response")):
            kp[name] = f5[:, j]
        kp["octave"], kp["class_id"] = i2[:, 0], i2[:, 1]
        out[f"image{t}"], out[f"kp{t}"], out[f"desc{t}"] = np.load(base + "_image.npy"), kp, np.load(base + "_desc.npy")
        for p in sorted(glob.glob(base + "_pyr*.npy")):
            out[f"pyr{t}_{os.path.basename(p)[len(os.path.basename(base)) + 4:-4]}"] = np.load(p)
    if out:
        np.savez_compressed(os.path.join(dst, f"ref_orb_{tag}.npz"), **out)
        print("wrote", os.path.join(dst, f"ref_orb_{tag}.npz"))
m = {k: np.load(os.path.join(src, f"ref_{k}.npy")) for k in ("match_a", "match_b", "match_d32", "match_d64", "angle_in", "angle_diff", "trig")
     if os.path.exists(os.path.join(src, f"ref_{k}.npy"))}
if m:
    np.savez_compressed(os.path.join(dst, "ref_match_base.npz"), **m)
    print("wrote", os.path.join(dst, "ref_match_base.npz"))

b = {k: np.load(os.path.join(src, f"ref_ba_config3_{k}.npy")) for k in ("pose_cw", "points", "outlier", "stats") if os.path.exists(os.path.join(src, f"ref_ba_config3_{k}.npy"))}
if len(b) == 4:
    np.savez_compressed(os.path.join(dst, "ref_ba_config3.npz"), **b)
    print("wrote", os.path.join(dst, "ref_ba_config3.npz"))
w = {k: np.load(os.path.join(src, f"ref_{k}.npy")) for k in ("bow_vec", "bow_feat_vec") if os.path.exists(os.path.join(src, f"ref_{k}.npy"))}
if len(w) == 2:
    np.savez_compressed(os.path.join(dst, "ref_bow.npz"), **w)
    print("wrote", os.path.join(dst, "ref_bow.npz"))
