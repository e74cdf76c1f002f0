#!/usr/bin/env python3
"""Writes the seeded config-3 BA scene (stella_vslam_amd/synthetic.py ba_scene) in the flat binary format oracle/ref_recipe/dump_ba_g2o.cc reads.
usage: tools/export_ba_scene.py oracle/_ref/fixtures/ba_config3.bin"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import synthetic  # noqa: E402

sc = synthetic.ba_scene()
with open(sys.argv[1], "wb") as f:
    P, L, E = len(sc["pose_cw"]), len(sc["points"]), len(sc["obs_pose"])
    np.array([P, L, E], "<i4").tofile(f)
    np.ascontiguousarray(sc["pose_cw"], "<f8").tofile(f)
    np.ascontiguousarray(sc["pose_fixed"], "u1").tofile(f)
    np.ascontiguousarray(sc["points"], "<f8").tofile(f)
    np.ascontiguousarray(sc["obs_pose"], "<i4").tofile(f)
    np.ascontiguousarray(sc["obs_point"], "<i4").tofile(f)
    np.ascontiguousarray(sc["obs_uvr"], "<f4").tofile(f)
    np.ascontiguousarray(sc["obs_inv_sigma_sq"], "<f4").tofile(f)
    np.ascontiguousarray(sc["intr"], "<f8").tofile(f)
print("wrote", sys.argv[1], P, L, E)
