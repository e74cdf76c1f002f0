#!/usr/bin/env python3
"""Workload for rocprofv3 kernel traces of the bundle adjusters: `local` = 5 calls of config 3, `global` = 2 calls of config 5, `large` = 2 calls of the 9.6 M-observation leg."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "local"
ba = optimize.local_bundle_adjuster()
if which == "local":
    sc = synthetic.ba_scene()
    for _ in range(6):
        ba.optimize_flat(sc)
else:
    sg = synthetic.ba_scene_large(num_lm=1600000) if which == "large" else synthetic.ba_scene_large()
    for _ in range(2):
        ba.optimize_global_flat(sg, num_iter=10)
