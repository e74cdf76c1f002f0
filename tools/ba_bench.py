#!/usr/bin/env python3
"""Time svgpu_local_ba on the config-3 scene (20 KF / 10k landmarks / ~60k obs)."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
sc = synthetic.ba_scene()
ba = optimize.local_bundle_adjuster()
ba.optimize_flat(sc)
t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    r = ba.optimize_flat(sc)
dt = (time.perf_counter() - t0) / n
print("ms/call", dt * 1e3, r["stats"])
