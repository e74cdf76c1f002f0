#!/usr/bin/env python3
"""Config-3 local BA: ms per call (median of batches), statistics.  usage: tools/local_exp.py [calls]"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ba = optimize.local_bundle_adjuster()
sc = synthetic.ba_scene()
for _ in range(20):
    r = ba.optimize_flat(sc)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(n // 5):
        r = ba.optimize_flat(sc)
    ts.append((time.perf_counter() - t0) / (n // 5))
ts.sort()
st = r["stats"]
print(f"local BA config 3: {ts[2] * 1e3:.3f} ms/call (min {ts[0] * 1e3:.3f})  chi2 {st['chi2_final']:.9f} it {st['iters_stage1']}+{st['iters_stage2']} trials {st['lm_trials']}", flush=True)
