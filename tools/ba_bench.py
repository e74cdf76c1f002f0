#!/usr/bin/env python3
"""Time the bundle adjusters: config 3 (local BA, 20 KF / 10k landmarks / ~60k obs) and, with --global, config 5
(global BA, 500 KF / 200k landmarks / 1.2M obs) with each linear solver.  SVGPU_BA_TRACE=1 prints the host-side laps."""
import sys, time, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
n = 5
out = {}
sc = synthetic.ba_scene()
for name, solver in (("auto", optimize.SOLVER_AUTO), ("cholesky", optimize.SOLVER_CHOLESKY), ("cholesky_mfma", optimize.SOLVER_CHOLESKY_MFMA),
                     ("pcg", optimize.SOLVER_PCG), ("pcg_multi", optimize.SOLVER_PCG_MULTI)):
    ba = optimize.local_bundle_adjuster().set_solver(solver)
    ba.optimize_flat(sc)
    t0 = time.perf_counter()
    for _ in range(n):
        r = ba.optimize_flat(sc)
    dt = (time.perf_counter() - t0) / n
    out["local_" + name] = dict(ms_per_call=dt * 1e3, stats=r["stats"])
    print("local", name, "ms/call", dt * 1e3, r["stats"], flush=True)
if "--global" in sys.argv:
    sg = synthetic.ba_scene_large()
    for name, solver in (("auto", optimize.SOLVER_AUTO),) + ((("dense", optimize.SOLVER_DENSE),) if "--dense" in sys.argv else ()):
        ba = optimize.local_bundle_adjuster().set_solver(solver)
        ba.optimize_global_flat(sg, num_iter=10)
        t0 = time.perf_counter()
        for _ in range(3):
            r = ba.optimize_global_flat(sg, num_iter=10)
        dt = (time.perf_counter() - t0) / 3
        out["global_" + name] = dict(ms_per_call=dt * 1e3, stats=r["stats"])
        print("global", name, "ms/call", dt * 1e3, r["stats"], flush=True)
json.dump(out, open("gpurun_out/ba_bench.json", "w"), indent=1)
