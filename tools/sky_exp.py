#!/usr/bin/env python3
"""Config-5 global BA under the envelope-factorisation variants: ms per call, ms per solve (HIP events around ba_solve), plan chosen.
usage: tools/sky_exp.py [cuts ...]   (cuts: 'auto', 'one', 'two', or a number)"""
import ctypes as C, os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
from stella_vslam_amd._lib import lib
L = lib()
sg = synthetic.ba_scene_large()
for v in (sys.argv[1:] or ["one", "two", "auto", "3", "4", "6", "8"]):
    for k in ("SVGPU_SKY_ONE_SIDED", "SVGPU_SKY_SEGMENTS"):
        os.environ.pop(k, None)
    if v == "one":
        os.environ["SVGPU_SKY_ONE_SIDED"] = "1"
    elif v == "two":
        os.environ["SVGPU_SKY_SEGMENTS"] = "0"
    elif v != "auto":
        os.environ["SVGPU_SKY_SEGMENTS"] = v
    ba = optimize.local_bundle_adjuster()
    r = ba.optimize_global_flat(sg, num_iter=10)
    plan = ba.last_envelope_plan()
    t0 = time.perf_counter()
    for _ in range(3):
        r = ba.optimize_global_flat(sg, num_iter=10)
    dt = (time.perf_counter() - t0) / 3
    L.svgpu_profile_select(ba.ctx.handle, b"ba_solve")
    r = ba.optimize_global_flat(sg, num_iter=10)
    ms, n = C.c_double(), C.c_longlong()
    L.svgpu_profile_read(ba.ctx.handle, C.byref(ms), C.byref(n))
    L.svgpu_profile_select(ba.ctx.handle, None)
    print(f"{v:>5}: {dt * 1e3:7.2f} ms/call  solve {ms.value / max(n.value, 1) * 1e3:7.1f} us x {n.value}  chi2 {r['stats']['chi2_final']:.6f} it {r['stats']['iters_stage1']} fail {r['stats']['cholesky_failures']}  {plan}", flush=True)
