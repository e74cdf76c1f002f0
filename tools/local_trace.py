"""Stage timing of a config-3 local BA call (SVGPU_BA_TRACE), steady state."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stella_vslam_amd import optimize, synthetic
sc = synthetic.ba_scene()
ba = optimize.local_bundle_adjuster()
for _ in range(20):
    ba.optimize_flat(sc)
os.environ["SVGPU_BA_TRACE"] = "1"
for _ in range(2):
    t0 = time.perf_counter()
    r = ba.optimize_flat(sc)
    print("ms", (time.perf_counter() - t0) * 1e3, r["stats"])
