"""Stage timing of a config-5 global BA call (SVGPU_BA_TRACE), steady state (third call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stella_vslam_amd import optimize, synthetic
sc = synthetic.ba_scene_large()
ba = optimize.local_bundle_adjuster()
ba.optimize_global_flat(sc, num_iter=10)
ba.optimize_global_flat(sc, num_iter=10)
os.environ["SVGPU_BA_TRACE"] = "1"
t0 = time.perf_counter()
r = ba.optimize_global_flat(sc, num_iter=10)
print("ms", (time.perf_counter() - t0) * 1e3, r["stats"])
