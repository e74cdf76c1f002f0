import sys, os, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import optimize, synthetic
sc = synthetic.ba_scene()
ba = optimize.local_bundle_adjuster()
os.environ.pop("SVGPU_BA_TRACE", None)
ba.optimize_flat(sc); ba.optimize_flat(sc)
os.environ["SVGPU_BA_TRACE"] = "1"
r = ba.optimize_flat(sc)
print(r["stats"])
