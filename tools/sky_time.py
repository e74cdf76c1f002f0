import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stella_vslam_amd import optimize, synthetic
from stella_vslam_amd._lib import lib
sc = synthetic.ba_scene_large()
ba = optimize.local_bundle_adjuster()
ba.optimize_global_flat(sc, num_iter=10)
L = lib()
L.svgpu_profile_select(ba.ctx.handle, b"ba_solve")
r = ba.optimize_global_flat(sc, num_iter=10)
ms, n = C.c_double(), C.c_longlong()
L.svgpu_profile_read(ba.ctx.handle, C.byref(ms), C.byref(n))
print("ba_solve: %.3f ms per solve (%d), trials %d" % (ms.value / max(n.value, 1), n.value, r["stats"]["lm_trials"]))
