"""Global BA (config 5, or the 9.6 M-observation scene with BIG=1) on the bench scene as generated (landmark ids at random around the loop) or with the
landmarks renumbered by first observer before the call (COHERENT=first | min).  With SVGPU_BA_NO_RENUMBER=1 / SVGPU_BA_NO_UNITS=1 /
SVGPU_BA_CHUNK_SHIFT=n and SVGPU_BA_DBG=schur (unit lives of the last Schur launch) this is the A/B harness behind DESIGN section 6's figures."""
import sys, os, time, pathlib
ROOT = str(pathlib.Path(__file__).resolve().parent.parent)
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from stella_vslam_amd import optimize, synthetic
big = os.environ.get('BIG')
sg = synthetic.ba_scene_large(num_lm=1600000) if big else synthetic.ba_scene_large()
if os.environ.get('COHERENT'):
    L = sg['points'].shape[0]; k = len(sg['obs_point']) // L
    first = sg['obs_pose'].reshape(L, k).min(1) if os.environ['COHERENT'] == 'min' else sg['obs_pose'].reshape(L, k)[:, 0]
    order = np.argsort(first, kind='stable')
    sg = dict(sg)
    sg['points'] = np.ascontiguousarray(sg['points'][order]); sg['points_gt'] = np.ascontiguousarray(sg['points_gt'][order])
    for name, w in (('obs_pose', 1), ('obs_uvr', 3), ('obs_inv_sigma_sq', 1), ('obs_huber', 1)):
        a = sg[name].reshape(L, k, *([3] if w == 3 else []))
        sg[name] = np.ascontiguousarray(a[order].reshape(sg[name].shape))
ba = optimize.local_bundle_adjuster()
ba.optimize_global_flat(sg, num_iter=10)
for it in (10, 10):
    t0 = time.perf_counter()
    r = ba.optimize_global_flat(sg, num_iter=it)
    print("iters", it, "ms", (time.perf_counter() - t0) * 1e3, r["stats"]["chi2_final"] if "stats" in r else "", flush=True)
