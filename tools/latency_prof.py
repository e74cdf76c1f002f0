#!/usr/bin/env python3
"""Workload for a rocprofv3 kernel trace of the single-frame host-buffer entry points: `reps` x (svgpu_orb_extract, svgpu_match_bruteforce)."""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from stella_vslam_amd import feature, match, synthetic
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seq = synthetic.frame_sequence(3, 640, 480, seed=0x5EED)
ext = feature.orb_extractor(feature.orb_params())
m = match.robust(0.8, True, ext.ctx)
k0, d0 = ext.extract(seq[0])
for i in range(reps):
    k1, d1 = ext.extract(seq[1 + (i & 1)])
    m.brute_force_match(d1, k1["angle"], d0, k0["angle"])
