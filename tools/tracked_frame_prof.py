#!/usr/bin/env python3
"""Workload for a rocprofv3 kernel trace of the tracked-frame chain through the drop-in classes: `reps` tracked frames (after the map set-up and
one warm-up frame).  Two traces with different `reps` give the launches per tracked frame by difference (tools/profile_round.sh)."""
import ctypes as C, os, sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from stella_vslam_amd import synthetic
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
resident = int(sys.argv[2]) if len(sys.argv) > 2 else 1
host = C.CDLL(str(ROOT / "stella_vslam_amd" / "host" / "libsvgpu_host.so"))
seq = np.ascontiguousarray(synthetic.frame_sequence(4, 640, 480, seed=0x5EED))
ms, cnt = np.zeros(8), np.zeros(8, np.int32)
rc = host.svgpu_host_tracked_frame(C.c_void_p(seq.ctypes.data), 4, 640, 480, reps, resident, C.c_void_p(ms.ctypes.data), C.c_void_p(cnt.ctypes.data))
print(rc, ms.tolist(), cnt.tolist())
