"""Phase clocks of k_fast.  Build the library with the clocks first:
    touch stella_vslam_amd/csrc/orb_kernels.hip && make -C stella_vslam_amd/csrc EXTRA=-DSV_FAST_PROF
then run this on the GPU and rebuild without the flag.  Prints the share of wave-time (s_memtime, summed over the waves) per phase."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from stella_vslam_amd import synthetic  # noqa: E402
from stella_vslam_amd._lib import lib  # noqa: E402
from stella_vslam_amd.pipeline import BatchExtractor  # noqa: E402

W, H, B = 640, 480, 64
ex = BatchExtractor(W, H, B)
ex.upload(synthetic.frame_sequence(B, W, H, seed=0x5EED))
for _ in range(3):
    ex.extract()
ex.ctx.synchronize()
out = (C.c_ulonglong * 8)()
lib().svgpu_debug_fast_prof(out)
ex.extract()
ex.ctx.synchronize()
lib().svgpu_debug_fast_prof(out)
v = np.array(list(out), float)
names = ["ROI load + sync", "pass A (quick test + queue)", "pass B (arc score)", "barrier after B", "NMS + arg-max", "barrier after NMS", "tail (global atomics)", "-"]
for n, x in zip(names, v):
    print(f"{n:32s} {x / v.sum() * 100:6.2f} %   {x / (B * 216 * 4):9.1f} clocks / wave")
