"""The two-stream front-end step, eight times in one process, with the matcher on ONE stream kept for all calls (MODE=cached, default) or on a fresh
torch.cuda.Stream per call (MODE=fresh): some streams of torch's pool land on a hardware queue that serialises against the extraction stream's --
calls 2 and 5 drop from 212-215 k to 163-166 k frames/s at 256 frames per step (round 5; bench.matcher_stream keeps one stream per process)."""
import sys, os, pathlib
ROOT = str(pathlib.Path(__file__).resolve().parent.parent)
sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch; torch.cuda.set_device(0)
import bench, numpy as np
from stella_vslam_amd import feature, synthetic
from stella_vslam_amd._lib import lib
ctx = feature.Context(0, priority=1); L = lib()
fr = synthetic.frame_sequence(256, 640, 480, seed=0x5EED)
orig = torch.cuda.Stream
mode = os.environ.get('MODE', 'cached')
cache = {}
def mk(*a, **k):
    if mode == 'cached':
        if 's' not in cache: cache['s'] = orig(*a, **k)
        return cache['s']
    return orig(*a, **k)
torch.cuda.Stream = mk
for i in range(8):
    fe = bench.run_front_end(ctx, L, fr, 256, 100, 3, lambda: None, 1, profile=True)
    print(i, round(256*100/fe["dt"]), round(fe["ms_per_step_unprofiled"],4), {k: round(v[0]/max(v[1],1),3) for k,v in fe["per_kernel"].items() if k in ('k_resize','k_select','k_describe')}, flush=True)
