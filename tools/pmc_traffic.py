#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) of
`bench.py` into profiles/<tag>_traffic.json: mean HBM-side bytes per launch of every kernel.
FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a streaming read
(MI355X_MICROARCH.md, section HBM, states it for 16 B/lane; tools/ubench/pmc_calib.hip measured the same factor for 1, 4, 8
and 16 B/lane on this toolchain -- profiles/r01_pmc_calibration.json) and WRITE_SIZE is exact: fetch is doubled here.
usage: tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_traffic.json [frames per launch, default 256]
"""
import json, os, sys
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # the kernel sources these counters were taken on: bench.py reports them only while the sources still hash to this

def per_kernel(d, counter):
    t = pd.read_csv(f"{d}/p_counter_collection.csv")
    t["k"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"^void\s+", "", regex=True).str.replace(r"[<(].*$", "", regex=True)  # "void k_blur<64>(...)" -> "k_blur": templated kernels carry their return type and arguments
    t = t[t.Counter_Name == counter]
    return t.groupby("k")["Counter_Value"].mean()

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
FETCH_CORRECTION = 2.0  # profiles/r01_pmc_calibration.json
out = {"csrc_hash": csrc_hash(), "batch": int(sys.argv[4]) if len(sys.argv) > 4 else 256, "unit": "bytes per launch (FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024; mean over the launches of the run)",
       "caveat": "gfx950: FETCH_SIZE counts half of the streamed bytes at every access width (calibrated, factor 2 applied); "
                 "Infinity-Cache hits are included", "kernels": {}}
alias = {"k_pyramid": "k_resize", "k_pyramid_lds": "k_resize", "k_bf_mfma": "k_bf_topk", "k_describe_bands": "k_describe"}
for k in sorted(set(f.index) | set(w.index)):
    if not k.startswith("k_"):
        continue
    fe, wr = float(f.get(k, 0)) * 1024 * FETCH_CORRECTION, float(w.get(k, 0)) * 1024
    out["kernels"][alias.get(k, k)] = {"fetch": round(fe), "write": round(wr), "total": round(fe + wr)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
