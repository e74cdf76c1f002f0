#!/usr/bin/env python3
"""LDS and matrix-core view of a rocprofv3 PMC pass (`--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE`, its own run) of `bench.py`, per kernel and launch:
  lds_bank_conflict_frac   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of the LDS unit's busy cycles lost to bank conflicts
  lds_util                 SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs)                     (rocprofv3's LdsUtil)
  mfma_busy_frac           SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)           (rocprofv3's MfmaUtil)
  mfma_i8_ops              SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 integer operations
kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.  usage: tools/pmc_lds_mfma.py <pmc dir> profiles/r02_lds_mfma.json [frames per launch]"""
import json, os, sys
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash

t = pd.read_csv(f"{sys.argv[1]}/p_counter_collection.csv")
t["k"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"^void\s+", "", regex=True).str.replace(r"[<(].*$", "", regex=True)  # "void k_blur<64>(...)" -> "k_blur": templated kernels carry their return type and arguments
g = t.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
alias = {"k_pyramid": "k_resize", "k_pyramid_lds": "k_resize", "k_bf_mfma": "k_bf_topk", "k_describe_bands": "k_describe"}
out = {"csrc_hash": csrc_hash(), "batch": int(sys.argv[3]) if len(sys.argv) > 3 else 256, "kernels": {}}
for k, r in g.iterrows():
    if not k.startswith("k_"):
        continue
    cyc = float(r["GRBM_GUI_ACTIVE"]) / 8.0
    idx = float(r.get("SQ_LDS_IDX_ACTIVE", 0.0))
    out["kernels"][alias.get(k, k)] = {
        "kernel_cycles": round(cyc),
        "lds_bank_conflict_frac": round(float(r.get("SQ_LDS_BANK_CONFLICT", 0.0)) / idx, 4) if idx > 0 else None,
        "lds_util": round(idx / (cyc * 256), 4) if cyc > 0 else None,
        "mfma_busy_frac": round(float(r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / (cyc * 1024), 4) if cyc > 0 else None,
        "mfma_i8_ops": int(float(r.get("SQ_INSTS_VALU_MFMA_MOPS_I8", 0.0)) * 512)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
