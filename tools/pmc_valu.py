#!/usr/bin/env python3
"""VALU-issue view of a rocprofv3 PMC pass (`--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE`, its own run with
--kernel-trace/--stats absent, as MI355X_MICROARCH.md prescribes) of `bench.py`: for every kernel the wave-level VALU instructions per
launch, the kernel's duration in shader cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the fraction of the chip's
VALU issue slots those instructions fill at 4 cycles per wave64 integer / packed-16 instruction on 256 CUs x 4 SIMDs
(measured: SQ_ACTIVE_INST_VALU ~= SQ_INSTS_VALU quad-cycles for these kernels; tools/ubench/bcnt.hip: 4.6 cycles per xor/bcnt).
The cost per instruction is calibrated, not assumed: tools/ubench/valu_issue.hip (profiles/r02_ubench_valu.json) measures 3.7-4.2
cycles per wave64 instruction per SIMD for the opcodes these kernels are made of (v_perm, v_pk_*, v_mad_i24, v_dot*, v_bcnt, v_fma;
v_add_u32 / v_xor_b32: 2.2), and an optional second pass (`--pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU`) gives the busy quad-cycles per
instruction of every kernel (`active_cycles_per_valu_inst` = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU).  Under the guide's nominal
2 cycles per instruction every `valu_issue_frac` below would halve.
usage: tools/pmc_valu.py gpurun_out/pmc_gi profiles/r02_valu_issue.json [frames per launch] [gpurun_out/pmc_active]"""
import json, os, sys
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # the kernel sources these counters were taken on: bench.py reports them only while the sources still hash to this

t = pd.read_csv(f"{sys.argv[1]}/p_counter_collection.csv")
t["k"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"^void\s+", "", regex=True).str.replace(r"[<(].*$", "", regex=True)  # "void k_blur<64>(...)" -> "k_blur": templated kernels carry their return type and arguments
g = t.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
alias = {"k_pyramid": "k_resize", "k_pyramid_lds": "k_resize", "k_bf_mfma": "k_bf_topk", "k_describe_bands": "k_describe"}
SIMDS, CYC = 256 * 4, 4
out = {"csrc_hash": csrc_hash(), "batch": int(sys.argv[3]) if len(sys.argv) > 3 else 256, "simds": SIMDS, "cycles_per_valu_wave_inst": CYC,
       "note": "valu_issue_frac = SQ_INSTS_VALU * 4 / (1024 SIMDs * kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; both legs of the "
               "two-stream pipeline run while a kernel is measured, so its cycles include the share the other stream's kernels take",
       "kernels": {}}
for k, r in g.iterrows():
    if not k.startswith("k_"):
        continue
    cyc = float(r["GRBM_GUI_ACTIVE"]) / 8.0
    out["kernels"][alias.get(k, k)] = {"valu_wave_insts": round(float(r["SQ_INSTS_VALU"])), "salu_wave_insts": round(float(r["SQ_INSTS_SALU"])),
                                       "lds_wave_insts": round(float(r["SQ_INSTS_LDS"])), "kernel_cycles": round(cyc),
                                       "valu_issue_frac": round(float(r["SQ_INSTS_VALU"]) * CYC / (SIMDS * cyc), 4)}
if len(sys.argv) > 4:
    a = pd.read_csv(f"{sys.argv[4]}/p_counter_collection.csv")
    a["k"] = a["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"^void\s+", "", regex=True).str.replace(r"[<(].*$", "", regex=True)  # "void k_blur<64>(...)" -> "k_blur": templated kernels carry their return type and arguments
    ga = a.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
    for k, r in ga.iterrows():
        kk = alias.get(k, k)
        if kk in out["kernels"] and float(r["SQ_INSTS_VALU"]) > 0:
            out["kernels"][kk]["active_cycles_per_valu_inst"] = round(4.0 * float(r["SQ_ACTIVE_INST_VALU"]) / float(r["SQ_INSTS_VALU"]), 3)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v["valu_issue_frac"] for k, v in out["kernels"].items() if v["valu_wave_insts"] > 1e6}, indent=1))
