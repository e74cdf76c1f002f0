#!/usr/bin/env python3
"""Per-kernel instruction counters of the largest launches in a rocprofv3 PMC pass (SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
GRBM_GUI_ACTIVE): usage tools/pmc_kernels.py <dir with p_counter_collection.csv>"""
import sys
import pandas as pd
t = pd.read_csv(f"{sys.argv[1]}/p_counter_collection.csv")
t["k"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.split("(").str[0]
for k in sorted(t.k.unique()):
    if not k.startswith("k_"):
        continue
    f = t[t.k == k]
    gmax = f.Grid_Size.max()
    g = f[f.Grid_Size == gmax].groupby("Counter_Name")["Counter_Value"].mean()
    cyc = g["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:16s} grid {gmax:9d} VALU {g['SQ_INSTS_VALU'] / 1e6:7.1f}M SALU {g['SQ_INSTS_SALU'] / 1e6:6.1f}M LDS {g['SQ_INSTS_LDS'] / 1e6:6.1f}M "
          f"cycles {cyc / 1e3:7.1f}k valu_frac {g['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.3f}")
