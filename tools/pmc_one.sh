#!/bin/bash
# SQ instruction counters of the front-end kernels for one library build: tools/pmc_one.sh TAG [lib.so]   (run on the GPU box from the repo root)
TAG=$1; LIB=${2:-}
R=$PWD; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
[ -n "$LIB" ] && export SVGPU_LIB_PATH=$R/$LIB
cd /tmp && export TMPDIR=/tmp
ORB_B=${ORB_B:-256} timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/p -o p --output-format csv -- python $R/tools/orb_kernel_times.py k_fast > $OUT/log 2>&1
cd $R
python - <<PY
import pandas as pd
t = pd.read_csv("$OUT/p/p_counter_collection.csv")
t["k"] = t["Kernel_Name"].str.replace("(anonymous namespace)::", "", regex=False).str.replace(r"^void\s+", "", regex=True).str.replace(r"[<(].*$", "", regex=True)
g = t.groupby(["k", "Counter_Name"])["Counter_Value"].mean().unstack()
print("$TAG"); print(g.to_string())
PY
rm -rf $OUT/p
