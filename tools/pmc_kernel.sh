#!/bin/bash
# Memory-path / issue diagnosis of ONE front-end kernel: separate PMC passes (TA, TCP, SQ waits, instruction levels) over the 1 024-frame
# extraction.  usage (GPU box, repo root): tools/pmc_kernel.sh TAG k_describe [lib.so]   -> gpurun_out/pmck_TAG.txt
TAG=$1; KERNEL=${2:-k_describe}; LIB=${3:-}
R=$PWD; OUT=$R/gpurun_out/pmck_$TAG; mkdir -p $OUT
[ -n "$LIB" ] && export SVGPU_LIB_PATH=$R/$LIB
cd /tmp && export TMPDIR=/tmp
export ORB_B=${ORB_B:-1024}
i=0
for set in "TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python $R/tools/orb_kernel_times.py $KERNEL > $OUT/log$i 2>&1
done
cd $R
python - <<PY > gpurun_out/pmck_$TAG.txt
import pandas as pd, glob
rows = {}
for f in sorted(glob.glob("$OUT/p*/p_counter_collection.csv")):
    t = pd.read_csv(f)
    t = t[t["Kernel_Name"].str.contains("$KERNEL")]
    g = t.groupby("Counter_Name")["Counter_Value"].mean()
    for k, v in g.items():
        rows[k] = v
print("$TAG $KERNEL per launch (mean over launches), ORB_B=$ORB_B")
for k, v in rows.items():
    print(f"{k:45s} {v:18.1f}")
PY
cat gpurun_out/pmck_$TAG.txt
tail -2 $OUT/log1
rm -rf $OUT
