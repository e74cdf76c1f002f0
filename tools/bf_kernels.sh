# the matcher kernels alone: durations and VALU instruction counts.  usage: sh tools/bf_kernels.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/bfk /tmp/bfp
timeout -k 5 120 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/bfk -o k --output-format csv -- python $R/tools/bf_alone.py 10 > /dev/null 2>&1
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --truncate-kernels -d /tmp/bfp -o p --output-format csv -- python $R/tools/bf_alone.py 3 > /dev/null 2>&1
python3 - <<PY
import csv,glob
import pandas as pd
f=glob.glob("/tmp/bfk/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if r['Name'].startswith('k_bf'): print("%-20s calls %4s avg %8.1f us min %8.1f"%(r['Name'][:20],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3))
t=pd.read_csv(glob.glob("/tmp/bfp/**/p_counter_collection.csv",recursive=True)[0])
g=t[t.Kernel_Name.str.startswith('k_bf')].groupby(["Kernel_Name","Counter_Name"])["Counter_Value"].mean().unstack()
print(g.round(0).to_string())
PY
