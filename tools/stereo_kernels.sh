R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/stp -o k --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-ba --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/stp/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "stereo" in r["Name"] or "k_fast" in r["Name"]:
        print("%-28s calls %5s avg %8.1f us  min %8.1f max %8.1f total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
