# kernel stats of the single-frame host-buffer entry points (extract + one brute-force pair)
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/latp -o k --output-format csv -- python $R/tools/latency_prof.py 30 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/latp/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-28s calls %5s avg %8.1f us  min %8.1f max %8.1f total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
