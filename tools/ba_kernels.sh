# per-kernel times of the bundle adjusters: `sh tools/ba_kernels.sh [local|global]`
R=$PWD; W=${1:-local}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/bak
timeout -k 5 200 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/bak -o k --output-format csv -- python $R/tools/ba_prof.py $W > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/bak/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-28s calls %5s avg %8.1f us  min %8.1f max %8.1f total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
