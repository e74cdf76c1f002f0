# per-kernel times of the tracked-frame path through the drop-in classes: `sh tools/tracked_frame_kernels.sh [mode]` (2 = the chain, default)
R=$PWD; M=${1:-2}; cd /tmp; export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --truncate-kernels -d /tmp/tfp -o k --output-format csv -- python $R/tools/tracked_frame_prof.py 100 $M > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/tfp/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-28s calls %5s avg %8.1f us  min %8.1f max %8.1f total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['TotalDurationNs'])/1e6))
for f in glob.glob("/tmp/tfp/**/k_memory_copy_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-28s calls %5s avg %8.1f us  total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
