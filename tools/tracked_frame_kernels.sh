R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/tfp -o k --output-format csv -- python $R/tools/tracked_frame_prof.py 30 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/tfp/**/k_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-28s calls %5s avg %8.1f us  min %8.1f max %8.1f total %7.2f ms"%(r['Name'][:28],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
cd $R; SVGPU_MATCH_TRACE=1 timeout 100 python tools/tracked_frame_prof.py 2 1 2>&1 | grep "\[match\]" | tail -14
