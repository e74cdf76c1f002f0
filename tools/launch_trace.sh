# per-launch kernel durations of one BA call (in launch order): tools/launch_trace.sh global|local  -> anomalies that averages hide
R=$PWD; cd /tmp; export TMPDIR=/tmp
w=${1:-global}
timeout 200 rocprofv3 --kernel-trace --truncate-kernels -d /tmp/lt_$w -o k --output-format csv -- python $R/tools/ba_prof.py $w > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/lt_$w/**/k_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# the last call = everything after the last k_ba_begin-preceding upload; print the tail half
n=len(rows); rows=rows[n//2:]
t0=int(rows[0]["Start_Timestamp"]); prev_end=t0
for r in rows:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.1f us  gap %7.1f  dur %8.1f  %-28s grid %s" % ((s-t0)/1e3, (s-prev_end)/1e3, (e-s)/1e3, r["Kernel_Name"][:28], r.get("Grid_Size_X","")))
    prev_end=e
PY
