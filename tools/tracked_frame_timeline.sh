# timeline of ONE tracked frame of the chain (kernels + copies with the gaps between them): `sh tools/tracked_frame_timeline.sh [mode]`
R=$PWD; M=${1:-2}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tft
timeout -k 5 120 rocprofv3 --kernel-trace --memory-copy-trace --truncate-kernels -d /tmp/tft -o k --output-format csv -- python $R/tools/tracked_frame_prof.py 60 $M > /dev/null 2>&1
python3 - <<PY
import csv,glob
ev=[]
for f in glob.glob("/tmp/tft/**/k_kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:30]))
for f in glob.glob("/tmp/tft/**/k_memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Direction'][:30]))
ev.sort()
# frames start with the host-to-device copy in front of k_pyramid_lds
idx=[i for i,e in enumerate(ev) if e[2].startswith('k_pyramid')]
i0=idx[len(idx)//2]-1; i1=idx[len(idx)//2+1]-1
t0=ev[i0][0]; prev=None
for s,e,n in ev[i0:i1]:
    print("%8.1f us  +%6.1f gap  %-30s %7.1f us"%((s-t0)/1e3, 0 if prev is None else (s-prev)/1e3, n, (e-s)/1e3)); prev=e
print("frame period %.1f us"%((ev[i1][0]-t0)/1e3))
PY
