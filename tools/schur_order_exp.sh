# k_ba_schur_rhs at config 5 under both unit orders: duration (kernel trace) and L2 / fabric traffic (PMC passes)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for o in ${ORDERS:-0 1}; do
  export SVGPU_BA_SCHUR_ORDER=$o
  echo "== order $o"
  timeout 200 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/so_$o -o k --output-format csv -- python $R/tools/ba_prof.py global > /dev/null 2>&1
  grep "k_ba_schur_rhs" $(find /tmp/so_$o -name "k_kernel_stats.csv" | head -1) | cut -d, -f1-4
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --pmc $set --truncate-kernels -d /tmp/sop_${o}_$tag -o p --output-format csv -- python $R/tools/ba_prof.py global > /dev/null 2>&1
    python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/sop_${o}_$tag/**/p_counter_collection.csv",recursive=True)
if not f: print("no output for $set"); raise SystemExit
acc=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"]!="k_ba_schur_rhs": continue
    acc[r["Counter_Name"]]+=float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
print("  ".join("%s/launch=%.4g"%(c,v/max(1,len(seen))) for c,v in acc.items()))
PY
  done
done
