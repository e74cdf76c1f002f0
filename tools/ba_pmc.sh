# PMC passes over one global-BA run (tools/ba_prof.py global): HBM bytes, L2 hit rate, VALU share per BA kernel -> gpurun_out/ba_pmc.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
w=${1:-global}
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --truncate-kernels -d /tmp/bapmc_$tag -o p --output-format csv -- python $R/tools/ba_prof.py $w > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/bapmc_$tag/**/p_counter_collection.csv",recursive=True)
if not f: print("no output for $set"); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    key=(r["Dispatch_Id"]); 
    if key not in seen: seen.add(key); cnt[k]+=1
for k in ["k_ba_schur_rhs","k_ba_lin","k_sky_band","k_ba_update","k_ba_chi2","k_seg_backward","k_pose_major","k_pair_emit"]:
    if k in acc: print("%-16s n=%3d "%(k,cnt[k]) + "  ".join("%s/launch=%.4g"%(c,v/cnt[k]) for c,v in acc[k].items()))
PY
done
