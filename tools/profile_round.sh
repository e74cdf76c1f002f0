#!/bin/bash
# Everything the bench line's roofline object rests on, taken on the GPU box in one go (run from the repository root):
#   kernel trace + stats of bench.py, three PMC passes (FETCH_SIZE / WRITE_SIZE / instruction counters, each its own run as the
#   guide prescribes), the VALU busy-cycle pass, kernel stats of the two bundle adjusters, then the bench line itself (which now
#   finds traffic / valu_issue files stamped with the hash of the sources it runs).  usage: tools/profile_round.sh r02
set -u
TAG=${1:-r04}
BATCH=${BATCH:-1024}   # frames per launch of the headline legs (bench.py's default batch)
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
exec < /dev/null
HEAD="--steps 3 --warmup 1 --no-ba --no-cpu-baseline --no-extra --batch2 0"  # (one batch size per trace: the averages below are per-launch figures)
# headline legs only: every launch of a front-end / matcher kernel in this trace is a $BATCH-frame launch, so the averages of the stats file
# are the per-launch durations the bench line quotes
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ks -o k --output-format csv -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-ba --batch2 0 > $OUT/bench_under_rocprof.json 2> $OUT/ks.log
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o p --output-format csv -- python $R/bench.py $HEAD > /dev/null 2> $OUT/pmc_f.log
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w -o p --output-format csv -- python $R/bench.py $HEAD > /dev/null 2> $OUT/pmc_w.log
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_gi -o p --output-format csv -- python $R/bench.py $HEAD > /dev/null 2> $OUT/pmc_gi.log
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc_act -o p --output-format csv -- python $R/bench.py $HEAD > /dev/null 2> $OUT/pmc_act.log
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o p --output-format csv -- python $R/bench.py $HEAD > /dev/null 2> $OUT/pmc_lds.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ba_local -o k --output-format csv -- python $R/tools/ba_prof.py local > /dev/null 2> $OUT/ba_local.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ba_global -o k --output-format csv -- python $R/tools/ba_prof.py global > /dev/null 2> $OUT/ba_global.log
# launches per tracked frame through the drop-in chain (mode 2: svgpu_track_motion + svgpu_track_local_map): two traces that differ by 20 frames;
# runtime copies count as launches (memory-copy trace)
for n in 10 30; do
  timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/tf_$n -o k --output-format csv -- python $R/tools/tracked_frame_prof.py $n 2 > /dev/null 2> $OUT/tf_$n.log
done
cd $R
python - <<PYEOF > $OUT/tracked_frame_launches.json
import csv, json, os
def calls(n, what):
    f = "$OUT/tf_%d/k_%s_stats.csv" % (n, what)
    if not os.path.exists(f):
        return {}
    return {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(f))}
def per_frame(what):
    a, b = calls(10, what), calls(30, what)
    out = {}
    for k in b:
        if b.get(k, 0) != a.get(k, 0):  # template instances of one kernel (k_track_cand<0> / <1>, k_pose_opt<..>) add up under its bare name
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            out[name] = out.get(name, 0.0) + (b.get(k, 0) - a.get(k, 0)) / 20.0
    return out
k, c = per_frame("kernel"), per_frame("memory_copy")
print(json.dumps({"what": "kernel launches and runtime copies per tracked frame through the drop-in chain (tracked_frame_chain: one submission per half of tracking_module's per-frame chain): difference of two rocprofv3 traces 20 frames apart",
                  "launches_per_frame": round(sum(k.values()) + sum(c.values()), 2), "kernels_per_frame": round(sum(k.values()), 2), "copies_per_frame": round(sum(c.values()), 2),
                  "host_syncs_per_frame": "2 (counted by the tracker itself: svgpu_tracker_counters, printed in the bench line as tracked_frame.chain.host_syncs)", "by_kernel": k, "by_copy": c}, indent=1))
PYEOF
cp $OUT/tracked_frame_launches.json profiles/${TAG}_tracked_frame_launches.json
python tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w profiles/${TAG}_traffic.json $BATCH > /dev/null
python tools/pmc_valu.py $OUT/pmc_gi profiles/${TAG}_valu_issue.json $BATCH $OUT/pmc_act > /dev/null
python tools/pmc_lds_mfma.py $OUT/pmc_lds profiles/${TAG}_lds_mfma.json $BATCH > $OUT/lds_mfma.log 2>&1
cp profiles/${TAG}_traffic.json profiles/${TAG}_valu_issue.json profiles/${TAG}_lds_mfma.json $OUT/
timeout 400 python bench.py 2> $OUT/bench.log < /dev/null | tail -1 > $OUT/bench.json
cp bench_detail.json $OUT/bench_detail.json 2>/dev/null
timeout 300 python tools/ba_bench.py --global > $OUT/ba_bench.log 2>&1 < /dev/null
cp gpurun_out/ba_bench.json $OUT/ba_bench.json 2>/dev/null
# keep the merge-back small: the raw traces stay on the box, the summaries travel
rm -f $OUT/ks/*kernel_trace.csv $OUT/ba_local/*kernel_trace.csv $OUT/ba_global/*kernel_trace.csv $OUT/tf_*/*kernel_trace.csv
rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_gi $OUT/pmc_act $OUT/pmc_lds
ls -la $OUT $OUT/ks | head -40
tail -c 600 $OUT/bench.json
