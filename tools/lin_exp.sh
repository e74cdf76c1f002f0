R=$PWD; mkdir -p gpurun_out/r03l; cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-base}; do
  if [ $v = base ]; then unset SVGPU_LIB_PATH; else export SVGPU_LIB_PATH=$R/stella_vslam_amd/variants/libsvgpu_$v.so; fi
  for w in global local; do
    timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03l/${v}_$w -o k --output-format csv -- python $R/tools/ba_prof.py $w > /dev/null 2> $R/gpurun_out/r03l/${v}_$w.log
    echo "== $v $w"; python3 - <<PY
import csv
for r in list(csv.DictReader(open("$R/gpurun_out/r03l/${v}_$w/k_kernel_stats.csv")))[:9]:
    print("  %-28s calls %4s avg %8.1f us total %7.3f ms" % (r["Name"].replace("(anonymous namespace)::","").split("(")[0][:28], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
    rm -f $R/gpurun_out/r03l/${v}_$w/*trace.csv
  done
done
