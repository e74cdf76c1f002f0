R=$PWD
for i in 1 2; do
for v in new head; do
  export SVGPU_LIB_PATH=$R/stella_vslam_amd/variants/libsvgpu_$v.so
  echo "$v: $(timeout 100 python tools/sky_time.py 2>&1 | tail -1) | $(timeout 100 python tools/ba_bench.py --global 2>&1 | grep '^global' | cut -c1-40)"
done; done
export SVGPU_LIB_PATH=$R/stella_vslam_amd/variants/libsvgpu_new.so
timeout 600 python -m pytest tests/test_gpu_ba.py -x -q -m gpu -k "global or envelope or alternative" > gpurun_out/t.log 2>&1; grep -a "passed\|failed\|rror" gpurun_out/t.log | tail -5
