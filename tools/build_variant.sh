#!/bin/bash
# A/B builds of libsvgpu.so for kernel experiments: tools/build_variant.sh NAME "-DBLUR_ROWS=64 ..." -> stella_vslam_amd/variants/libsvgpu_NAME.so
# (git-ignored like every .so, shipped to the GPU box by gpurun; selected at run time with SVGPU_LIB_PATH=<that file>).
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=/tmp/svgpu_variants/$NAME
mkdir -p $OBJ $ROOT/stella_vslam_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -I$ROOT/include -I$ROOT/stella_vslam_amd/csrc $EXTRA"
pids=()
for f in $ROOT/stella_vslam_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $OBJ/$(basename ${f%.hip}).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/stella_vslam_amd/variants/libsvgpu_$NAME.so $OBJ/*.o -ldl
echo built $ROOT/stella_vslam_amd/variants/libsvgpu_$NAME.so
