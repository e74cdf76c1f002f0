# k_ba_schur_rhs at config 5 against the share size (pairs per unit)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for sp in ${SHARES:-128 192 256 320 448}; do
  export SVGPU_BA_SHARE_PAIRS=$sp
  timeout 200 rocprofv3 --kernel-trace --stats --truncate-kernels -d /tmp/ss_$sp -o k --output-format csv -- python $R/tools/ba_prof.py global > /dev/null 2>&1
  echo "share_pairs $sp: $(grep 'k_ba_schur_rhs\|k_ba_sys_fin' $(find /tmp/ss_$sp -name 'k_kernel_stats.csv' | head -1) | cut -d, -f1,4 | tr '\n' ' ')"
done
