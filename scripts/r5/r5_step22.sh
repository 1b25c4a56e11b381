#!/bin/bash
# round 5, step 22: which of the two one-launch reductions costs time at the shard (kernel stats of each form)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "reductions or scale_and_cor" > gpurun_out/r5_s22_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 gpurun_out/r5_s22_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s22_tests.log; exit 1; fi
for v in new old; do
  if [ $v = old ]; then export SGL_ROWSUM_TWO_KERNELS=1 SGL_COR_TWO_KERNELS=1; else unset SGL_ROWSUM_TWO_KERNELS SGL_COR_TWO_KERNELS; fi
  rm -rf gpurun_out/s22_$v.d
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s22_$v.d -- python3 bench.py --cells 125000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r5_s22_$v.json 2> gpurun_out/r5_s22_$v.err
  python3 scripts/pmc_summary.py $(find gpurun_out/s22_$v.d -name "*.db" | head -1) > gpurun_out/r5_s22_${v}_stats.csv 2>&1
  rm -rf gpurun_out/s22_$v.d
  echo "== $v"; grep -i "rowsum\|cor_\|partial_sum_small\|scale_kernel" gpurun_out/r5_s22_${v}_stats.csv | cut -c1-120
done
