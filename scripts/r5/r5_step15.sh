#!/bin/bash
# round 5, step 15: the two-lane NNLS (ranks 65 - 128) as generated asm: bit-identity against the compiled kernel, then timing
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
KS="${HALF_KS:-97 or 98 or 99 or 100}"
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "two_lane_solve and ($KS)" > gpurun_out/r5_s15_tests.log 2>&1; rc=$?
echo "two-lane asm tests rc=$rc"; tail -5 gpurun_out/r5_s15_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s15_tests.log; exit 1; fi
for k in ${HALF_BENCH_KS:-100}; do
  for v in asm compiled; do
    if [ $v = compiled ]; then export SGL_NNLS_NO_ASM=1; else unset SGL_NNLS_NO_ASM; fi
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5_s15_k${k}_$v.json 2> gpurun_out/r5_s15_k${k}_$v.err
    python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s15_k${k}_$v.json').read().strip().splitlines()[-1])
print('$v', 'k=$k', round(d['ms_per_step'],2), {a: round(b,2) for a,b in d['phases_ms_per_step'].items() if b})
PY
  done
done
unset SGL_NNLS_NO_ASM
