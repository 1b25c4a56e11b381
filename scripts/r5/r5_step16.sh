#!/bin/bash
# round 5, step 16: two-lane asm solve with sweep-count packing; head-per-tail ratio of the interleave (1 / 2 / 3) A/B
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "(two_lane_solve and (97 or 98 or 99 or 100)) or packing_by_sweep" > gpurun_out/r5_s16_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r5_s16_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s16_tests.log; exit 1; fi
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python3 bench.py --k 100 --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_s16_$label.json 2> gpurun_out/r5_s16_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s16_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['ms_per_step'],2), {a: round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, d['nnls_mean_sweeps'])
PY
}
run hpt2_pack X=1
run hpt2_nopack SGL_NNLS_NO_PACK=1
run hpt1_pack SGL_LIB_PATH=$PWD/build/lib_half_hpt1.so
run hpt3_pack SGL_LIB_PATH=$PWD/build/lib_half_hpt3.so
run compiled SGL_NNLS_NO_ASM=1
run hpt2_pack_again X=1
