#!/bin/bash
# round 5, step 17: every instance of the generated two-lane solve (68 ... 128): bit-identity, then the k sweep at 200 000 cells
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "two_lane_solve or packing_by_sweep" > gpurun_out/r5_s17_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 gpurun_out/r5_s17_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s17_tests.log; exit 1; fi
run() {  # label, k, env...
  local label=$1 k=$2; shift 2
  env "$@" timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_s17_$label.json 2> gpurun_out/r5_s17_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s17_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['ms_per_step'],2), {a: round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, round(d['nnls_mean_sweeps']['h'],1), round(d['nnls_mean_sweeps']['h_per_wave'],1))
PY
}
for k in 66 70 80 90 96 100 104 112 120 128; do
  run asm_k$k $k X=1
  run compiled_k$k $k SGL_NNLS_NO_ASM=1
done
