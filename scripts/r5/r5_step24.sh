#!/bin/bash
# round 5, step 24: packed single-round H solve with the second layer of workgroups in ascending order (snake) vs launch order
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_nmf.py tests/test_gpu_ops.py -x -q -m gpu -k "packing_by_sweep or generated_sweep" > gpurun_out/r5_s24_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 gpurun_out/r5_s24_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s24_tests.log; exit 1; fi
run() {  # label, env, args
  local label=$1 e=$2; shift 2
  env $e timeout 300 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r5_s24_$label.json 2> gpurun_out/r5_s24_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s24_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['value'],2), round(d['ms_per_step'],3), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b}, d['nnls_mean_sweeps']['h_per_wave'])
PY
}
for rep in 1 2; do
  run snake_125k_$rep X=1 --cells 125000 --steps 40 --warmup 5
  run plain_125k_$rep SGL_NNLS_NO_SNAKE=1 --cells 125000 --steps 40 --warmup 5
done
run snake_100k X=1 --cells 100000 --steps 40 --warmup 5
run plain_100k SGL_NNLS_NO_SNAKE=1 --cells 100000 --steps 40 --warmup 5
run snake_70k_k30 X=1 --cells 70000 --k 30 --steps 40 --warmup 5
run plain_70k_k30 SGL_NNLS_NO_SNAKE=1 --cells 70000 --k 30 --steps 40 --warmup 5
