#!/bin/bash
# round 5, step 21: one-launch row sums / cor with LDS-staged partials: bits, then the shard and default bench
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py tests/test_gpu_native_team.py -x -q -m gpu -k "reductions or scale_and_cor or c_nmf or team" > gpurun_out/r5_s21_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r5_s21_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s21_tests.log; exit 1; fi
run() {  # label, args..., env via X=
  local label=$1; shift
  timeout 300 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r5_s21_$label.json 2> gpurun_out/r5_s21_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s21_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['value'],2), round(d['ms_per_step'],3), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b})
PY
}
run shard_new --cells 125000 --steps 40 --warmup 5
SGL_ROWSUM_TWO_KERNELS=1 SGL_COR_TWO_KERNELS=1 run shard_old --cells 125000 --steps 40 --warmup 5
run shard_new2 --cells 125000 --steps 40 --warmup 5
run config3 --steps 20 --warmup 5
run config2 --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5
