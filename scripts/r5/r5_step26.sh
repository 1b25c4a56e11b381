#!/bin/bash
# round 5, step 26: re-pack threshold of the two-lane solve (32 columns per wave: twice the waves of the lane solve per column)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
run() {  # label, env, args
  local label=$1 e=$2; shift 2
  env $e timeout 300 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r5_s26_$label.json 2> gpurun_out/r5_s26_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s26_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['ms_per_step'],3), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b}, round(d['nnls_mean_sweeps']['h_per_wave'],1))
PY
}
for k in 100 128 70; do
  run k${k}_default X=1 --k $k --cells 200000 --steps 10 --warmup 3
  run k${k}_repack128k SGL_NNLS_REPACK_MIN_COLS=131072 --k $k --cells 200000 --steps 10 --warmup 3
  run k${k}_repack64k SGL_NNLS_REPACK_MIN_COLS=65536 --k $k --cells 200000 --steps 10 --warmup 3
done
