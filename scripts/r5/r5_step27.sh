#!/bin/bash
# round 5, step 27: default re-pack threshold of the two-lane solve; NNLS tests; k = 100 / 128 check
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_nmf.py tests/test_gpu_ops.py -x -q -m gpu -k "nnls or c_nmf" > gpurun_out/r5_s27_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 gpurun_out/r5_s27_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s27_tests.log; exit 1; fi
for k in 100 128; do
  timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_s27_k$k.json 2> gpurun_out/r5_s27_k$k.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s27_k$k.json').read().strip().splitlines()[-1])
print('k=$k', round(d['ms_per_step'],3), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b}, round(d['nnls_mean_sweeps']['h_per_wave'],1))
PY
done
