#!/bin/bash
# round 5, step 19: full GPU suite with ranks 97 - 128 as four quad passes and the generated two-lane solve; k sweep record
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_s19_fullsuite.log 2>&1; rc=$?
echo "full GPU suite rc=$rc"; tail -3 gpurun_out/r5_s19_fullsuite.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s19_fullsuite.log; exit 1; fi
echo "# python3 bench.py --k K --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline; ms per iteration, phases, mean sweeps per column (H) / executed per 64 columns" > gpurun_out/r5_k_sweep_200k_cells_v2.txt
for k in 50 64 66 70 80 90 96 100 104 112 120 128; do
  timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_s19_k$k.json 2> gpurun_out/r5_s19_k$k.err
  python3 - >> gpurun_out/r5_k_sweep_200k_cells_v2.txt <<PY
import json
d=json.loads(open('gpurun_out/r5_s19_k$k.json').read().strip().splitlines()[-1])
print('k=$k', round(d['ms_per_step'],2), {a: round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, round(d['nnls_mean_sweeps']['h'],1), round(d['nnls_mean_sweeps']['h_per_wave'],1))
PY
done
cat gpurun_out/r5_k_sweep_200k_cells_v2.txt
