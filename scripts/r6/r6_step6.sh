#!/bin/bash
# round 6, GPU call 6: the four-lanes-per-column solve of ranks 129 - 256: bit-identity tests, then the rank sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "test_nnls" > gpurun_out/r6_s6_tests.log 2>&1; tail -4 gpurun_out/r6_s6_tests.log
python -m pytest tests/test_gpu_nmf.py -m gpu -x -q > gpurun_out/r6_s6_tests_nmf.log 2>&1; tail -3 gpurun_out/r6_s6_tests_nmf.log
for k in 130 160 200 256; do
  for nq in 0 1; do
    if [ $nq = 1 ]; then export SGL_NNLS_NO_QUARTER=1; else unset SGL_NNLS_NO_QUARTER; fi
    python bench.py --k $k --cells 200000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k no_quarter=$nq', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0}, d['nnls_mean_sweeps'])"
  done
done > gpurun_out/r6_s6_k_above_128.txt 2>&1
cat gpurun_out/r6_s6_k_above_128.txt
