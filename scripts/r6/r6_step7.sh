#!/bin/bash
# round 6, GPU call 7: the four-lanes-per-column solve against per-column Grams: parity / bit-identity tests, then A/B per rank
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_degenerate.py -m gpu -x -q > gpurun_out/r6_s7_tests.log 2>&1; tail -5 gpurun_out/r6_s7_tests.log
for k in 34 40 48 50 56 64 70 80 90 100 112 128; do
  for nq in 1 0; do
    if [ $nq = 1 ]; then export SGL_NNLS_NO_QUARTER_PERCOL=1; else unset SGL_NNLS_NO_QUARTER_PERCOL; fi
    python scripts/ard_rate.py 200000 30000 $k 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms_per_iter']; print('k=$k no_quarter_percol=$nq ms/iter', round(d['ms_per_iter'],1), 'nnls_h', round(p['nnls_h'],2), 'nnls_w', round(p['nnls_w'],2), 'mask', round(p['mask'],1), 'mse', '%.12g' % d['test_mse'][-1], d['sweeps_per_column_per_iter'])"
  done
done > gpurun_out/r6_s7_quarter_percol_ab.txt 2>&1
cat gpurun_out/r6_s7_quarter_percol_ab.txt
