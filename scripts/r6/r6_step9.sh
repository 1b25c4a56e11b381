#!/bin/bash
# round 6, GPU call 9: the VALU remainder rows of the list downdate kernel: op-level and fit-level tests, then A/B per rank
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mask_gram" > gpurun_out/r6_s9_tests.log 2>&1; tail -3 gpurun_out/r6_s9_tests.log
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_degenerate.py tests/test_gpu_config5.py -m gpu -x -q > gpurun_out/r6_s9_tests2.log 2>&1; tail -3 gpurun_out/r6_s9_tests2.log
for k in 20 34 40 44 50 56 60 66 70 72; do
  for nv in 1 0; do
    if [ $nv = 1 ]; then export SGL_MASK_GRAM_NO_REMV=1; else unset SGL_MASK_GRAM_NO_REMV; fi
    python scripts/ard_rate.py 200000 30000 $k 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms_per_iter']; print('k=$k no_remv=$nv ms/iter', round(d['ms_per_iter'],1), 'mask', round(p['mask'],2), 'nnls_h', round(p['nnls_h'],2), 'mse', '%.12g' % d['test_mse'][-1])"
  done
done > gpurun_out/r6_s9_remv_ab.txt 2>&1
cat gpurun_out/r6_s9_remv_ab.txt
