#!/bin/bash
# round 6, GPU call 13: the four-lane solve packed by sweep counts: bit-identity, then the rank sweep with / without packing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nmf.py -m gpu -x -q -k "packing or 130 or 200 or test_c_nmf_parity" > gpurun_out/r6_s13_tests.log 2>&1; tail -3 gpurun_out/r6_s13_tests.log
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "test_nnls" > gpurun_out/r6_s13_tests2.log 2>&1; tail -2 gpurun_out/r6_s13_tests2.log
for k in 130 160 200 256; do
  for np in 1 0; do
    if [ $np = 1 ]; then export SGL_NNLS_NO_PACK=1; else unset SGL_NNLS_NO_PACK; fi
    python bench.py --k $k --cells 200000 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k no_pack=$np', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0}, round(d['nnls_mean_sweeps']['h'],2), round(d['nnls_mean_sweeps']['h_per_wave']/4,2))"
  done
done > gpurun_out/r6_s13_k_above_128.txt 2>&1
cat gpurun_out/r6_s13_k_above_128.txt
