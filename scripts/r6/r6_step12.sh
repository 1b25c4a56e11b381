#!/bin/bash
# round 6, GPU call 12: the four-lane solve with its row buffers by parity (no copies): bit-identity tests, rank sweep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "test_nnls" > gpurun_out/r6_s12_tests.log 2>&1; tail -3 gpurun_out/r6_s12_tests.log
for k in 130 160 200 256; do
  python bench.py --k $k --cells 200000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0})"
done > gpurun_out/r6_s12_k_above_128.txt 2>&1
cat gpurun_out/r6_s12_k_above_128.txt
