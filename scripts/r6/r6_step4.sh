#!/bin/bash
# round 6, GPU call 4: the Gram of ranks 129 - 256 on the matrix cores: tests, then the rank sweep above 128 again
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "test_gram or test_nnls_ranks_129" > gpurun_out/r6_s4_tests.log 2>&1; tail -3 gpurun_out/r6_s4_tests.log
python -m pytest tests/test_gpu_nmf.py tests/test_gpu_pool.py -m gpu -x -q > gpurun_out/r6_s4_tests_nmf.log 2>&1; tail -3 gpurun_out/r6_s4_tests_nmf.log
for k in 128 130 160 200 256 300; do
  python bench.py --k $k --cells 200000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0})"
done > gpurun_out/r6_s4_k_above_128.txt 2>&1
cat gpurun_out/r6_s4_k_above_128.txt
