#!/bin/bash
# round 6, GPU call 2: where the set-up seconds go, ranks above 128, the loopback-8 masked iteration, the bench with the new CPU window
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "test_rhs_tiled_kernel_all_ranks or test_rhs_both" > gpurun_out/r6_s2_tests.log 2>&1; tail -2 gpurun_out/r6_s2_tests.log
python -m pytest tests/test_gpu_nmf.py -m gpu -x -q > gpurun_out/r6_s2_tests_nmf.log 2>&1; tail -2 gpurun_out/r6_s2_tests_nmf.log
SGL_TRACE_SETUP=1 python scripts/one_shot_rate.py > gpurun_out/r6_s2_one_shot_trace.json 2> gpurun_out/r6_s2_one_shot_trace.err
grep -c "sgl setup" gpurun_out/r6_s2_one_shot_trace.err
for k in 128 130 160 200 256; do
  for mk in 128 1024; do
    SGL_TILED_MAX_K=$mk python bench.py --k $k --cells 200000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k tiled_max_k=$mk', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0})"
  done
done > gpurun_out/r6_s2_k_above_128.txt 2>&1
cat gpurun_out/r6_s2_k_above_128.txt
python scripts/r6/team_masked_rate.py > gpurun_out/r6_s2_team_masked.json 2> gpurun_out/r6_s2_team_masked.err; tail -c 1500 gpurun_out/r6_s2_team_masked.json; tail -3 gpurun_out/r6_s2_team_masked.err
python bench.py > gpurun_out/r6_s2_bench.json 2> gpurun_out/r6_s2_bench.err; tail -c 1200 gpurun_out/r6_s2_bench.json
