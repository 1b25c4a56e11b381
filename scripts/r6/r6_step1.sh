#!/bin/bash
# round 6, GPU call 1: the whole GPU suite incl. the new config-4 team tests, the one-shot call at configs 2 and 3, the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r6_s1_tests.log 2>&1
tail -5 gpurun_out/r6_s1_tests.log
python scripts/one_shot_rate.py --genes 20000 --cells 50000 --k 30 --staged 8 > gpurun_out/r6_one_shot_config2.json 2> gpurun_out/r6_one_shot_config2.err
python scripts/one_shot_rate.py --staged 8 > gpurun_out/r6_one_shot_config3.json 2> gpurun_out/r6_one_shot_config3.err
tail -c 600 gpurun_out/r6_one_shot_config3.json; tail -3 gpurun_out/r6_one_shot_config3.err
python bench.py > gpurun_out/r6_s1_bench.json 2> gpurun_out/r6_s1_bench.err
tail -c 400 gpurun_out/r6_s1_bench.json
