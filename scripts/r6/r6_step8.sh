#!/bin/bash
# round 6, GPU call 8: the whole GPU suite and the default bench on the (nearly) final tree
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r6_s8_tests.log 2>&1
tail -12 gpurun_out/r6_s8_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r6_s8_bench.json 2> gpurun_out/r6_s8_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r6_s8_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['phases_ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['iterations_timed'])"
