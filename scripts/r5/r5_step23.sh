#!/bin/bash
# round 5, step 23: full GPU suite + smoke + the default bench command on the last tree
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_s23_fullsuite.log 2>&1; rc=$?
echo "full GPU suite rc=$rc"; tail -3 gpurun_out/r5_s23_fullsuite.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s23_fullsuite.log; exit 1; fi
timeout 60 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python3 bench.py > gpurun_out/r5_bench_end_of_round.json 2> gpurun_out/r5_bench_end_of_round.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_end_of_round.json').read().strip().splitlines()[-1])
print(d['value'], d['steps'], d['warmup'], {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b}, d['roofline']['frac'], d['cpu_baseline']['value'])
PY
timeout 300 python3 bench.py --cells 125000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r5_shard_125k_end_of_round.json 2>/dev/null
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_shard_125k_end_of_round.json').read().strip().splitlines()[-1])
print('shard', d['value'], {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b})
PY
