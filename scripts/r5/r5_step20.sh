#!/bin/bash
# round 5, step 20: config 5 at full size on the final tree (bench.py --workload ard), and the default bench line
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 1500 python3 bench.py --workload ard > gpurun_out/r5_config5_full_size_v2.json 2> gpurun_out/r5_config5_full_size_v2.err; echo "ard rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_config5_full_size_v2.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['roofline']['frac'], d['cpu_baseline']['value'])
for r in d['per_rank']: print(r['k'], round(r['seconds'],2) if 'seconds' in r else r)
PY
timeout 600 python3 bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
tail -1 gpurun_out/r5_bench_final.json | cut -c1-400
