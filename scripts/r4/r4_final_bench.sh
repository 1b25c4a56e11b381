#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 bench.py > $O/r4_bench_end_of_round.json 2> $O/r4_bench_end_of_round.err; echo "rc=$?"
timeout 600 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline > $O/r4_bench_config2_end_of_round.json 2>/dev/null
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r4_bench_20steps.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("r4_bench_end_of_round","r4_bench_config2_end_of_round","r4_bench_20steps"):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d["value"],2), round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, round(d["roofline"]["frac"],4), d["roofline"]["traffic"], (d.get("cpu_baseline") or {}).get("value"))
PY
