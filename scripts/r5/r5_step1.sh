#!/bin/bash
# round 5, first GPU call: team tests (new watchdog test), this box's baseline (config 3, 125k shard), loopback 8 with per-rank block
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_native_team.py tests/test_gpu_sharded.py -x -q -m gpu > $O/r5_s1_team_tests.log 2>&1; echo "team tests rc=$?" 
tail -3 $O/r5_s1_team_tests.log
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/r5_s1_bench_c3.json 2> $O/r5_s1_bench_c3.err; echo "c3 rc=$?"
timeout 600 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --cells 125000 > $O/r5_s1_bench_125k.json 2> $O/r5_s1_bench_125k.err; echo "125k rc=$?"
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --gpus 8 --loopback > $O/r5_s1_bench_lb8.json 2> $O/r5_s1_bench_lb8.err; echo "lb8 rc=$?"
python3 - <<'PY'
import json
for n in ("c3","125k","lb8"):
    try:
        d=json.loads(open('gpurun_out/r5_s1_bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, round(d["value"],2), "it/s", round(d["ms_per_step"],3), "ms", {k:round(v,3) for k,v in d["phases_ms_per_step"].items()}, "frac", round(d["roofline"]["frac"],4))
        if "per_rank" in d: print("  per_rank sums:", [round(r["sum_of_phases_ms"],2) for r in d["per_rank"]], d["rank_imbalance"])
    except Exception as e: print(n, "FAILED", e)
PY
