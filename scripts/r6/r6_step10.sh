#!/bin/bash
# round 6, GPU call 10: end-of-round records on the last tree: the whole GPU suite, smoke, the default bench, the shard proxy,
# config 5, the one-shot call, ranks above 128
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r6_gpu_tests_end_of_round.log 2>&1
tail -12 gpurun_out/r6_gpu_tests_end_of_round.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r6_bench_end_of_round.json 2> gpurun_out/r6_bench_end_of_round.err
python bench.py --cells 125000 --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/r6_shard_125k_end_of_round.json 2>/dev/null
python bench.py --workload ard --no-cpu-baseline > gpurun_out/r6_config5_full_size_end_of_round.json 2>/dev/null
python scripts/one_shot_rate.py > gpurun_out/r6_one_shot_config3_end_of_round.json 2>/dev/null
python - <<'PY'
import json
def last(f): return json.loads(open(f).read().strip().splitlines()[-1])
d=last("gpurun_out/r6_bench_end_of_round.json"); print("bench", round(d["value"],2), {p: round(v,2) for p,v in d["phases_ms_per_step"].items() if v>0}, round(d["roofline"]["frac"],4), d["cpu_baseline"]["value"], d["cpu_baseline"]["iterations_timed"])
d=last("gpurun_out/r6_shard_125k_end_of_round.json"); print("shard", round(d["value"],1), {p: round(v,3) for p,v in d["phases_ms_per_step"].items() if v>0})
d=last("gpurun_out/r6_config5_full_size_end_of_round.json"); print("config5", round(d["value"],1), [(q["k"], round(q["sec_per_masked_iter"],3)) for q in d["per_rank"]])
d=json.load(open("gpurun_out/r6_one_shot_config3_end_of_round.json")); print("one-shot", [(round(c["wall_s"],2), round(c["setup_s"],3)) for c in d["calls"]])
PY
