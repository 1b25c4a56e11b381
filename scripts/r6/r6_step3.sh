#!/bin/bash
# round 6, GPU call 3: the device-memory pool and the four-columns-per-wave solve at ranks 129 - 256 under the whole GPU suite;
# the one-shot call again; ranks above 128; the config-5 grid
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r6_s3_tests.log 2>&1
tail -14 gpurun_out/r6_s3_tests.log
SGL_TRACE_SETUP=1 python scripts/one_shot_rate.py > gpurun_out/r6_s3_one_shot_config3.json 2> gpurun_out/r6_s3_one_shot_trace.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_s3_one_shot_config3.json"))
for c in d["calls"]:
    s=c["library_split_s"]; print("%-60s wall %.3f setup %.3f h2d %.3f fit %.3f loop %.3f"%(c["call"][:60],c["wall_s"],c["setup_s"],s["h2d_s"],s["fit_init_s"],s["iterate_s"]))
PY
python scripts/one_shot_rate.py --genes 20000 --cells 50000 --k 30 > gpurun_out/r6_s3_one_shot_config2.json 2>/dev/null
for k in 130 160 200 256; do
  python bench.py --k $k --cells 200000 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k', round(d['ms_per_step'],2), {p: round(v,2) for p,v in d['phases_ms_per_step'].items() if v>0})"
done > gpurun_out/r6_s3_k_above_128.txt 2>&1
cat gpurun_out/r6_s3_k_above_128.txt
python bench.py --workload ard --no-cpu-baseline > gpurun_out/r6_s3_config5.json 2> gpurun_out/r6_s3_config5.err
python -c "
import json; d=json.loads(open('gpurun_out/r6_s3_config5.json').read().strip().splitlines()[-1]); print('grid', d['value']); print([ (q['k'], [round(x,2) for x in q['fit_wall_s']]) for q in d['per_rank']])"
