#!/bin/bash
# round 5: config 5 measured like config 3 (bench.py --workload ard at full size), the test of the mode, and the CPU baseline on the full matrix
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_config5.py -x -q -m gpu -k "bench_ard" > $O/r5_s3_test.log 2>&1; echo "ard mode test rc=$?"; tail -3 $O/r5_s3_test.log
timeout 1500 python3 bench.py --workload ard > $O/r5_config5_full_size.json 2> $O/r5_config5_full_size.err; echo "config5 rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_config5_full_size.json').read().strip().splitlines()[-1])
print("grid", round(d["value"],1), "s; roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k in ("achieved","frac","ms_per_iteration")})
for q in d["per_rank"]: print(q["k"], round(q["sec_per_masked_iter"],3), [round(w,2) for w in q["fit_wall_s"]], {p:round(v,1) for p,v in q["phases_ms_per_iter_last_replicate"].items()}, "TF", round(q["downdate"]["achieved_tflops"] or 0,1))
cb=d.get("cpu_baseline",{}); print("cpu", cb.get("value"), cb.get("cores"), [ (r["k"], round(r["est_full_sec_per_masked_iter"],1)) for r in cb.get("per_rank",[])])
PY
timeout 1200 python3 bench.py --cpu-sample-cells 0 --steps 20 --warmup 3 > $O/r5_bench_cpu_full_size.json 2> $O/r5_bench_cpu_full_size.err; echo "cpu full rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_cpu_full_size.json').read().strip().splitlines()[-1])
print(round(d["value"],2), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sample"][:200])
PY
