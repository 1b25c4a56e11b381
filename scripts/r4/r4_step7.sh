#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
SGL_TILED_DEEP=1 timeout 1500 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs" > $O/r4_s7_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s7_tests.log | tail -4
for rep in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_m3_$rep.json 2>/dev/null
SGL_TILED_DEEP=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_m5_$rep.json 2>/dev/null
done
for f in r4_bench_c3_m3_1 r4_bench_c3_m5_1 r4_bench_c3_m3_2 r4_bench_c3_m5_2; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "frac", round(d["roofline"]["frac"],4))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
