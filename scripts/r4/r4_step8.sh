#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "rhs or padding or golden or c_nmf_parity" > $O/r4_s8_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s8_tests.log | tail -5
for rep in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_tab_$rep.json 2>/dev/null
SGL_TILED_NO_TABLE=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_notab_$rep.json 2>/dev/null
done
C2="--genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline"
timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_tab.json 2>/dev/null
SGL_TILED_NO_TABLE=1 timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_notab.json 2>/dev/null
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 125000 > $O/r4_bench_125k_tab.json 2>/dev/null
for f in r4_bench_c3_tab_1 r4_bench_c3_notab_1 r4_bench_c3_tab_2 r4_bench_c3_notab_2 r4_bench_config2_tab r4_bench_config2_notab r4_bench_125k_tab; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "frac", round(d["roofline"]["frac"],4))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
