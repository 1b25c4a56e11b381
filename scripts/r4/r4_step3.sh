#!/bin/bash
# quad layout (k <= 32): parity, then config 2 A/B against the pair layout
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs or padding" > $O/r4_s3_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s3_tests.log | tail -25
C2="--genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline"
timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_quad.json 2> $O/r4_bench_config2_quad.err; echo "quad rc=$?"
SGL_TILED_NO_QUAD=1 timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_pair.json 2> $O/r4_bench_config2_pair.err; echo "pair rc=$?"
for k in 10 20 32; do
 timeout 600 python3 bench.py --genes 30000 --cells 200000 --k $k --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_k${k}_quad.json 2>/dev/null
 SGL_TILED_NO_QUAD=1 timeout 600 python3 bench.py --genes 30000 --cells 200000 --k $k --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_k${k}_pair.json 2>/dev/null
done
for f in r4_bench_config2_quad r4_bench_config2_pair r4_k10_quad r4_k10_pair r4_k20_quad r4_k20_pair r4_k32_quad r4_k32_pair; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "epn", {a:round(b,3) for a,b in d["roofline"]["entries_per_nonzero"].items()})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 $O/r4_bench_config2_quad.err
