#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "rhs or padding or golden or c_nmf_parity" > $O/r4_s4_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s4_tests.log | tail -5
C2="--genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline"
timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_quad.json 2> $O/r4_bench_config2_quad.err; echo "quad rc=$?"
SGL_TILED_NO_QUAD=1 timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_pair.json 2> $O/r4_bench_config2_pair.err; echo "pair rc=$?"
SGL_TILED_NO_QUAD=1 SGL_TILED_OLD_SPLIT=1 timeout 600 python3 bench.py $C2 > $O/r4_bench_config2_pair_oldsplit.json 2>/dev/null
for k in 10 32; do
 timeout 600 python3 bench.py --genes 30000 --cells 200000 --k $k --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_k${k}_quad.json 2>/dev/null
done
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3.json 2>/dev/null
SGL_TILED_OLD_SPLIT=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_oldsplit.json 2>/dev/null
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 125000 > $O/r4_bench_125k.json 2>/dev/null
SGL_TILED_OLD_SPLIT=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 125000 > $O/r4_bench_125k_oldsplit.json 2>/dev/null
for f in r4_bench_config2_quad r4_bench_config2_pair r4_bench_config2_pair_oldsplit r4_k10_quad r4_k32_quad r4_bench_c3 r4_bench_c3_oldsplit r4_bench_125k r4_bench_125k_oldsplit; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    L=d["roofline"]["stream_layouts"]
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "R", L["rhs_h"]["tile_ranges"], L["rhs_w"]["tile_ranges"], "T", L["rhs_h"]["tiles"], L["rhs_w"]["tiles"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
