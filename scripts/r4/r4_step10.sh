#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_config5.py -x -q -m gpu -k "rhs or padding or golden or c_nmf_parity or config3 or slices or config5" > $O/r4_s10_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s10_tests.log | tail -8
for rep in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_tail_$rep.json 2>/dev/null
SGL_TILED_NO_TAIL=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_notail_$rep.json 2>/dev/null
done
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 200000 > $O/r4_bench_200k_tail.json 2>/dev/null
SGL_TILED_NO_TAIL=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 200000 > $O/r4_bench_200k_notail.json 2>/dev/null
for f in r4_bench_c3_tail_1 r4_bench_c3_notail_1 r4_bench_c3_tail_2 r4_bench_c3_notail_2 r4_bench_200k_tail r4_bench_200k_notail; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    L=d["roofline"]["stream_layouts"]
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "R", L["rhs_h"]["tile_ranges"], L["rhs_w"]["tile_ranges"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
