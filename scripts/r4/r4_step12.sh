#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_nmf.py tests/test_gpu_ops.py tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "nnls or packing or config3 or c_nmf_parity or golden" > $O/r4_s12_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s12_tests.log | tail -8
for rep in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_pack_$rep.json 2>/dev/null
SGL_NNLS_NO_PACK=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 10 > $O/r4_bench_c3_nopack_$rep.json 2>/dev/null
done
timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 125000 > $O/r4_bench_125k_pack.json 2>/dev/null
SGL_NNLS_NO_PACK=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --cells 125000 > $O/r4_bench_125k_nopack.json 2>/dev/null
for f in r4_bench_c3_pack_1 r4_bench_c3_nopack_1 r4_bench_c3_pack_2 r4_bench_c3_nopack_2 r4_bench_125k_pack r4_bench_125k_nopack; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, {a:round(b,2) for a,b in d["nnls_mean_sweeps"].items()})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
