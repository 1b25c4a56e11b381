#!/bin/bash
# round 5: L2 prefetch of the entry stream in the chunk loop (gen_acc_tiled.py SGL_GEN_PF): correctness on the default build, A/B 0 / 2 / 4 laps
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs or four_columns" > $O/r5_s4_rhs_tests.log 2>&1; rc=$?; echo "rhs + quad-shared nnls tests (pf2 = default build) rc=$rc"; tail -2 $O/r5_s4_rhs_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s4_rhs_tests.log | head -20; exit 1; fi
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -x -q -m gpu -k "c_nmf_parity or project" > $O/r5_s4_nmf_tests.log 2>&1; rc=$?; echo "c_nmf parity tests rc=$rc"; tail -2 $O/r5_s4_nmf_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s4_nmf_tests.log | head -20; exit 1; fi
for rep in 1 2; do
for v in pf0 pf2 pf4; do
  export SGL_LIB_PATH=$PWD/build/lib_$v.so
  timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r5_s4_${v}_$rep.json 2>/dev/null
  python3 - "$O/r5_s4_${v}_$rep.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"],2), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if k.startswith("rhs") or k.startswith("nnls")})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done; done
for v in pf0 pf2; do
  export SGL_LIB_PATH=$PWD/build/lib_$v.so
  timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 --cells 125000 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 125k', round(d['value'],1), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"
  timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 --genes 20000 --cells 50000 --k 30 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v config2', round(d['value'],1), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"
done
