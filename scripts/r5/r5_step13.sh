#!/bin/bash
# round 5: generated NNLS sweep: register plans that buy a third / fourth wave per SIMD at ranks up to 30; interleave density A/B at k = 50
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generated_sweep or nnls" > $O/r5_s13_ops.log 2>&1; rc=$?; echo "nnls op tests rc=$rc"; tail -2 $O/r5_s13_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s13_ops.log | head -20; exit 1; fi
for k in 10 20 24 28 30 32; do
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('asm k=$k', round(d['ms_per_step'],3), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if a.startswith('nnls')})"
done
timeout 300 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2', round(d['value'],1), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if b})"
for rep in 1 2; do for v in default nnls_stride1 nnls_stride3; do
  if [ $v = default ]; then unset SGL_LIB_PATH; else export SGL_LIB_PATH=$PWD/build/lib_$v.so; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v c3', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if k.startswith('nnls')})"
done; done
unset SGL_LIB_PATH
