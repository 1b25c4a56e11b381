#!/bin/bash
# round 5: generated NNLS sweep at more ranks: bit identity, then nnls_h per rank against the hipcc-scheduled kernels (200 000 cells)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generated_sweep or nnls" > $O/r5_s10_ops.log 2>&1; rc=$?; echo "nnls op tests rc=$rc"; tail -2 $O/r5_s10_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s10_ops.log | head -20; exit 1; fi
for k in 10 20 30 40 44 48 50; do
  for v in asm hipcc; do
    if [ $v = hipcc ]; then export SGL_NNLS_NO_ASM=1; else unset SGL_NNLS_NO_ASM; fi
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v k=$k', round(d['ms_per_step'],3), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if a.startswith('nnls') or a.startswith('rhs')})"
  done
done
unset SGL_NNLS_NO_ASM
timeout 300 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 asm', round(d['value'],1), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if b})"
SGL_NNLS_NO_ASM=1 timeout 300 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 hipcc', round(d['value'],1), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if b})"
