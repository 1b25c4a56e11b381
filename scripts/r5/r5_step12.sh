#!/bin/bash
# round 5: generated NNLS sweep at ranks 51 - 64 (x in the accumulator registers, one wave per SIMD): bit identity + A/B
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generated_sweep or nnls" > $O/r5_s12_ops.log 2>&1; rc=$?; echo "nnls op tests rc=$rc"; tail -2 $O/r5_s12_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s12_ops.log | head -20; exit 1; fi
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -x -q -m gpu -k "c_nmf_parity" > $O/r5_s12_nmf.log 2>&1; rc=$?; echo "c_nmf parity rc=$rc"; tail -2 $O/r5_s12_nmf.log
for k in 52 56 60 64; do
  for v in asm hipcc; do
    if [ $v = hipcc ]; then export SGL_NNLS_NO_ASM=1; else unset SGL_NNLS_NO_ASM; fi
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v k=$k', round(d['ms_per_step'],3), {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if a.startswith('nnls') or a.startswith('rhs')})"
  done
done
