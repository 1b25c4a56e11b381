#!/bin/bash
# round 5: the lane NNLS sweep as generated asm (k = 49 / 50): bit identity, then A/B against the hipcc-scheduled kernel
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "generated_sweep or nnls" > $O/r5_s9_ops.log 2>&1; rc=$?; echo "nnls op tests rc=$rc"; tail -2 $O/r5_s9_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s9_ops.log | head -20; exit 1; fi
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -x -q -m gpu -k "c_nmf_parity or packing or config2 or golden" > $O/r5_s9_nmf.log 2>&1; rc=$?; echo "nmf tests rc=$rc"; tail -2 $O/r5_s9_nmf.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s9_nmf.log | head -20; exit 1; fi
run() { local label=$1; shift
  timeout 600 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],2), 'it/s', {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v}, {k:round(v,1) for k,v in d['nnls_mean_sweeps'].items()})"
}
for rep in 1 2; do
  unset SGL_NNLS_NO_ASM
  run "c3 asm" --steps 10 --warmup 3
  SGL_NNLS_NO_ASM=1 run "c3 hipcc" --steps 10 --warmup 3
  run "125k asm" --steps 40 --warmup 5 --cells 125000
  SGL_NNLS_NO_ASM=1 run "125k hipcc" --steps 40 --warmup 5 --cells 125000
done
run "k=49 200k asm" --steps 10 --warmup 3 --cells 200000 --k 49
SGL_NNLS_NO_ASM=1 run "k=49 200k hipcc" --steps 10 --warmup 3 --cells 200000 --k 49
