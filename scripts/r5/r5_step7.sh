#!/bin/bash
# round 5: lane refill of the H-side solve: bit identity + A/B at config 3 and at the 125 000-cell shard
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -x -q -m gpu -k "refill or packing or c_nmf_parity" > $O/r5_s7_tests.log 2>&1; rc=$?; echo "refill tests rc=$rc"; tail -2 $O/r5_s7_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s7_tests.log | head -20; exit 1; fi
run() { # label, bench args...
  local label=$1; shift
  timeout 600 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],2), 'it/s', {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v}, {k:round(v,1) for k,v in d['nnls_mean_sweeps'].items()})"
}
for rep in 1 2; do
  unset SGL_NNLS_NO_REFILL SGL_NNLS_REFILL_MIN
  run "c3 refill(8)" --steps 10 --warmup 3
  SGL_NNLS_REFILL_MIN=16 run "c3 refill(16)" --steps 10 --warmup 3
  SGL_NNLS_REFILL_MIN=4 run "c3 refill(4)" --steps 10 --warmup 3
  SGL_NNLS_NO_REFILL=1 run "c3 passes" --steps 10 --warmup 3
  run "125k refill(8)" --steps 40 --warmup 5 --cells 125000
  SGL_NNLS_REFILL_MIN=16 run "125k refill(16)" --steps 40 --warmup 5 --cells 125000
  SGL_NNLS_REFILL_MIN=4 run "125k refill(4)" --steps 40 --warmup 5 --cells 125000
  SGL_NNLS_NO_REFILL=1 run "125k one pass" --steps 40 --warmup 5 --cells 125000
done
run "k=30 200k refill" --steps 20 --warmup 3 --cells 200000 --k 30
SGL_NNLS_NO_REFILL=1 run "k=30 200k passes" --steps 20 --warmup 3 --cells 200000 --k 30
run "k=64 200k refill" --steps 20 --warmup 3 --cells 200000 --k 64
SGL_NNLS_NO_REFILL=1 run "k=64 200k passes" --steps 20 --warmup 3 --cells 200000 --k 64
