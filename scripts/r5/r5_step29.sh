#!/bin/bash
# round 5, step 29: tile staging by load-to-LDS (global_load_lds_dwordx4) against load + ds_write, config 3 and the shard
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
export SGL_LIB_PATH=$PWD/build/lib_tiled_dma.so
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "rhs or c_nmf_parity" > gpurun_out/r5_s29_tests.log 2>&1; rc=$?
unset SGL_LIB_PATH
echo "tests (DMA build) rc=$rc"; tail -2 gpurun_out/r5_s29_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s29_tests.log; exit 1; fi
run() {  # label, env, args
  local label=$1 e=$2; shift 2
  env $e timeout 300 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r5_s29_$label.json 2> gpurun_out/r5_s29_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s29_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['value'],2), round(d['ms_per_step'],3), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b})
PY
}
for rep in 1 2; do
  run dma_c3_$rep SGL_LIB_PATH=$PWD/build/lib_tiled_dma.so --steps 20 --warmup 5
  run std_c3_$rep X=1 --steps 20 --warmup 5
done
run dma_shard SGL_LIB_PATH=$PWD/build/lib_tiled_dma.so --cells 125000 --steps 40 --warmup 5
run std_shard X=1 --cells 125000 --steps 40 --warmup 5
