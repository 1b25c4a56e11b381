#!/bin/bash
# round 5: LDS wait granularity of the chunk loop (one s_waitcnt per 4 / 2 / 1 entry tuples) + LDS FIFO counters
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
export SGL_LIB_PATH=$PWD/build/lib_wait1.so
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs" > $O/r5_s6_ops.log 2>&1; rc=$?; echo "rhs ops on wait1 rc=$rc"; tail -2 $O/r5_s6_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s6_ops.log | head -20; exit 1; fi
for rep in 1 2; do
for v in wait4 wait2 wait1; do
  export SGL_LIB_PATH=$PWD/build/lib_$v.so
  timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if k.startswith('rhs') or k.startswith('nnls')})"
done; done
for v in wait4 wait1; do
  export SGL_LIB_PATH=$PWD/build/lib_$v.so
  timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 --genes 20000 --cells 50000 --k 30 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v config2', round(d['value'],1), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"
done
unset SGL_LIB_PATH
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for set in "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  name=r5_pmc_$(echo $set | cut -d' ' -f1)
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $O/$name.d -- $BENCH > $O/$name.json 2> $O/$name.err
  db=$(find $O/$name.d -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 scripts/pmc_summary.py $db > $O/$name.csv 2>&1; grep "acc_tiled_kernel" $O/$name.csv | cut -c1-120; else echo "$name: no database"; tail -3 $O/$name.err; fi
  rm -rf $O/$name.d
done
