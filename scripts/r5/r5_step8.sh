#!/bin/bash
# round 5: k sweep (three quad passes), small problems and short solves (four columns per wave), in-flight counters, LDS wait granularity
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r5_s8_fullsuite.log 2>&1; rc=$?; echo "full GPU suite rc=$rc"; tail -3 $O/r5_s8_fullsuite.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s8_fullsuite.log | head -20; exit 1; fi
echo "--- k sweep at 200k cells: three quad passes (default) vs two pair passes"
for k in 64 66 70 80 90 96 100; do
  for v in quad3 pair2; do
    if [ $v = pair2 ]; then export SGL_TILED_NO_QUAD3=1; else unset SGL_TILED_NO_QUAD3; fi
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v k=$k', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, round(d['roofline']['entries_per_nonzero']['rhs_h'],3))"
  done
done
unset SGL_TILED_NO_QUAD3
echo "--- small problems: four columns per wave on a shared Gram (default up to 8192 columns) vs the lane kernel"
for v in quad lane; do
  if [ $v = lane ]; then export SGL_NNLS_QUAD_SHARED_MAX_COLS=0; else unset SGL_NNLS_QUAD_SHARED_MAX_COLS; fi
  echo "[$v]"; timeout 600 python3 scripts/r4/r4_small.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
done
unset SGL_NNLS_QUAD_SHARED_MAX_COLS
echo "--- the W-side solve of a team rank's gene block: 3750 genes x 125 000 cells (nnls_w phase), k = 50 / 30 / 64 / 16"
for k in 50 30 64 16; do
  for v in quad lane; do
    if [ $v = lane ]; then export SGL_NNLS_QUAD_SHARED_MAX_COLS=0; else unset SGL_NNLS_QUAD_SHARED_MAX_COLS; fi
    timeout 300 python3 bench.py --genes 3750 --cells 125000 --k $k --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v 3750 genes k=$k', round(d['ms_per_step'],3), 'ms/iter', {a:round(b,3) for a,b in d['phases_ms_per_step'].items() if b}, {a:round(b,1) for a,b in d['nnls_mean_sweeps'].items()})"
  done
done
unset SGL_NNLS_QUAD_SHARED_MAX_COLS
echo "--- in-flight counters of acc_tiled_kernel (config 3)"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  name=r5_pmc_$(echo $set | cut -d' ' -f1)
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $O/$name.d -- $BENCH > $O/$name.json 2> $O/$name.err
  db=$(find $O/$name.d -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 scripts/pmc_summary.py $db > $O/$name.csv 2>&1; grep "acc_tiled_kernel" $O/$name.csv | cut -c1-120; else echo "$name: no database"; tail -3 $O/$name.err; fi
  rm -rf $O/$name.d
done

export SGL_LIB_PATH=$PWD/build/lib_wait1.so
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs" > $O/r5_s6_ops.log 2>&1; rc=$?; echo "rhs ops on wait1 rc=$rc"; tail -2 $O/r5_s6_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s6_ops.log | head -20; fi
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
