#!/bin/bash
# PMC passes focused on the NNLS lane kernel (scalar-cache / instruction-cache behaviour, issue stalls)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
CMD="python3 bench.py --cells 250000 --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_SMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" ; do
  i=$((i+1))
  rm -rf $O/nn_pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/nn_pmc_$i -- $CMD > $O/nn_pmc_$i.json 2> $O/nn_pmc_$i.err
  db=$(find $O/nn_pmc_$i -name "*.db" | head -1)
  python3 scripts/pmc_summary.py $db nnls > $O/nn_pmc_$i.csv 2>&1
  rm -rf $O/nn_pmc_$i
  cat $O/nn_pmc_$i.csv
done
