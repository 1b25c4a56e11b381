#!/bin/bash
# PMC passes over a reduced-size bench run (200k cells); summaries land in gpurun_out/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CMD="python3 bench.py --cells 200000 --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_IFETCH" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_$i
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmc_$i -- $CMD > gpurun_out/pmc_$i.json 2> gpurun_out/pmc_$i.err
  db=$(find gpurun_out/pmc_$i -name "*.db" | head -1)
  python3 scripts/pmc_summary.py $db "$1" > gpurun_out/pmc_$i.csv 2>&1
  rm -rf gpurun_out/pmc_$i
done
cat gpurun_out/pmc_*.csv
