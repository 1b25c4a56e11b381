#!/bin/bash
# Round-1 profile set (run on the GPU box through gpurun): the default bench command under
# rocprofv3 --kernel-trace --stats, then separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ).
# Summaries land in gpurun_out/r1_*.csv; copy the ones to keep into profiles/.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/r1_trace
timeout 900 rocprofv3 --kernel-trace --stats -d $O/r1_trace -- python3 bench.py > $O/r1_trace_bench.json 2> $O/r1_trace.err
db=$(find $O/r1_trace -name "*.db" | head -1)
python3 scripts/pmc_summary.py $db > $O/r1_kernel_stats.csv 2>&1
rm -rf $O/r1_trace
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" ; do
  i=$((i+1))
  rm -rf $O/r1_pmc_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $O/r1_pmc_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r1_pmc_$i.json 2> $O/r1_pmc_$i.err
  db=$(find $O/r1_pmc_$i -name "*.db" | head -1)
  python3 scripts/pmc_summary.py $db > $O/r1_pmc_$i.csv 2>&1
  rm -rf $O/r1_pmc_$i
done
tail -n 40 $O/r1_kernel_stats.csv
cat $O/r1_pmc_1.csv $O/r1_pmc_2.csv
