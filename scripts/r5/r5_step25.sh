#!/bin/bash
# round 5, step 25: empty columns through every solve kernel; kernel stats + SQ counters of a k = 100 plain fit (generated two-lane solve)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -x -q -m gpu -k "empty_columns" > $O/r5_s25_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/r5_s25_tests.log
if [ $rc -ne 0 ]; then tail -40 $O/r5_s25_tests.log; exit 1; fi
BENCH="python3 bench.py --k 100 --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline"
rm -rf $O/k100.d
timeout 600 rocprofv3 --kernel-trace --stats -d $O/k100.d -- $BENCH > $O/r5_k100_under_rocprof.json 2> $O/r5_k100.err
python3 scripts/pmc_summary.py $(find $O/k100.d -name "*.db" | head -1) > $O/r5_k100_kernel_stats.csv 2>&1
rm -rf $O/k100.d
head -12 $O/r5_k100_kernel_stats.csv | cut -c1-150
rm -rf $O/k100p.d
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/k100p.d -- $BENCH > /dev/null 2> $O/r5_k100p.err
python3 scripts/pmc_summary.py $(find $O/k100p.d -name "*.db" | head -1) > $O/r5_k100_pmc_sq_cycles.csv 2>&1
rm -rf $O/k100p.d
grep "nnls_half_asm" $O/r5_k100_pmc_sq_cycles.csv | cut -c1-150
rm -rf $O/k100q.d
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --kernel-trace -d $O/k100q.d -- $BENCH > /dev/null 2> $O/r5_k100q.err
python3 scripts/pmc_summary.py $(find $O/k100q.d -name "*.db" | head -1) > $O/r5_k100_pmc_sq_insts.csv 2>&1
rm -rf $O/k100q.d
grep "nnls_half_asm" $O/r5_k100_pmc_sq_insts.csv | cut -c1-150
