#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun).  The program goes directly after `--`.
#  1. the default bench command under rocprofv3 --kernel-trace --stats           -> r3_kernel_stats.csv
#  2. separate PMC passes of the same workload: FETCH_SIZE, WRITE_SIZE, two SQ sets, and the matrix-core set
#     (FP64 MFMA op count + MFMA busy cycles: the Gram kernels)                   -> r3_pmc_*.csv
#  3. a masked (c_ard_nmf) iteration at 30 000 genes x 200 000 cells, k = 50: kernel stats, FETCH / WRITE and
#     the matrix-core set (mask_gram_mfma_kernel)                                  -> r3_ard_*.csv
# Summaries land in gpurun_out/; copy the ones to keep into profiles/.  `prof_r3.sh ard` runs part 3 only, `prof_r3.sh main` parts 1 - 2.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
run_stats() {  # name, command...
  local name=$1; shift
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/$name.d -- "$@" > $O/$name.json 2> $O/$name.err
  python3 scripts/pmc_summary.py $(find $O/$name.d -name "*.db" | head -1) > $O/$name.csv 2>&1
  rm -rf $O/$name.d
}
run_pmc() {  # name, "counters", command...
  local name=$1 set=$2; shift 2
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $O/$name.d -- "$@" > $O/$name.json 2> $O/$name.err
  python3 scripts/pmc_summary.py $(find $O/$name.d -name "*.db" | head -1) > $O/$name.csv 2>&1
  rm -rf $O/$name.d
}
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
if [ "$1" != "ard" ]; then
run_stats r3_kernel_stats python3 bench.py --no-cpu-baseline
run_pmc r3_pmc_fetch_size "FETCH_SIZE" $BENCH
run_pmc r3_pmc_write_size "WRITE_SIZE" $BENCH
run_pmc r3_pmc_sq_cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" $BENCH
run_pmc r3_pmc_sq_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" $BENCH
run_pmc r3_pmc_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $BENCH
fi
if [ "$1" != "ard" ]; then
# secondary records: the per-rank proxy of the 8-GPU run (125 000 cells), RCCL inside the library as a team of one,
# the skewed generator
python3 bench.py --cells 125000 --no-cpu-baseline --steps 20 > $O/r3_shard_125k.json 2> $O/r3_shard_125k.err
python3 bench.py --native-comm --no-cpu-baseline --steps 10 > $O/r3_bench_native_comm.json 2> $O/r3_bench_native_comm.err
python3 bench.py --data skewed --no-cpu-baseline --steps 10 > $O/r3_bench_skewed.json 2> $O/r3_bench_skewed.err
SGL_TILED_SORT=0 python3 bench.py --data skewed --no-cpu-baseline --steps 10 > $O/r3_bench_skewed_matrix_order.json 2> $O/r3_bench_skewed_matrix_order.err
python3 scripts/make_traffic_json.py $O/r3_pmc_fetch_size.csv $O/r3_pmc_write_size.csv $O/r3_pmc_fetch_size.json $O/traffic.json > /dev/null
fi
if [ "$1" == "main" ]; then head -12 $O/r3_kernel_stats.csv; grep acc_tiled $O/r3_pmc_fetch_size.csv $O/r3_pmc_write_size.csv; exit 0; fi
ARD="python3 scripts/ard_rate.py 200000 30000 50 2"
run_stats r3_ard_kernel_stats $ARD
run_pmc r3_ard_pmc_fetch_size "FETCH_SIZE" $ARD
run_pmc r3_ard_pmc_write_size "WRITE_SIZE" $ARD
run_pmc r3_ard_pmc_mfma "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $ARD
head -12 $O/r3_kernel_stats.csv
grep -i "gram" $O/r3_pmc_mfma.csv | head; grep -i "gram" $O/r3_ard_pmc_mfma.csv | head
grep acc_tiled $O/r3_pmc_fetch_size.csv $O/r3_pmc_write_size.csv
