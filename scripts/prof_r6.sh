#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun).  The program goes directly after `--`.
#  1. the default bench command under rocprofv3 --kernel-trace --stats             -> r6_kernel_stats.csv
#  2. separate PMC passes of the same workload: FETCH_SIZE, WRITE_SIZE, L2 hit / miss, two SQ sets, the matrix-core set
#  3. config 2 (quad layout) and the 125 000-cell shard under --kernel-trace --stats
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
python3 bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err
run_stats r6_kernel_stats python3 bench.py --no-cpu-baseline
run_pmc r6_pmc_fetch_size "FETCH_SIZE" $BENCH
run_pmc r6_pmc_write_size "WRITE_SIZE" $BENCH
run_pmc r6_pmc_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" $BENCH
run_pmc r6_pmc_sq_cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" $BENCH
run_pmc r6_pmc_sq_insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" $BENCH
python3 scripts/make_traffic_json.py $O/r6_pmc_fetch_size.csv $O/r6_pmc_write_size.csv $O/r6_pmc_fetch_size.json $O/traffic.json > /dev/null
C2="--genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline"
run_stats r6_config2_kernel_stats python3 bench.py $C2
run_stats r6_shard_125k_kernel_stats python3 bench.py --cells 125000 --no-cpu-baseline --steps 20
python3 bench.py --cells 125000 --no-cpu-baseline --steps 20 > $O/r6_shard_125k.json 2> /dev/null
head -14 $O/r6_kernel_stats.csv
grep acc_tiled $O/r6_pmc_fetch_size.csv $O/r6_pmc_write_size.csv $O/r6_pmc_l2.csv $O/r6_config2_pmc_sq_insts.csv
tail -1 $O/r6_bench_default.json | cut -c1-600

# one masked fit per rank under the profiler (config 5's kernels)
for k in 50 100; do
  rm -rf $O/r6_ard$k.d
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/r6_ard$k.d -- python3 scripts/ard_rate.py 200000 30000 $k 5 > $O/r6_ard${k}.json 2> $O/r6_ard${k}.err
  python3 scripts/pmc_summary.py $(find $O/r6_ard$k.d -name "*.db" | head -1) > $O/r6_ard${k}_kernel_stats.csv 2>&1
  rm -rf $O/r6_ard$k.d
  head -8 $O/r6_ard${k}_kernel_stats.csv | cut -c1-160
done
