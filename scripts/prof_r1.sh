cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof1 gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -- python3 bench.py --cells 200000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/bench.json 2> gpurun_out/prof1/err.txt
find gpurun_out/prof1 -name "*stats*" | head
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc1 -- python3 bench.py --cells 200000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1/bench.json 2> gpurun_out/pmc1/err.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --kernel-trace -d gpurun_out/pmc2 -- python3 bench.py --cells 200000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2/bench.json 2> gpurun_out/pmc2/err.txt
ls -R gpurun_out/pmc1 | head -20
