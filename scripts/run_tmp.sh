cd $GRAFT_REPO_ROOT
export SGL_BENCH_FORCE_DEVICE=0
timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cells 60000 --no-cpu-baseline > gpurun_out/two_ranks_one_gpu.log 2>&1
echo "exit code $?"
grep -v "^$" gpurun_out/two_ranks_one_gpu.log | tail -25 | cut -c1-400
