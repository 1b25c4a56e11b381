#!/bin/bash
# rehearsals of the launcher forms on the one-GPU box
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 600 python3 bench.py --native-comm --no-cpu-baseline --steps 10 > $O/r4_bench_native_comm.json 2> $O/r4_bench_native_comm.err; echo "native-comm rc=$?"
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --steps 5 > $O/r4_bench_torchrun1.json 2> $O/r4_bench_torchrun1.err; echo "torchrun1 rc=$?"
SGL_BENCH_FORCE_DEVICE=0 timeout 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --cells 200000 > $O/r4_two_ranks_one_gpu.log 2>&1; echo "two ranks on one gpu (must fail cleanly) rc=$?"
timeout 900 python3 bench.py --gpus 8 --loopback --steps 5 --warmup 1 > $O/r4_bench_loopback_8.json 2> $O/r4_bench_loopback_8.err; echo "lb8 rc=$?"
timeout 600 python3 bench.py --gpus 1 --single-process --no-cpu-baseline --steps 10 > $O/r4_bench_single_process_1.json 2> $O/r4_bench_single_process_1.err; echo "sp1 rc=$?"
for f in r4_bench_native_comm r4_bench_torchrun1 r4_bench_loopback_8 r4_bench_single_process_1; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "n_gpus", d["n_gpus"], d["comm"]["mode"], d["comm"]["rccl_nranks"], d.get("loopback"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -4 $O/r4_two_ranks_one_gpu.log
