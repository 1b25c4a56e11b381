#!/bin/bash
# round 5, step 31: the launch forms of bench.py on the last tree: under torch.distributed.run (N = 1), one-process team of one, loopback 8
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'], 2), d['n_gpus'], d.get('comm', {}).get('mode'), d.get('comm', {}).get('rccl_nranks'), 'per_rank' in d, d.get('rank_imbalance', {}).get('max_over_min_compute'))
except Exception as e:
    print(sys.argv[1], 'NO JSON', e)
PY
}
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_s31_torchrun1.json 2> gpurun_out/r5_s31_torchrun1.err; echo "torchrun rc=$?"; show gpurun_out/r5_s31_torchrun1.json
timeout 600 python3 bench.py --gpus 1 --single-process --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_s31_single_process.json 2> gpurun_out/r5_s31_single_process.err; echo "single-process rc=$?"; show gpurun_out/r5_s31_single_process.json
timeout 600 python3 bench.py --gpus 8 --loopback --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_s31_loopback8.json 2> gpurun_out/r5_s31_loopback8.err; echo "loopback rc=$?"; show gpurun_out/r5_s31_loopback8.json
timeout 120 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5_s31_gpus2.log 2>&1; echo "gpus 2 on one device rc=$? (1 expected)"; tail -2 gpurun_out/r5_s31_gpus2.log
