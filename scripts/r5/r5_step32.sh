#!/bin/bash
# round 5, step 32: the process-per-GPU form of bench.py with TWO processes on a 1-GPU box (both ranks on device 0, the hook's
# all-reduce through gloo): exercises every world > 1 branch of the host side (id broadcast aside) -- not a scaling point
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 SGL_BENCH_FORCE_DEVICE=0 SGL_BENCH_HOOK_BACKEND=gloo
for n in 2 4; do
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --gpus $n --comm hook --cells 400000 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5_s32_two_procs_$n.json 2> gpurun_out/r5_s32_two_procs_$n.err; echo "torchrun $n rc=$?"
  tail -3 gpurun_out/r5_s32_two_procs_$n.err | cut -c1-300
  python3 - $n <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open('gpurun_out/r5_s32_two_procs_%s.json' % n).read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], d['comm']['mode'], d['comm']['host_coordination'], d['comm']['tol_bit_identical_across_ranks'], [ (r['rank'], r['cells'], round(r['sum_of_phases_ms'],2)) for r in d['per_rank']], d['rank_imbalance'])
PY
done
