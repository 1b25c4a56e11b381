#!/bin/bash
# per-rank proxy of the strong-scaling curve: one GPU running the shard a rank of an N-GPU team would hold
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
for n in 1 2 4 8; do
  cells=$((1000000 / n))
  timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --cells $cells > $O/r4_proxy_n$n.json 2>/dev/null
done
python3 - <<'PY'
import json
rows=[]
for n in (1,2,4,8):
    d=json.loads(open('gpurun_out/r4_proxy_n%d.json'%n).read().strip().splitlines()[-1])
    rows.append({"n_gpus_emulated": n, "cells_per_rank": d["config"]["cells"], "ms_per_iteration": d["ms_per_step"], "phases_ms": d["phases_ms_per_step"]})
base=rows[0]["ms_per_iteration"]
for r in rows: r["compute_only_speedup_vs_1gpu"]=base/r["ms_per_iteration"]
json.dump({"what":"bench.py --cells 1000000/N on ONE GPU: the compute a rank of an N-GPU team does per iteration (its W solve still covers all genes here; on a team it covers 1/N of them); collectives not included","rows":rows}, open('gpurun_out/r4_shard_proxy.json','w'), indent=1)
for r in rows: print(r["n_gpus_emulated"], r["cells_per_rank"], round(r["ms_per_iteration"],3), round(r["compute_only_speedup_vs_1gpu"],2), {k:round(v,3) for k,v in r["phases_ms"].items() if v})
PY
