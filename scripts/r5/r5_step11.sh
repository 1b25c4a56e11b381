#!/bin/bash
# round 5: the full GPU suite on the tree with the generated NNLS sweep at every even rank up to 50, then the round's records
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/r5_s11_fullsuite.log 2>&1; rc=$?; echo "full GPU suite rc=$rc"; tail -3 $O/r5_s11_fullsuite.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s11_fullsuite.log | head -20; exit 1; fi
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/r5_bench.json 2> $O/r5_bench.err; echo "bench rc=$?"
timeout 600 python3 bench.py --cells 125000 --steps 40 --warmup 5 --no-cpu-baseline > $O/r5_shard_125k.json 2>/dev/null
timeout 600 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline > $O/r5_bench_config2.json 2>/dev/null
timeout 600 python3 bench.py --gpus 8 --loopback --steps 10 --warmup 2 --no-cpu-baseline > $O/r5_bench_loopback_8.json 2>/dev/null
timeout 600 python3 scripts/r4/r4_small.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" > $O/r5_small_problems_final.txt
for n in 1 2 4 8; do
  timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 --cells $((1000000 / n)) > $O/r5_proxy_n$n.json 2>/dev/null
done
python3 - <<'PY'
import json
def last(p): return json.loads(open(p).read().strip().splitlines()[-1])
for n in ("r5_bench","r5_shard_125k","r5_bench_config2","r5_bench_loopback_8"):
    try:
        d=last('gpurun_out/%s.json'%n); print(n, round(d["value"],2), "it/s", round(d["ms_per_step"],3), "ms", {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, "frac", round(d["roofline"]["frac"],4), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(n, "FAILED", e)
rows=[]
for n in (1,2,4,8):
    d=last('gpurun_out/r5_proxy_n%d.json'%n)
    rows.append({"n_gpus_emulated": n, "cells_per_rank": d["config"]["cells"], "ms_per_iteration": d["ms_per_step"], "phases_ms": d["phases_ms_per_step"]})
base=rows[0]["ms_per_iteration"]
for r in rows: r["compute_only_speedup_vs_1gpu"]=base/r["ms_per_iteration"]
json.dump({"what":"bench.py --cells 1000000/N on ONE GPU: the compute a rank of an N-GPU team does per iteration (its W solve still covers all genes here; on a team it covers 1/N of them, see r5_team_gene_block_solve.txt); collectives not included","rows":rows}, open('gpurun_out/r5_shard_proxy.json','w'), indent=1)
for r in rows: print(r["n_gpus_emulated"], r["cells_per_rank"], round(r["ms_per_iteration"],3), round(r["compute_only_speedup_vs_1gpu"],2), {k:round(v,3) for k,v in r["phases_ms"].items() if v})
PY
cat $O/r5_small_problems_final.txt
echo "--- crossover of the four-columns-per-wave shared-Gram solve (nnls_w, k = 50 / 30), genes = columns of the W side"
for k in 50 30; do for g in 6000 8192 12000 16384 24000; do
  for v in quad lane; do
    if [ $v = lane ]; then export SGL_NNLS_QUAD_SHARED_MAX_COLS=0; else export SGL_NNLS_QUAD_SHARED_MAX_COLS=100000; fi
    timeout 300 python3 bench.py --genes $g --cells 60000 --k $k --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v genes=$g k=$k nnls_w', round(d['phases_ms_per_step']['nnls_w'],3), 'nnls_h', round(d['phases_ms_per_step']['nnls_h'],3))"
  done; done; done
