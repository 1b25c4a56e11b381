#!/bin/bash
# round 5, step 18: ranks 97 - 128 as FOUR quad passes over the entry stream against two pair passes
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
run() {  # label, k, env...
  local label=$1 k=$2; shift 2
  env "$@" timeout 300 python3 bench.py --k $k --cells 200000 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_s18_$label.json 2> gpurun_out/r5_s18_$label.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/r5_s18_$label.json').read().strip().splitlines()[-1])
print('$label', round(d['ms_per_step'],2), {a: round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, d['roofline'].get('stream_layouts'))
PY
}
for k in 100 112 128; do
  run pair2_k$k $k X=1
  run quad4_k$k $k SGL_TILED_QUAD4=1
done
run k64 64 X=1
run k50 50 X=1
