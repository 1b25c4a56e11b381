#!/bin/bash
# ms per ALS iteration and per phase for several ranks on a 200k-cell shard (fast paths stop at k = 64)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for k in "$@"; do
  timeout 300 python bench.py --k $k --cells 200000 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('k=$k', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['phases_ms_per_step'].items()}, {a:round(b,1) for a,b in d['nnls_mean_sweeps'].items()})"
done
