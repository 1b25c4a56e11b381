#!/bin/bash
# round-3 tree (build/r3tree, a worktree of a43f1b3 built in place) against the current one on the SAME box, alternating
cd "${GRAFT_REPO_ROOT:?}" || exit 1
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"; }
for rep in 1 2; do
  (cd build/r3tree && timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | show r3)
  timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | show r4
  SGL_TILED_NO_TABLE=1 timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | show r4-notable
done
