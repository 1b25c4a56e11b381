#!/bin/bash
# where the global-memory quad solve overtakes the LDS-triangle one (masked path)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], 'iters', d['iters'], round(d['ms_per_iter'],1), {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if v.__class__ is float and k.startswith('nnls')})"; }
for k in 16 24 32 36 40 44 48; do
  for from in 49 1; do
    SGL_NNLS_QUAD_GLOBAL_FROM=$from timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 6 2>/dev/null | showa "from=$from"
  done
done
