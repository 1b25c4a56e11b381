#!/bin/bash
# restructured quad-global solve (ranks 65 - 128 of the masked path): parity tests, then masked rates against build/lib_oldstep.so
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests -q -m gpu -x -k "ard or mask or config5 or quad or nnls or fullsize" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], 'iters', d['iters'], round(d['ms_per_iter'],1), {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if v})"; }
for v in old new; do
  if [ $v = old ]; then export SGL_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_oldstep.so; else unset SGL_LIB_PATH; fi
  for k in 100 80 70 128; do for it in 3 10; do timeout 600 python3 scripts/ard_rate.py 200000 30000 $k $it 2>/dev/null | showa $v; done; done
done
