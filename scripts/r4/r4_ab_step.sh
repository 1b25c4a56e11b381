#!/bin/bash
# A/B on one box: the select-per-quantity NNLS step (build/lib_oldstep.so) against the one-expression step
cd "${GRAFT_REPO_ROOT:?}" || exit 1
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"; }
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], round(d['ms_per_iter'],1), {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if v})"; }
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export SGL_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_oldstep.so; else unset SGL_LIB_PATH; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | show "$v config3"
  timeout 600 python3 bench.py --no-cpu-baseline --steps 20 --warmup 2 --cells 125000 2>/dev/null | show "$v 125k"
  if [ $rep = 1 ]; then
  for k in 100 70 50 20; do timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 8 2>/dev/null | showa $v; done
  fi
done
done
