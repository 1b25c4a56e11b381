#!/bin/bash
# time the accumulate phases of bench.py for every library variant in build/abl/ (kernel experiments)
cd $GRAFT_REPO_ROOT
for f in build/abl/lib_*.so; do
  v=$(basename $f .so)
  SGL_LIB_PATH=$PWD/$f timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']
print('$v', 'rhs_h %.2f rhs_w %.2f nnls_h %.2f total %.2f tol %.6g' % (p['rhs_h'], p['rhs_w'], p['nnls_h'], d['ms_per_step'], d['tol_last']))"
done
