#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], 'iters', d['iters'], {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if k=='mask'})"; }
for k in 24 28 37 44 56 69 72 76 88; do
  for v in tiles quads; do
    if [ $v = tiles ]; then export SGL_MASK_GRAM_NO_REM8=1; else unset SGL_MASK_GRAM_NO_REM8; fi
    timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 4 2>/dev/null | showa $v
  done
done
