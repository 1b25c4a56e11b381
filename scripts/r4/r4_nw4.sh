#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for v in 8 4; do
  if [ $v = 8 ]; then unset SGL_LIB_PATH; else export SGL_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_nw4.so; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves per workgroup $v', {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if k.startswith('rhs')}, d['roofline']['stream_layouts']['rhs_w']['tile_ranges'])"
done
