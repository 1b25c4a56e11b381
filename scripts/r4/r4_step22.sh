#!/bin/bash
# remainder rows as two / three quads in the list Gram downdate: op and fit parity, then the masked rates per rank
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py tests/test_gpu_config5.py tests/test_gpu_fullsize_oracle.py -q -x -k "mask or ard or config5" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], 'iters', d['iters'], round(d['ms_per_iter'],1), {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if v})"; }
for k in 40 60 70 90; do
  for v in tiles quads; do
    if [ $v = tiles ]; then export SGL_MASK_GRAM_NO_REM8=1; else unset SGL_MASK_GRAM_NO_REM8; fi
    timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 6 2>/dev/null | showa $v
  done
done
