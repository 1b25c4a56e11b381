#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
timeout 1500 python3 scripts/config5_sweep.py 1000000 30000 10 5 > gpurun_out/r4_config5_full_size_e.json 2> gpurun_out/r4_config5_e.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4_config5_full_size_e.json'))
print('grid', d['grid_wall_s'])
import collections
t=collections.defaultdict(float)
for f in d['fits']: t[f['k']]+=f['wall_s']
print({k:round(v,2) for k,v in t.items()})"
