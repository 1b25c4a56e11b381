#!/bin/bash
# kept mask lists: the config-5 tests, then the full-size grid
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1200 python3 -m pytest tests/test_gpu_config5.py tests/test_gpu_sharded.py tests/test_gpu_native_team.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15
timeout 1500 python3 scripts/config5_sweep.py 1000000 30000 10 5 > gpurun_out/r4_config5_full_size_b.json 2> gpurun_out/r4_config5_b.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4_config5_full_size_b.json'))
print('grid', d['grid_wall_s'])
import collections
t=collections.defaultdict(float)
for f in d['fits']: t[f['k']]+=f['wall_s']
print({k:round(v,2) for k,v in t.items()})"
