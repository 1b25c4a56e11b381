#!/bin/bash
# bare DPP broadcasts in the LDS-triangle quad solve (masked path, k <= 64): full GPU suite, masked rates, the config-5 grid
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
showa() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k', d['k'], 'iters', d['iters'], round(d['ms_per_iter'],1), {k:round(v,2) for k,v in d['phases_ms_per_iter'].items() if v})"; }
for k in 60 50 20 10; do timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 10 2>/dev/null | showa new; done
timeout 1500 python3 scripts/config5_sweep.py 1000000 30000 10 5 > gpurun_out/r4_config5_full_size_d.json 2> gpurun_out/r4_config5_d.err
python3 -c "
import json
d=json.load(open('gpurun_out/r4_config5_full_size_d.json'))
print('grid', d['grid_wall_s'])
import collections
t=collections.defaultdict(float)
for f in d['fits']: t[f['k']]+=f['wall_s']
print({k:round(v,2) for k,v in t.items()})"
