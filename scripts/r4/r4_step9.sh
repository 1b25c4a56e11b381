#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1200 python3 scripts/config5_sweep.py 1000000 30000 10 5 > $O/r4_config5_full_size.json 2> $O/r4_config5_full_size.err
python3 -c "
import json
d=json.loads(open('$O/r4_config5_full_size.json').read().strip().splitlines()[-1])
print('grid', round(d['grid_wall_s'],1))
by={}
for f in d['fits']: by.setdefault(f['k'],[]).append(f['wall_s'])
print({k:[round(x,2) for x in v] for k,v in by.items()})
print({k:round(v*1e3,1) for k,v in d['sec_per_plain_iter'].items()})
"
