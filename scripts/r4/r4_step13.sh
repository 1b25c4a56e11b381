#!/bin/bash
# the one-expression NNLS step: GPU suite (bit-identity and oracle parity), then config 3, the shard, a masked fit
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5
for a in "" "--cells 125000"; do
timeout 600 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"
done
timeout 600 python3 scripts/ard_rate.py 200000 30000 100 5
timeout 600 python3 scripts/ard_rate.py 200000 30000 50 5
timeout 600 python3 scripts/ard_rate.py 200000 30000 20 5
