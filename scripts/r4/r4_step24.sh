#!/bin/bash
# records: the config-5 grid at full size, kernel statistics of masked fits at ranks 10 / 50 / 70 / 100, default bench + shard
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 scripts/config5_sweep.py 1000000 30000 10 5 > $O/r4_config5_full_size.json 2> $O/r4_config5.err
python3 -c "
import json
d=json.load(open('$O/r4_config5_full_size.json'))
print('grid', d['grid_wall_s'], 'plain timing', d['plain_timing_wall_s'])
import collections
t=collections.defaultdict(float)
for f in d['fits']: t[f['k']]+=f['wall_s']
print({k:round(v,2) for k,v in t.items()})"
for k in 10 50 70 100; do
  name=r4_ard${k}_kernel_stats
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/$name.d -- python3 scripts/ard_rate.py 200000 30000 $k 5 > $O/$name.json 2> $O/$name.err
  python3 scripts/pmc_summary.py $(find $O/$name.d -name "*.db" | head -1) > $O/$name.csv 2>&1
  rm -rf $O/$name.d
  echo "== k=$k"; head -8 $O/$name.csv | cut -c1-150
done
python3 bench.py > $O/r4_bench_step.json 2> /dev/null; tail -1 $O/r4_bench_step.json | cut -c1-400
python3 bench.py --cells 125000 --no-cpu-baseline --steps 20 > $O/r4_shard_125k_step.json 2> /dev/null; tail -1 $O/r4_shard_125k_step.json | cut -c1-300
