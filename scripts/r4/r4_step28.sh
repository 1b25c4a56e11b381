#!/bin/bash
# end-of-round secondary records: the skewed matrix, ranks 10 ... 128 at 200 000 cells, config 2
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out
timeout 600 python3 bench.py --data skewed --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/r4_bench_skewed.json
python3 -c "
import json
d=json.loads(open('$O/r4_bench_skewed.json').read()); print('skewed', round(d['value'],2), {k:round(v,2) for k,v in d['phases_ms_per_step'].items() if v})"
timeout 600 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r4_bench_config2_final.json
python3 -c "
import json
d=json.loads(open('$O/r4_bench_config2_final.json').read()); print('config2', round(d['value'],2), {k:round(v,3) for k,v in d['phases_ms_per_step'].items() if v})"
bash scripts/k_sweep.sh 10 20 30 32 40 50 64 70 80 100 128 > $O/r4_k_sweep_200k_cells.txt 2>&1
cat $O/r4_k_sweep_200k_cells.txt
