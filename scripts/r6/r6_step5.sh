#!/bin/bash
# round 6, GPU call 5: the round's profile set (scripts/prof_r6.sh), the rate micro-benchmarks behind the two priced levers,
# config 5 with its CPU baseline, the loopback-8 bench, config 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash scripts/prof_r6.sh > gpurun_out/r6_s5_prof.log 2>&1
tail -25 gpurun_out/r6_s5_prof.log | cut -c1-200
( scripts/ubench/valu_rate; scripts/ubench/lds_rate ) > gpurun_out/r6_ubench_rates.txt 2>&1
grep -E "cvt|v_fmac_f64 |b64|b128" gpurun_out/r6_ubench_rates.txt | head -20
python bench.py --workload ard > gpurun_out/r6_config5_full_size.json 2> gpurun_out/r6_config5_full_size.err
python -c "
import json; d=json.loads(open('gpurun_out/r6_config5_full_size.json').read().strip().splitlines()[-1]); print('grid', d['value'], 'cpu', d.get('cpu_baseline',{}).get('value'), d['roofline']['frac'])"
python bench.py --gpus 8 --loopback --no-cpu-baseline > gpurun_out/r6_bench_loopback_8.json 2>/dev/null; tail -c 300 gpurun_out/r6_bench_loopback_8.json
python bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r6_bench_config2.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_config2.json').read().strip().splitlines()[-1]); print('config2', d['value'])"
