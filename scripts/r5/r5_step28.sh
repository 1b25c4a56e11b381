#!/bin/bash
# round 5, step 28: cor's final stage out of LDS, batched loads in the short partial sum: bits (tests) and kernel times at the shard
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_nmf.py -x -q -m gpu -k "scale_and_cor or c_nmf_parity or gram" > gpurun_out/r5_s28_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 gpurun_out/r5_s28_tests.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s28_tests.log; exit 1; fi
rm -rf gpurun_out/s28.d
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/s28.d -- python3 bench.py --cells 125000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r5_s28.json 2> gpurun_out/r5_s28.err
python3 scripts/pmc_summary.py $(find gpurun_out/s28.d -name "*.db" | head -1) > gpurun_out/r5_s28_stats.csv 2>&1
rm -rf gpurun_out/s28.d
grep -i "rowsum\|cor_\|partial_sum\|scale_kernel\|pad_gram\|add_diag\|copyBuffer" gpurun_out/r5_s28_stats.csv | cut -c1-120
timeout 300 python3 bench.py --cells 125000 --steps 40 --warmup 5 --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {a: round(b,3) for a,b in d['phases_ms_per_step'].items() if b})"
