#!/bin/bash
# kernel statistics of one masked (c_ard_nmf) fit at 30 000 genes x 200 000 cells, ranks 10 / 20 / 50 / 100 (round-4 kernels)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for k in 10 20 50 100; do
  name=r4_ard${k}_kernel_stats
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/$name.d -- python3 scripts/ard_rate.py 200000 30000 $k 5 > $O/$name.json 2> $O/$name.err
  python3 scripts/pmc_summary.py $(find $O/$name.d -name "*.db" | head -1) > $O/$name.csv 2>&1
  rm -rf $O/$name.d
  echo "== k=$k"; head -12 $O/$name.csv | cut -c1-150
done
