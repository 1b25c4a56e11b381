#!/bin/bash
# round 5, step 14: the full GPU suite on the last tree, then the round-5 profile set (scripts/prof_r5.sh)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5_s14_fullsuite.log 2>&1; rc=$?
echo "full GPU suite rc=$rc"; tail -3 gpurun_out/r5_s14_fullsuite.log
if [ $rc -ne 0 ]; then tail -40 gpurun_out/r5_s14_fullsuite.log; exit 1; fi
timeout 60 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/prof_r5.sh
