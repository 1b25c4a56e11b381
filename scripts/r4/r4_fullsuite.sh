#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 3000 python3 -m pytest tests/ -x -q -m gpu --durations=15 > $O/r4_gpu_suite.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_gpu_suite.log | tail -30
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
