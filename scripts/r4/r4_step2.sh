#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_ops.py -x -q -m gpu -k "fullsize or slices or mse_test_op" --durations=8 > $O/r4_s2_tests.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/r4_s2_tests.log | tail -40
