#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests/test_gpu_ops.py -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | grep -E "^FAILED|passed|failed|Error" | head -20
timeout 300 python3 examples/pbmc3k_run_nmf.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
