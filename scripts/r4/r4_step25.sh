#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 600 python3 scripts/r4/r4_small.py 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | grep -E "^FAILED|passed|failed" | head -40
