#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 600 python3 -m pytest tests/test_gpu_nmf.py -q -x -k "test_c_ard_nmf_parity and (85 or 87 or 88 or 69 or 53)" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | grep -E "Error|assert|passed|failed|rel|E  " | head -30
echo "--- tiles"
SGL_MASK_GRAM_NO_REM8=1 timeout 600 python3 -m pytest tests/test_gpu_nmf.py -q -x -k "test_c_ard_nmf_parity and (85 or 87 or 88 or 69 or 53)" 2>&1 | grep -E "Error|assert|passed|failed|E  " | head -30
