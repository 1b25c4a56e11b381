#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 900 python3 scripts/r4/r4_sparsity.py 2>/dev/null | tail -1
