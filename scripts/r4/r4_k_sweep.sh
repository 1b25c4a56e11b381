#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
bash scripts/k_sweep.sh 10 20 30 32 34 40 50 60 64 70 80 90 100 104 112 120 128 > gpurun_out/r4_k_sweep_200k_cells.txt 2>&1
cat gpurun_out/r4_k_sweep_200k_cells.txt
