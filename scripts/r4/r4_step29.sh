#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_nmf.py -q -x -k "packing or nnls or c_nmf_parity or config2" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -2
rm -rf gpurun_out/scan.d
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/scan.d -- python3 bench.py --no-cpu-baseline > gpurun_out/scan.json 2>/dev/null
python3 scripts/pmc_summary.py $(find gpurun_out/scan.d -name "*.db" | head -1) | grep -E "pack|kernel,"
rm -rf gpurun_out/scan.d
tail -1 gpurun_out/scan.json | cut -c1-200
