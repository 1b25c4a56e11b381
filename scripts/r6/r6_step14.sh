#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pool.py -m gpu -x -q > gpurun_out/r6_s14_tests.log 2>&1; tail -5 gpurun_out/r6_s14_tests.log
