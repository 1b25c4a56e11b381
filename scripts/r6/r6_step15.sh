#!/bin/bash
# round 6, last GPU call: the whole GPU suite on the last tree of the round
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r6_gpu_tests_end_of_round.log 2>&1
tail -12 gpurun_out/r6_gpu_tests_end_of_round.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
