#!/bin/bash
# round 6, GPU call 11: triangles in the sharded masked exchange (team tests, config-4 team tests), the k = 160 full-size oracle slices
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native_team.py tests/test_gpu_sharded.py tests/test_gpu_config4_team.py -m gpu -x -q --durations=5 > gpurun_out/r6_s11_tests.log 2>&1; tail -9 gpurun_out/r6_s11_tests.log
python -m pytest tests/test_gpu_fullsize_oracle.py -m gpu -x -q --durations=5 > gpurun_out/r6_s11_tests2.log 2>&1; tail -9 gpurun_out/r6_s11_tests2.log
python scripts/r6/team_masked_rate.py 1000000 50 8 > gpurun_out/r6_s11_team_masked.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r6_s11_team_masked.json')); print(d['one_context']['sec_per_masked_iter'], d['loopback_team_8']['sec_per_masked_iter'], d['team_over_one_context'])"
