#!/bin/bash
# round 5: the degenerate masked solves (strict non-finite semantics in the per-column solves) + what the strict form costs
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_degenerate.py -x -q -m gpu > $O/r5_s2_degenerate.log 2>&1; echo "degenerate rc=$?"; tail -15 $O/r5_s2_degenerate.log
for k in 20 50 100; do
  for lib in r4 new; do
    if [ $lib = r4 ]; then export SGL_LIB_PATH=$PWD/build/lib_r4.so; else unset SGL_LIB_PATH; fi
    timeout 600 python3 scripts/ard_rate.py 200000 30000 $k 6 2>/dev/null | tail -1 > $O/r5_s2_ard_${lib}_k$k.json
    python3 -c "
import json; d=json.load(open('$O/r5_s2_ard_${lib}_k$k.json')); print('$lib', $k, round(d['ms_per_iter'],1), {p:round(v,1) for p,v in d['phases_ms_per_iter'].items() if v})"
  done
done
unset SGL_LIB_PATH
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/r5_s2_fullsuite.log 2>&1; echo "full suite rc=$?"; tail -5 $O/r5_s2_fullsuite.log
