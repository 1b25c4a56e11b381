#!/bin/bash
# mse_test from listed matrix values: tests, then one masked fit per rank with traces
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_config5.py tests/test_gpu_nmf.py -q -x -k "mse or config5 or ard or mask" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6
for v in novals vals; do
  if [ $v = novals ]; then export SGL_MSE_NO_VALS=1; else unset SGL_MSE_NO_VALS; fi
  for k in 10 50 100; do
  python3 - $k $v <<'PY'
import sys, time
sys.path.insert(0, '.')
import singlet_amd as sa
k = int(sys.argv[1])
c = sa.Context(0); c.synth(30000, 200000, 20)
for rep in range(2):
    c.fit_init(k, None)
    t0 = time.perf_counter()
    r = c.ard_run(0.0, 4, 0.01, 0.0, 123, 20, 1e9, 1)   # a trace after every iteration
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    c.fit_init(k, None)
    r2 = c.ard_run(0.0, 4, 0.01, 0.0, 123, 20, 1e9, 4)  # one trace
    dt2 = time.perf_counter() - t1
print(sys.argv[2], 'k', k, 'ms per trace', round((dt - dt2) / 3 * 1e3, 2), 'test_mse', r['test_mse'][-1], r2['test_mse'][-1])
PY
  done
done
