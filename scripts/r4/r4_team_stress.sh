#!/bin/bash
# the threaded one-process team, many times over: 16 ranks / 8 ranks / 3 ranks on device 0, plain and masked
cd "${GRAFT_REPO_ROOT:?}" || exit 1
fail=0
for i in $(seq 1 12); do
  for n in 16 8 3; do
    timeout 120 python3 bench.py --gpus $n --loopback --genes 2000 --cells 30000 --k 10 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || { echo "bench failed at round $i n=$n rc=$?"; fail=1; }
  done
done
timeout 600 python3 - <<'PY' || fail=1
import numpy as np, singlet_amd as sa
from oracle import oracle as ora
A = ora.synth_csc(300, 2000, 10)
dA = sa.dgCMatrix(A.x, A.i, A.p, (A.nrow, A.ncol))
ref = None
for rep in range(25):
    with sa.Multi([0] * 5) as M:
        M.upload(dA); M.fit_init(8, ora.synth_winit(8, 300))
        r = M.ard_run(0.0, 3, 0.01, 0.0, 7, 10, 1e9, 1)
        W, d, H = M.get_factors()
    if ref is None: ref = (W, H, r["test_mse"])
    assert np.array_equal(W, ref[0]) and np.array_equal(H, ref[1]) and np.array_equal(r["test_mse"], ref[2]), rep
print("25 masked team fits bit-identical")
PY
echo "stress done fail=$fail"
