#!/bin/bash
# round 5: ranks 65 - 96 as three quad passes; four columns per wave on a shared Gram for short solves; in-flight counters of the accumulate
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rhs or four_columns" > $O/r5_s5_ops.log 2>&1; rc=$?; echo "rhs / nnls ops rc=$rc"; tail -2 $O/r5_s5_ops.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s5_ops.log | head -20; exit 1; fi
timeout 900 python3 -m pytest tests/test_gpu_nmf.py tests/test_gpu_native_team.py -x -q -m gpu > $O/r5_s5_nmf.log 2>&1; rc=$?; echo "nmf + team tests rc=$rc"; tail -2 $O/r5_s5_nmf.log
if [ $rc -ne 0 ]; then grep -E "Error|error|assert" $O/r5_s5_nmf.log | head -20; exit 1; fi
echo "--- k sweep at 200k cells: three quad passes (default) vs two pair passes"
for k in 64 66 70 80 90 96 100; do
  for v in quad3 pair2; do
    if [ $v = pair2 ]; then export SGL_TILED_NO_QUAD3=1; else unset SGL_TILED_NO_QUAD3; fi
    timeout 300 python3 bench.py --k $k --cells 200000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v k=$k', round(d['ms_per_step'],2), {a:round(b,2) for a,b in d['phases_ms_per_step'].items() if b}, round(d['roofline']['entries_per_nonzero']['rhs_h'],3))"
  done
done
unset SGL_TILED_NO_QUAD3
echo "--- small problems: four columns per wave on a shared Gram (default up to 8192 columns) vs the lane kernel"
for v in quad lane; do
  if [ $v = lane ]; then export SGL_NNLS_QUAD_SHARED_MAX_COLS=0; else unset SGL_NNLS_QUAD_SHARED_MAX_COLS; fi
  echo "[$v]"; timeout 600 python3 scripts/r4/r4_small.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr"
done
unset SGL_NNLS_QUAD_SHARED_MAX_COLS
echo "--- the W-side solve of a team rank: 3750 genes, k = 50 / 30 (op level, 20 calls each)"
python3 - <<'PY'
import os, time, numpy as np, sys
sys.path.insert(0, os.getcwd())
import singlet_amd as sa
c = sa.Context(0)
rng = np.random.default_rng(3)
for k in (50, 30, 16, 64):
    F = rng.random((4 * k, k)); G = F.T @ F / (4 * k) + 1e-15 * np.eye(k)
    for ncols in (3750, 8192, 30000):
        B = rng.normal(size=(ncols, k)) + 0.5; X0 = np.zeros((ncols, k))
        res = {}
        for name, env in (("lane", None), ("quad", "1")):
            if env: os.environ["SGL_OP_NNLS_QUAD_SHARED"] = env
            else: os.environ.pop("SGL_OP_NNLS_QUAD_SHARED", None)
            c.op_nnls(G, B, X0, 0.01, 0.0)
            c.timing_enable(True)
            # op_nnls includes uploads: time the kernel through the sweeps per second of repeated calls instead
            t0 = time.perf_counter()
            for _ in range(10): X, s = c.op_nnls(G, B, X0, 0.01, 0.0)
            res[name] = ((time.perf_counter() - t0) / 10, s)
        print("k=%d cols=%d  lane %.3f ms  quad %.3f ms (calls incl. upload/download; sweeps %d / %d)" % (k, ncols, 1e3 * res["lane"][0], 1e3 * res["quad"][0], res["lane"][1], res["quad"][1]))
PY
echo "--- in-flight counters of acc_tiled_kernel (config 3)"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  name=r5_pmc_$(echo $set | cut -d' ' -f1)
  rm -rf $O/$name.d
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $O/$name.d -- $BENCH > $O/$name.json 2> $O/$name.err
  db=$(find $O/$name.d -name "*.db" | head -1)
  if [ -n "$db" ]; then python3 scripts/pmc_summary.py $db > $O/$name.csv 2>&1; grep "acc_tiled_kernel" $O/$name.csv | cut -c1-120; else echo "$name: no database"; tail -3 $O/$name.err; fi
  rm -rf $O/$name.d
done
