set -e
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "nnls" 2>&1 | tail -3
for rep in 1 2; do
for v in base pipe; do
  if [ $v = base ]; then export SGL_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/build/lib_base.so; else unset SGL_LIB_PATH; fi
  python bench.py --no-cpu-baseline --steps 8 > gpurun_out/ab_$v.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$v", round(j["value"],2), {k:round(x,2) for k,x in j["phases_ms_per_step"].items()}, j["tol_last"], j["nnls_mean_sweeps"])
PY
done; done
for k in 30 40 64; do for v in base pipe; do
  if [ $v = base ]; then export SGL_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/build/lib_base.so; else unset SGL_LIB_PATH; fi
  python bench.py --no-cpu-baseline --steps 5 --cells 200000 --k $k > gpurun_out/ab_$v.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("k=$k $v", round(j["value"],2), {k:round(x,2) for k,x in j["phases_ms_per_step"].items()}, j["tol_last"])
PY
done; done
