# usage: ab_libs.sh "<bench args>" lib1 lib2 ...   (libs under build/, "cur" = the in-tree library); two rounds
args="$1"; shift
for rep in 1 2; do for v in "$@"; do
  if [ $v = cur ]; then unset SGL_LIB_PATH; else export SGL_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/build/lib_$v.so; fi
  python bench.py --no-cpu-baseline $args > gpurun_out/ab_$v.json 2>/dev/null
  python - <<PY
import json
j=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("%-8s"%"$v", round(j["value"],2), {k:round(x,2) for k,x in j["phases_ms_per_step"].items()}, j["tol_last"])
PY
done; done
