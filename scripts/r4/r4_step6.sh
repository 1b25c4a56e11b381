#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for mc in 0 131072 65536 32768 16384; do
  if [ $mc = 0 ]; then unset SGL_NNLS_REPACK_MIN_COLS; else export SGL_NNLS_REPACK_MIN_COLS=$mc; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 30 --cells 125000 > $O/r4_125k_mc$mc.json 2>/dev/null
  python3 - "$O/r4_125k_mc$mc.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "it/s", round(d["value"],2), "ms", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if v}, d["nnls_mean_sweeps"])
PY
done
