#!/bin/bash
# timing ablations of the schedule-table chunk loop (results are WRONG by construction; only the accumulate phases are read)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for v in full nofma noread noadd nofma_noadd; do
  if [ $v = full ]; then unset SGL_LIB_PATH; else export SGL_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_abl_$v.so; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/r4_abl_$v.json 2>/dev/null
  python3 - "$O/r4_abl_$v.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], {k:round(v,3) for k,v in d["phases_ms_per_step"].items() if k.startswith("rhs")})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
