cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in a0 a1 a2 a3 a4; do
  export SGL_LIB_PATH=/root/repo/build/lib_$v.so
  timeout 120 python bench.py --no-cpu-baseline --steps 6 --warmup 1 --cells 200000 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {a:round(b,3) for a,b in j['phases_ms_per_step'].items() if a.startswith('nnls')}, j['nnls_mean_sweeps']['h_per_wave'])"
done; done
