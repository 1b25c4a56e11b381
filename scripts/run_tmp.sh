cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "ard or mask or config5" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for k in 10 30 50 64 80 100 128; do
  for v in pf1 pf2; do
    export SGL_LIB_PATH=/root/repo/build/lib_$v.so
    echo "k=$k $v $(python scripts/ard_rate.py 200000 30000 $k 2 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_iter'],1), {a:round(b,1) for a,b in j['phases_ms_per_iter'].items() if b>0.05}, j['test_mse'][-1])")"
  done
done
