cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "ard or mask or config5 or team" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for k in 10 30 50 64 80 100 128; do
    echo "k=$k $(python scripts/ard_rate.py 200000 30000 $k 2 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_iter'],1), {a:round(b,1) for a,b in j['phases_ms_per_iter'].items() if b>0.05}, j['test_mse'][-1])")"
done
