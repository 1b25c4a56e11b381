cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "ard or mask or config5" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
for k in 56 64 80 100 112 128; do
    unset SGL_NNLS_QUAD_GLOBAL_128 SGL_NNLS_NO_QUAD_GLOBAL
    export SGL_NNLS_QUAD_GLOBAL_128=1
    echo "k=$k quad_global $(python scripts/ard_rate.py 200000 30000 $k 2 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_iter'],1), {a:round(b,1) for a,b in j['phases_ms_per_iter'].items() if b>0.05}, j['test_mse'][-1])")"
done
