cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
python bench.py --no-cpu-baseline --steps 10 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],2), {a:round(b,3) for a,b in j['phases_ms_per_step'].items()})"
python bench.py --no-cpu-baseline --steps 30 --cells 125000 | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],2), {a:round(b,3) for a,b in j['phases_ms_per_step'].items()})"
