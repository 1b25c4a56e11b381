#!/bin/bash
# Round 4, first GPU visit: threaded one-process team (loopback tests), reference-pinned rng tests, bench rehearsals
# of the launcher-free N > 1 form, the config-2 record.
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_native_team.py tests/test_gpu_sharded.py tests/test_gpu_ops.py -x -q -m gpu -k "team or rccl or singlet_ngpu or reference or sharded" > $O/r4_s1_tests.log 2>&1
tail -5 $O/r4_s1_tests.log
timeout 600 python3 bench.py --gpus 1 --single-process --no-cpu-baseline --steps 10 > $O/r4_bench_single_process_1.json 2> $O/r4_bench_single_process_1.err; echo "sp1 rc=$?"
timeout 900 python3 bench.py --gpus 8 --loopback --steps 5 --warmup 1 > $O/r4_bench_loopback_8.json 2> $O/r4_bench_loopback_8.err; echo "lb8 rc=$?"
timeout 900 python3 bench.py --gpus 2 --loopback --steps 5 --warmup 1 > $O/r4_bench_loopback_2.json 2> $O/r4_bench_loopback_2.err; echo "lb2 rc=$?"
SGL_MULTI_SERIAL=1 timeout 900 python3 bench.py --gpus 8 --loopback --steps 5 --warmup 1 > $O/r4_bench_loopback_8_serial.json 2> $O/r4_bench_loopback_8_serial.err; echo "lb8s rc=$?"
timeout 300 python3 bench.py --gpus 2 --steps 2 > $O/r4_bench_gpus2_on_one_device.out 2> $O/r4_bench_gpus2_on_one_device.err; echo "gpus2 (must fail) rc=$?"
timeout 600 python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline > $O/r4_bench_config2.json 2> $O/r4_bench_config2.err; echo "cfg2 rc=$?"
rm -rf $O/c2.d
timeout 600 rocprofv3 --kernel-trace --stats -d $O/c2.d -- python3 bench.py --genes 20000 --cells 50000 --k 30 --steps 50 --warmup 5 --no-cpu-baseline > $O/r4_config2_under_rocprof.json 2> $O/r4_config2_rocprof.err
python3 scripts/pmc_summary.py $(find $O/c2.d -name "*.db" | head -1) > $O/r4_config2_kernel_stats.csv 2>&1
rm -rf $O/c2.d
head -14 $O/r4_config2_kernel_stats.csv
for f in r4_bench_single_process_1 r4_bench_loopback_8 r4_bench_loopback_2 r4_bench_loopback_8_serial r4_bench_config2; do
  python3 - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", round(d["value"],2), "ms", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), {k:round(v,3) for k,v in d["phases_ms_per_step"].items()}, d["comm"]["mode"], d.get("loopback"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 $O/r4_bench_gpus2_on_one_device.err
