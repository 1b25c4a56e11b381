#!/bin/bash
# power / clock samples (hwmon sysfs, 20 ms) while the benchmark iterates: is the iteration power-limited?
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
ls /sys/class/drm/ > gpurun_out/power_sysfs.txt 2>&1
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $h"; ls $h; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input temp1_input temp2_input temp3_input; do [ -r $h/$f ] && echo "$f $(cat $h/$f)"; done; done >> gpurun_out/power_sysfs.txt 2>&1
cat > /tmp/sample.py <<'PY'
import glob, sys, time, json
hs = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')
def rd(p):
    try: return int(open(p).read().strip())
    except Exception: return None
out = []
t0 = time.time()
stop = sys.argv[1]
import os
while not os.path.exists(stop):
    row = {'t': round(time.time() - t0, 3)}
    for i, h in enumerate(hs):
        for f in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input'):
            v = rd(h + '/' + f)
            if v is not None: row['%d_%s' % (i, f)] = v
    out.append(row); time.sleep(0.02)
json.dump(out, open(sys.argv[2], 'w'))
PY
for v in tab notab; do
  rm -f /tmp/stop_$v
  python3 /tmp/sample.py /tmp/stop_$v gpurun_out/power_$v.json &
  sp=$!
  if [ $v = notab ]; then export SGL_TILED_NO_TABLE=1; fi
  timeout 600 python3 bench.py --no-cpu-baseline --steps 150 --warmup 2 2>/dev/null | tail -1 > gpurun_out/power_bench_$v.json
  touch /tmp/stop_$v; wait $sp
done
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 >> gpurun_out/power_sysfs.txt
python3 - <<'PY'
import json
for v in ('tab','notab'):
    d=json.load(open('gpurun_out/power_%s.json'%v)); b=json.loads(open('gpurun_out/power_bench_%s.json'%v).read())
    keys=[k for k in d[0] if k!='t']
    print(v, 'it/s', round(b['value'],2), {k:round(x,3) for k,x in b['phases_ms_per_step'].items() if x})
    n=len(d)
    for k in keys:
        xs=[r[k] for r in d if k in r]
        tail=xs[-(len(xs)//3):]   # last third = inside the timed loop
        print('  ',k,'all-mean',sum(xs)/len(xs),'last-third mean',sum(tail)/len(tail),'max',max(xs))
PY
