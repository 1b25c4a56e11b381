#!/bin/bash
# round 5, step 30: lock-step waste of the masked quad solves: executed sweeps per wave (4 columns) against needed per column
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out
for k in 50 100; do
python3 - <<PY
import sys, json, time
import singlet_amd as sa
cells, genes, k, iters = 200000, 30000, $k, 10
ctx = sa.Context(0)
ctx.synth(genes, cells, 20)
ctx.fit_init(k, None)
ctx.ard_run(0.0, 1, 0.01, 0.0, 123, 20, 1e9, 1)
ctx.fit_init(k, None)
ctx.timing_enable(True); ctx.timing_get(reset=True); ctx.sweeps_get(reset=True)
r = ctx.ard_run(0.0, iters, 0.01, 0.0, 123, 20, 1e9, iters)
ph = ctx.timing_get(reset=True); sw = ctx.sweeps_get(reset=True)
print(k, {p: round(v[0] / iters, 2) for p, v in ph.items() if v[0]}, {a: b for a, b in sw.items()})
h_need = sw["h_sweeps"] / iters / cells; h_exec = sw["h_wave_sweeps"] / iters / (cells / 4)
w_need = sw["w_sweeps"] / iters / genes; w_exec = sw["w_wave_sweeps"] / iters / (genes / 4)
print("k=%d  H: needed %.1f executed per quad %.1f (x%.2f)   W: needed %.1f executed %.1f (x%.2f)" % (k, h_need, h_exec, h_exec / h_need, w_need, w_exec, w_exec / w_need))
PY
done
