#!/usr/bin/env python3
"""Round-5 verdict, item 2: one masked (c_ard_nmf) iteration at k = 50 on an 8-rank loopback team (all ranks on device 0) against
the plain one-context masked iteration, per cell.  python scripts/r6/team_masked_rate.py [cells] [k] [ranks]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import singlet_amd as sa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
GENES, INV, L1, SEED = 30000, 20, 0.01, 1001
out = {"config": "%d genes x %d cells, k=%d, masked fit (test density 1/20), %d iterations timed after a 2-iteration fit" % (GENES, cells, k, 6)}


def timed(obj, label):
    obj.fit_init(k, None)
    obj.ard_run(0.0, 2, L1, 0.0, SEED, INV, 1e9, 5)          # lists, workspaces
    obj.fit_init(k, None)
    ctxs = [obj] if isinstance(obj, sa.Context) else [obj.rank_ctx(r) for r in range(ranks)]
    for c in ctxs:
        c.timing_enable(True)
        c.timing_get(reset=True)
    t0 = time.perf_counter()
    r = obj.ard_run(0.0, 6, L1, 0.0, SEED, INV, 1e9, 100)    # one trace row at the end only
    dt = time.perf_counter() - t0
    ph = [c.timing_get(reset=True) for c in ctxs]
    for c in ctxs:
        c.timing_enable(False)
    worst = {p: max(q[p][0] for q in ph) / 6.0 for p in ph[0]}
    out[label] = {"sec_per_masked_iter": dt / 6.0, "test_mse": float(r["test_mse"][-1]), "phases_ms_per_iter_max_over_ranks": {p: v for p, v in worst.items() if v > 0}}


with sa.Context(0) as c:
    c.synth(GENES, cells, INV)
    timed(c, "one_context")
with sa.Multi([0] * ranks) as M:
    M.synth(GENES, cells, INV)
    timed(M, "loopback_team_%d" % ranks)
a, b = out["one_context"]["sec_per_masked_iter"], out["loopback_team_%d" % ranks]["sec_per_masked_iter"]
out["team_over_one_context"] = b / a
print(json.dumps(out))
