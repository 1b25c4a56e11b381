#!/usr/bin/env python3
"""share of exact zeros in h and w of masked (c_ard_nmf) fits on the synthetic matrix: p0 of the active-set estimate in DESIGN.md"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import singlet_amd as sa  # noqa: E402

c = sa.Context(0)
c.synth(30000, 100000, 20)
out = {}
for k in (10, 50, 100):
    c.fit_init(k, None)
    c.ard_run(0.0, 6, 0.01, 0.0, 1001, 20, 1e9, 3)
    W, d, H = c.get_factors()
    # per wave of four consecutive columns: share of coordinates that rest at zero in all four
    z = (H == 0)
    q = z[: (z.shape[0] // 4) * 4].reshape(-1, 4, k).all(axis=1).mean()
    out[k] = {"h_zero_share": float(z.mean()), "w_zero_share": float((W == 0).mean()), "h_all_four_zero_share": float(q)}
print(json.dumps(out))
