#!/usr/bin/env python3
"""ALS iterations/sec of the c_nmf loop on the synthetic config-3 matrix
(30 000 genes x 1 000 000 cells, 5 % non-zeros, k = 50), BASELINE.json's metric.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one ALS iteration (H-update, scale, W-update, scale, cor) over the
whole matrix, inputs resident in HBM (generated on the device by the hash
generator of SURVEY.md 8(d)).  For N > 1 the cells are sharded over the ranks
(strong scaling: the same 1M x 30k problem) with two all-reduces per iteration
over RCCL: k row sums, then [k x genes right-hand sides | k x k Gram].
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0    # same table: float4 copy, 79 %


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genes", type=int, default=30000)
    ap.add_argument("--cells", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--inv-density", type=int, default=20)
    ap.add_argument("--L1", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=16000)
    ap.add_argument("--force-allreduce", action="store_true",
                    help="with one rank: still create the RCCL process group and route the two per-iteration sums through "
                         "the all-reduce hook (plumbing check of the multi-GPU path on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (restatement of singlet's OpenMP path) timed on this host's cores on a
    bounded sample: the first `cpu_sample_cells` cells of the same synthetic matrix, all genes,
    same k / penalties.  Per-cell phases are scaled to the full cell count; the per-gene part
    (the W-side NNLS, independent of the number of cells) is separated with a second,
    half-size run and NOT scaled."""
    import numpy as np
    from oracle import oracle as ora
    ns = min(args.cpu_sample_cells, args.cells)
    t0 = time.perf_counter()
    full = ora.synth_csc(args.genes, ns, args.inv_density)
    half = ora.CSC(full.x[:full.p[ns // 2]], full.i[:full.p[ns // 2]], full.p[:ns // 2 + 1], args.genes, ns // 2)
    w0 = ora.synth_winit(args.k, args.genes)
    gen_s = time.perf_counter() - t0
    res = {}
    for name, A in (("full", full), ("half", half)):
        At = A.t()
        # 1 warm-up iteration then 2 timed (fresh call each: the oracle has no resume)
        r1 = ora.c_nmf(A, At, 0.0, 1, args.L1, args.L1, 0.0, 0.0, 0, w0, timing=True)
        r3 = ora.c_nmf(A, At, 0.0, 3, args.L1, args.L1, 0.0, 0.0, 0, w0, timing=True)
        res[name] = (r3["phase_sec"] - r1["phase_sec"]) / 2.0, (r3["sweeps"] - r1["sweeps"]) / 2.0
    (pf, sf), (ph, _) = res["full"], res["half"]
    scale = args.cells / ns
    # predict(At) = a * cells + b  (b = the m NNLS solves of the W-update)
    a = (pf[2] - ph[2]) / (ns - ns // 2)
    b = max(pf[2] - a * ns, 0.0)
    t_full = (pf[0] + pf[1]) * scale + a * args.cells + b + pf[3]
    cores = ora.lib().ora_max_threads()
    return {
        "value": 1.0 / t_full, "unit": "iter/s", "cores": int(cores), "kind": "port",
        "sample": ("oracle/singlet_oracle.c (gcc -O2 -fopenmp, restatement of singlet's OpenMP path), first %d of %d "
                   "cells x %d genes, k=%d; ALS iterations 2-3 timed (%.2f s/iter on the sample: predict(A) %.2f, "
                   "scale(h) %.3f, predict(At) %.2f, scale(w)+cor %.3f); per-cell phases scaled x%.1f, the W-side NNLS "
                   "(%.2f s, from a half-size run) not scaled; mean NNLS sweeps H %.1f W %.1f; generation %.1f s not timed"
                   % (ns, args.cells, args.genes, args.k, float(pf.sum()), pf[0], pf[1], pf[2], pf[3], scale, b,
                      sf[0] / ns, sf[1] / args.genes, gen_s)),
        "sample_sec_per_iter": float(pf.sum()), "est_full_sec_per_iter": float(t_full),
    }


def measured_traffic(args, world, dom):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, separate runs, corrected as MI355X_MICROARCH.md prescribes): counters cannot be read
    from inside this process, so the figure measured by scripts/prof_r1.sh on this exact workload is
    kept in profiles/traffic.json and reported only when the workload matches; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if world == 1 and all(w[key] == getattr(args, key) for key in ("genes", "cells", "k", "inv_density")):
            return t["bytes_per_launch"][dom]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import singlet_amd as sa

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_allreduce:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = sa.Context(local_rank)
    # one explicit (non-default) stream shared by the kernels and torch: RCCL orders its collective against
    # torch's CURRENT stream, so the library must launch on exactly that one
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    from singlet_amd.sharded import shard_by_count, torch_allreduce_hook
    if dist is not None:
        ctx.set_allreduce(torch_allreduce_hook(dist, torch.device("cuda", local_rank)))

    # contiguous equal-count cell blocks: the synthetic columns are i.i.d., so equal counts are equal
    # non-zeros to within 0.1 %
    lo, n_local = shard_by_count(args.cells, world, rank)
    t0 = time.perf_counter()
    ctx.synth(args.genes, n_local, args.inv_density, cell_offset=lo, ncells_total=args.cells)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    m, n_loc, nnz_local = ctx.dims()
    ctx.fit_init(args.k, None)

    def step():
        ctx.step_begin()
        ctx.step_h(args.L1, 0.0)
        ctx.step_scale_h()
        ctx.step_w(args.L1, 0.0)
        return ctx.step_scale_w()

    for _ in range(args.warmup):
        step()
    ctx.sweeps_get(reset=True)
    ctx.timing_enable(True)
    ctx.timing_get(reset=True)
    barrier()
    t0 = time.perf_counter()
    tols = [step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    phases = ctx.timing_get(reset=True)
    ctx.timing_enable(False)
    sweeps = ctx.sweeps_get(reset=True)
    layout = ctx.layout_get()

    nnz_total = nnz_local
    if dist is not None:
        tt = torch.tensor([elapsed, float(nnz_local)], dtype=torch.float64, device="cuda")
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        nnz_total = int(sm[1])

    if rank == 0:
        k, n = args.k, args.cells
        ms_step = 1e3 * elapsed / args.steps
        # algorithmic bytes: values f64 (vb = 8) + int32 row index per non-zero, column pointers,
        # factor read + right-hand-side write; local shard on this rank
        vb = 8
        bytes_h = nnz_local * (vb + 4) + 4 * (n_loc + 1) + k * n_loc * 8 + k * m * 8
        bytes_w = nnz_local * (vb + 4) + 4 * (m + 1) + k * n_loc * 8 + k * m * 8
        bytes_iter = 2 * nnz_total * (vb + 4) + 4 * (n + m + 2) + 2 * k * n * 8 + 3 * k * m * 8
        ph_ms = {p: (v[0] / args.steps) for p, v in phases.items()}
        rhs_h_ms, rhs_w_ms = ph_ms["rhs_h"], ph_ms["rhs_w"]
        dom = "rhs_h" if rhs_h_ms >= rhs_w_ms else "rhs_w"
        dom_ms = max(rhs_h_ms, rhs_w_ms)
        dom_bytes = bytes_h if dom == "rhs_h" else bytes_w
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # what the kernel really has to move: the padded entry stream (12 B per stored entry), one byte per
        # (column pair, tile), the output, and the factor tiles every workgroup stages (mostly L2 / MALL hits)
        lay = layout["A" if dom == "rhs_h" else "At"]
        ncols_dom = n_loc if dom == "rhs_h" else m
        nrows_dom = m if dom == "rhs_h" else n_loc
        stream_bytes = lay["entries"] * 12 + lay["col_blocks"] * lay["tiles"] * (32 + 8)
        out_bytes = 8 * k * ncols_dom * (lay["tile_ranges"] + (2 if lay["tile_ranges"] > 1 else 0))
        staged_bytes = 8 * k * nrows_dom * ((lay["col_blocks"] + 7) // 8)
        traffic = measured_traffic(args, world, dom)
        out = {
            "metric": "ALS iterations/sec (1M cells x 30k genes, 5% nnz, k=50)",
            "value": args.steps / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic CSC %d genes x %d cells, 1/%d non-zero (nnz %d), k=%d, L1=%g, tol=0 "
                                   "(no early stop), cells sharded over %d GPU(s)"
                                   % (m, n, args.inv_density, nnz_total, k, args.L1, world),
                       "genes": m, "cells": n, "k": k, "nnz": nnz_total, "parallelism": "cells/%d" % world},
            "roofline": {"bound": "hbm", "kernel": "acc_tiled_kernel (%s: sparse accumulate of predict, one launch = one pass over the matrix)" % dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured_copy": achieved / HBM_MEASURED_GBS, "traffic": traffic,
                         "stream_layout": dict(lay, stream_bytes=stream_bytes, output_bytes=out_bytes,
                                               factor_tile_bytes_staged_from_l2=staged_bytes),
                         "algorithmic_bytes_per_pass": dom_bytes, "ms_per_pass": dom_ms,
                         "whole_iteration": {"algorithmic_bytes": bytes_iter, "achieved": bytes_iter / (ms_step * 1e-3) / 1e9 / world,
                                             "frac": bytes_iter / (ms_step * 1e-3) / 1e9 / world / HBM_PEAK_GBS}},
            "phases_ms_per_step": ph_ms,
            "tol_last": tols[-1], "generate_s": gen_s,
        }
        # mean sweeps per column, and mean sweeps each 64-column wave actually ran (its slowest column)
        out["nnls_mean_sweeps"] = {"h": sweeps["h_sweeps"] / (args.steps * n_loc), "w": sweeps["w_sweeps"] / (args.steps * m),
                                   "h_per_wave": sweeps["h_wave_sweeps"] / (args.steps * ((n_loc + 63) // 64)),
                                   "w_per_wave": sweeps["w_wave_sweeps"] / (args.steps * ((m + 63) // 64))}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # noqa: BLE001 - the baseline is reported, never required
                out["cpu_baseline"] = {"error": repr(e)}
        line = json.dumps(out)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio; push it out first so that the JSON line is the
        # last thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
