#!/usr/bin/env python3
"""ALS iterations/sec of the c_nmf loop on the synthetic config-3 matrix
(30 000 genes x 1 000 000 cells, 5 % non-zeros, k = 50), BASELINE.json's metric.

  python bench.py [--gpus N] [--steps K] [--warmup W]                  (no launcher: ONE process drives the N devices)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one process per GPU)

Which form runs is decided by the environment (plan()): with WORLD_SIZE set (torch.distributed.run) every process is
one rank and joins the library's RCCL communicator (sgl_comm_init_rank); without it `--gpus N` (N > 1) takes the
library's one-process team -- sgl_multi_create (ncclCommInitAll), one host thread per device inside the library --
which is what an R session uses (no launcher, no torch).  `--single-process` forces that form at N = 1 (an RCCL team
of one); `--loopback` puts all N ranks on device 0 (the exchange is then a HIP kernel, RCCL refuses duplicate
devices): the team logic end to end on a 1-GPU box, marked "loopback": true.

A step = one ALS iteration (H-update, scale, W-update, scale, cor) over the
whole matrix, inputs resident in HBM (generated on the device by the hash
generator of SURVEY.md 8(d)).  For N > 1 the cells are sharded over the ranks
(strong scaling: the same 1M x 30k problem); the exchange runs inside the
library over RCCL (--comm native, the default for N > 1): one grouped
collective per iteration (reduce-scatter of the k x genes right-hand sides by
gene blocks + all-reduce of [k x k Gram | k row sums]) and one all-gather of
the solved W blocks.  --comm hook keeps round 1's path (two all-reduces through
torch.distributed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # v_mfma_f64_*: 32 FLOP / clk / SIMD x 1024 SIMDs x 2.4 GHz = the FP64 vector peak (AMD's MI355X figure;
                              # MI355X_MICROARCH.md tabulates the f32-input MFMA the same way: "runs at the vector rate")
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBS = 6290.0    # same table: float4 copy, 79 %


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("als", "ard"), default="als",
                    help="als = the headline: plain ALS iterations of c_nmf on config 3; ard = BASELINE config 5 on one GPU: the "
                         "(rank, replicate) grid of masked fits behind ard_nmf / cross_validate_nmf on one resident matrix, with "
                         "the Gram downdate's FP64-MFMA roofline and the CPU oracle's c_ard_nmf timed beside it")
    ap.add_argument("--ranks", default="10,20,30,40,50,60,70,80,90,100", help="--workload ard: the ranks of the grid")
    ap.add_argument("--replicates", type=int, default=3, help="--workload ard: restarts per rank (R's n_replicates)")
    ap.add_argument("--maxit", type=int, default=10, help="--workload ard: iterations per masked fit")
    ap.add_argument("--trace", type=int, default=5, help="--workload ard: trace_test_mse (mse_test every this many iterations)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)   # 2 warm-up + 10 timed iterations at fixed maxit
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genes", type=int, default=30000)
    ap.add_argument("--cells", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--inv-density", type=int, default=20)
    ap.add_argument("--L1", type=float, default=0.01)
    ap.add_argument("--data", choices=("iid", "skewed"), default="iid",
                    help="iid = BASELINE's synthetic matrix (the headline); skewed = the same generator with log-normal weights "
                         "per cell (sigma 0.5) and per gene (sigma 1.5): heavy-tailed row and column counts (secondary record)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=None,
                    help="cells of the CPU baseline sample (0 = all); default 200 000 (als) / 10 000 (ard)")
    ap.add_argument("--comm", choices=("auto", "native", "hook", "none"), default="auto",
                    help="exchange between the ranks: native = RCCL inside the library (sgl_comm_init_rank), hook = "
                         "torch.distributed all-reduce through sgl_set_allreduce, auto = none for one rank, native otherwise")
    ap.add_argument("--native-comm", action="store_true", help="same as --comm native (with one rank: an RCCL team of one)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus devices through the library's own team (sgl_multi_*); the default "
                         "for --gpus N > 1 when no launcher set WORLD_SIZE")
    ap.add_argument("--loopback", action="store_true",
                    help="single-process team with all ranks on device 0 (exchange through a HIP kernel instead of RCCL): "
                         "rehearsal of the N-rank team logic on a 1-GPU box; the JSON line carries \"loopback\": true")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="with one rank: still create the RCCL process group and route the two per-iteration sums through "
                         "the all-reduce hook (plumbing check of the hook path on a 1-GPU box)")
    a = ap.parse_args(argv)
    if a.cpu_sample_cells is None:
        a.cpu_sample_cells = 200000 if a.workload == "als" else 10000
    return a


def plan(args, env):
    """How this invocation runs, from the flags and the launcher's environment only (no GPU, no imports):
      {'form': 'process-per-gpu', 'world': WORLD_SIZE, 'rank': RANK, 'local_rank': LOCAL_RANK}   under torch.distributed.run
      {'form': 'single-process', 'world': N, 'devices': [...], 'loopback': bool}                  one process, N devices
      {'form': 'one-gpu', 'world': 1}                                                            plain context, no team
    A launcher's WORLD_SIZE > 1 wins over --gpus (the driver passes both, equal).  With no launcher, or a launcher
    world of ONE process, --gpus N > 1 is the one-process team.  --single-process / --loopback under a launcher with
    more than one rank is a contradiction and refused."""
    ws = env.get("WORLD_SIZE")
    world = 1
    if ws is not None and str(ws).strip() != "":
        world = int(ws)
        if world < 1:
            raise SystemExit("bench.py: WORLD_SIZE=%r" % ws)
    if world > 1:
        if args.single_process or args.loopback:
            raise SystemExit("bench.py: --single-process / --loopback drive all devices from ONE process; do not start them "
                             "under a launcher with WORLD_SIZE=%d" % world)
        return {"form": "process-per-gpu", "world": world, "rank": int(env.get("RANK", "0")),
                "local_rank": int(env.get("LOCAL_RANK", "0"))}
    n = int(args.gpus)
    if n < 1 or n > 16:
        raise SystemExit("bench.py: --gpus %d out of range (1..16)" % n)
    if n > 1 or args.single_process or args.loopback:
        if args.comm in ("hook", "none") or args.force_allreduce:
            raise SystemExit("bench.py: the one-process team exchanges inside the library (--comm native); --comm hook / none "
                             "and --force-allreduce belong to the one-rank-per-process form")
        if args.data != "iid":
            raise SystemExit("bench.py: --data skewed is a one-GPU record")
        return {"form": "single-process", "world": n, "devices": [0] * n if args.loopback else list(range(n)),
                "loopback": bool(args.loopback), "rank": 0, "local_rank": 0}
    return {"form": "one-gpu", "world": 1, "rank": 0, "local_rank": 0}


def pick_device(local_rank, visible, forced=None):
    """Device index of a process-per-GPU rank, from counts only (no GPU call): the launcher's LOCAL_RANK when that many devices are
    visible; with FEWER visible than LOCAL_RANK + 1 -- a launcher that isolates one device per rank (HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES set per process) -- the rank's own device is index LOCAL_RANK modulo the visible count, i.e. 0 when one
    is visible.  -> (device, how).  `forced` = SGL_BENCH_FORCE_DEVICE (rehearsals on a 1-GPU box)."""
    if forced is not None and str(forced).strip() != "":
        return int(forced), "forced by SGL_BENCH_FORCE_DEVICE"
    visible = int(visible)
    if visible < 1:
        raise SystemExit("bench.py: no gfx950 device visible to local rank %d" % local_rank)
    if local_rank < visible:
        return int(local_rank), "LOCAL_RANK"
    return int(local_rank % visible), ("LOCAL_RANK %d but only %d device(s) visible to this process (devices isolated per rank): "
                                       "index %d" % (local_rank, visible, local_rank % visible))


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, w_start=None, it_start=0):
    """The oracle (line-by-line restatement of singlet's OpenMP path, oracle/singlet_oracle.c) MEASURED on
    this host's cores on a bounded sample of the same workload: the first `cpu_sample_cells` cells
    (default 200 000 = the slice SURVEY.md 8d prescribes; --cpu-sample-cells 0 = all cells) x all genes,
    same k / penalties, A and t(A) in host memory.  One warm-up iteration, then two ALS iterations timed
    inside the C code (omp_get_wtime per phase).  w_start (round 6): the w of the GPU fit after its `it_start`
    iterations -- the CPU then times iterations it_start + 2 .. it_start + 3 of the SAME fit, where the NNLS needs
    the sweeps the GPU's timed iterations needed (from the generator's initial w it would time iterations 2 - 3:
    46 sweeps per cell against the GPU window's 29, ~10 % in the GPU's favour: round-5 verdict).  Two builds of the same source: gcc -O2 (R's default
    level, what `value` reports) and -O3 -march=native (the generous variant).  When the sample is not
    the whole matrix the per-cell phases are scaled to the full cell count; the part of predict(At)
    that does not grow with the cells (the m NNLS solves of the W-update) is separated with a second,
    half-size run and not scaled."""
    import ctypes as C
    import subprocess
    import numpy as np
    from oracle import oracle as ora
    ns = args.cells if args.cpu_sample_cells <= 0 else min(args.cpu_sample_cells, args.cells)
    t0 = time.perf_counter()
    full = ora.synth_csc(args.genes, ns, args.inv_density)
    w0 = ora.synth_winit(args.k, args.genes) if w_start is None else np.ascontiguousarray(w_start, dtype=np.float64)
    assert w0.shape == (args.genes, args.k)
    full_t = full.t()
    gen_s = time.perf_counter() - t0

    def timed_run(A, At, lib=None):
        """-> (phase seconds per iteration [4], sweeps per iteration [2]) of ALS iterations 2-3"""
        L = lib or ora.lib()
        L.ora_set_timing_skip(1)
        try:
            if lib is None:
                r = ora.c_nmf(A, At, 0.0, 3, args.L1, args.L1, 0.0, 0.0, 0, w0, timing=True)
                return r["phase_sec"] / 2.0, r["sweeps"] / 2.0
            f64p, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
            L.ora_c_nmf.restype = C.c_int
            L.ora_c_nmf.argtypes = ora.lib().ora_c_nmf.argtypes
            w = np.array(w0, dtype=np.float64, order="C")
            h, d = np.empty((A.ncol, args.k)), np.empty(args.k)
            tr, ph, sw = np.zeros(3), np.zeros(4), np.zeros(2, dtype=np.int64)
            P = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
            L.ora_c_nmf(P(A.x, f64p), P(A.i, i32p), P(A.p, i32p), P(At.x, f64p), P(At.i, i32p), P(At.p, i32p), A.nrow,
                        A.ncol, 0.0, 3, args.L1, args.L1, 0.0, 0.0, 0, args.k, P(w, f64p), P(h, f64p), P(d, f64p),
                        P(tr, f64p), P(ph, f64p), P(sw, i64p))
            return ph / 2.0, sw / 2.0
        finally:
            L.ora_set_timing_skip(0)

    pf, sf = timed_run(full, full_t)
    scale = args.cells / ns
    b = 0.0
    if ns < args.cells:
        nh = ns // 2
        half = ora.CSC(full.x[:full.p[nh]], full.i[:full.p[nh]], full.p[:nh + 1], args.genes, nh)
        ph, _ = timed_run(half, half.t())
        # predict(At) = a * cells + b  (b = the m NNLS solves of the W-update)
        a = (pf[2] - ph[2]) / (ns - nh)
        b = min(max(pf[2] - a * ns, 0.0), pf[2])

    def full_size(p, bb):
        return (p[0] + p[1]) * scale + (p[2] - bb) * scale + bb + p[3]

    t_full = full_size(pf, b)
    cores = ora.lib().ora_max_threads()
    out = {
        "value": 1.0 / t_full, "unit": "iter/s", "cores": int(cores), "kind": "port", "cpu_model": _cpu_model(),
        "build": "gcc -O2 -fopenmp -ffp-contract=off",
        "sample_sec_per_iter": float(pf.sum()), "est_full_sec_per_iter": float(t_full),
        "sample_phases_sec": {"predict_A": float(pf[0]), "scale_h": float(pf[1]), "predict_At": float(pf[2]),
                              "scale_w_cor": float(pf[3])},
        "nnls_mean_sweeps": {"h": float(sf[0]) / ns, "w": float(sf[1]) / args.genes},
    }
    try:  # the generous build, compiled for THIS host
        odir = os.path.join(ROOT, "oracle")
        subprocess.check_call(["make", "-C", odir, "libsinglet_oracle_native.so"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        Ln = C.CDLL(os.path.join(odir, "libsinglet_oracle_native.so"))
        Ln.ora_set_timing_skip.argtypes = [C.c_int]
        pn, _ = timed_run(full, full_t, Ln)
        bn = b * (pn[2] / pf[2]) if pf[2] > 0 else 0.0
        out["native_build"] = {"build": "gcc -O3 -march=native -fopenmp -ffp-contract=off", "value": 1.0 / full_size(pn, bn),
                               "sample_sec_per_iter": float(pn.sum())}
    except Exception as e:  # noqa: BLE001
        out["native_build"] = {"error": repr(e)}
    what = "all %d cells (full)" % ns if ns == args.cells else "%d cells (first %d of %d)" % (ns, ns, args.cells)
    out["iterations_timed"] = ("2-3 of a fit from the generator's initial w" if w_start is None else
                               "%d-%d of the fit the GPU timed (w after the GPU's iteration %d handed over)" % (it_start + 2, it_start + 3, it_start))
    out["sample"] = ("oracle/singlet_oracle.c, restatement of singlet's OpenMP path, %d threads on %s; %s x %d genes, k=%d; "
                     "1 warm-up + ALS iterations %s timed in C: %.2f s/iter on the sample (predict(A) %.2f, scale(h) %.3f, "
                     "predict(At) %.2f, scale(w)+cor %.3f)%s; generation + transpose %.1f s not timed"
                     % (cores, out["cpu_model"], what, args.genes, args.k, out["iterations_timed"], float(pf.sum()), pf[0], pf[1], pf[2], pf[3],
                        "" if ns == args.cells else "; per-cell phases scaled x%.1f, the W-side NNLS (%.2f s, from a "
                        "half-size run) not scaled" % (scale, b), gen_s))
    return out


def measured_traffic(args, world, dom, lay):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE, separate runs, corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be read
    from inside this process, so the figure measured by scripts/prof_r2.sh is kept in
    profiles/traffic.json together with the workload AND the entry-stream layout it was measured on
    (entries, tiles, tile ranges): it is reported only when both match this run, otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if world != 1 or not all(w[key] == getattr(args, key) for key in ("genes", "cells", "k", "inv_density")):
            return None
        want = t["layout"][dom]
        if all(int(want[key]) == int(lay[key]) for key in ("entries", "tiles", "tile_rows", "tile_ranges")):
            return t["bytes_per_launch"][dom]
    except (OSError, KeyError, ValueError, TypeError):
        pass
    return None


COMM_PER_ITERATION = {
    "none": "no exchange (one shard)",
    "native": "RCCL inside the library: 1 grouped collective (reduce-scatter k x genes by gene blocks + all-reduce [k x k | k]) "
              "+ 1 all-gather of the W blocks",
    "native-single-process": "RCCL inside the library, one process driving all devices (sgl_multi_*: ncclCommInitAll, one host "
                             "thread per device): 1 grouped collective (reduce-scatter k x genes by gene blocks + all-reduce "
                             "[k x k | k]) + 1 all-gather of the W blocks",
    "loopback": "all ranks on one device: the same exchange steps as summing HIP kernels (no RCCL, no xGMI)",
    "hook": "2 all-reduces through torch.distributed (k row sums; [k x genes | k x k])",
}


def report(args, run):
    """The JSON record from what a run measured.  run: world, elapsed, steps' tols, rank-0 phases / sweeps / layout /
    dims, nnz_total, comm dict, gen_s."""
    world, elapsed = run["world"], run["elapsed"]
    m, n_loc, nnz_local = run["dims"]
    nnz_total, phases, sweeps, layout = run["nnz_total"], run["phases"], run["sweeps"], run["layout"]
    k, n = args.k, args.cells
    ms_step = 1e3 * elapsed / args.steps
    # algorithmic bytes: values f64 (vb = 8) + int32 row index per non-zero, column pointers,
    # factor read + right-hand-side write; local shard on this rank
    vb = 8
    bytes_h = nnz_local * (vb + 4) + 4 * (n_loc + 1) + k * n_loc * 8 + k * m * 8
    bytes_w = nnz_local * (vb + 4) + 4 * (m + 1) + k * n_loc * 8 + k * m * 8
    bytes_iter = 2 * nnz_total * (vb + 4) + 4 * (n + m + 2) + 2 * k * n * 8 + 3 * k * m * 8
    # every rank's hipEvent phases (rank order); the headline phases are the MAX over the ranks -- an iteration ends when
    # the slowest rank has delivered its part of the exchange -- and `per_rank` keeps each rank's own figures, so that a
    # straggler or a slow collective shows in the first multi-GPU record
    phases_all = run.get("phases_all") or [phases]
    per_rank_ms = [{p: (v[0] / args.steps) for p, v in ph.items()} for ph in phases_all]
    ph_ms = {p: max(r[p] for r in per_rank_ms) for p in per_rank_ms[0]}
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
    traffic = measured_traffic(args, world, dom, lay) if args.data == "iid" else None
    # Why 0.60 of the HBM peak is out of reach for this formulation in FP64 at k = 50 (DESIGN.md "The sparse update
    # against its rooflines"): every entry tuple (2 entries; 4 at ranks up to 32) costs one ds_read_b128 -- 4.1 LDS cycles
    # per CU by SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, the chip at ~2.05 GHz inside this kernel -- one broadcast address add and
    # two FP64 FMAs (4 cycles each per SIMD).  Round-4 ablations of the full problem (profiles/README.md): the loop without
    # its FMAs 8.35 ms, without its LDS reads 8.38 ms, complete 10.57 ms per pass.
    nsl = 4 if (k <= 32 and not os.environ.get("SGL_TILED_NO_QUAD")) else 2
    tuples = lay["entries"] / float(nsl)
    clk = 2.05e9
    ceiling = {"columns_per_lds_instruction": nsl,
               "lds_floor_ms": tuples * 4.1 / clk / 256 * 1e3,
               "fp64_fma_floor_ms": tuples * 2 * 4.0 / clk / 1024 * 1e3,
               "valu_floor_ms": tuples * 3 * 4.0 / clk / 1024 * 1e3,
               "loop_floor_ms": tuples * 2.35e-9 / 256 * 1e3,
               "hbm_floor_ms": dom_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
               "source": "profiles/r4_pmc_sq_insts.csv, r4_pmc_sq_cycles.csv (4.1 LDS cycles per ds_read_b128, ~2.05 GHz, 4 cycles per "
                         "VALU instruction) and the round-4 ablations in profiles/README.md: 2.35 ns per entry tuple per CU for the loop "
                         "with either its FMAs or its LDS reads removed",
               "note": "the kernel is bound by LDS operand delivery and FP64 issue, which overlap incompletely at two waves per "
                       "SIMD, not by HBM: the frac below cannot exceed hbm_floor_ms / loop_floor_ms with one LDS read per entry tuple.  "
                       "CLOSED (round 6, DESIGN.md): three factors per lane (ranks 33 - 48) prices at 10.6 ms per pass against 10.2 -- an LDS "
                       "read costs per instruction, ds_read_b64 as much as ds_read_b128 (profiles/r6_ubench_rates.txt) -- and an f32 copy of the "
                       "operand in LDS at 10.6 ms too (a v_cvt_f64_f32 per factor at the FP64 issue rate)"}
    ceiling["max_frac_of_this_formulation"] = (ceiling["hbm_floor_ms"] / ceiling["loop_floor_ms"]) if tuples > 0 else None
    comm = run["comm"]
    default_shape = (args.genes, args.cells, args.k, args.inv_density) == (30000, 1000000, 50, 20)
    out = {
        "metric": "ALS iterations/sec (1M cells x 30k genes, 5% nnz, k=50)" if default_shape else
                  "ALS iterations/sec (%d cells x %d genes, 1/%d nnz, k=%d)" % (args.cells, args.genes, args.inv_density, args.k),
        "value": args.steps / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic" if args.data == "iid" else "synthetic-skewed",
        "config": {"workload": "synthetic CSC %d genes x %d cells, %s (nnz %d), k=%d, L1=%g, tol=0 "
                               "(no early stop), cells sharded over %d GPU(s)"
                               % (m, n, "1/%d non-zero" % args.inv_density if args.data == "iid" else
                                  "mean density 1/%d with log-normal cell (sigma 0.5) and gene (sigma 1.5) weights" % args.inv_density,
                                  nnz_total, k, args.L1, world),
                   "genes": m, "cells": n, "k": k, "nnz": nnz_total, "parallelism": "cells/%d" % world},
        "roofline": {"bound": "hbm", "kernel": "acc_tiled_kernel (%s: sparse accumulate of predict, one launch = one pass over the matrix)" % dom,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_of_measured_copy": achieved / HBM_MEASURED_GBS, "traffic": traffic, "ceiling": ceiling,
                     "entries_per_nonzero": {"rhs_h": layout["A"]["entries"] / max(nnz_local, 1),
                                             "rhs_w": layout["At"]["entries"] / max(nnz_local, 1)},
                     "stream_layout": dict(lay, stream_bytes=stream_bytes, output_bytes=out_bytes,
                                           factor_tile_bytes_staged_from_l2=staged_bytes),
                     "stream_layouts": {"rhs_h": layout["A"], "rhs_w": layout["At"]},
                     "algorithmic_bytes_per_pass": dom_bytes, "ms_per_pass": dom_ms,
                     "whole_iteration": {"algorithmic_bytes": bytes_iter, "achieved": bytes_iter / (ms_step * 1e-3) / 1e9 / world,
                                         "frac": bytes_iter / (ms_step * 1e-3) / 1e9 / world / HBM_PEAK_GBS}},
        "phases_ms_per_step": ph_ms,
        "phases_are": "max over the %d rank(s) of each phase's hipEvent time per iteration" % len(per_rank_ms),
        "comm": dict(comm, per_iteration=COMM_PER_ITERATION[comm["mode"]]),
        "tol_last": run["tols"][-1], "generate_s": run["gen_s"],
    }
    if len(per_rank_ms) > 1 or run.get("rank_info"):
        info = run.get("rank_info") or [{} for _ in per_rank_ms]
        out["per_rank"] = [dict(info[r], rank=r, phases_ms_per_step=per_rank_ms[r], sum_of_phases_ms=sum(per_rank_ms[r].values()),
                                comm_ms=per_rank_ms[r].get("comm", 0.0)) for r in range(len(per_rank_ms))]
        # compute = everything but the exchange; a rank's comm phase holds its wait for the slowest rank plus the transfer
        sums = [r["sum_of_phases_ms"] - r["comm_ms"] for r in out["per_rank"]]
        out["rank_imbalance"] = {"slowest_rank_by_compute": int(max(range(len(sums)), key=lambda q: sums[q])),
                                 "max_over_min_compute": max(sums) / max(min(sums), 1e-12),
                                 "note": "comm_ms = wait for the slowest rank + transfer: the straggler shows the SMALLEST comm_ms",
                                 "comm_ms_max": max(r["comm_ms"] for r in out["per_rank"]), "comm_ms_min": min(r["comm_ms"] for r in out["per_rank"])}
    if run.get("loopback"):
        out["loopback"] = True
        out["config"]["parallelism"] = "cells/%d, all ranks on ONE device (loopback rehearsal, not a scaling point)" % world
    # mean sweeps per column, and mean sweeps each 64-column wave actually ran (its slowest column)
    out["nnls_mean_sweeps"] = {"h": sweeps["h_sweeps"] / (args.steps * n_loc), "w": sweeps["w_sweeps"] / (args.steps * max(run["w_cols_rank0"], 1)),
                               "h_per_wave": sweeps["h_wave_sweeps"] / (args.steps * ((n_loc + 63) // 64)),
                               "w_per_wave": sweeps["w_wave_sweeps"] / (args.steps * ((run["w_cols_rank0"] + 63) // 64))}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, run.get("w_start"), args.warmup + args.steps)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:  # noqa: BLE001 - the baseline is reported, never required
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def emit(line):
    # RCCL writes a version banner through C stdio; push it out first so that the JSON line is the
    # last thing on stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    print(line, flush=True)


def run_single_process(args, pl):
    """One process, N devices, no launcher and no torch: the library's own team (sgl_multi_create -> ncclCommInitAll over
    the N devices, or the loopback exchange when all ranks sit on device 0), one host thread per device inside the
    library.  sgl_multi_iterate returns when EVERY rank has delivered its tol (each rank's stream is synchronised by that
    read), so the K timed iterations are bracketed by complete device idleness on both sides: the barrier + synchronize
    of the bench contract."""
    import singlet_amd as sa
    from singlet_amd._lib import SingletHipError
    world = pl["world"]
    have = sa.device_count()
    if not pl["loopback"] and have < world:
        # same words as sgl_multi_create's own refusal, before anything is allocated
        raise SystemExit("bench.py: --gpus %d but %d gfx950 device(s) visible (sgl_device_count); use --loopback to rehearse the "
                         "team logic on one device" % (world, have))
    try:
        M = sa.Multi(pl["devices"])
    except SingletHipError as e:
        raise SystemExit("bench.py: sgl_multi_create(%r) failed: %s" % (pl["devices"], e))
    with M:
        t0 = time.perf_counter()
        M.synth(args.genes, args.cells, args.inv_density)
        gen_s = time.perf_counter() - t0
        M.fit_init(args.k, None)
        c0 = M.rank_ctx(0)
        ranks = [M.rank_ctx(r) for r in range(world)]
        dims = [c.dims() for c in ranks]

        def step():
            return M.iterate(args.L1, args.L1, 0.0, 0.0)

        for _ in range(args.warmup):
            step()
        info = c0.comm_info()
        if not pl["loopback"] and (not info["is_rccl"] or info["nranks"] != world):
            raise SystemExit("bench.py: the library's communicator spans %d ranks (rccl=%s), asked for %d"
                             % (info["nranks"], info["is_rccl"], world))
        for c in ranks:
            c.sweeps_get(reset=True)      # reads the rank's counters behind a stream synchronise: EVERY rank's stream is idle
            c.timing_enable(True)         # before t0, whatever --warmup was (0 included: fit_init's asynchronous tail)
            c.timing_get(reset=True)
        t0 = time.perf_counter()
        tols = [step() for _ in range(args.steps)]
        elapsed = time.perf_counter() - t0    # iterate() returns when every rank's tol has been read: all streams idle again
        phases_all = [c.timing_get(reset=True) for c in ranks]
        for c in ranks:
            c.timing_enable(False)
        phases = phases_all[0]
        m = dims[0][0]
        mb = (m + world - 1) // world if world > 1 else m
        run = {"world": world, "elapsed": elapsed, "tols": tols, "dims": dims[0], "nnz_total": int(sum(d[2] for d in dims)),
               "phases": phases, "phases_all": phases_all,
               "rank_info": [{"device": int(pl["devices"][r]), "cells": int(dims[r][1]), "nnz": int(dims[r][2])} for r in range(world)],
               "sweeps": c0.sweeps_get(reset=True), "layout": c0.layout_get(), "gen_s": gen_s,
               "w_cols_rank0": min(mb, m), "loopback": pl["loopback"],
               "comm": {"mode": "loopback" if pl["loopback"] else "native-single-process", "note": None,
                        "rccl_nranks": info["nranks"] if info["is_rccl"] else None, "rccl_path": info["path"] or None,
                        "host_coordination": "none (one process; one library thread per device%s)"
                                             % (", SGL_MULTI_SERIAL: one thread for all" if os.environ.get("SGL_MULTI_SERIAL") else ""),
                        "devices": pl["devices"],
                        "tol_bit_identical_across_ranks": True if world > 1 else None}}   # checked by the library on every iteration
        out = report(args, run)
    emit(json.dumps(out))


def cpu_baseline_ard(args, ranks):
    """The oracle's c_ard_nmf (oracle/singlet_oracle.c: predict_mask / mse_test / the loop of src/singlet.cpp:1090-1152 restated)
    MEASURED on this host's cores on a cell slice of the same synthetic matrix: per rank one 2-iteration masked fit (one trace
    row) on the first S cells and one on the first S / 2 -- both the per-cell work (hashing, right-hand sides, every cell's and
    every gene's Gram downdate) and the part that does not grow with the cells (the genes' NNLS solves) are then known:
    t(cells) = a * cells + b per masked iteration, evaluated at the full cell count."""
    import numpy as np
    from oracle import oracle as ora
    ns = args.cells if args.cpu_sample_cells <= 0 else min(args.cpu_sample_cells, args.cells)
    nh = max(ns // 2, 1)
    t0 = time.perf_counter()
    full = ora.synth_csc(args.genes, ns, args.inv_density)
    full_t = full.t()
    half = ora.CSC(full.x[:full.p[nh]], full.i[:full.p[nh]], full.p[:nh + 1], args.genes, nh)
    half_t = half.t()
    gen_s = time.perf_counter() - t0
    rows, total = [], 0.0
    fits = args.replicates * args.maxit
    ora.c_ard_nmf(half, half_t, 0.0, 1, args.L1, 0.0, 0, ora.synth_winit(ranks[0], args.genes), 1001, args.inv_density, 1e9, 1)   # thread pool, page faults
    for k in ranks:
        w0 = ora.synth_winit(k, args.genes)

        def timed(A, At):
            t = time.perf_counter()
            ora.c_ard_nmf(A, At, 0.0, 2, args.L1, 0.0, 0, w0, 1001, args.inv_density, 1e9, 2)
            return (time.perf_counter() - t) / 2.0
        ts = timed(full, full_t)
        if ns < args.cells and nh < ns:
            th = timed(half, half_t)
            a = (ts - th) / (ns - nh)
            if a <= 0.0:            # timer noise on a tiny sample: everything scales
                a = ts / ns
            b = min(max(ts - a * ns, 0.0), ts)
        else:
            a, b = ts / ns, 0.0
        est = a * args.cells + b
        rows.append({"k": k, "sample_sec_per_masked_iter": ts, "est_full_sec_per_masked_iter": est, "not_scaled_sec": b})
        total += fits * est
    cores = int(ora.lib().ora_max_threads())
    return {"value": total, "unit": "s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "build": "gcc -O2 -fopenmp -ffp-contract=off", "per_rank": rows,
            "sample": "oracle/singlet_oracle.c c_ard_nmf (restatement of src/singlet.cpp:1090-1152 with predict_mask :436-466 and "
                      "mse_test :536-568), %d threads on %s; per rank a 2-iteration masked fit with one trace row on the first %d and "
                      "the first %d of %d cells x %d genes, seconds per masked iteration extrapolated linearly in the cells with "
                      "the part that does not grow (the genes' solves) kept; value = sum over the grid's %d fits x %d iterations; "
                      "generation + transposes %.1f s not timed.  A port: its Gram downdate AAt(wsub) is a plain rank-one "
                      "loop where the reference calls Eigen's blocked rankUpdate -- the reference itself is likely several times "
                      "faster than this figure" % (cores, _cpu_model(), ns, nh, args.cells, args.genes,
                                                   len(ranks) * args.replicates, args.maxit, gen_s)}


def run_ard(args):
    """BASELINE config 5 on ONE GPU: the (rank, replicate) grid of 10-iteration masked fits that cross_validate_nmf / the rank
    search of ard_nmf run (R/cross_validate_nmf.R:69-104, R/ard_nmf.R:95-160), on one resident matrix: sgl_fit_init +
    sgl_ard_run per (k, replicate), fit set-up (entry streams per rank, mask lists per seed) inside the timed region.
    value = wall seconds of the grid.  roofline = the per-column Gram downdates of predict_mask (src/singlet.cpp:458-463) against
    the FP64 matrix pipe: one masked iteration sums w_r w_r^T over every drawn (cell, gene) pair, once per cell and once per
    gene -- pairs x k (k + 1) flop on the symmetric half -- and the hipEvent time of the phase that holds exactly those kernels
    (SGL_PH_MASK) divides it; taken from the last replicate of each rank, whose mask lists exist already."""
    import singlet_amd as sa
    ranks = [int(v) for v in str(args.ranks).split(",") if v.strip()]
    if not ranks or min(ranks) < 1:
        raise SystemExit("bench.py: --ranks %r" % args.ranks)
    ctx = sa.Context(0)
    t0 = time.perf_counter()
    ctx.synth(args.genes, args.cells, args.inv_density)
    gen_s = time.perf_counter() - t0
    m, n, nnz = ctx.dims()
    ctx.fit_init(ranks[0], None)
    ctx.ard_run(0.0, 1, args.L1, 0.0, 1, args.inv_density, 1e9, 1)      # warm-up: module load, workspaces (its mask is not the grid's)
    ctx.timing_enable(True)
    fits, per_rank = [], []
    t_all = time.perf_counter()
    for k in ranks:
        rr = []
        for rep in range(1, args.replicates + 1):
            ctx.timing_get(reset=True)
            t0 = time.perf_counter()
            ctx.fit_init(k, None, synth_seed=0x5EED + rep)              # a different initial w per replicate
            r = ctx.ard_run(1e-4, args.maxit, args.L1, 0.0, 1000 + rep, args.inv_density, 1e-4, args.trace)
            dt = time.perf_counter() - t0
            ph = ctx.timing_get(reset=True)
            pairs = ctx.mask_pairs()
            it = max(int(r["n_iter"]), 1)
            row = {"k": k, "rep": rep, "wall_s": dt, "iters": int(r["n_iter"]), "traces": len(r["test_mse"]),
                   "sec_per_masked_iter": dt / it, "test_mse": float(r["test_mse"][-1]),
                   "phases_ms": {p: v[0] for p, v in ph.items() if v[0] > 0}, "mask_pairs": {"per_cell": pairs[0], "per_gene": pairs[1]}}
            rr.append(row)
            fits.append(row)
            print({q: row[q] for q in ("k", "rep", "wall_s", "iters", "sec_per_masked_iter")}, file=sys.stderr, flush=True)
        last = rr[-1]
        npairs = last["mask_pairs"]["per_cell"] + last["mask_pairs"]["per_gene"]
        mask_s = last["phases_ms"].get("mask", 0.0) * 1e-3
        flops = float(npairs) * k * (k + 1) * max(last["iters"], 1)
        tf = flops / mask_s / 1e12 if (mask_s > 0 and npairs > 0) else None
        per_rank.append({"k": k, "sec_per_masked_iter": sum(q["sec_per_masked_iter"] for q in rr) / len(rr),
                         "fit_wall_s": [q["wall_s"] for q in rr], "iters": [q["iters"] for q in rr],
                         "phases_ms_per_iter_last_replicate": {p: v / max(last["iters"], 1) for p, v in last["phases_ms"].items()},
                         "downdate": {"pairs_per_iteration": npairs, "flop_per_iteration": float(npairs) * k * (k + 1),
                                      "ms_per_iteration": 1e3 * mask_s / max(last["iters"], 1), "achieved_tflops": tf,
                                      "frac_of_fp64_mfma_peak": (tf / FP64_MFMA_PEAK_TFLOPS) if tf else None}})
    total = time.perf_counter() - t_all
    ctx.timing_enable(False)
    with_tf = [q for q in per_rank if q["downdate"]["achieved_tflops"]]
    dom = max(with_tf, key=lambda q: q["downdate"]["ms_per_iteration"]) if with_tf else None
    nfits = len(fits)
    out = {
        "metric": "rank-sweep wall seconds (ard_nmf + cross_validate_nmf grid k in {%s}, %d restarts, %d cells x %d genes, %d-iteration "
                  "masked fits)" % (",".join(str(k) for k in ranks), args.replicates, n, m, args.maxit),
        "value": total, "unit": "s", "n_gpus": 1, "steps": nfits, "warmup": 1, "ms_per_step": 1e3 * total / max(nfits, 1),
        "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config 5 on one GPU: synthetic CSC %d genes x %d cells, 1/%d non-zero (nnz %d); grid of %d masked fits "
                               "(c_ard_nmf: cv_tol 1e-4, maxit %d, trace_test_mse %d, test density 1/%d, L1 %g), one resident context; a step = one fit, "
                               "set-up (entry streams per rank, mask lists per seed) included"
                               % (m, n, args.inv_density, nnz, nfits, args.maxit, args.trace, args.inv_density, args.L1),
                   "genes": m, "cells": n, "ranks": ranks, "replicates": args.replicates, "nnz": nnz, "parallelism": "one GPU (replica sweep over N GPUs: SINGLET_REPLICA_GPUS)"},
        "roofline": None if dom is None else {
            "bound": "mfma", "kernel": "mask_gram_list_kernel: per-column Gram downdates of predict_mask, rank %d (the rank with the longest downdate phase)" % dom["k"],
            "achieved": dom["downdate"]["achieved_tflops"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["downdate"]["frac_of_fp64_mfma_peak"], "traffic": None,
            "algorithmic_flop_per_iteration": dom["downdate"]["flop_per_iteration"], "ms_per_iteration": dom["downdate"]["ms_per_iteration"],
            "algorithmic_flop_is": "drawn (cell, gene) pairs listed per cell + per gene, x k (k + 1): the symmetric half of every rank-one downdate, 2 flop per FMA",
            "time_is": "hipEvent time of the SGL_PH_MASK phase (the downdate kernels of both half-iterations) in the last replicate's fit, per iteration",
            "note": "FP64 MFMA and FP64 VALU share one rate on this part and do not overlap (DESIGN.md, Masked path): the fraction also prices the tile padding of k not a multiple of 16"},
        "per_rank": per_rank, "fits": fits, "generate_s": gen_s,
        "grid_is": "the masked fits only, in rank order; mask lists of the %d seeds are built in the first rank's fits and kept" % args.replicates,
    }
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline_ard(args, ranks)
            out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["value"] / out["value"]
        except Exception as e:  # noqa: BLE001 - the baseline is reported, never required
            out["cpu_baseline"] = {"error": repr(e)}
    ctx.close()
    emit(json.dumps(out))


def main():
    args = parse()
    if args.workload == "ard":
        if args.gpus != 1 or "WORLD_SIZE" in os.environ and int(os.environ.get("WORLD_SIZE") or 1) > 1:
            raise SystemExit("bench.py: --workload ard is a one-GPU record (the grid's fits are independent: the replica sweep, not a team)")
        return run_ard(args)
    pl = plan(args, os.environ)
    if pl["form"] == "single-process":
        return run_single_process(args, pl)
    rank, local_rank, world = pl["rank"], pl["local_rank"], pl["world"]
    args.gpus = world

    import torch
    import singlet_amd as sa

    # SGL_BENCH_FORCE_DEVICE: plumbing tests on a 1-GPU box, every rank on one device (RCCL then refuses the team).  Otherwise the
    # launcher's LOCAL_RANK -- unless it isolates the devices per rank, where every process sees its own as index 0
    # (torch.cuda.device_count() reads the count without initialising the runtime)
    local_rank, device_how = pick_device(local_rank, torch.cuda.device_count(), os.environ.get("SGL_BENCH_FORCE_DEVICE"))
    torch.cuda.set_device(local_rank)
    mode = args.comm
    if args.native_comm:
        mode = "native"
    if args.force_allreduce:
        mode = "hook"
    if mode == "auto":
        mode = "native" if world > 1 else "none"
    if world > 1 and mode == "none":
        raise SystemExit("--comm none needs a single rank")
    # Host-side coordination (the 128-byte id, flags, the timing maximum) runs over a gloo group on CPU tensors when
    # the exchange is the library's own RCCL communicator: no second RCCL communicator set in the process.  Only the
    # hook path (--comm hook, or the fallback when the native initialisation fails) creates a torch NCCL group.
    dist, hook_group = None, None
    if world > 1 or mode == "hook":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if mode == "native" or os.environ.get("SGL_BENCH_HOOK_BACKEND") == "gloo":
            # (SGL_BENCH_HOOK_BACKEND=gloo with --comm hook and SGL_BENCH_FORCE_DEVICE=0: a rehearsal of the process-per-GPU form on a
            #  1-GPU box -- every rank its own context on the one device, the hook's all-reduce through gloo; not a scaling point)
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    host_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_host(vals, op):
        """all-reduce of a few host doubles over the coordination group"""
        if dist is None:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=host_dev)
        dist.all_reduce(t, op=op)
        return [float(v) for v in t.cpu()]

    ctx = sa.Context(local_rank)
    # one explicit (non-default) stream shared by the kernels and torch: RCCL orders its collective against
    # torch's CURRENT stream, so the library must launch on exactly that one
    stream = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    from singlet_amd.sharded import shard_by_count, torch_allreduce_hook
    comm_note = None
    if mode == "native":
        # (1) every rank checks that it can bind RCCL at all (no communication) and the ranks agree on it: the
        #     communicator initialisation below is collective and would hang on the ranks that did join;
        # (2) rank 0 makes the RCCL id, the host side broadcasts its 128 bytes, every rank joins with its context;
        # a failure makes ALL ranks take the hook path, and the JSON line says so
        err = None
        if "SGL_RCCL_PATH" not in os.environ:   # the librccl this process has already mapped (torch's), not a second copy
            try:
                with open("/proc/self/maps") as f:
                    libs = sorted({ln.split()[-1] for ln in f if "librccl" in ln and ln.split()[-1].startswith("/")})
                if libs:
                    os.environ["SGL_RCCL_PATH"] = libs[0]
            except OSError:
                pass
        ok, rccl_path = sa.comm_available()
        if not ok:
            err = "rank %d cannot bind RCCL" % rank
        if reduce_host([0.0 if ok else 1.0], dist.ReduceOp.MAX if dist else None)[0] == 0.0:
            my_id = None
            if rank == 0:
                try:
                    my_id = sa.comm_unique_id()
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
            ids = [my_id]
            if dist is not None:
                dist.broadcast_object_list(ids, src=0)      # always: the other ranks are waiting in it
            if ids[0] is None:
                err = err or "rank 0 could not create an RCCL id"
            else:
                try:
                    ctx.comm_init_rank(world, rank, ids[0])
                except Exception as e:  # noqa: BLE001
                    err = repr(e)
        else:
            err = err or "RCCL cannot be bound on another rank"
        failed = reduce_host([1.0 if err else 0.0], dist.ReduceOp.MAX if dist else None)[0] != 0.0
        if failed:
            if world == 1:
                raise SystemExit("native comm failed: %s" % err)
            print("bench.py: native comm failed on some rank (%s): ALL ranks use the torch all-reduce hook" % err,
                  file=sys.stderr, flush=True)
            comm_note = "native init failed (%s)" % (err or "on another rank")
            ctx.close()
            ctx = sa.Context(local_rank)
            ctx.set_stream(stream.cuda_stream)
            mode = "hook"
            hook_group = None if os.environ.get("SGL_BENCH_HOOK_BACKEND") == "gloo" else dist.new_group(backend="nccl")
    if mode == "hook":
        ctx.set_allreduce(torch_allreduce_hook(dist, torch.device("cuda", local_rank), group=hook_group))

    # contiguous equal-count cell blocks: the synthetic columns are i.i.d., so equal counts are equal
    # non-zeros to within 0.1 %
    lo, n_local = shard_by_count(args.cells, world, rank)
    t0 = time.perf_counter()
    ctx.synth(args.genes, n_local, args.inv_density, cell_offset=lo, ncells_total=args.cells,
              skew=(0.5, 1.5) if args.data == "skewed" else None)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t0
    m, n_loc, nnz_local = ctx.dims()
    ctx.fit_init(args.k, None)

    def step():
        return ctx.nmf_iterate(args.L1, args.L1, 0.0, 0.0)

    warm_tols = [step() for _ in range(args.warmup)]
    comm_info = ctx.comm_info()
    if dist is not None and warm_tols:
        # W is replicated: tol = cor(w, w_prev) must come out bit-identical on every rank.  One 8-byte gather.
        import struct
        mine = struct.pack("<d", warm_tols[-1])
        got = [None] * world
        dist.all_gather_object(got, (mine, comm_info["nranks"]))
        if any(g[0] != got[0][0] for g in got) or any(g[1] != got[0][1] for g in got):
            raise SystemExit("bench.py: ranks disagree after warm-up (tol bits / communicator sizes): %r"
                             % [(struct.unpack("<d", g[0])[0], g[1]) for g in got])
    if mode == "native" and comm_info["nranks"] != world:
        raise SystemExit("bench.py: the library's communicator spans %d ranks, launched with %d" % (comm_info["nranks"], world))
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
    phases_all, rank_info = [phases], [{"device": int(local_rank), "device_from": device_how, "cells": int(n_loc), "nnz": int(nnz_local)}]
    if dist is not None:
        elapsed = reduce_host([elapsed], dist.ReduceOp.MAX)[0]
        nnz_total = int(reduce_host([float(nnz_local)], dist.ReduceOp.SUM)[0])
        got = [None] * world
        dist.all_gather_object(got, (phases, rank_info[0]))
        phases_all, rank_info = [g[0] for g in got], [g[1] for g in got]

    if rank == 0:
        mb = (m + world - 1) // world if (world > 1 and mode == "native") else m
        w_start = ctx.get_factors(h=False)[0] if (world == 1 and not args.no_cpu_baseline) else None   # for the CPU baseline's window
        out = report(args, {"world": world, "w_start": w_start, "elapsed": elapsed, "tols": tols, "dims": (m, n_loc, nnz_local), "nnz_total": nnz_total,
                            "phases": phases, "phases_all": phases_all, "rank_info": rank_info if world > 1 else None,
                            "sweeps": sweeps, "layout": layout, "gen_s": gen_s, "w_cols_rank0": min(mb, m),
                            "comm": {"mode": mode, "note": comm_note, "rccl_nranks": comm_info["nranks"] if comm_info["is_rccl"] else None,
                                     "rccl_path": comm_info["path"] or None,
                                     "host_coordination": None if dist is None else dist.get_backend(),
                                     "tol_bit_identical_across_ranks": None if dist is None else True}})
        line = json.dumps(out)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
