"""The library's own cell-sharded path (sgl_set_allreduce: global row sums, all-reduced right-hand
sides + Gram, global per-gene counts, global hash indices, global group sums) on ONE GPU: two
processes = two shards sharing the device, with an all-reduce hook that exchanges the buffers through
a pipe and sums them in a fixed order.  Must reproduce the unsharded fit and the oracle."""
import ctypes
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from conftest import rel_fro, to_dgc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class PipeSum:
    """all-reduce over two single-GPU processes: device -> host, swap over a pipe, sum in rank order."""

    def __init__(self, rank, conn):
        self.rank, self.conn = rank, conn
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def __call__(self, ptr, count):
        mine = np.empty(count)
        assert self.hip.hipMemcpy(mine.ctypes.data, ptr, 8 * count, 2) == 0     # device -> host (synchronous)
        if self.rank == 0:   # ordered exchange: two simultaneous sends larger than the pipe buffer would deadlock
            self.conn.send_bytes(mine.tobytes())
            other = np.frombuffer(self.conn.recv_bytes(), dtype=np.float64)
        else:
            other = np.frombuffer(self.conn.recv_bytes(), dtype=np.float64)
            self.conn.send_bytes(mine.tobytes())
        total = (mine + other) if self.rank == 0 else (other + mine)             # same order on both ranks
        assert self.hip.hipMemcpy(ptr, total.ctypes.data, 8 * count, 1) == 0    # host -> device


def _shard(A, ora, lo, hi):
    s, e = A.p[lo], A.p[hi]
    return ora.CSC(A.x[s:e], A.i[s:e], A.p[lo:hi + 1] - A.p[lo], A.nrow, hi - lo), s, e


def _worker(rank, conn, q, job):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import singlet_amd as sa
        from oracle import oracle as ora
        from singlet_amd.sharded import nmf_loop
        kind, m, n, k, split = job
        A = ora.synth_csc(m, n, 20 if kind.startswith("nmf") else 10)
        if kind == "nmf_late_hook":   # gene 5 only in the second shard, gene 9 nowhere
            keep = ~(((A.i == 5) & (np.repeat(np.arange(n), np.diff(A.p)) < split)) | (A.i == 9))
            cnt = np.bincount(np.repeat(np.arange(n), np.diff(A.p))[keep], minlength=n)
            A = ora.CSC(A.x[keep], A.i[keep], np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), m, n)
        lo, hi = (0, split) if rank == 0 else (split, n)
        Ash, s, e = _shard(A, ora, lo, hi)
        c = sa.Context(0)
        try:
            c.upload(sa.dgCMatrix(Ash.x, Ash.i, Ash.p, (Ash.nrow, Ash.ncol)), None, cell_offset=lo, ncells_total=n)
            if kind != "nmf_late_hook":
                c.set_allreduce(PipeSum(rank, conn))
            if kind == "nmf_late_hook":
                # the hook installed AFTER sgl_fit_init: the global gene counts must be rebuilt through it on the
                # next W-update (a gene without entries in one shard would otherwise keep a stale column there)
                c.fit_init(k, ora.synth_winit(k, m))
                c.set_allreduce(PipeSum(rank, conn))
                it, tols = nmf_loop(c, 0.0, 4, 0.01, 0.01, 0.0, 0.0)
                W, d, H = c.get_factors()
                q.put((rank, "ok", (W, d, H, tols)))
            elif kind == "nmf":
                c.fit_init(k, ora.synth_winit(k, m))
                it, tols = nmf_loop(c, 0.0, 4, 0.01, 0.01, 0.0, 0.0)
                W, d, H = c.get_factors()
                q.put((rank, "ok", (W, d, H, tols)))
            else:
                sb = np.random.default_rng(9).integers(0, 3, n).astype(np.int32)
                c.weight_by_split(sb[lo:hi], 3)
                q.put((rank, "ok", (c.download(0)[0], s, e)))
        finally:
            c.close()
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, "err", traceback.format_exc() + repr(exc)))


def _run_two(job):
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, conn, q, job)) for r, conn in ((0, a), (1, b))]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, status, payload = q.get(timeout=90)
        assert status == "ok", payload
        res[rank] = payload
    for p in procs:
        p.join(60)
    return res[0], res[1]


@pytest.mark.timeout(120)
@pytest.mark.parametrize("m,n,k,split", [(300, 1000, 8, 430), (257, 700, 30, 1), (500, 640, 50, 320)])
def test_two_shards_one_gpu_match_unsharded(sa, ora, m, n, k, split):
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf(A, At, 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, w0)
    one = sa.c_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    (W0, d0, H0, t0), (W1, d1, H1, t1) = _run_two(("nmf", m, n, k, split))
    assert np.array_equal(W0, W1) and np.array_equal(d0, d1) and np.array_equal(t0, t1)   # replicated bit-for-bit
    H = np.vstack([H0, H1])
    assert rel_fro(W0, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9 and rel_fro(d0, ref["d"]) < 1e-9
    assert rel_fro(W0, one["w"].T) < 1e-11 and rel_fro(H, one["h"].T) < 1e-11


@pytest.mark.timeout(120)
def test_hook_installed_after_fit_init_still_uses_global_gene_counts(sa, ora):
    m, n, k, split = 120, 500, 6, 260
    A = ora.synth_csc(m, n, 20)
    col = np.repeat(np.arange(n), np.diff(A.p))
    keep = ~(((A.i == 5) & (col < split)) | (A.i == 9))
    cnt = np.bincount(col[keep], minlength=n)
    A2 = ora.CSC(A.x[keep], A.i[keep], np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), m, n)
    ref = ora.c_nmf(A2, A2.t(), 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, ora.synth_winit(k, m))
    (W0, d0, H0, t0), (W1, d1, H1, t1) = _run_two(("nmf_late_hook", m, n, k, split))
    assert np.array_equal(W0, W1) and np.array_equal(d0, d1)          # no rank keeps a stale column
    assert rel_fro(W0, ref["w"]) < 1e-9 and rel_fro(np.vstack([H0, H1]), ref["h"]) < 1e-9


@pytest.mark.timeout(120)
def test_two_shards_weight_by_split_uses_global_group_sums(sa, ora):
    """weight_by_split needs the group totals over ALL cells: each shard contributes its part through
    the all-reduce hook, and the rescaled shard must equal the slice of the unsharded result."""
    m, n, split = 200, 500, 170
    A = ora.synth_csc(m, n, 10)
    sb = np.random.default_rng(9).integers(0, 3, n).astype(np.int32)
    ref = ora.weight_by_split(A, sb, 3)
    for x, s, e in _run_two(("wbs", m, n, 0, split)):
        assert rel_fro(x, ref.x[s:e]) < 1e-14
