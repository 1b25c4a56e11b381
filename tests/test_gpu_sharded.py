"""The library's own cell-sharded path (sgl_set_allreduce: global row sums, all-reduced right-hand
sides + Gram, global per-gene counts, global hash indices) on ONE GPU: two contexts = two shards in
one process, driven by two threads, with an all-reduce hook that sums the two device buffers in a
fixed order.  Must reproduce the unsharded fit of the same matrix and the oracle."""
import threading

import numpy as np
import pytest

from conftest import rel_fro, to_dgc

pytestmark = pytest.mark.gpu


class TwoShardSum:
    """all-reduce over two in-process shards: every call is a rendezvous of both threads."""

    def __init__(self):
        import torch
        self.torch = torch
        self.bar = threading.Barrier(2, timeout=120)
        self.slots = [None, None]

    def hook(self, rank):
        from singlet_amd.sharded import DevView

        def fn(ptr, count):
            t = self.torch.as_tensor(DevView(ptr, count), device="cuda")
            self.slots[rank] = t.cpu()          # synchronises with the kernels that produced it
            self.bar.wait()
            total = self.slots[0] + self.slots[1]  # fixed order on both ranks -> bit-identical sums
            self.bar.wait()
            t.copy_(total)
            self.torch.cuda.synchronize()
        return fn


@pytest.mark.timeout(300)
@pytest.mark.parametrize("m,n,k,split", [(300, 1000, 8, 430), (257, 700, 30, 1), (500, 640, 50, 320)])
def test_two_shards_one_gpu_match_unsharded(sa, ora, m, n, k, split):
    torch = pytest.importorskip("torch")
    torch.cuda.init()                      # in the main thread, before the two shard threads use it
    assert torch.cuda.device_count() >= 1
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    maxit, L1 = 4, 0.01
    ref = ora.c_nmf(A, At, 0.0, maxit, L1, L1, 0.0, 0.0, 0, w0)
    one = sa.c_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, maxit, False, L1, L1, 0.0, 0.0, 0, w0.T)

    bounds = [0, split, n]
    red = TwoShardSum()
    out, errs = [None, None], []

    def worker(r):
        try:
            from singlet_amd.sharded import nmf_loop
            lo, hi = bounds[r], bounds[r + 1]
            s, e = A.p[lo], A.p[hi]
            Ash = ora.CSC(A.x[s:e], A.i[s:e], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)
            c = sa.Context(0)
            try:
                c.upload(to_dgc(sa, Ash), None, cell_offset=lo, ncells_total=n)   # transpose built on the device
                c.set_allreduce(red.hook(r))
                c.fit_init(k, w0)
                it, tols = nmf_loop(c, 0.0, maxit, L1, L1, 0.0, 0.0)
                W, d, H = c.get_factors()
                out[r] = (W, d, H, tols)
            finally:
                c.close()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            red.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(280)
    assert not errs, errs
    (W0, d0, H0, t0), (W1, d1, H1, t1) = out
    assert np.array_equal(W0, W1) and np.array_equal(d0, d1) and np.array_equal(t0, t1)   # replicated bit-for-bit
    H = np.vstack([H0, H1])
    assert rel_fro(W0, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9 and rel_fro(d0, ref["d"]) < 1e-9
    assert rel_fro(W0, one["w"].T) < 1e-11 and rel_fro(H, one["h"].T) < 1e-11


@pytest.mark.timeout(300)
def test_two_shards_weight_by_split_uses_global_group_sums(sa, ora):
    """weight_by_split needs the group totals over ALL cells: each shard contributes its part through
    the all-reduce hook, and the rescaled shard must equal the slice of the unsharded result."""
    torch = pytest.importorskip("torch")
    torch.cuda.init()
    m, n, split = 200, 500, 170
    A = ora.synth_csc(m, n, 10)
    sb = np.random.default_rng(9).integers(0, 3, n).astype(np.int32)
    ref = ora.weight_by_split(A, sb, 3)
    bounds = [0, split, n]
    red = TwoShardSum()
    out, errs = [None, None], []

    def worker(r):
        try:
            lo, hi = bounds[r], bounds[r + 1]
            s, e = A.p[lo], A.p[hi]
            Ash = ora.CSC(A.x[s:e], A.i[s:e], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)
            c = sa.Context(0)
            try:
                c.upload(to_dgc(sa, Ash), None, cell_offset=lo, ncells_total=n)
                c.set_allreduce(red.hook(r))
                c.weight_by_split(sb[lo:hi], 3)
                out[r] = (c.download(0)[0], s, e)
            finally:
                c.close()
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)
            red.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(280)
    assert not errs, errs
    for x, s, e in out:
        assert rel_fro(x, ref.x[s:e]) < 1e-14
