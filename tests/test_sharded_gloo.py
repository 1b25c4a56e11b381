"""The N > 1 path on CPU: two gloo ranks, cells sharded, the two per-iteration sums through
torch.distributed.  Checks (a) the host logic shared with the GPU driver (sharding, loop,
all-reduce plumbing) and (b) that the sharded decomposition reproduces the unsharded loop."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as ora
    from oracle_backend import OracleShardContext
    from singlet_amd.sharded import nmf_loop, shard_by_nnz

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, n, k = 120, 301, 7
    A = ora.synth_csc(m, n, 10)
    # an all-zero gene and an empty cell exercise the skip rules across shards
    bounds = shard_by_nnz(A.p, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    s, e = A.p[lo], A.p[hi]
    Ash = ora.CSC(A.x[s:e], A.i[s:e], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)
    ctx = OracleShardContext(ora)
    ctx.upload(Ash, Ash.t(), cell_offset=lo, ncells_total=n)

    def allreduce(arr):
        t = torch.from_numpy(arr)
        dist.all_reduce(t)

    ctx.set_allreduce(allreduce)
    w0 = ora.synth_winit(k, m)
    ctx.fit_init(k, w0)
    it, tols = nmf_loop(ctx, 0.0, 4, 0.01, 0.01, 0.0, 0.0)
    W, d, H = ctx.get_factors()
    q.put((rank, lo, hi, W, d, H, tols))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_loop_matches_unsharded():
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    m, n, k = 120, 301, 7
    A = ora.synth_csc(m, n, 10)
    ref = ora.c_nmf(A, A.t(), 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, ora.synth_winit(k, m))
    (_, lo0, hi0, W0, d0, H0, t0), (_, lo1, hi1, W1, d1, H1, t1) = res
    assert (lo0, hi1) == (0, n) and hi0 == lo1
    assert np.array_equal(W0, W1) and np.array_equal(d0, d1) and np.array_equal(t0, t1)  # replicated bit-for-bit
    H = np.vstack([H0, H1])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(W0, ref["w"]) < 1e-11 and rel(H, ref["h"]) < 1e-11 and rel(d0, ref["d"]) < 1e-12
    assert np.allclose(t0, ref["tol"], rtol=1e-9)


def _team_worker(rank, world, port, q, m=121):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as ora
    from oracle_backend import OracleTeamRank
    from singlet_amd.sharded import shard_by_nnz

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, k = 301, 7          # 121 genes over 2 ranks / 123 over 4: unequal last block (the padding path)
    A = ora.synth_csc(m, n, 10)
    bounds = shard_by_nnz(A.p, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    s, e = A.p[lo], A.p[hi]
    Ash = ora.CSC(A.x[s:e], A.i[s:e], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)

    def all_reduce(arr):
        dist.all_reduce(torch.from_numpy(arr))

    def reduce_scatter(full):      # rows [rank * mb, (rank + 1) * mb) of the sum over the ranks
        blocks = [torch.from_numpy(np.ascontiguousarray(b)) for b in np.split(full, world)]
        out = torch.empty_like(blocks[rank])
        try:
            dist.reduce_scatter(out, blocks)
        except (RuntimeError, NotImplementedError):   # a gloo build without reduce_scatter: same result by parts
            for r, b in enumerate(blocks):
                dist.reduce(b, dst=r)
            out = blocks[rank]
        return out.numpy()

    def all_gather(block):
        outs = [torch.empty_like(torch.from_numpy(block)) for _ in range(world)]
        dist.all_gather(outs, torch.from_numpy(np.ascontiguousarray(block)))
        return np.vstack([o.numpy() for o in outs])

    t = OracleTeamRank(ora, rank, world, reduce_scatter, all_reduce, all_gather)
    t.upload(Ash, Ash.t())
    t.fit_init(k, ora.synth_winit(k, m))
    tols = [t.iterate(0.01, 0.01, 0.0, 0.0) for _ in range(4)]
    W, d, H = t.get_factors()
    q.put((rank, lo, hi, W.copy(), d, H, np.array(tols)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_team_exchange_pattern_matches_unsharded():
    """The exchange pattern of the library's native team (multi.hip): unscaled partials, ONE grouped exchange
    (reduce-scatter by gene blocks + all-reduce of [Gram | row sums]), gene blocks solved per rank, all-gather.
    Two gloo ranks with the oracle's operators reproduce the unsharded loop to rounding."""
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_team_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    m, n, k = 121, 301, 7
    A = ora.synth_csc(m, n, 10)
    ref = ora.c_nmf(A, A.t(), 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, ora.synth_winit(k, m))
    (_, lo0, hi0, W0, d0, H0, t0), (_, lo1, hi1, W1, d1, H1, t1) = res
    assert (lo0, hi1) == (0, n) and hi0 == lo1
    assert np.array_equal(W0, W1) and np.array_equal(d0, d1) and np.array_equal(t0, t1)  # replicated bit-for-bit
    H = np.vstack([H0, H1])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(W0, ref["w"]) < 1e-11 and rel(H, ref["h"]) < 1e-11 and rel(d0, ref["d"]) < 1e-12
    assert np.allclose(t0, ref["tol"], rtol=1e-8)


@pytest.mark.timeout(300)
def test_four_rank_team_exchange_uneven_last_gene_block():
    """World size 4 with 123 genes: gene blocks of 31, the last one short (30 real genes + 1 pad row) -- the
    reduce-scatter / all-gather units of multi.hip are padded to equal blocks and the pad must never leak into W."""
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    world, m, n, k = 4, 123, 301, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_team_worker, args=(r, world, port, q, m)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    A = ora.synth_csc(m, n, 10)
    ref = ora.c_nmf(A, A.t(), 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, ora.synth_winit(k, m))
    assert res[0][1] == 0 and res[-1][2] == n and all(res[r][2] == res[r + 1][1] for r in range(world - 1))
    for r in range(1, world):   # replicated bit-for-bit
        assert np.array_equal(res[0][3], res[r][3]) and np.array_equal(res[0][4], res[r][4]) and np.array_equal(res[0][6], res[r][6])
    H = np.vstack([t[5] for t in res])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert res[0][3].shape == (m, k)
    assert rel(res[0][3], ref["w"]) < 1e-11 and rel(H, ref["h"]) < 1e-11 and rel(res[0][4], ref["d"]) < 1e-12
    assert np.allclose(res[0][6], ref["tol"], rtol=1e-8)


def test_cell_split_keeps_every_block_non_empty(sa):
    """sgl_split_cells_by_nnz (the split of sgl_multi_upload_csc): a heavy LAST cell used to push the last boundary
    past ncol (round-2 advice: n = 2, p = [0, 1, 2, 10] gave [0, 3, 4])."""
    assert list(sa.split_cells_by_nnz([0, 1, 2, 10], 2)) == [0, 2, 3]
    rng = np.random.default_rng(5)
    for n in range(2, 9):
        for trial in range(40):
            ncol = int(rng.integers(n, 3 * n + 2))
            cnt = rng.integers(0, 4, size=ncol)
            which = trial % 4
            if which == 0:
                cnt[-1] = 1000       # heavy last cell
            elif which == 1:
                cnt[0] = 1000        # heavy first cell
            elif which == 2:
                cnt[:] = 0           # an all-empty matrix
            p = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            lo = sa.split_cells_by_nnz(p, n)
            assert lo[0] == 0 and lo[-1] == ncol and np.all(np.diff(lo) >= 1), (n, p.tolist(), lo.tolist())
    with pytest.raises(sa.SingletHipError):
        sa.split_cells_by_nnz([0, 1, 2], 3)


def test_shard_helpers():
    from singlet_amd.sharded import shard_by_count, shard_by_nnz
    assert [shard_by_count(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    p = np.array([0, 10, 10, 30, 31, 60, 100])
    b = shard_by_nnz(p, 2)
    assert b[0] == 0 and b[-1] == 6 and 0 < b[1] < 6
    left = p[b[1]] - p[0]
    assert abs(left - 50) <= 30
    b8 = shard_by_nnz(p, 8)
    assert len(b8) == 9 and all(b8[i] <= b8[i + 1] for i in range(8))
