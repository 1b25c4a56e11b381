"""Cell-sharded ALS: host-side pieces shared by bench.py, the multi-GPU driver and
the gloo tests.

The reference has no distribution at all; its closest relative is the
column-chunked list variant that carries a running column `offset`
(src/singlet.cpp:384-402, 469-503).  Here every rank owns a contiguous block of
cells (columns of A) and the matching rows of H; W, its Gram and d are
replicated.  Per ALS iteration two sums over ranks are needed:

  scale(h, d)        k row sums                         (src/singlet.cpp:651)
  predict(At, h, w)  [k x genes right-hand sides | k x k Gram of H]   (:654)

A "context" is anything with the step API of include/singlet_hip.h section 2
(singlet_amd.Context on a GPU; tests/oracle_backend.py on the CPU); the sums run
through the all-reduce hook the context was given.
"""
import numpy as np


def shard_by_count(ncols, world, rank):
    """Contiguous equal-count blocks (i.i.d. synthetic columns: equal counts = equal work)."""
    base, rem = divmod(int(ncols), int(world))
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


def shard_by_nnz(p, world):
    """Contiguous column blocks with (nearly) equal non-zero counts, from the column pointer
    array of a dgCMatrix.  Returns `world + 1` boundaries b, rank r owns columns [b[r], b[r+1])."""
    p = np.asarray(p, dtype=np.int64)
    ncol = p.shape[0] - 1
    total = int(p[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r // world
        c = int(np.searchsorted(p, target, side="left"))
        c = min(max(c, bounds[-1]), ncol)
        bounds.append(c)
    bounds.append(ncol)
    return bounds


def nmf_loop(ctx, tol, maxit, L1_w, L1_h, L2_w, L2_h, log=None):
    """c_nmf_base's loop (src/singlet.cpp:647-664) over the step API.  Every rank runs the same
    loop; tol_ comes out identical on all ranks because W is replicated bit-for-bit."""
    tol_ = 1.0
    it = 0
    tols = []
    while it < maxit and tol_ > tol:
        ctx.step_begin()
        ctx.step_h(L1_h, L2_h)
        ctx.step_scale_h()
        ctx.step_w(L1_w, L2_w)
        tol_ = ctx.step_scale_w()
        tols.append(tol_)
        it += 1
        if log is not None:
            log(it, tol_)
    return it, np.array(tols)


class DevView:
    """__cuda_array_interface__ view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def torch_allreduce_hook(dist, device, group=None):
    """All-reduce hook for singlet_amd.Context.set_allreduce: sums `count` doubles at a device
    pointer over the default process group (RCCL when the backend is "nccl").  The context must
    launch on torch's current stream (Context.set_stream) so the collective is ordered after the
    kernels that produced the buffer and before the ones that consume it, with no host sync."""
    import torch
    views = {}

    def hook(ptr, count):
        t = views.get((ptr, count))
        if t is None:
            t = torch.as_tensor(DevView(ptr, count), device=device)
            views[(ptr, count)] = t
        dist.all_reduce(t, group=group)

    return hook
