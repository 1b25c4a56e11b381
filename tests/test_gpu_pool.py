"""The library's device-memory pool (singlet_amd/csrc/pool.hip, round 6): blocks of 64 MB and more freed by a context serve the
next request instead of going back to the driver -- the one-shot calls behind R's run_nmf / ard_nmf (R/run_nmf.R:39-59,
R/ard_nmf.R:95-160) create and destroy a context per call, and a hipMalloc right after such a free takes seconds at config 3."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cached(sa):
    out = C.c_int64()
    assert sa._lib.load().sgl_pool_info(C.byref(out)) == 0
    return out.value


def test_blocks_freed_by_a_one_shot_call_serve_the_next_one(sa):
    L = sa._lib.load()
    L.sgl_cache_release()
    assert _cached(sa) == 0
    genes, cells, k = 5000, 60000, 12
    with sa.Context(0) as c:                       # the host matrix: generated on the device, downloaded into pageable arrays
        c.synth(genes, cells, 20)
        c.fit_init(k, None)
        w0, _, _ = c.get_factors(h=False)
        x, i, p = c.download(0)
    A = sa.dgCMatrix(x, i, p.astype(np.int32), (genes, cells))
    assert A.x.nbytes > (64 << 20)                 # large enough to be pooled
    kept_by_the_generator = _cached(sa)
    assert kept_by_the_generator >= A.x.nbytes     # the context above is gone, its large blocks are not
    r1 = sa.c_nmf(A, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    after_1 = _cached(sa)
    r2 = sa.c_nmf(A, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    after_2 = _cached(sa)
    # a block handed out again holds whatever its last owner left in it: the results must not care
    for key in ("w", "d", "h", "tol"):
        assert np.array_equal(r1[key], r2[key]), key
    assert after_1 >= A.x.nbytes and after_2 == after_1        # the second call allocated nothing new: same blocks, same sizes
    # a masked fit on the same matrix reuses them too and adds its own workspace
    r3 = sa.c_ard_nmf(A, None, 0.0, 2, False, 0.01, 0.0, 0, w0.T, 7, 20, 1e9, 1)
    assert np.all(np.isfinite(r3["w"])) and _cached(sa) >= after_2
    assert L.sgl_cache_release() == 0
    assert _cached(sa) == 0


def test_pool_counts_as_free_memory_and_is_given_back_on_demand(sa):
    """The budgets derived from the free memory (mask lists, the chunk of per-column Grams) see cached blocks as free, and an
    allocation the driver cannot serve empties the cache before it fails."""
    L = sa._lib.load()
    L.sgl_cache_release()
    with sa.Context(0) as c:
        c.synth(4000, 80000, 20)                   # 16 M non-zeros: x 128 MB
    held = _cached(sa)
    assert held > (64 << 20)
    with sa.Context(0) as c:                       # a different shape: nothing fits, the old blocks stay cached beside the new ones
        c.synth(9000, 30000, 20)
        c.fit_init(8, None)
        assert np.isfinite(c.nmf_iterate(0.01, 0.01, 0.0, 0.0))
    assert _cached(sa) >= held
    L.sgl_cache_release()
    assert _cached(sa) == 0


def test_call_times_split_the_one_shot_call(sa):
    """sgl_call_times_get: the wall-clock split of the calling thread's last one-shot call (what scripts/one_shot_rate.py reports
    at configs 2 and 3): host -> device bytes = the dgCMatrix slots, the parts add up to the whole, and with SINGLET_HIP_CACHE=1 the
    second call on the same host slots uploads nothing."""
    import os
    L = sa._lib.load()
    genes, cells, k = 3000, 20000, 8
    with sa.Context(0) as c:
        c.synth(genes, cells, 20)
        c.fit_init(k, None)
        w0, _, _ = c.get_factors(h=False)
        x, i, p = c.download(0)
    A = sa.dgCMatrix(x, i, p.astype(np.int32), (genes, cells))
    os.environ.pop("SINGLET_HIP_CACHE", None)
    r = sa.c_nmf(A, None, 0.0, 5, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    t = sa.call_times()
    assert r["iter"] == 5 and t["cached"] == 0.0
    assert t["h2d_bytes"] == 12.0 * A.nnz + 4.0 * (cells + 1)                       # x (8 B) + i (4 B) per non-zero + p; t(A) is built on the device
    parts = t["h2d_s"] + t["validate_s"] + t["transpose_s"] + t["fit_init_s"] + t["iterate_s"] + t["d2h_s"]
    assert 0.0 < parts <= t["total_s"] * 1.0001 and t["iterate_s"] > 0.0 and t["transpose_s"] > 0.0
    os.environ["SINGLET_HIP_CACHE"] = "1"
    try:
        sa.c_nmf(A, None, 0.0, 2, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
        assert sa.call_times()["cached"] == 0.0 and sa.call_times()["h2d_bytes"] > 0    # this call uploaded and filled the cache
        r2 = sa.c_nmf(A, None, 0.0, 5, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
        t2 = sa.call_times()
        assert t2["cached"] == 1.0 and t2["h2d_bytes"] == 0.0 and t2["transpose_s"] == 0.0
        for key in ("w", "d", "h"):
            assert np.array_equal(r[key], r2[key]), key
    finally:
        os.environ.pop("SINGLET_HIP_CACHE", None)
        L.sgl_cache_release()
