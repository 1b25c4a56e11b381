import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import singlet_amd._lib as L
        return L.load().sgl_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ora():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sa():
    """The product package.  On a fresh checkout the shared library is not there yet (build products are
    not in git): build it once (hipcc cross-compiles without a GPU) instead of failing every test."""
    import singlet_amd
    from singlet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return singlet_amd


@pytest.fixture(scope="session")
def ctx(sa):
    """One device context shared by the GPU tests; fails loudly (no skip) when the
    HIP library or the device is missing, because -m gpu is only run on a GPU box."""
    c = sa.Context(0)
    yield c
    c.close()


def to_dgc(sa, A):
    return sa.dgCMatrix(A.x, A.i, A.p, (A.nrow, A.ncol))


def rel_fro(a, b):
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / (nb if nb > 0 else 1.0))


def same_zero_pattern(a, b, eps=1e-12):
    """Zero / non-zero pattern identical except where |value| < eps * max (SURVEY 8d parity gate)."""
    a, b = np.asarray(a), np.asarray(b)
    thr = eps * max(np.abs(b).max(), 1e-300)
    diff = (a == 0) != (b == 0)
    return bool(np.all(~diff | (np.abs(a) < thr) & (np.abs(b) < thr)))


@pytest.fixture(autouse=True)
def _no_pending_hip_error(request):
    """After every GPU test the HIP runtime must be left without a sticky error: a pending error
    makes the next library that probes the runtime in this process (torch, RCCL) see 'no GPU'."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return
    err = hip.hipPeekAtLastError()
    assert err == 0, "test left HIP error %d pending" % err
