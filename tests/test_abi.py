"""The drop-in boundary on the CPU: libsinglet_hip.so loads, exports every symbol
include/singlet_hip.h declares, and fails loudly without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "singlet_hip.h")).read()
    return sorted(set(re.findall(r"SGL_API\s+[\w\s\*]+?\b(sgl_[a-z0-9_]+)\s*\(", h)))


def test_header_declares_the_three_reference_entry_points():
    names = _declared()
    for n in ("sgl_c_nmf", "sgl_c_ard_nmf", "sgl_c_project_model", "sgl_last_error"):
        assert n in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol(sa):
    from singlet_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in _declared():
        assert hasattr(L, n), "libsinglet_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == _declared(), "python binding and header disagree"
    assert _lib.load().sgl_abi_version() == 2


def test_no_cpu_fallback(sa):
    from singlet_amd import _lib
    if _lib.load().sgl_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sa.SingletHipError) as e:
        sa.Context(0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)
    import numpy as np
    A = sa.dgCMatrix([1.0, 2.0], [0, 1], [0, 1, 2], (2, 2))
    with pytest.raises(sa.SingletHipError):
        sa.c_nmf(A, None, 0.0, 1, False, 0, 0, 0, 0, 0, np.ones((1, 2)))
    with pytest.raises(sa.SingletHipError):
        sa.c_project_model(A, np.ones((2, 1)), 0, 0, 0)


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "singlet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"^\s*(from|import)\s+oracle", r"libsinglet_oracle", r"\bora_[a-z]", r"oracle[./]oracle",
                            r"np_transcription"):
                    assert not re.search(pat, src, flags=re.M), (os.path.join(dirpath, f), pat)


def test_dgCMatrix_validation(sa):
    with pytest.raises(ValueError):
        sa.dgCMatrix([1.0], [0], [0, 1, 1], (2, 3))
    with pytest.raises(ValueError):
        sa.dgCMatrix([1.0, 2.0], [0], [0, 1], (2, 1))
    M = sa.dgCMatrix.from_dense([[0, 1.5], [2.0, 0]])
    assert M.nnz == 2 and M.Dim == (2, 2) and list(M.i) == [1, 0]
    assert M.col_slice(1, 2).nnz == 1


def test_r_shim_matches_the_abi():
    """singlet_amd/r/singlet_hip_shim.c (the .Call layer an R user loads) cannot be built here (no R);
    syntax-check its calls into include/singlet_hip.h against prototype-only R API declarations."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-cast-function-type",
                        "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "tests", "r_api_stub"),
                        os.path.join(root, "singlet_amd", "r", "singlet_hip_shim.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_backend_r_rebinds_every_symbol_the_shim_registers():
    """singlet_amd/r/backend.R must rebind one R wrapper per `.Call` symbol of the shim, with the arity the shim registers
    (and the reference's R/RcppExports.R declares): a symbol registered but not rebound would leave that wrapper on
    the CPU path silently (round-3 finding: weight_by_split)."""
    shim = open(os.path.join(ROOT, "singlet_amd", "r", "singlet_hip_shim.c")).read()
    backend = open(os.path.join(ROOT, "singlet_amd", "r", "backend.R")).read()
    registered = dict((n, int(a)) for n, a in re.findall(r'\{"(_singlet_\w+)",\s*\(DL_FUNC\)&\w+,\s*(\d+)\}', shim))
    assert len(registered) >= 10
    for sym, arity in registered.items():
        m = re.search(r'rebind\("%s",\s*function\(([^)]*)\)\s*\.Call\(dll\[\["%s"\]\],([^)]*)\)' % (sym[len("_singlet_"):], sym), backend, flags=re.S)
        assert m, "backend.R does not rebind %s" % sym
        formals = [a.strip() for a in m.group(1).split(",")]
        passed = [a.strip() for a in m.group(2).split(",")]
        assert len(formals) == arity and passed == formals, (sym, arity, formals, passed)


def test_reference_build_products_stay_out_of_history():
    """oracle/_ref/ (the reference's `rng` class cut out of the reference tree at build time, and its build) is a build
    product: git-ignored, never committed; only the recipe (make_ref.sh), the shim and the fixture it generated are."""
    import subprocess
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/_ref"], capture_output=True, text=True)
    if tracked.returncode != 0:
        pytest.skip("not a git checkout")
    assert tracked.stdout.strip() == ""
    ign = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in ign
    # the built librng_ref.so travels to the GPU box like the other built libraries; the extracted reference text (*.inc) need not
    # (round-5 verdict): .gpurunignore may name oracle/_ref/*.inc, never the directory or the library
    gi = open(os.path.join(ROOT, ".gpurunignore")).read().split() if os.path.exists(os.path.join(ROOT, ".gpurunignore")) else []
    assert all(ln == "oracle/_ref/*.inc" for ln in gi if "oracle/_ref" in ln), gi
    for f in ("oracle/make_ref.sh", "oracle/ref_rng_shim.cpp", "tests/golden/make_rng_ref.py", "tests/golden/rng_ref.npz"):
        assert os.path.exists(os.path.join(ROOT, f)), f
    # the shim holds no reference text: it includes the extracted class
    shim = open(os.path.join(ROOT, "oracle", "ref_rng_shim.cpp")).read()
    assert '#include "_ref/rng_class.inc"' in shim and "class rng" not in shim
