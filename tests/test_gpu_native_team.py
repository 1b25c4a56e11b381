"""The library's native multi-GPU path (include/singlet_hip.h section 2b, multi.hip): cells sharded over
the ranks of a team, ONE grouped collective (reduce-scatter of the W-side right-hand sides by gene
blocks + all-reduce of [Gram | row sums] of the unscaled h) and one all-gather of the w blocks per
iteration.  On a 1-GPU box the team logic runs with the ranks sharing the device (exchange by a HIP
kernel instead of RCCL, which refuses duplicate devices) and RCCL itself runs as a team of one."""
import os

import numpy as np
import pytest

from conftest import rel_fro, same_zero_pattern, to_dgc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,ranks", [(300, 1000, 8, 2), (257, 700, 30, 3), (500, 640, 50, 2), (130, 900, 70, 4), (96, 400, 5, 7), (210, 520, 100, 2),
                                          (230, 500, 120, 3), (300, 1000, 8, 8), (263, 1100, 50, 8), (150, 640, 70, 8)])
def test_team_on_one_device_matches_the_oracle_and_the_single_shard(sa, ora, m, n, k, ranks):
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf(A, At, 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, w0)
    one = sa.c_nmf(to_dgc(sa, A), None, 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    with sa.Multi([0] * ranks) as M:
        M.upload(to_dgc(sa, A))
        M.fit_init(k, w0)
        it, tols = M.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0)
        W, d, H = M.get_factors()
        # every rank holds the same w, d bit for bit
        for r in range(1, ranks):
            Wr, dr, _ = M.rank_ctx(r).get_factors(h=False)
            assert np.array_equal(Wr, W) and np.array_equal(dr, d)
    assert it == 4
    assert rel_fro(W, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9 and rel_fro(d, ref["d"]) < 1e-9
    assert same_zero_pattern(W, ref["w"]) and same_zero_pattern(H, ref["h"])
    # against the one-shard GPU fit: the scaling is applied after the sums -> rounding only
    assert rel_fro(W, one["w"].T) < 1e-11 and rel_fro(H, one["h"].T) < 1e-11
    assert np.allclose(tols, one["tol"], rtol=1e-9, atol=0)


def test_team_skips_genes_empty_in_all_shards_only(sa, ora):
    """A gene with no entry in ONE shard but entries elsewhere must still be solved (global counts decide,
    src/singlet.cpp:340); a gene empty everywhere keeps its stale column on every rank."""
    m, n, k = 60, 400, 6
    A = ora.synth_csc(m, n, 4)
    x, i, p = A.x.copy(), A.i.copy(), A.p.copy()
    keep = np.ones(x.shape[0], dtype=bool)
    col_of = np.repeat(np.arange(n), np.diff(p))
    keep &= ~((i == 7) & (col_of < n // 2))     # gene 7: only in the second half of the cells
    keep &= ~(i == 11)                          # gene 11: nowhere
    cnt = np.bincount(col_of[keep], minlength=n)
    A2 = ora.CSC(x[keep], i[keep], np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32), m, n)
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf(A2, A2.t(), 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    with sa.Multi([0, 0]) as M:
        M.upload(to_dgc(sa, A2))
        M.fit_init(k, w0)
        M.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
        W, d, H = M.get_factors()
    assert rel_fro(W, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9


def test_rccl_team_of_one_matches_plain_run(sa, ora):
    """RCCL itself (ncclCommInitAll / ncclCommInitRank, grouped reduce-scatter + all-reduce, all-gather on
    the compute stream) with the one device this box has: same factors as the plain loop to rounding."""
    m, n, k = 400, 900, 24
    A = ora.synth_csc(m, n, 20)
    w0 = ora.synth_winit(k, m)
    one = sa.c_nmf(to_dgc(sa, A), None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    with sa.Multi([0]) as M:                      # ncclCommInitAll
        M.upload(to_dgc(sa, A))
        M.fit_init(k, w0)
        M.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
        W, d, H = M.get_factors()
    assert rel_fro(W, one["w"].T) < 1e-11 and rel_fro(H, one["h"].T) < 1e-11 and rel_fro(d, one["d"]) < 1e-11
    c = sa.Context(0)                             # ncclCommInitRank
    try:
        c.comm_init_rank(1, 0, sa.comm_unique_id())
        c.upload(to_dgc(sa, A))
        c.fit_init(k, w0)
        tols = [c.nmf_iterate(0.01, 0.01, 0.0, 0.0) for _ in range(3)]
        W2, d2, H2 = c.get_factors()
    finally:
        c.close()
    assert np.array_equal(W2, W) and np.array_equal(H2, H) and np.array_equal(d2, d)
    assert np.allclose(tols, one["tol"], rtol=1e-9, atol=0)


def test_c_nmf_honours_singlet_ngpu(sa, ora, monkeypatch):
    """SINGLET_NGPU above the device count is an error (no silent fallback); = 1 is the plain path."""
    A = ora.synth_csc(50, 80, 5)
    w0 = ora.synth_winit(4, 50)
    monkeypatch.setenv("SINGLET_NGPU", "64")
    with pytest.raises(sa.SingletHipError):
        sa.c_nmf(to_dgc(sa, A), None, 0.0, 2, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    monkeypatch.setenv("SINGLET_NGPU", "1")
    r = sa.c_nmf(to_dgc(sa, A), None, 0.0, 2, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    assert r["iter"] == 2


@pytest.mark.parametrize("which,ranks", [("both", 2), ("h_only", 3), ("w_only", 4)])
def test_linked_nmf_on_a_team(sa, ora, which, ranks):
    """c_linked_nmf (src/singlet.cpp:1059-1086) with the cells sharded: link_h's columns follow their cells, link_w
    multiplies each rank's gene block of the summed right-hand sides."""
    m, n, k = 263, 530, 9          # 263 genes: uneven last gene block
    A = ora.synth_csc(m, n, 15)
    w0 = ora.synth_winit(k, m)
    rng = np.random.default_rng(4)
    lh = (rng.random((k, n)) < 0.7) * (0.5 + rng.random((k, n)))
    lw = (rng.random((k, m)) < 0.8).astype(np.float64)
    off = np.ones((1, 1))
    link_h = lh if which in ("both", "h_only") else off
    link_w = lw if which in ("both", "w_only") else off
    ref = ora.c_linked_nmf(A, A.t(), 0.0, 4, 0.01, 0.0, 0, w0, link_h, link_w)
    with sa.Multi([0] * ranks) as M:
        M.upload(to_dgc(sa, A))
        M.fit_init(k, w0)
        M.set_links(link_h, link_w)
        M.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0)
        W, d, H = M.get_factors()
    assert rel_fro(W, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9 and rel_fro(d, ref["d"]) < 1e-9
    assert same_zero_pattern(W, ref["w"]) and same_zero_pattern(H, ref["h"])
    if which in ("both", "h_only"):
        assert np.all(H[lh.T == 0] == 0)


def test_team_refuses_what_it_does_not_support(sa, ora):
    A = ora.synth_csc(40, 90, 5)
    with sa.Multi([0, 0]) as M:
        M.upload(to_dgc(sa, A))
        M.fit_init(4, ora.synth_winit(4, 40))
        c = M.rank_ctx(0)
        with pytest.raises(sa.SingletHipError):
            c.set_allreduce(lambda p, n: None)          # a team rank cannot take a hook
        with pytest.raises(sa.SingletHipError):
            c.step_w(0.0, 0.0)                          # step API is per shard; the team iterates as a whole
        with pytest.raises(sa.SingletHipError):
            c.ard_run(0.0, 2, 0.01, 0.0, 1, 20, 1e9, 1)  # a rank of a one-process team does not run the loop alone


@pytest.mark.parametrize("m,n,k,ranks,inv", [(300, 900, 8, 2, 20), (257, 700, 30, 3, 10), (400, 520, 50, 2, 20), (420, 640, 70, 4, 10),
                                             (300, 900, 8, 8, 20), (263, 800, 50, 8, 10), (150, 640, 70, 7, 10)])
def test_sharded_masked_path_matches_the_oracle_and_the_single_shard(sa, ora, m, n, k, ranks, inv):
    """c_ard_nmf with the cells sharded (src/singlet.cpp:469-503, 571-607 are the reference's own chunked
    forms, `i + offset` at :485, `j + offset` at :590): per-gene right-hand sides and Gram downdates are
    reduce-scattered by gene blocks, the loss sum all-reduced.  Same trace rows, same factors."""
    A = ora.synth_csc(m, n, 10)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    seed = 977
    ref = ora.c_ard_nmf(A, At, 0.0, 5, 0.01, 0.0, 0, w0, seed, inv, 1e9, 2)
    one = sa.c_ard_nmf(to_dgc(sa, A), None, 0.0, 5, False, 0.01, 0.0, 0, w0.T, seed, inv, 1e9, 2)
    with sa.Multi([0] * ranks) as M:
        M.upload(to_dgc(sa, A))
        M.fit_init(k, w0)
        r = M.ard_run(0.0, 5, 0.01, 0.0, seed, inv, 1e9, 2)
        W, d, H = M.get_factors()
    assert list(r["iter"]) == list(ref["iter"]) == [0, 2, 4, 5]
    assert rel_fro(r["test_mse"], ref["test_mse"]) < 1e-9 and rel_fro(r["tol"], ref["tol"]) < 1e-7
    assert rel_fro(W, ref["w"]) < 1e-9 and rel_fro(H, ref["h"]) < 1e-9 and rel_fro(d, ref["d"]) < 1e-9
    assert rel_fro(W, one["w"].T) < 1e-10 and rel_fro(H, one["h"].T) < 1e-10
    assert rel_fro(r["test_mse"], one["test_mse"]) < 1e-11


def test_sharded_masked_overfit_break_and_rccl_team_of_one(sa, ora):
    """The overfit break (score > threshold, l.1124) happens at the same iteration on the team; and the masked
    loop runs through RCCL itself as a team of one."""
    m, n, k = 200, 500, 12
    A = ora.synth_csc(m, n, 10)
    w0 = ora.synth_winit(k, m)
    one = sa.c_ard_nmf(to_dgc(sa, A), None, 0.0, 30, False, 0.0, 0.0, 0, w0.T, 5, 10, 1e-5, 1)
    for devs in ([0, 0], [0]):
        with sa.Multi(devs) as M:
            M.upload(to_dgc(sa, A))
            M.fit_init(k, w0)
            r = M.ard_run(0.0, 30, 0.0, 0.0, 5, 10, 1e-5, 1)
        assert list(r["iter"]) == list(one["iter"]) and rel_fro(r["test_mse"], one["test_mse"]) < 1e-10
        assert rel_fro(r["score_overfit"], one["score_overfit"]) < 1e-6


@pytest.mark.parametrize("masked", [False, True])
def test_threaded_and_serial_team_drives_are_bit_identical(sa, ora, masked, monkeypatch):
    """sgl_multi_* drives its ranks from one host thread per device (round 4; the default) or, with SGL_MULTI_SERIAL=1,
    from the calling thread with grouped collectives (rounds 2 - 3): same kernels, same exchange, same sums in the same
    order -- the factors and traces must agree bit for bit, plain and masked, here with three ranks on one device."""
    m, n, k = 180, 700, 12
    A = to_dgc(sa, ora.synth_csc(m, n, 10))
    w0 = ora.synth_winit(k, m)
    out = {}
    for serial in (False, True):
        if serial:
            monkeypatch.setenv("SGL_MULTI_SERIAL", "1")
        else:
            monkeypatch.delenv("SGL_MULTI_SERIAL", raising=False)
        with sa.Multi([0, 0, 0]) as M:
            M.upload(A)
            M.fit_init(k, w0)
            if masked:
                r = M.ard_run(0.0, 4, 0.01, 0.0, 99, 8, 1e9, 2)
                trace = (r["test_mse"], r["tol"], r["iter"])
            else:
                it, tols = M.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0)
                trace = (tols,)
            out[serial] = (M.get_factors(), trace)
    for a, b in zip(out[False][0], out[True][0]):
        assert np.array_equal(a, b)
    for a, b in zip(out[False][1], out[True][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k,ranks", [(7, 2), (50, 3), (70, 8), (100, 4)])
def test_sharded_masked_path_moves_triangles_with_the_same_bits(sa, ora, k, ranks, monkeypatch):
    """Round 6: the per-gene Gram downdates of the sharded masked W-update are symmetric, so their reduce-scatter moves the lower
    triangles -- k (k + 1) / 2 doubles per gene instead of k^2 (2.4 -> 1.2 GB per rank and iteration at k = 100).  The sums over the
    ranks are element-wise either way: the fit must come out bit for bit as with the full blocks (SGL_TEAM_FULL_S=1)."""
    m, n = 260, 700
    A = to_dgc(sa, ora.synth_csc(m, n, 10))
    w0 = ora.synth_winit(k, m)
    out = {}
    for full in (True, False):
        if full:
            monkeypatch.setenv("SGL_TEAM_FULL_S", "1")
        else:
            monkeypatch.delenv("SGL_TEAM_FULL_S", raising=False)
        with sa.Multi([0] * ranks) as M:
            M.upload(A)
            M.fit_init(k, w0)
            r = M.ard_run(0.0, 3, 0.01, 0.0, 31, 8, 1e9, 1)
            out[full] = (M.get_factors(), (r["test_mse"], r["tol"], r["iter"]))
    for a, b in zip(out[True][0], out[False][0]):
        assert np.array_equal(a, b)
    for a, b in zip(out[True][1], out[False][1]):
        assert np.array_equal(a, b)


def test_team_error_reaches_the_caller_from_a_worker_thread(sa, ora):
    """A failure inside a rank's worker thread (here: a rank above the library's limit, refused by every rank) comes back
    as the call's error with the rank's message, and the team stays usable."""
    A = to_dgc(sa, ora.synth_csc(120, 300, 10))
    with sa.Multi([0, 0]) as M:
        M.upload(A)
        with pytest.raises(sa.SingletHipError) as e:
            M.fit_init(2000, None)
        assert "unsupported" in str(e.value) or "rank" in str(e.value)
        M.fit_init(6, ora.synth_winit(6, 120))
        it, tols = M.nmf_run(0.0, 2, 0.01, 0.01, 0.0, 0.0)
        assert it == 2 and np.all(np.isfinite(tols))


@pytest.mark.parametrize("argv,mode,n", [(["--gpus", "4", "--loopback"], "loopback", 4),
                                          (["--gpus", "1", "--single-process"], "native-single-process", 1)])
def test_bench_launcher_free_forms_print_one_json_line(argv, mode, n):
    """`python bench.py --gpus N` without a launcher is the library's one-process team: run it end to end on a small
    shape (all ranks on device 0, or the RCCL team of one) and hold the JSON line to the bench contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv +
                       ["--genes", "3000", "--cells", "40000", "--k", "12", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "comm"):
        assert key in d, key
    assert d["n_gpus"] == n and d["steps"] == 3 and d["comm"]["mode"] == mode and d["value"] > 0
    assert d.get("loopback", False) == (mode == "loopback")
    if mode == "native-single-process":
        assert d["comm"]["rccl_nranks"] == 1
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])


@pytest.mark.parametrize("nproc", [2, 3])
def test_bench_process_per_gpu_form_with_several_processes_on_one_device(nproc):
    """The driver's N > 1 command -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with every rank's
    context on device 0 (SGL_BENCH_FORCE_DEVICE) and the hook's all-reduce through gloo (SGL_BENCH_HOOK_BACKEND): a rehearsal of
    the process-per-GPU form on a 1-GPU box.  Everything on the host side that only runs with world > 1 does run: the sharding by
    rank, the agreement of the ranks' tol bits after warm-up, the gather of every rank's phases, the max-over-ranks clock, one JSON
    line from rank 0 with `per_rank` and `rank_imbalance`.  (RCCL itself refuses two ranks on one device: that part stays with the
    team of one and the loopback team.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SGL_BENCH_FORCE_DEVICE="0", SGL_BENCH_HOOK_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                        "--master-port", str(29560 + nproc), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--comm", "hook",
                        "--genes", "3000", "--cells", "40001", "--k", "12", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == nproc and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["comm"]["mode"] == "hook" and d["comm"]["host_coordination"] == "gloo" and d["comm"]["tol_bit_identical_across_ranks"] is True
    assert [q["rank"] for q in d["per_rank"]] == list(range(nproc))
    assert sum(q["cells"] for q in d["per_rank"]) == 40001 and max(q["cells"] for q in d["per_rank"]) - min(q["cells"] for q in d["per_rank"]) <= 1
    assert all(q["phases_ms_per_step"]["comm"] > 0 for q in d["per_rank"])
    assert d["rank_imbalance"]["slowest_rank_by_compute"] in range(nproc)


def test_team_call_times_out_instead_of_hanging():
    """A rank whose worker is blocked on the host (test hook SGL_TEAM_TEST_STALL) keeps its peers waiting for its part of the
    exchange.  The watchdog of the team call (SGL_TEAM_TIMEOUT_S) must release them and report SGL_ECOMM instead of hanging
    the host, and the team must refuse further calls (its communicators count as aborted).  Own process: both switches
    are read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np
import singlet_amd as sa
from oracle import oracle as ora
A = ora.synth_csc(150, 400, 10)
dA = sa.dgCMatrix(A.x, A.i, A.p, (A.nrow, A.ncol))
M = sa.Multi([0, 0, 0])
M.upload(dA)
t0 = time.time()
try:
    M.fit_init(6, ora.synth_winit(6, 150))     # its gene-count all-reduce is the first exchange step
    print("NO-ERROR")
except sa.SingletHipError as e:
    print("ERR1", round(time.time() - t0, 1), str(e))
try:
    M.iterate(0.01, 0.01, 0.0, 0.0)
    print("NO-ERROR-2")
except sa.SingletHipError as e:
    print("ERR2", str(e))
M.close()
print("CLOSED")
''' % root
    env = dict(os.environ, SGL_TEAM_TEST_STALL="1:6", SGL_TEAM_TIMEOUT_S="1.5")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    out = r.stdout
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERR1" in out and "did not come back" in out, out
    assert "ERR2" in out and "aborted" in out, out
    assert "CLOSED" in out and "NO-ERROR" not in out, out
