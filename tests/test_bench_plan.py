"""bench.py's dispatch (plan()): which form an invocation takes is decided from the flags and the launcher's
environment alone -- checked here without a GPU, because the first multi-GPU run happens on the driver's node."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def plan(argv, env=None):
    return bench.plan(bench.parse(argv), env or {})


def test_plain_invocation_is_one_gpu_without_a_team():
    assert plan([]) == {"form": "one-gpu", "world": 1, "rank": 0, "local_rank": 0}
    assert plan(["--gpus", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})["form"] == "one-gpu"


@pytest.mark.parametrize("n", [2, 4, 8])
def test_plain_gpus_n_takes_the_one_process_team(n):
    """`python bench.py --gpus N` with no launcher (the shape of the driver's 1-GPU command) must run, not ask for torchrun."""
    p = plan(["--gpus", str(n), "--steps", "3", "--warmup", "1"])
    assert p["form"] == "single-process" and p["world"] == n and p["devices"] == list(range(n)) and not p["loopback"]
    # a launcher world of ONE process asking for N devices is the same thing
    assert plan(["--gpus", str(n)], {"WORLD_SIZE": "1", "RANK": "0"})["form"] == "single-process"


@pytest.mark.parametrize("n", [2, 8])
def test_launcher_world_wins(n):
    env = {"WORLD_SIZE": str(n), "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"}
    p = plan(["--gpus", str(n)], env)
    assert p == {"form": "process-per-gpu", "world": n, "rank": 1, "local_rank": 1}
    assert plan(["--gpus", "1"], env)["world"] == n   # WORLD_SIZE > 1 overrides a stale --gpus


def test_rehearsal_switches():
    assert plan(["--gpus", "1", "--single-process"]) == {"form": "single-process", "world": 1, "devices": [0], "loopback": False,
                                                         "rank": 0, "local_rank": 0}
    p = plan(["--gpus", "4", "--loopback"])
    assert p["devices"] == [0, 0, 0, 0] and p["loopback"] and p["form"] == "single-process"


@pytest.mark.parametrize("argv,env", [
    (["--gpus", "4", "--loopback"], {"WORLD_SIZE": "4"}),          # loopback under a launcher
    (["--gpus", "2", "--single-process"], {"WORLD_SIZE": "2"}),
    (["--gpus", "0"], {}),
    (["--gpus", "17"], {}),
    (["--gpus", "2", "--comm", "hook"], {}),                       # the hook belongs to one rank per process
    (["--gpus", "2", "--data", "skewed"], {}),
    (["--gpus", "2"], {"WORLD_SIZE": "0"}),
])
def test_contradictions_are_refused(argv, env):
    with pytest.raises(SystemExit) as e:
        plan(argv, env)
    assert e.value.code not in (0, None)


def test_too_few_devices_exits_non_zero_with_a_message():
    """On a box with fewer than N devices (here: none) `bench.py --gpus N` must fail loudly, before any allocation,
    and print no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""      # also on a GPU box: hide the devices
    env["ROCR_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "device" in r.stderr and "{" not in r.stdout


def _fake_run(world, slow_rank=None):
    names = ("gram", "rhs_h", "nnls_h", "rhs_w", "nnls_w", "scale", "comm", "mask")
    base = {"gram": 0.9, "rhs_h": 12.4, "nnls_h": 9.1, "rhs_w": 12.0, "nnls_w": 3.0, "scale": 1.4, "comm": 2.0, "mask": 0.0}
    phases_all = []
    for r in range(world):
        ph = {n: (base[n] * (1.5 if (r == slow_rank and n in ("rhs_h", "rhs_w")) else 1.0), 10) for n in names}
        if slow_rank is not None and r != slow_rank:
            ph["comm"] = (base["comm"] + 12.0, 20)      # the others wait for the straggler inside the collective
        phases_all.append(ph)
    lay = {"entries": 1000, "tiles": 3, "tile_rows": 408, "tile_ranges": 1, "col_blocks": 8}
    return {"world": world, "elapsed": 0.5, "tols": [0.1] * 10, "dims": (300, 1000 // world, 15000 // world), "nnz_total": 15000,
            "phases": phases_all[0], "phases_all": phases_all,
            "rank_info": [{"device": r, "cells": 1000 // world, "nnz": 15000 // world} for r in range(world)],
            "sweeps": {"h_sweeps": 100, "w_sweeps": 10, "h_wave_sweeps": 10, "w_wave_sweeps": 4}, "layout": {"A": lay, "At": lay},
            "gen_s": 0.1, "w_cols_rank0": 300 // world,
            "comm": {"mode": "native-single-process", "note": None, "rccl_nranks": world, "rccl_path": "librccl.so.1",
                     "host_coordination": "none", "devices": list(range(world)), "tol_bit_identical_across_ranks": True}}


def test_report_carries_every_ranks_phases_and_the_comm_phase():
    """The first N > 1 record must diagnose itself: headline phases = max over the ranks, `per_rank` = each rank's own
    hipEvent phases incl. `comm`, and the slowest rank is named."""
    import json
    args = bench.parse(["--gpus", "4", "--steps", "10", "--genes", "300", "--cells", "1000", "--k", "5", "--no-cpu-baseline"])
    out = bench.report(args, _fake_run(4, slow_rank=2))
    json.dumps(out)
    assert len(out["per_rank"]) == 4 and [r["rank"] for r in out["per_rank"]] == [0, 1, 2, 3]
    assert out["per_rank"][2]["phases_ms_per_step"]["rhs_h"] == pytest.approx(1.86)
    assert out["phases_ms_per_step"]["rhs_h"] == pytest.approx(1.86)          # the max over ranks, not rank 0's 1.24
    assert out["phases_ms_per_step"]["comm"] == pytest.approx(1.4)            # a waiting rank's collective
    assert out["per_rank"][2]["comm_ms"] == pytest.approx(0.2)                # the straggler itself does not wait
    assert out["rank_imbalance"]["comm_ms_max"] == pytest.approx(1.4) and out["rank_imbalance"]["comm_ms_min"] == pytest.approx(0.2)
    assert out["roofline"]["ms_per_pass"] == pytest.approx(1.86)
    # one rank: no per-rank block unless the run brings rank_info
    one = _fake_run(1)
    one["rank_info"] = None
    assert "per_rank" not in bench.report(args, one)


def test_report_names_the_straggler_by_its_compute_time():
    args = bench.parse(["--gpus", "4", "--steps", "10", "--genes", "300", "--cells", "1000", "--k", "5", "--no-cpu-baseline"])
    out = bench.report(args, _fake_run(4, slow_rank=2))
    assert out["rank_imbalance"]["slowest_rank_by_compute"] == 2 and out["rank_imbalance"]["max_over_min_compute"] > 1.2


def test_process_per_gpu_device_pick_survives_a_launcher_that_isolates_the_devices():
    """The driver's N > 1 command gives every process a LOCAL_RANK.  Normally all N devices are visible and the rank takes device
    LOCAL_RANK; a launcher that sets HIP_VISIBLE_DEVICES per process leaves ONE device visible, index 0 -- ranks > 0 must not ask
    for a device that does not exist (round-5 verdict: they died before any RCCL call)."""
    for lr in range(8):
        assert bench.pick_device(lr, 8) == (lr, "LOCAL_RANK")
        dev, how = bench.pick_device(lr, 1)
        assert dev == 0 and (how == "LOCAL_RANK" if lr == 0 else "isolated" in how)
    assert bench.pick_device(5, 4)[0] == 1                      # fewer devices than ranks: wraps (RCCL then refuses, hook path)
    assert bench.pick_device(3, 8, forced="0") == (0, "forced by SGL_BENCH_FORCE_DEVICE")
    assert bench.pick_device(3, 8, forced="") == (3, "LOCAL_RANK")
    with pytest.raises(SystemExit):
        bench.pick_device(0, 0)
