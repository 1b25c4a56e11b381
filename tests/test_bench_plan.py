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
