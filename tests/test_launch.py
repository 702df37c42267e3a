"""Self-launch of the per-GPU ranks (segmantic_amd/seg/launch.py): what ``pl.Trainer(devices=N)``
does for the reference (monai_unet.py:529-538).  CPU-only: the ranks meet over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ECHO = str(ROOT / "tests" / "helpers" / "rank_echo.py")


def test_runtime_env_is_set_at_package_import():
    # a fresh interpreter: importing the package must export the queue setting before any device call
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import segmantic_amd, torch; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES'), segmantic_amd.HW_QUEUES_EFFECTIVE, torch.cuda.is_initialized())")
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    for k in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE"):
        env.pop(k, None)
    # a rank of a multi-process run (WORLD_SIZE is known at import time): 8 queues, before any device call
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, WORLD_SIZE="2"), capture_output=True, text=True,
                         check=True).stdout
    assert out.split() == ["8", "True", "False"]
    # single-GPU process: the runtime default stays (measured faster, segmantic_amd/__init__.py)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert out.split() == ["None", "True", "False"]
    # an exported value wins
    out = subprocess.run([sys.executable, "-c", "import os, segmantic_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"],
                         env=dict(env, GPU_MAX_HW_QUEUES="16", WORLD_SIZE="2"), capture_output=True, text=True, check=True).stdout
    assert out.strip() == "16"


def test_torchrun_command_shape():
    from segmantic_amd.seg import launch
    cmd = launch.torchrun_command(4, ["bench.py", "--gpus", "4"], port=1234)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "1234" and cmd[-3:] == ["bench.py", "--gpus", "4"]
    cmd = launch.torchrun_command(2, ["-m", "pkg.mod", "x"], port=1)
    assert cmd[-3:] == ["-m", "pkg.mod", "x"]


def test_spawn_ranks_runs_the_ranks_and_returns_their_code(capfd, monkeypatch):
    from segmantic_amd.seg import launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = launch.spawn_ranks(2, [ECHO, "--tag", "a"])
    out = capfd.readouterr().out
    assert rc == 0
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["world"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--tag", "a"]
    assert line["hwq"] == os.environ.get("GPU_MAX_HW_QUEUES", "8")     # spawn_ranks exports it to its ranks
    # a failing rank fails the launch
    rc = launch.spawn_ranks(2, [ECHO, "--fail-rank", "1"])
    capfd.readouterr()
    assert rc != 0


def test_spawn_ranks_refuses_inside_a_rank_and_for_one_rank(monkeypatch):
    from segmantic_amd.seg import launch
    with pytest.raises(ValueError):
        launch.spawn_ranks(1, [ECHO])
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError):
        launch.spawn_ranks(2, [ECHO])


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2` without a launcher must start two ranks itself.  Without a GPU the
    ranks stop at "needs an MI355X" -- which proves they were started (one message per rank) and that
    the launcher's failure code comes back."""
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    env.pop("WORLD_SIZE", None)
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_ddp_gpu.py on a GPU box")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert (p.stdout + p.stderr).count("bench.py needs an MI355X") == 2
