"""Self-launch of the one-process-per-GPU ranks.

The reference turns ``gpu_ids: [0, ..., 7]`` into data-parallel training by itself:
``pl.Trainer(devices=len(gpu_ids))`` (``src/segmantic/seg/monai_unet.py:529-538``) makes Lightning's
DDP strategy start one process per device.  Here the same happens with ``torch.distributed.run`` as a
**child process**: the parent never touches the GPU (a process that has initialised the HIP runtime
must not be replaced or forked into ranks on this platform), it only waits for the children, relays
their output (inherited stdio) and returns their exit code.

Used by ``bench.py --gpus N``, ``segmantic-unet train / train-config / predict`` with several
``gpu_ids`` and by ``monai_unet.train()`` / ``predict()`` called from Python.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence

# Ranks get 8 hardware queues (a rank drives more HIP streams -- training, weight gradients, re-pack, sampler,
# gradient buckets + RCCL's own -- than ROCm's default 4, and streams that alias one queue execute in order);
# segmantic_amd/__init__.py sets the same for any process that is imported with WORLD_SIZE > 1.
HW_QUEUES_ENV = "GPU_MAX_HW_QUEUES"
HW_QUEUES_DEFAULT = "8"


def under_launcher() -> bool:
    """True inside a rank started by torchrun / this module (WORLD_SIZE is part of its contract)."""
    return "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def torchrun_command(nproc: int, target: Sequence[str], port: Optional[int] = None) -> List[str]:
    """``python -m torch.distributed.run`` command line for ``nproc`` ranks on this node.
    ``target``: script path + arguments, or ``["-m", module, ...]``."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    target = list(target)
    if target and target[0] == "-m":
        cmd += ["-m", target[1]] + target[2:]
    else:
        cmd += target
    return cmd


def spawn_ranks(nproc: int, target: Sequence[str], env: Optional[dict] = None) -> int:
    """Start ``nproc`` ranks of ``target`` as children of this process and wait for them.

    Refuses to run from a process whose HIP runtime is already up: the caller must decide to go
    multi-process *before* its first ``torch.cuda`` call (``bench.py`` / the CLI do so at the top of
    ``main``)."""
    if nproc < 2:
        raise ValueError("spawn_ranks is for 2 or more ranks")
    if under_launcher():
        raise RuntimeError("spawn_ranks called from inside a launched rank")
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        raise RuntimeError(
            "segmantic_amd: cannot start per-GPU ranks from a process that has already initialised the GPU; "
            "request several gpu_ids before the first device call, or launch with torch.distributed.run")
    e = dict(os.environ if env is None else env)
    e.setdefault(HW_QUEUES_ENV, HW_QUEUES_DEFAULT)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("OMP_NUM_THREADS", "4")
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e["PYTHONPATH"] = os.pathsep.join([root] + ([e["PYTHONPATH"]] if e.get("PYTHONPATH") else []))
    cmd = torchrun_command(nproc, target)
    proc = subprocess.Popen(cmd, env=e)
    try:
        return proc.wait()
    except BaseException:
        proc.terminate()            # exactly the child we started, never a pattern
        try:
            proc.wait(timeout=30)
        except Exception:
            proc.kill()
        raise
