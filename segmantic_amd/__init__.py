"""segmantic_amd -- MI355X-native hot path for segmantic's 3D UNet train / predict surface.

Importing the package loads ``csrc/libsegmi.so`` (hand-written gfx950 HIP kernels behind the
C-ABI of ``include/segmi.h``).  A missing library is an ImportError: there is no CPU fallback.
"""
import os as _os
import sys as _sys

# Runtime environment first: the HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (the first
# device call of the process), so it has to be in place before anything below can touch the GPU.
# Multi-rank runs (WORLD_SIZE > 1, known at import time under torchrun / seg/launch.py) get 8 hardware
# queues: a rank drives more streams than ROCm's default 4 (training, weight gradients, re-pack, sampler,
# gradient buckets + RCCL's own) and streams that alias one queue run in order, which would put the bucket
# all-reduces behind the backward they are meant to overlap.  Single-GPU runs keep the default: measured
# on one MI355X in round 3, 8 queues are 0.5-1 % SLOWER for the training step (5.71 / 5.65 / 5.89 vs 5.64 /
# 5.64 / 5.83 ms, alternating runs) and 4-5 % slower for the fit leg that follows the other legs of
# `bench.py` (6.15 vs 5.92 ms) -- more queues let the side streams contend harder.  An exported value wins.
if int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1:
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_t = _sys.modules.get("torch")
HW_QUEUES_EFFECTIVE = not (_t is not None and _t.cuda.is_initialized())
"""False when the importing process had already initialised the GPU: the queue setting above is
then ignored by the runtime (bench.py reports this flag next to its overlap figures)."""
del _t

from . import _lib  # noqa: F401,E402  (fails loudly when the native library is missing)

__version__ = "0.1.0"
