"""segmantic_amd -- MI355X-native hot path for segmantic's 3D UNet train / predict surface.

Importing the package loads ``csrc/libsegmi.so`` (hand-written gfx950 HIP kernels behind the
C-ABI of ``include/segmi.h``).  A missing library is an ImportError: there is no CPU fallback.
"""
import os as _os
import sys as _sys

# Runtime environment first: the HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (the first
# device call of the process), so it has to be in place before anything below can touch the GPU --
# for every run, single- or multi-rank (seg/launch.py explains the setting; an exported value wins).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_t = _sys.modules.get("torch")
HW_QUEUES_EFFECTIVE = not (_t is not None and _t.cuda.is_initialized())
"""False when the importing process had already initialised the GPU: the queue setting above is
then ignored by the runtime (bench.py reports this flag next to its overlap figures)."""
del _t

from . import _lib  # noqa: F401,E402  (fails loudly when the native library is missing)

__version__ = "0.1.0"
