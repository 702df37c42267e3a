"""segmantic_amd -- MI355X-native hot path for segmantic's 3D UNet train / predict surface.

Importing the package loads ``csrc/libsegmi.so`` (hand-written gfx950 HIP kernels behind the
C-ABI of ``include/segmi.h``).  A missing library is an ImportError: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the native library is missing)

__version__ = "0.1.0"
