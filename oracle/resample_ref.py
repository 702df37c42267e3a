"""Oracle (test infrastructure): ITK ``ResampleImageFilter`` semantics in numpy float64.

Reference call sites: ``src/segmantic/image/processing.py:49-71`` (``resample``: size' =
ceil(size * spacing / target), same origin / direction, identity transform, sitkLinear or
sitkNearestNeighbor, default pixel 0, output pixel type = input pixel type), ``:74-98``
(``apply_transform``: output grid = fixed image, transform maps fixed -> moving) and
``:101-120`` (``resample_to_ref``).

ITK (via SimpleITK, unpinned, absent here) semantics restated from its published source:
  for each output index i:  p = O_out + Dir_out (i * sp_out);  q = T(p);
  c = (Dir_in^-1 (q - O_in)) / sp_in  (continuous index in the moving image);
  inside  <=>  -0.5 <= c_d < size_d - 0.5 for every d  (ImageFunction::IsInsideBuffer);
  outside -> default pixel value (0);
  linear: base = floor(c) clamped to [0, size-1], upper neighbour clamped to size-1 (so the
          half-voxel border replicates the edge value), weights from c - floor(c);
  nearest: index = floor(c + 0.5)  (RoundHalfIntegerUp);
  integer pixel types: clamp to the type's range, then C-cast (truncation toward zero).
Arrays are numpy [z, y, x] (SimpleITK ``GetArrayFromImage`` order); size / spacing / origin
are (x, y, z).  PARITY UNPINNED for voxel values (SimpleITK absent; the reference's tests pin
only the geometry, ``tests/image/test_image.py:33-52``, which ``tests/test_oracle.py`` checks).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np


def resample_size(size: Sequence[int], spacing: Sequence[float],
                  target: Sequence[float]) -> Tuple[int, ...]:
    """processing.py:54-58"""
    return tuple(int(math.ceil(s * sp / t)) for s, sp, t in zip(size, spacing, target))


def ref_resample_grid(arr: np.ndarray, in_spacing, in_origin, in_direction,
                      out_size, out_spacing, out_origin, out_direction,
                      nearest: bool = False, transform: Optional[np.ndarray] = None,
                      default: float = 0.0, return_real: bool = False) -> np.ndarray:
    """arr [z,y,x] (or [y,x]); geometry tuples in (x,y,z) order; ``transform`` optional 4x4
    (3x3 in 2-D) homogeneous matrix mapping output-physical -> input-physical points."""
    nd = arr.ndim
    size_in = np.array(arr.shape[::-1], dtype=np.int64)
    sp_in = np.asarray(in_spacing, np.float64)
    sp_out = np.asarray(out_spacing, np.float64)
    o_in = np.asarray(in_origin, np.float64)
    o_out = np.asarray(out_origin, np.float64)
    d_in = np.asarray(in_direction, np.float64).reshape(nd, nd)
    d_out = np.asarray(out_direction, np.float64).reshape(nd, nd)
    out_size = [int(s) for s in out_size]
    idx = np.stack(np.meshgrid(*[np.arange(s, dtype=np.float64) for s in out_size[::-1]],
                               indexing="ij"), -1)[..., ::-1]          # [..., (x,y,z)]
    p = o_out + (idx * sp_out) @ d_out.T
    if transform is not None:
        tm = np.asarray(transform, np.float64)
        p = p @ tm[:nd, :nd].T + tm[:nd, nd]
    c = ((p - o_in) @ np.linalg.inv(d_in).T) / sp_in                    # continuous index
    inside = np.all((c >= -0.5) & (c < size_in - 0.5), axis=-1)
    a = arr.astype(np.float64)
    if nearest:
        ii = np.floor(c + 0.5).astype(np.int64)
        ii = np.clip(ii, 0, size_in - 1)
        val = a[tuple(ii[..., d] for d in range(nd - 1, -1, -1))]
    else:
        base = np.floor(c)
        frac = c - base
        base = base.astype(np.int64)
        lo = np.clip(base, 0, size_in - 1)
        hi = np.clip(base + 1, 0, size_in - 1)
        # where base was clamped from below (c in [-0.5, 0)), ITK treats distance <= 0
        frac = np.where(base < 0, 0.0, frac)
        val = np.zeros(c.shape[:-1], np.float64)
        for corner in range(1 << nd):
            w = np.ones(c.shape[:-1], np.float64)
            ix = []
            for d in range(nd):
                if (corner >> d) & 1:
                    w = w * frac[..., d]
                    ix.append(hi[..., d])
                else:
                    w = w * (1.0 - frac[..., d])
                    ix.append(lo[..., d])
            val = val + w * a[tuple(ix[::-1])]
    val = np.where(inside, val, float(default))
    if return_real:
        return val
    if np.issubdtype(arr.dtype, np.integer):
        info = np.iinfo(arr.dtype)
        val = np.trunc(np.clip(val, info.min, info.max))
    return val.astype(arr.dtype)


def ref_resample(arr, spacing, target_spacing, nearest=False, origin=None, direction=None):
    """``processing.resample`` -> (array, new_spacing)."""
    nd = arr.ndim
    origin = [0.0] * nd if origin is None else origin
    direction = np.eye(nd) if direction is None else direction
    size = arr.shape[::-1]
    new_size = resample_size(size, spacing, target_spacing)
    out = ref_resample_grid(arr, spacing, origin, direction, new_size, target_spacing, origin,
                            direction, nearest)
    return out, tuple(float(t) for t in target_spacing)
