"""Random draws of the training augmentation (reference ``src/segmantic/seg/monai_unet.py:178-217``).

The arithmetic runs in ``csrc/augment.hip``; this module only draws the parameters the way
MONAI's transforms do and composes the spatial ones into one index map:

* ``augment_spatial``: ``RandRotated(prob=0.2, range_z=0.4)``, ``range_x``, ``range_y`` (one
  rotation each, about the volume centre, angle ~ U(-0.4, 0.4) rad), then
  ``RandZoomd(prob=0.2, min_zoom=0.8, max_zoom=1.3, keep_size=True)`` (one factor for all axes).
  The reference resamples the whole volume once per transform; here the four index maps are
  composed and applied inside the patch gather (one interpolation instead of up to four, no
  whole-volume passes; "area" zoom interpolation of the image is approximated by trilinear).
* ``augment_intensity``: ``RandAdjustContrastd(prob=0.2, gamma=(0.5, 4.5))``,
  ``RandHistogramShiftd(prob=0.2, num_control_points=10)``, ``RandBiasFieldd(prob=0.2)``,
  ``RandGibbsNoised(prob=0.2, alpha=(0, 1))``, ``RandKSpaceSpikeNoised(prob=0.2)`` per patch (the
  spike intensity's default range, 0.95..1.1 x 2.5 x mean log|K|, is evaluated on the device from a
  host-drawn U(0,1)).

Spatial axes: the cached volumes are [C, d0, d1, d2]; ``range_x`` rotates about d0, ``range_y``
about d1, ``range_z`` about d2, as MONAI names the axes of a channel-first array.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def _rot(axis: int, angle: float) -> np.ndarray:
    """4x4 rotation about spatial axis `axis` (0,1,2 = d0,d1,d2) of coordinates ordered (d0,d1,d2)."""
    c, s = np.cos(angle), np.sin(angle)
    m = np.eye(4)
    a, b = [(1, 2), (0, 2), (0, 1)][axis]
    m[a, a], m[a, b], m[b, a], m[b, b] = c, -s, s, c
    return m


def draw_spatial(rng: np.random.RandomState, shape) -> Optional[np.ndarray]:
    """4x4 map from an index of the augmented volume to the source index, both in (d0, d1, d2)
    order, or None when no spatial transform fired."""
    ctr = (np.asarray(shape, dtype=np.float64) - 1.0) / 2.0
    to_c, from_c = np.eye(4), np.eye(4)
    to_c[:3, 3], from_c[:3, 3] = -ctr, ctr
    m = np.eye(4)
    fired = False
    # reference order: rotate about z (d2), then x (d0), then y (d1), then zoom.  The augmented
    # image is Z(Ry(Rx(Rz(I)))), so an output index is pulled back through zoom, Ry, Rx, Rz.
    stages = []
    for axis in (2, 0, 1):
        if rng.rand() < 0.2:
            stages.append(_rot(axis, -float(rng.uniform(-0.4, 0.4))))   # pull-back = inverse rotation
            fired = True
        else:
            stages.append(None)
    zoom = None
    if rng.rand() < 0.2:
        zoom = float(rng.uniform(0.8, 1.3))
        fired = True
    if not fired:
        return None
    if zoom is not None:
        z = np.eye(4)
        z[0, 0] = z[1, 1] = z[2, 2] = 1.0 / zoom
        m = z @ m
    for st in reversed(stages):          # Ry, Rx, Rz pull-backs
        if st is not None:
            m = st @ m
    return from_c @ m @ to_c


def to_index_map_xyz(m_d012: np.ndarray) -> np.ndarray:
    """(d0,d1,d2)-ordered 4x4 -> the kernel's row-major 3x4 over (x=d2, y=d1, z=d0)."""
    p = np.zeros((4, 4))
    p[0, 2] = p[1, 1] = p[2, 0] = p[3, 3] = 1.0       # (x,y,z,1) -> (d0,d1,d2,1)
    mm = p.T @ m_d012 @ p                              # p is its own inverse (a swap)
    return mm[:3, :].copy()


def forward_point(m_d012: np.ndarray, pt) -> np.ndarray:
    """Source index -> index in the augmented volume (inverse of the pull-back map)."""
    inv = np.linalg.inv(m_d012)
    return (inv @ np.array([pt[0], pt[1], pt[2], 1.0]))[:3]


def draw_intensity(rng: np.random.RandomState, n: int, roi=None):
    """Per-patch draws: (contrast, hist, bias) for ``ops.intensity_augment`` and, when ``roi`` is
    given, (gibbs, spike) for ``ops.kspace_augment``."""
    con = (rng.rand(n) < 0.2).astype(np.uint8)
    gam = rng.uniform(0.5, 4.5, n).astype(np.float32)
    hon = (rng.rand(n) < 0.2).astype(np.uint8)
    ctrl = np.tile(np.linspace(0.0, 1.0, 10), (n, 1))
    for i in range(n):
        for k in range(1, 9):                         # RandHistogramShift.randomize
            ctrl[i, k] = rng.uniform(ctrl[i, k - 1], ctrl[i, k + 1])
    bon = (rng.rand(n) < 0.2).astype(np.uint8)
    coef = rng.uniform(0.0, 0.1, (n, 20)).astype(np.float32)
    out = ((con, gam), (hon, ctrl.astype(np.float32)), (bon, coef))
    if roi is None:
        return out
    gon = (rng.rand(n) < 0.2).astype(np.uint8)
    alpha = rng.uniform(0.0, 1.0, n).astype(np.float32)
    son = (rng.rand(n) < 0.2).astype(np.uint8)
    loc = np.stack([rng.randint(0, int(r), n) for r in roi], 1).astype(np.int32)
    u = rng.rand(n).astype(np.float32)
    return out + ((gon, alpha), (son, loc, u))
