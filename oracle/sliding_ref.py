"""Oracle (test infrastructure): MONAI ``sliding_window_inference`` restated for CPU.

Reference call sites: ``src/segmantic/seg/monai_unet.py:354-356`` (validation, roi 160^d,
sw_batch 4, overlap 0.25 default, constant blend), ``:637-639`` (predict,
``SlidingWindowInferer(roi_size=net.spatial_size, sw_batch_size=4)``), ``:840-842`` (ensemble,
roi 96, overlap 0.5).  Algorithm restated from MONAI's published
``monai.inferers.utils.sliding_window_inference`` / ``monai.data.utils.dense_patch_slices``:

  1. if an image dim < roi: symmetric constant-0 pad (half = diff // 2 low, rest high)
  2. interval_d = roi_d if roi_d == img_d else max(int(roi_d * (1 - overlap)), 1)
  3. n_d = ceil((img_d - roi_d) / interval_d) + 1 ; start_k = min(k * interval_d, img_d - roi_d)
  4. windows enumerated with the first spatial dim slowest; groups of ``sw_batch_size`` are
     passed to ``predictor``; for each window IN ORDER: out[slice] += w * pred, cnt[slice] += w
  5. out /= cnt ; crop the padding
  w == 1 for mode "constant"; "gaussian": separable exp(-0.5 ((x - c)/sigma)^2) with
  sigma = 0.125 * roi, clamped from below at max(min nonzero, 1e-3).

PARITY UNPINNED (no golden vectors in the reference's tests; MONAI absent).
"""
from __future__ import annotations

import math
from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def scan_intervals(image_size: Sequence[int], roi: Sequence[int], overlap: float) -> List[int]:
    out = []
    for i, r in zip(image_size, roi):
        if r == i:
            out.append(int(r))
        else:
            iv = int(r * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return out


def window_starts(image_size: Sequence[int], roi: Sequence[int], overlap: float):
    """Per-dim start lists and the full window list (first spatial dim slowest)."""
    iv = scan_intervals(image_size, roi, overlap)
    per_dim = []
    for i, r, s in zip(image_size, roi, iv):
        n = int(math.ceil(float(i - r) / s)) + 1 if s > 0 else 1
        per_dim.append([min(k * s, i - r) for k in range(n)])
    grid = np.stack(np.meshgrid(*per_dim, indexing="ij"), -1).reshape(-1, len(roi))
    return per_dim, [tuple(int(v) for v in row) for row in grid]


def importance_map(roi: Sequence[int], mode: str = "constant", sigma_scale: float = 0.125):
    if mode == "constant":
        return torch.ones(tuple(roi), dtype=torch.float32)
    if mode != "gaussian":
        raise ValueError(mode)
    w = torch.ones(tuple(roi), dtype=torch.float32)
    for d, r in enumerate(roi):
        sigma = sigma_scale * r
        x = torch.arange(-(r - 1) / 2.0, (r - 1) / 2.0 + 1, dtype=torch.float32)
        g = torch.exp(x ** 2 / (-2 * sigma ** 2))
        shape = [1] * len(roi)
        shape[d] = r
        w = w * g.reshape(shape)
    mn = max(float(w[w != 0].min()), 1e-3)
    return torch.clamp(w, min=mn)


def ref_sliding_window_inference(inputs: torch.Tensor, roi_size: Sequence[int],
                                 sw_batch_size: int,
                                 predictor: Callable[[torch.Tensor], torch.Tensor],
                                 overlap: float = 0.25, mode: str = "constant"):
    """inputs [B, C, *sp] -> [B, K, *sp]; returns (output, count_map, windows)."""
    nd = inputs.dim() - 2
    B = inputs.shape[0]
    orig = list(inputs.shape[2:])
    roi = [int(r) if r else int(o) for r, o in zip(roi_size, orig)]
    image_size = [max(o, r) for o, r in zip(orig, roi)]
    pad = []
    for k in range(nd - 1, -1, -1):
        diff = max(roi[k] - orig[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    if any(pad):
        inputs = F.pad(inputs, pad, mode="constant", value=0.0)
    _, wins = window_starts(image_size, roi, overlap)
    w = importance_map(roi, mode)
    total = len(wins) * B
    out = None
    cnt = None
    for g in range(0, total, sw_batch_size):
        idx = range(g, min(g + sw_batch_size, total))
        sl = []
        for i in idx:
            b, wi = divmod(i, len(wins))
            st = wins[wi]
            sl.append((slice(b, b + 1), slice(None)) +
                      tuple(slice(s, s + r) for s, r in zip(st, roi)))
        data = torch.cat([inputs[s] for s in sl], 0)
        pred = predictor(data)
        if out is None:
            K = pred.shape[1]
            out = torch.zeros([B, K] + image_size, dtype=torch.float32)
            cnt = torch.zeros([1, 1] + image_size, dtype=torch.float32)
        for j, s in enumerate(sl):
            out[s] += w * pred[j:j + 1].float()
            if s[0].start == 0:
                cnt[(slice(0, 1), slice(None)) + s[2:]] += w
    out = out / cnt
    crop: Tuple = (slice(None), slice(None))
    for k in range(nd):
        lo = pad[2 * (nd - 1 - k)]
        crop = crop + (slice(lo, lo + orig[k]),)
    return out[crop], cnt[crop], wins
