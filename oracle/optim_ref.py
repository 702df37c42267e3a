"""CPU restatement of ``adabelief_pytorch.AdaBelief.step`` as the reference configures it.
TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Reference call site: ``src/segmantic/seg/monai_unet.py:305-314``::

    AdaBelief(params, lr=..., eps=optimizer["epsilon"], betas=(0.9, 0.999),
              weight_decouple=optimizer["weight_decouple"], fixed_decay=False, rectify=False)

``adabelief_pytorch`` is a third-party dependency (``pyproject.toml:27``, unpinned, absent from
``/root/reference`` and from this image).  Its published update rule (adabelief-pytorch 0.2.x,
``AdaBelief.step``; amsgrad False, weight_decay default 0) per parameter tensor, float32:

    if weight_decouple:      p *= 1 - lr * weight_decay          (fixed_decay=False)
    elif weight_decay != 0:  g += weight_decay * p
    step += 1;  bc1 = 1 - beta1**step;  bc2 = 1 - beta2**step
    m = beta1 * m + (1 - beta1) * g
    r = g - m
    s = beta2 * s + (1 - beta2) * r * r
    denom = sqrt(s.add_(eps)) / sqrt(bc2) + eps     # NOTE: add_ is IN PLACE -- eps accumulates
    p -= (lr / bc1) * m / denom                     #       in the stored second moment

PARITY UNPINNED: no fixture of the reference holds an AdaBelief trajectory.
"""
from __future__ import annotations

import math

import numpy as np


class RefAdaBelief:
    def __init__(self, n: int, lr: float, eps: float = 1e-16, betas=(0.9, 0.999),
                 weight_decay: float = 0.0, weight_decouple: bool = True):
        self.lr64, self.wd64 = float(lr), float(weight_decay)   # Python doubles, as the optimiser holds them
        self.lr, self.eps, self.betas = np.float32(lr), np.float32(eps), betas
        self.wd, self.decouple = np.float32(weight_decay), bool(weight_decouple)
        self.m = np.zeros(n, np.float32)
        self.s = np.zeros(n, np.float32)
        self.t = 0

    def step(self, p: np.ndarray, g: np.ndarray) -> np.ndarray:
        """p, g float32 [n]; returns the updated p (float32 arithmetic, one rounding per op as
        torch's eager tensor ops would do)."""
        f = np.float32
        p = p.astype(np.float32).copy()
        g = g.astype(np.float32).copy()
        b1, b2 = f(self.betas[0]), f(self.betas[1])
        if self.decouple:
            p = p * f(1.0 - self.lr64 * self.wd64)
        elif self.wd != 0:
            g = g + self.wd * p
        self.t += 1
        bc1 = 1.0 - self.betas[0] ** self.t
        bc2 = 1.0 - self.betas[1] ** self.t
        self.m = self.m * b1 + g * f(1.0 - self.betas[0])
        r = g - self.m
        self.s = self.s * b2 + (r * r) * f(1.0 - self.betas[1])
        self.s = self.s + self.eps                       # in place in adabelief_pytorch
        denom = np.sqrt(self.s) / f(math.sqrt(bc2)) + self.eps
        step_size = f(self.lr64 / bc1)
        return (p - step_size * (self.m / denom)).astype(np.float32)
