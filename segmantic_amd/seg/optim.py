"""Optimisers over the flat parameter arena + the reference's LR schedules.

Replaces ``torch.optim.Adam / SGD`` and ``adabelief_pytorch.AdaBelief`` as configured at
reference ``src/segmantic/seg/monai_unet.py:292-314`` and the three schedulers of ``:316-337``
(stepped once per validation epoch, ``:375-379``).  One fused HIP kernel updates all 4.8 M
parameters (the reference's optimiser walks 148 small tensors).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .. import ops


class FlatOptimizer:
    def __init__(self, flat: torch.Tensor, flat_grad: torch.Tensor, lr: float):
        self.flat, self.flat_grad = flat, flat_grad
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.steps = 0

    def zero_grad(self):  # gradients are overwritten by the engine's backward
        pass

    def state_dict(self):
        return {"lr": self.lr, "steps": self.steps}

    def step(self, grad_scale: float = 1.0, lo: int = 0, hi: Optional[int] = None, advance: bool = True):
        """One update of the arena range ``[lo, hi)`` (default: all of it).  A training step that updates
        the arena in two pieces (``UNetEngine.finish_carried``) passes ``advance=False`` with the second
        one: both pieces belong to the same step (bias corrections, SGD's first-step rule)."""
        raise NotImplementedError

    def _range(self, lo, hi):
        hi = self.flat.numel() if hi is None else int(hi)
        lo = int(lo)
        if not 0 <= lo <= hi <= self.flat.numel():
            raise ValueError(f"optimizer range [{lo}, {hi}) outside the arena of {self.flat.numel()} parameters")
        return lo, hi


class FlatAdam(FlatOptimizer):
    def __init__(self, flat, flat_grad, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 amsgrad=False):
        super().__init__(flat, flat_grad, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.max_exp_avg_sq = torch.zeros_like(flat) if amsgrad else None

    def step(self, grad_scale: float = 1.0, lo: int = 0, hi: Optional[int] = None, advance: bool = True):
        lo, hi = self._range(lo, hi)
        if advance:
            self.steps += 1
        if hi > lo:
            ops.adam_step(self.flat[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                          None if self.max_exp_avg_sq is None else self.max_exp_avg_sq[lo:hi], self.lr,
                          self.betas[0], self.betas[1], self.eps, self.weight_decay, self.steps, grad_scale)


class FlatSGD(FlatOptimizer):
    def __init__(self, flat, flat_grad, lr, momentum=0.0, weight_decay=0.0):
        super().__init__(flat, flat_grad, lr)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.buf = torch.zeros_like(flat) if momentum != 0 else None

    def step(self, grad_scale: float = 1.0, lo: int = 0, hi: Optional[int] = None, advance: bool = True):
        lo, hi = self._range(lo, hi)
        if advance:
            self.steps += 1
        if hi > lo:
            ops.sgd_step(self.flat[lo:hi], self.flat_grad[lo:hi], None if self.buf is None else self.buf[lo:hi],
                         self.lr, self.momentum, self.weight_decay, self.steps == 1, grad_scale)


class FlatAdaBelief(FlatOptimizer):
    def __init__(self, flat, flat_grad, lr=1e-3, betas=(0.9, 0.999), eps=1e-16, weight_decay=0.0,
                 weight_decouple=True):
        super().__init__(flat, flat_grad, lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.weight_decouple = weight_decouple
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_var = torch.zeros_like(flat)

    def step(self, grad_scale: float = 1.0, lo: int = 0, hi: Optional[int] = None, advance: bool = True):
        lo, hi = self._range(lo, hi)
        if advance:
            self.steps += 1
        if hi > lo:
            ops.adabelief_step(self.flat[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi],
                               self.exp_avg_var[lo:hi], self.lr, self.betas[0], self.betas[1], self.eps,
                               self.weight_decay, self.weight_decouple, self.steps, grad_scale)


def make_optimizer(cfg: dict, flat, flat_grad) -> FlatOptimizer:
    """cfg keys as the reference's ``optimizer`` dict (monai_unet.py:83-90, 429-437)."""
    name = cfg.get("optimizer", "Adam")
    if name == "SGD":
        return FlatSGD(flat, flat_grad, lr=cfg["lr"], momentum=cfg.get("momentum", 0.9))
    if name == "Adam":
        return FlatAdam(flat, flat_grad, lr=cfg["lr"], amsgrad=bool(cfg.get("amsgrad", False)))
    if name == "AdaBelief":
        return FlatAdaBelief(flat, flat_grad, lr=cfg["lr"], eps=cfg.get("epsilon", 1e-8),
                             weight_decouple=bool(cfg.get("weight_decouple", False)))
    raise ValueError(f"unknown optimizer '{name}' (Adam, SGD, AdaBelief)")


# ---------------------------------------------------------------------------- schedulers
class ConstantLR:
    """torch ConstantLR(factor=1, total_iters=0): the learning rate never changes."""

    def __init__(self, opt: FlatOptimizer):
        self.opt = opt

    def step(self, metric: Optional[float] = None):
        pass


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", factor, patience) defaults:
    threshold 1e-4 (rel), cooldown 0, min_lr 0, eps 1e-8."""

    def __init__(self, opt: FlatOptimizer, factor=0.5, patience=10, threshold=1e-4, eps=1e-8,
                 min_lr=0.0):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.opt, self.factor, self.patience = opt, factor, patience
        self.threshold, self.eps, self.min_lr = threshold, eps, min_lr
        self.best = math.inf
        self.num_bad = 0

    def step(self, metric: float):
        cur = float(metric)
        if cur < self.best * (1.0 - self.threshold):
            self.best = cur
            self.num_bad = 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = max(self.opt.lr * self.factor, self.min_lr)
            if self.opt.lr - new_lr > self.eps:
                self.opt.lr = new_lr
                print(f"Reducing learning rate to {new_lr:.4e}.")
            self.num_bad = 0


class CosineAnnealingWarmRestarts:
    """torch CosineAnnealingWarmRestarts(T_0, T_mult, eta_min=0), stepped once per epoch."""

    def __init__(self, opt: FlatOptimizer, T_0: int, T_mult: int = 1, eta_min: float = 0.0):
        if T_0 <= 0 or not isinstance(T_0, int):
            raise ValueError(f"Expected positive integer T_0, but got {T_0}")
        if T_mult < 1 or not isinstance(T_mult, int):
            raise ValueError(f"Expected integer T_mult >= 1, but got {T_mult}")
        self.opt, self.T_0, self.T_i, self.T_mult, self.eta_min = opt, T_0, T_0, T_mult, eta_min
        self.T_cur = 0

    def step(self, metric: Optional[float] = None):
        self.T_cur += 1
        if self.T_cur >= self.T_i:
            self.T_cur -= self.T_i
            self.T_i *= self.T_mult
        self.opt.lr = self.eta_min + (self.opt.base_lr - self.eta_min) * \
            (1 + math.cos(math.pi * self.T_cur / self.T_i)) / 2


def make_scheduler(cfg: dict, opt: FlatOptimizer):
    name = cfg.get("scheduler", "Constant")
    if name == "Constant":
        return ConstantLR(opt)
    if name == "ReduceOnPlateau":
        return ReduceLROnPlateau(opt, factor=cfg.get("factor", 0.5), patience=cfg.get("patience", 10))
    if name == "Cosine":
        return CosineAnnealingWarmRestarts(opt, T_0=int(cfg.get("T_0", 50)),
                                           T_mult=int(cfg.get("T_multi", 1)))
    raise ValueError(f"unknown scheduler '{name}' (Constant, ReduceOnPlateau, Cosine)")
