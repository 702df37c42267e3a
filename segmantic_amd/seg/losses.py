"""Loss / metric front-ends over the fused HIP kernels.

``DiceLoss`` replaces ``monai.losses.DiceLoss(to_onehot_y=True, softmax=True)`` as constructed at
reference ``src/segmantic/seg/monai_unet.py:128`` and called at ``:344`` / ``:357``;
``DiceMetric`` replaces ``monai.metrics.DiceMetric(include_background=False, reduction="mean")``
(``:136-138``, ``:642-644``).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


def as_ndhwc(t: torch.Tensor) -> torch.Tensor:
    """Logical [N,C,D,H,W] tensor -> dense NDHWC storage [N,D,H,W,C] (no copy if already so)."""
    if t.dim() != 5:
        raise ValueError("expected a 5-D [N,C,D,H,W] tensor")
    p = t.permute(0, 2, 3, 4, 1)
    return p if p.is_contiguous() else p.contiguous()


class _DiceState:
    """Scratch owned by one loss instance (reused across calls)."""

    def __init__(self):
        self.partials: Optional[torch.Tensor] = None
        self.coef: Optional[torch.Tensor] = None
        self.loss: Optional[torch.Tensor] = None
        self.dlogits: Optional[torch.Tensor] = None

    def ensure(self, logits: torch.Tensor):
        n, k = logits.shape[0], logits.shape[4]
        chunks = ops.dice_chunks(logits)
        dev = logits.device
        if self.partials is None or self.partials.shape != (n, chunks, 3, k) or self.partials.device != dev:
            self.partials = torch.empty((n, chunks, 3, k), device=dev)
            self.coef = torch.empty((n, 2, k), device=dev)
        # a fresh scalar per call: callers keep the returned loss tensor
        self.loss = torch.empty(1, device=dev)


def dice_forward(state: _DiceState, logits_ndhwc: torch.Tensor, labels: torch.Tensor,
                 smooth_nr: float, smooth_dr: float) -> torch.Tensor:
    lab = labels.to(logits_ndhwc.device, torch.float32).contiguous().view(-1)
    if lab.numel() != logits_ndhwc.numel() // logits_ndhwc.shape[4]:
        raise ValueError("label volume does not match logits")
    state.ensure(logits_ndhwc)
    ops.softmax_dice_fwd(logits_ndhwc, lab, state.partials, state.coef, state.loss, smooth_nr,
                         smooth_dr)
    state.labels = lab
    return state.loss.view(())


def dice_backward(state: _DiceState, logits_ndhwc: torch.Tensor, grad_scale: float = 1.0,
                  out: Optional[torch.Tensor] = None,
                  bias_grad: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``bias_grad`` (f32[K]): also receives sum_voxels dlogits, i.e. the bias gradient of the conv
    that produced the logits (saves that layer a pass over the gradient tensor)."""
    if out is None or out.shape != logits_ndhwc.shape or out.dtype != logits_ndhwc.dtype:
        out = torch.empty_like(logits_ndhwc)
    ops.softmax_dice_bwd(logits_ndhwc, state.labels, state.coef, grad_scale, out,
                         scratch=state.partials if bias_grad is not None else None, bias_grad=bias_grad)
    return out


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, loss_mod):
        lg = as_ndhwc(logits)
        ctx.lg = lg
        ctx.mod = loss_mod
        ctx.state = loss_mod._state
        return dice_forward(loss_mod._state, lg, labels, loss_mod.smooth_nr, loss_mod.smooth_dr).clone()

    @staticmethod
    def backward(ctx, g):
        # g is the upstream scalar; fold it into the kernel on the host only when it is 1
        d = dice_backward(ctx.state, ctx.lg, 1.0)
        d = d.permute(0, 4, 1, 2, 3)
        gs = g.to(d.dtype)
        return (d if bool(gs == 1) else d * gs), None, None


class DiceLoss(torch.nn.Module):
    """Fused softmax + one-hot + Dice (MONAI defaults: include_background, smooth 1e-5, mean)."""

    def __init__(self, to_onehot_y: bool = True, softmax: bool = True, smooth_nr: float = 1e-5,
                 smooth_dr: float = 1e-5):
        super().__init__()
        if not (to_onehot_y and softmax):
            raise NotImplementedError("the HIP Dice kernel implements to_onehot_y=True, softmax=True "
                                      "(the configuration segmantic uses)")
        self.smooth_nr, self.smooth_dr = smooth_nr, smooth_dr
        self._state = _DiceState()

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """logits [N,K,D,H,W] (float32/bfloat16), labels [N,1,D,H,W] integer-valued."""
        if logits.requires_grad and torch.is_grad_enabled():
            return _DiceFn.apply(logits, labels, self)
        lg = as_ndhwc(logits)
        return dice_forward(self._state, lg, labels, self.smooth_nr, self.smooth_dr).clone()


class DiceMetric:
    """Label-overlap Dice: per class 2|P&T| / (|P|+|T|), NaN when the class is absent in T,
    nan-mean over classes then over the accumulated batch items (reduction="mean")."""

    def __init__(self, num_classes: int, include_background: bool = False):
        self.k = num_classes
        self.include_background = include_background
        self._items = []

    def __call__(self, pred_labels: torch.Tensor, true_labels: torch.Tensor) -> torch.Tensor:
        """pred/true: integer label volumes with a leading batch dim [N, ...]. Returns [N, C']."""
        out = []
        for b in range(pred_labels.shape[0]):
            p = pred_labels[b].reshape(-1).to(torch.int32).contiguous()
            t = true_labels[b].reshape(-1).to(device=p.device, dtype=torch.int32).contiguous()
            counts = torch.empty((self.k, 3), dtype=torch.int64, device=p.device)
            ops.label_counts(p, t, self.k, counts)
            c = counts.double()
            d = torch.where(c[:, 2] > 0, 2.0 * c[:, 0] / (c[:, 1] + c[:, 2]),
                            torch.full((self.k,), float("nan"), dtype=torch.double, device=p.device))
            if not self.include_background:
                d = d[1:]
            out.append(d.float())
        res = torch.stack(out)
        self._items.append(res)
        return res

    def aggregate(self) -> torch.Tensor:
        if not self._items:
            return torch.tensor(float("nan"))
        d = torch.cat(self._items)
        nn_ = ~torch.isnan(d)
        per_b = torch.where(nn_, d, torch.zeros_like(d)).sum(1) / nn_.sum(1).clamp(min=1)
        valid = nn_.sum(1) > 0
        # MONAI do_metric_reduction("mean"): 0 (not NaN) when no item has a valid class
        return per_b[valid].mean() if bool(valid.any()) else torch.zeros((), device=d.device)

    def reset(self):
        self._items = []


class ConfusionMatrixMetric:
    """MONAI ``ConfusionMatrixMetric(metric_name=[...])`` as configured at reference
    ``monai_unet.py:645-646`` (include_background=True, compute_sample=False, reduction="mean"):
    tp / fp / tn / fn per (volume, class) from the same ``segmi_label_counts`` pass the Dice metric
    uses, averaged over all accumulated (volume, class) items, then the ratios."""

    NAMES = ("sensitivity", "specificity", "precision", "accuracy")

    def __init__(self, num_classes: int, metric_name=NAMES):
        unknown = [m for m in metric_name if m not in self.NAMES]
        if unknown:
            raise NotImplementedError(f"confusion metrics implemented: {self.NAMES}, not {unknown}")
        self.k, self.metric_name = num_classes, list(metric_name)
        self._items = []

    def __call__(self, pred_labels: torch.Tensor, true_labels: torch.Tensor) -> torch.Tensor:
        """integer label volumes [N, ...] -> [N, K, 4] (tp, fp, tn, fn) float64."""
        out = []
        for b in range(pred_labels.shape[0]):
            p = pred_labels[b].reshape(-1).to(torch.int32).contiguous()
            t = true_labels[b].reshape(-1).to(device=p.device, dtype=torch.int32).contiguous()
            counts = torch.empty((self.k, 3), dtype=torch.int64, device=p.device)
            ops.label_counts(p, t, self.k, counts)
            c = counts.double()
            tp, fp, fn = c[:, 0], c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]
            tn = float(p.numel()) - tp - fp - fn
            out.append(torch.stack([tp, fp, tn, fn], 1))
        res = torch.stack(out)
        self._items.append(res)
        return res

    def aggregate(self):
        """-> one 0-d tensor per metric name (MONAI returns a list in ``metric_name`` order)."""
        if not self._items:
            return [torch.tensor(float("nan")) for _ in self.metric_name]
        tp, fp, tn, fn = torch.cat(self._items).mean((0, 1)).unbind()
        vals = {"sensitivity": tp / (tp + fn), "specificity": tn / (tn + fp),
                "precision": tp / (tp + fp), "accuracy": (tp + tn) / (tp + fp + tn + fn)}
        return [vals[m].float() for m in self.metric_name]

    def reset(self):
        self._items = []
