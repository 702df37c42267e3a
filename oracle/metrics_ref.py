"""Oracle (test infrastructure): discretisation, Dice metric, intensity normalisation.

Reference call sites in ``src/segmantic/seg/monai_unet.py``:
  * ``:129-134`` / ``:622`` / ``:673``  AsDiscrete(argmax=True[, to_onehot=K]) -- channel argmax,
    first index wins on ties (torch.argmax semantics)                     -> ``ref_argmax``
  * ``:136-138`` / ``:642-644``  DiceMetric(include_background=False, reduction="mean")
    per class 2|y^ & y| / (|y^| + |y|), NaN when the class is absent in y, nan-mean over
    classes then batch                                                     -> ``ref_dice_metric``
  * ``:164``  NormalizeIntensityd(nonzero=False, channel_wise=True): (x - mean) / std with the
    population std (ddof=0), std == 0 -> divide by 1                       -> ``ref_normalize``
PARITY UNPINNED (MONAI absent; the reference's tests never check these values).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def ref_argmax(logits: torch.Tensor) -> torch.Tensor:
    """[B,K,*sp] -> [B,1,*sp] int64, first max index on ties."""
    return torch.argmax(logits, dim=1, keepdim=True)


def ref_dice_metric(pred_label: torch.Tensor, true_label: torch.Tensor, num_classes: int,
                    include_background: bool = False):
    """pred_label / true_label [B,1,*sp] integer labels.  Returns (per [B,C'] dice, mean)."""
    p = F.one_hot(pred_label[:, 0].long(), num_classes).movedim(-1, 1).float()
    t = F.one_hot(true_label[:, 0].long(), num_classes).movedim(-1, 1).float()
    if not include_background:
        p, t = p[:, 1:], t[:, 1:]
    ax = list(range(2, p.dim()))
    inter = (p * t).sum(ax)
    yo = t.sum(ax)
    po = p.sum(ax)
    d = torch.where(yo > 0, 2.0 * inter / (yo + po), torch.full_like(yo, float("nan")))
    # reduction "mean": nan-mean over classes, then nan-mean over batch
    not_nan = ~torch.isnan(d)
    d0 = torch.where(not_nan, d, torch.zeros_like(d))
    per_b = d0.sum(1) / not_nan.sum(1).clamp(min=1)
    valid_b = not_nan.sum(1) > 0
    # MONAI do_metric_reduction: where no batch item has a valid class the result is 0, not NaN
    mean = per_b[valid_b].mean() if valid_b.any() else torch.tensor(0.0)
    return d, mean


def ref_confusion_metrics(pred_label: torch.Tensor, true_label: torch.Tensor, num_classes: int):
    """``ConfusionMatrixMetric(metric_name=["sensitivity", "specificity", "precision", "accuracy"])``
    as the reference builds it (``monai_unet.py:645-646``; MONAI defaults include_background=True,
    compute_sample=False, reduction="mean"): per (batch item, class) tp / fp / tn / fn of the
    one-hot volumes, averaged over batch and classes FIRST, then
    sensitivity = tp/(tp+fn), specificity = tn/(tn+fp), precision = tp/(tp+fp),
    accuracy = (tp+tn)/(tp+fp+tn+fn).  pred/true [B,1,*sp] integer labels -> 4 floats."""
    p = F.one_hot(pred_label[:, 0].long(), num_classes).movedim(-1, 1).double()
    t = F.one_hot(true_label[:, 0].long(), num_classes).movedim(-1, 1).double()
    ax = list(range(2, p.dim()))
    tp = (p * t).sum(ax)
    fp = (p * (1 - t)).sum(ax)
    fn = ((1 - p) * t).sum(ax)
    tn = ((1 - p) * (1 - t)).sum(ax)
    tp, fp, tn, fn = (float(v.mean()) for v in (tp, fp, tn, fn))
    div = lambda a, b: a / b if b != 0 else float("nan")
    return (div(tp, tp + fn), div(tn, tn + fp), div(tp, tp + fp), div(tp + tn, tp + fp + tn + fn))


def ref_normalize(x: np.ndarray) -> np.ndarray:
    """x [C,*sp] float32 -> channel-wise (x - mean) / std (population std; 0 -> 1)."""
    out = np.empty_like(x, dtype=np.float32)
    for c in range(x.shape[0]):
        v = torch.from_numpy(np.ascontiguousarray(x[c])).float()
        m = v.mean()
        s = v.std(unbiased=False)
        s = s if float(s) != 0.0 else torch.tensor(1.0)
        out[c] = ((v - m) / s).numpy()
    return out
