"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the reference's
``SelectBestEnsemble.__call__`` (``src/segmantic/seg/transforms.py:39-61``): per (tissue, model) pair
of ``label_model_dict``, in insertion order, ``out[pred[model] == tissue] = tissue``; multi-channel
(one-hot) inputs are arg-maxed first and the result is one-hot encoded again with
``num_classes = max(tissue ids) + 1``.

Pinned by the reference's own vector ``tests/seg/test_transforms.py:9-43`` (committed as
``tests/golden/reference_select_best.json``; ``tests/test_oracle.py``).  Voxels that no pair selects
are uninitialised memory in the reference (``torch.empty``); here they are 0, as in the HIP kernel.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch


def ref_select_best(preds: Sequence[torch.Tensor], label_model_dict: Dict[int, int]) -> torch.Tensor:
    """preds: E tensors [C, ...] -- C == 1: label maps, C > 1: one-hot / score maps.  Returns the
    combined map in the same form (float32, as the reference's ``torch.empty`` default dtype)."""
    img = torch.stack([torch.as_tensor(p) for p in preds])          # [E, C, ...]
    has_ch = img.dim() > 1 and img.shape[1] > 1
    if has_ch:
        img = torch.argmax(img, dim=1, keepdim=True)
    out = torch.zeros(img.shape[1:], dtype=torch.float32)
    for tissue, model in label_model_dict.items():
        out[img[model] == tissue] = float(tissue)
    if has_ch:
        k = max(label_model_dict.keys()) + 1
        oh = torch.zeros((k,) + tuple(out.shape[1:]), dtype=torch.float32)
        oh.scatter_(0, out.long(), 1.0)
        return oh
    return out
