"""Volume pre / post-processing around the network, on device.

Restates the reference's ``default_preprocessing`` (``src/segmantic/seg/monai_unet.py:151-176``)
and the predict-time inversion chain (``:612-625``) without MONAI:

  load (NIfTI, channel first, MONAI axis order [C, x, y, z])  -> Orientation("RAS")
  -> NormalizeIntensity(channel_wise)   [HIP kernel]          -> CropForeground(source > 0)
  -> float32                            -> optional Spacing(pixdim)  [HIP trilinear resample]

and for predictions:  invert Spacing (trilinear on the K-channel logits) -> invert the crop
(zero padding) -> invert the orientation -> argmax [HIP kernel] -> save with the source affine.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from ..data.nifti import read_nifti, write_nifti


# ---------------------------------------------------------------------------- orientation
def io_orientation(affine: np.ndarray) -> np.ndarray:
    """For each voxel axis: (closest world axis, sign) -- the axis-aligned part of an affine."""
    R = np.asarray(affine, np.float64)[:3, :3]
    R = R / np.maximum(np.sqrt((R ** 2).sum(0)), 1e-12)
    ornt = np.zeros((3, 2), dtype=np.int64)
    used_w, used_v = set(), set()
    for _ in range(3):
        best = None
        for v in range(3):
            if v in used_v:
                continue
            for w in range(3):
                if w in used_w:
                    continue
                if best is None or abs(R[w, v]) > best[0]:
                    best = (abs(R[w, v]), v, w)
        _, v, w = best
        ornt[v] = (w, 1 if R[w, v] >= 0 else -1)
        used_v.add(v)
        used_w.add(w)
    return ornt


def to_ras(vol: torch.Tensor, affine: np.ndarray):
    """vol [C, x, y, z] -> RAS-oriented volume, new affine, and the record needed to invert."""
    ornt = io_orientation(affine)
    flips = [d for d in range(3) if ornt[d, 1] < 0]
    A = np.array(affine, np.float64)
    shape = list(vol.shape[1:])
    if flips:
        vol = torch.flip(vol, [1 + d for d in flips])
        for d in flips:
            A[:3, 3] = A[:3, 3] + A[:3, d] * (shape[d] - 1)
            A[:3, d] = -A[:3, d]
    perm = [int(np.where(ornt[:, 0] == w)[0][0]) for w in range(3)]   # world axis w <- voxel axis
    vol = vol.permute([0] + [1 + p for p in perm]).contiguous()
    A = A[:, perm + [3]]
    return vol, A, {"flips": flips, "perm": perm}


def from_ras(vol: torch.Tensor, rec: Dict) -> torch.Tensor:
    inv = [0] * 3
    for w, p in enumerate(rec["perm"]):
        inv[p] = w
    vol = vol.permute([0] + [1 + i for i in inv])
    if rec["flips"]:
        vol = torch.flip(vol, [1 + d for d in rec["flips"]])
    return vol.contiguous()


# ---------------------------------------------------------------------------- spacing
def _affine_spacing(A: np.ndarray) -> np.ndarray:
    return np.sqrt((np.asarray(A, np.float64)[:3, :3] ** 2).sum(0))


def spacing_resample(vol: torch.Tensor, affine: np.ndarray, pixdim: Sequence[float],
                     out_shape: Optional[Sequence[int]] = None):
    """MONAI ``Spacing`` geometry: out_shape = round((n - 1) * in_sp / out_sp + 1), voxel i_out
    sits at continuous input index i_out * out_sp / in_sp; trilinear, edge replicate.
    vol [C, x, y, z] float32 on device."""
    in_sp = _affine_spacing(affine)
    out_sp = np.asarray(list(pixdim) + [1.0] * 3, np.float64)[:3]
    n = np.asarray(vol.shape[1:], np.float64)
    if out_shape is None:
        out_shape = [int(v) for v in np.round((n - 1) * in_sp / out_sp + 1.0)]
    m = np.zeros((3, 4))
    # kernel arrays are [z][y][x] = our dims (d0, d1, d2) -> kernel x = d2, y = d1, z = d0
    ratios = out_sp / in_sp
    m[0, 0], m[1, 1], m[2, 2] = ratios[2], ratios[1], ratios[0]
    outs = [ops.resample3d(vol[c].contiguous(), out_shape, m, nearest=False) for c in range(vol.shape[0])]
    A = np.array(affine, np.float64)
    A[:3, :3] = A[:3, :3] @ np.diag(ratios)
    return torch.stack(outs), A


# ---------------------------------------------------------------------------- pipeline
class PredictPipeline:
    """default_preprocessing + the inversion chain of ``predict`` (one volume at a time)."""

    def __init__(self, device, spacing: Sequence[float] = (), with_label: bool = False):
        self.device = torch.device(device)
        self.spacing = list(spacing) if spacing else []
        self.with_label = with_label

    # -- forward chain ----------------------------------------------------------------------
    def _load(self, path) -> tuple:
        arr, A = read_nifti(path)
        if arr.ndim == 3:
            vol = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 1, 0)))[None]   # [1,x,y,z]
        else:
            vol = torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 2, 1)))
        return vol, A

    def load(self, image_path, label_path=None) -> Dict:
        img, A0 = self._load(image_path)
        item: Dict = {"path": Path(image_path), "affine0": A0, "shape0": tuple(img.shape[1:])}
        img = img.to(self.device, torch.float32)
        img, A, rec = to_ras(img, A0)
        item["ornt"] = rec
        ops.normalize_intensity_(img)
        lab = None
        if label_path is not None:
            lab, Al = self._load(label_path)
            lab, _, _ = to_ras(lab.to(self.device, torch.float32), Al)
        src = lab if lab is not None else img
        fg = (src > 0).any(0)
        nz = torch.nonzero(fg)
        if nz.numel() == 0:
            lo, hi = [0, 0, 0], list(fg.shape)
        else:
            lo = [int(v) for v in nz.min(0).values]
            hi = [int(v) + 1 for v in nz.max(0).values]
        item["crop"] = (lo, hi, tuple(img.shape[1:]))
        sl = (slice(None),) + tuple(slice(l, h) for l, h in zip(lo, hi))
        img = img[sl].contiguous()
        A = A.copy()
        A[:3, 3] = A[:3, 3] + A[:3, :3] @ np.asarray(lo, np.float64)
        if lab is not None:
            lab = lab[sl].contiguous()
        item["affine_crop"] = A
        item["shape_crop"] = tuple(img.shape[1:])
        if self.spacing:
            img, A2 = spacing_resample(img, A, self.spacing)
            if lab is not None:   # the reference resamples the label with the same (bilinear) mode
                lab, _ = spacing_resample(lab, A, self.spacing)
            item["affine"] = A2
        item["image"] = img
        if lab is not None:
            item["label"] = lab
        return item

    # -- inverse chain ----------------------------------------------------------------------
    def invert_and_discretize(self, logits: torch.Tensor, item: Dict) -> torch.Tensor:
        """logits [K, x, y, z] (any storage) -> label volume [x0, y0, z0] uint8/int16 in the
        source image's voxel grid."""
        lg = logits.float().contiguous()
        if self.spacing:
            # inverse Spacing: back to the cropped grid
            inv_pix = _affine_spacing(item["affine_crop"])
            lg, _ = spacing_resample(lg, item["affine"], inv_pix, out_shape=item["shape_crop"])
        lo, hi, full = item["crop"]
        K = lg.shape[0]
        out = torch.zeros((K,) + tuple(full), dtype=torch.float32, device=lg.device)
        out[(slice(None),) + tuple(slice(l, h) for l, h in zip(lo, hi))] = lg
        out = from_ras(out, item["ornt"])
        nd = out.permute(1, 2, 3, 0).contiguous()[None]            # NDHWC
        lab = torch.empty(nd.shape[1:4], dtype=torch.uint8 if K <= 256 else torch.int16,
                          device=out.device)
        ops.argmax(nd, lab)
        return lab

    def save(self, label_vol: torch.Tensor, item: Dict, output_dir: Path) -> Path:
        name = item["path"].name
        stem = name[:-7] if name.endswith(".nii.gz") else Path(name).stem
        out = Path(output_dir) / f"{stem}.nii.gz"
        arr = label_vol.cpu().numpy().transpose(2, 1, 0)            # [z,y,x]
        write_nifti(out, arr, item["affine0"])
        return out
