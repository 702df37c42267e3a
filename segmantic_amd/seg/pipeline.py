"""Volume pre / post-processing around the network, on device.

Restates the reference's ``default_preprocessing`` (``src/segmantic/seg/monai_unet.py:151-176``)
and the predict-time inversion chain (``:612-625``) without MONAI:

  load (NIfTI / MetaImage / NRRD, channel first, MONAI axis order [C, x, y, z])  -> Orientation("RAS")
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
from ..data.imageio import read_image, strip_image_suffix, write_image


# ---------------------------------------------------------------------------- orientation
def io_orientation(affine: np.ndarray) -> np.ndarray:
    """(closest world axis, sign) per voxel axis, by the rule MONAI's ``Orientation`` takes from
    nibabel (``io_orientation``): the shear-free part of the direction matrix (polar factor of the
    unit-column matrix), then for voxel axis 0, 1, 2 in turn the world axis with the largest
    |component| among those not taken yet."""
    rzs = np.asarray(affine, np.float64)[:3, :3]
    norms = np.sqrt((rzs * rzs).sum(0))
    norms[norms == 0] = 1.0
    u, sv, vt = np.linalg.svd(rzs / norms, full_matrices=False)
    keep = sv > sv.max() * 3 * np.finfo(np.float64).eps
    rot = u[:, keep] @ vt[keep]
    ornt = np.zeros((3, 2), dtype=np.int64)
    free = np.ones(3, dtype=bool)
    for v in range(3):
        col = np.where(free, np.abs(rot[:, v]), -1.0)
        w = int(np.argmax(col))
        if col[w] <= 1e-8 * max(1.0, float(np.abs(rot).max())):
            raise ValueError("degenerate affine: a voxel axis has no world direction")
        ornt[v] = (w, -1 if rot[w, v] < 0 else 1)
        free[w] = False
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


def spacing_geometry(affine: np.ndarray, shape: Sequence[int], pixdim: Sequence[float]):
    """Output grid of MONAI ``Spacing(pixdim)`` (``diagonal=False``, ``scale_extent=False``) for an
    input grid (affine, shape): (new affine, new shape).

    The zooms of the direction matrix are replaced by ``pixdim`` while its rotation is kept (the
    upper-triangular Cholesky factor Z of M^T M carries zooms and shear: M = R Z, new M = R
    diag(sign(Z_ii) pixdim)); the new extent is the bounding box of the 8 old corner voxels in new
    index space, ``round(ptp + 1)`` voxels; the new origin is the corner that is minimal there."""
    A = np.asarray(affine, np.float64)
    in_sp = _affine_spacing(A)
    pix = [float(v) for v in pixdim][:3]
    out_sp = np.asarray(pix + list(in_sp[len(pix):]), np.float64)
    out_sp[out_sp == 0] = 1.0
    M = A[:3, :3]
    Z = np.linalg.cholesky(M.T @ M).T
    new = np.eye(4)
    new[:3, :3] = (M @ np.linalg.inv(Z)) @ np.diag(np.sign(np.diag(Z)) * np.abs(out_sp))
    n = np.asarray(shape, np.float64)
    corners = np.array([[i * (n[0] - 1), j * (n[1] - 1), k * (n[2] - 1), 1.0]
                        for i in (0, 1) for j in (0, 1) for k in (0, 1)]).T
    idx_new = np.linalg.solve(new, A @ corners)[:3]
    new_shape = [int(v) for v in np.round(idx_new.max(1) - idx_new.min(1) + 1.0)]
    # the corner that is minimal in every dimension of the new index space, within 1e-3 voxel
    # (MONAI's AFFINE_TOL; affines read from f32 NIfTI headers carry a 1e-7 shear after rotation)
    excess = np.array([-(idx_new - idx_new[:, c:c + 1]).min(1).min() for c in range(8)])
    first = int(np.argmin(excess))
    if excess[first] > 1e-3:
        raise ValueError("Spacing: affine with a shear that leaves no minimal corner")
    new[:3, 3] = (A @ corners)[:3, first]
    return new, new_shape


def affine_resample(vol: torch.Tensor, src_affine: np.ndarray, dst_affine: np.ndarray,
                    dst_shape: Sequence[int], nearest: bool = False) -> torch.Tensor:
    """MONAI ``SpatialResample`` (bilinear, border padding): destination voxel i reads the source
    at continuous index ``inv(src_affine) @ dst_affine @ i`` (the ``align_corners`` flag cancels
    between MONAI's index normalisation and ``grid_sample``, see ``oracle/pipeline_ref.py``).
    vol [C, x, y, z] float32 on device."""
    X = np.linalg.solve(np.asarray(src_affine, np.float64), np.asarray(dst_affine, np.float64))
    # kernel arrays are [z][y][x] = our dims (d0, d1, d2) -> kernel x = d2, y = d1, z = d0
    m = np.zeros((3, 4))
    for r in range(3):
        for c in range(3):
            m[r, c] = X[2 - r, 2 - c]
        m[r, 3] = X[2 - r, 3]
    outs = [ops.resample3d(vol[c].contiguous(), list(dst_shape), m, nearest=nearest, border=True, half_even=True)
            for c in range(vol.shape[0])]
    return torch.stack(outs)


def spacing_resample(vol: torch.Tensor, affine: np.ndarray, pixdim: Sequence[float], nearest: bool = False):
    """``Spacingd(pixdim)`` forward (reference ``monai_unet.py:173-174``): (volume, new affine).
    ``nearest``: a configured ``mode="nearest"`` for this key (the reference default is bilinear)."""
    new_affine, new_shape = spacing_geometry(affine, vol.shape[1:], pixdim)
    return affine_resample(vol, affine, new_affine, new_shape, nearest=nearest), new_affine


# ---------------------------------------------------------------------------- pipeline
class PredictPipeline:
    """default_preprocessing + the inversion chain of ``predict`` (one volume at a time)."""

    def __init__(self, device, spacing: Sequence[float] = (), with_label: bool = False,
                 label_nearest: bool = False):
        self.device = torch.device(device)
        self.spacing = list(spacing) if spacing else []
        self.with_label = with_label
        self.label_nearest = bool(label_nearest)       # Spacingd(mode=[bilinear, nearest]) from a bundle config

    # -- forward chain ----------------------------------------------------------------------
    def _load(self, path) -> tuple:
        arr, A = read_image(path)
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
            if lab is not None:   # the reference's default resamples the label with the same (bilinear) mode
                lab, _ = spacing_resample(lab, A, self.spacing, nearest=self.label_nearest)
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
            # inverse Spacing (Spacing.inverse = the same resample with the two grids swapped)
            lg = affine_resample(lg, item["affine"], item["affine_crop"], item["shape_crop"])
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
        # MONAI SaveImaged(output_postfix="", separate_folder=False): <stem>.nii.gz (the default output_ext)
        out = Path(output_dir) / f"{strip_image_suffix(item['path'].name)}.nii.gz"
        arr = label_vol.cpu().numpy().transpose(2, 1, 0)            # [z,y,x]
        write_image(out, arr, item["affine0"])
        return out
