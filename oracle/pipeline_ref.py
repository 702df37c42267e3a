"""CPU (numpy, float64 index math) restatement of the volume pre-/post-processing chain around the
network in ``predict``.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Reference call sites (``src/segmantic/seg/monai_unet.py``):

* ``default_preprocessing`` ``:151-176``: ``LoadImaged(ensure_channel_first)`` ->
  ``Orientationd("RAS")`` -> ``NormalizeIntensityd(channel_wise)`` -> ``CropForegroundd(source>0)``
  -> ``EnsureTyped(float32)`` -> optional ``Spacingd(pixdim)`` (defaults: bilinear, border padding,
  ``align_corners=False``, ``diagonal=False``, ``scale_extent=False``) -- for image AND label keys.
* post chain ``:612-625``: ``Invertd(pre_transforms, orig_keys="image", nearest_interp=False)`` on
  the K-channel logits (inverse Spacing with the forward's bilinear/border mode, inverse crop =
  zero padding, inverse orientation) -> ``AsDiscreted(argmax=True)``.

The arithmetic lives in MONAI (>= 1.2, unpinned, absent here) and, for the orientation codes, in
nibabel.  Their published algorithms as restated below:

``nibabel.orientations.io_orientation``: RS = RZS / column norms; polar decomposition R = P @ Qs of
  RS (SVD); for input axis 0,1,2 in turn: output axis = argmax |R[:, in]| among rows not yet used
  (used rows are zeroed), flip = sign of that entry.
``nibabel.orientations.inv_ornt_aff`` and MONAI ``Orientation``: new_affine = affine @
  inv_ornt_aff(ornt, shape); data = flip(input axes with flip -1) then transpose so that output
  axis w takes the input axis whose ornt[:, 0] == w.
MONAI ``Spacing.__call__``: new_affine = ``zoom_affine(affine, pixdim, diagonal=False)`` (rotation
  part of the RZS kept via Cholesky of RZS^T RZS, zooms replaced); ``compute_shape_offset``:
  out_shape = round(ptp(inv(new) @ old @ corners) + 1), offset = the world position of the corner
  that is minimal in the new index space; then ``SpatialResample``: output voxel i samples the
  input at continuous index ``inv(old_affine) @ new_affine @ i``.

  **Spacing convention** (the question VERDICT r1 raised): ``SpatialResample`` converts the index
  map to normalised ``grid_sample`` coordinates with ``to_norm_affine(..., align_corners=
  align_corners)`` and samples with the *same* ``align_corners`` flag, so the two normalisations
  cancel exactly: whatever ``align_corners`` is, output voxel i reads continuous input index
  ``xform @ i`` (voxel-centre to voxel-centre; index 0 maps to index 0 for an axis-aligned
  zoom because the offset is the minimal corner).  ``padding_mode="border"`` clamps the
  *coordinate* to [0, n-1] before the 8-tap trilinear interpolation.
``Spacing.inverse``: the same resample with src/dst affines swapped and ``spatial_size`` = the
  size recorded before the forward call.

PARITY UNPINNED: no reference test pins a resampled or inverted voxel value.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------- orientation
def ref_io_orientation(affine: np.ndarray) -> np.ndarray:
    affine = np.asarray(affine, np.float64)
    RZS = affine[:3, :3]
    zooms = np.sqrt(np.sum(RZS * RZS, axis=0))
    zooms[zooms == 0] = 1
    RS = RZS / zooms
    P, S, Qs = np.linalg.svd(RS, full_matrices=False)
    tol = S.max() * 3 * np.finfo(S.dtype).eps
    keep = S > tol
    R = np.dot(P[:, keep], Qs[keep])
    ornt = np.ones((3, 2)) * np.nan
    for in_ax in range(3):
        col = R[:, in_ax]
        if not np.allclose(col, 0):
            out_ax = int(np.argmax(np.abs(col)))
            ornt[in_ax, 0] = out_ax
            ornt[in_ax, 1] = -1 if col[out_ax] < 0 else 1
            R[out_ax, :] = 0
    return ornt


def ref_inv_ornt_aff(ornt: np.ndarray, shape: Sequence[int]) -> np.ndarray:
    ornt = np.asarray(ornt)
    p = ornt.shape[0]
    shape = np.array(shape[:p], np.float64)
    axis_transpose = [int(v) for v in ornt[:, 0]]
    flips = ornt[:, 1]
    undo_reorder = np.eye(p + 1)[axis_transpose + [p], :]
    undo_flip = np.diag(list(flips) + [1.0])
    center_trans = -(shape - 1) / 2.0
    undo_flip[:p, p] = (flips * center_trans) - center_trans
    return undo_flip @ undo_reorder


def ref_to_ras(vol: np.ndarray, affine: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """vol [C, d0, d1, d2] -> (RAS volume, new affine, ornt)."""
    ornt = ref_io_orientation(affine)
    new_affine = np.asarray(affine, np.float64) @ ref_inv_ornt_aff(ornt, vol.shape[1:])
    out = vol
    for ax in range(3):
        if ornt[ax, 1] == -1:
            out = np.flip(out, axis=1 + ax)
    order = np.argsort(ornt[:, 0])
    out = np.transpose(out, [0] + [1 + int(o) for o in order])
    return np.ascontiguousarray(out), new_affine, ornt


def ref_from_ras(vol: np.ndarray, ornt: np.ndarray) -> np.ndarray:
    """inverse of ``ref_to_ras`` on the data."""
    order = np.argsort(ornt[:, 0])
    inv = np.argsort(order)
    out = np.transpose(vol, [0] + [1 + int(o) for o in inv])
    for ax in range(3):
        if ornt[ax, 1] == -1:
            out = np.flip(out, axis=1 + ax)
    return np.ascontiguousarray(out)


# ------------------------------------------------------------------------------- intensity / crop
def ref_normalize(vol: np.ndarray) -> np.ndarray:
    out = np.empty_like(vol, dtype=np.float32)
    for c in range(vol.shape[0]):
        x = vol[c].astype(np.float64)
        m, s = x.mean(), x.std()
        out[c] = ((x - m) / (s if s != 0 else 1.0)).astype(np.float32)
    return out


def ref_foreground_bbox(src: np.ndarray):
    fg = (src > 0).any(0)
    nz = np.argwhere(fg)
    if nz.size == 0:
        return [0, 0, 0], list(fg.shape)
    return [int(v) for v in nz.min(0)], [int(v) + 1 for v in nz.max(0)]


# ------------------------------------------------------------------------------- spacing
def ref_zoom_affine(affine: np.ndarray, scale: Sequence[float]) -> np.ndarray:
    affine = np.asarray(affine, np.float64)
    norm = np.sqrt((affine[:3, :3] ** 2).sum(0))
    scale_np = np.asarray(list(scale)[:3] + list(norm[len(list(scale)[:3]):]), np.float64)
    scale_np[scale_np == 0] = 1.0
    rzs = affine[:3, :3]
    zs = np.linalg.cholesky(rzs.T @ rzs).T
    rotation = rzs @ np.linalg.inv(zs)
    s = np.sign(np.diag(zs)) * np.abs(scale_np)
    new_affine = np.eye(4)
    new_affine[:3, :3] = rotation @ np.diag(s)
    return new_affine


def ref_compute_shape_offset(shape, in_affine, out_affine):
    shape = np.asarray(shape, np.float64)
    in_coords = [(0.0, d - 1.0) for d in shape]
    corners = np.asarray(np.meshgrid(*in_coords, indexing="ij")).reshape((3, -1))
    corners = np.concatenate((corners, np.ones_like(corners[:1])))
    corners_out = np.linalg.solve(out_affine, in_affine) @ corners
    corners_w = in_affine @ corners
    all_dist = corners_out[:-1].copy()
    corners_out = corners_out[:-1] / corners_out[-1]
    out_shape = np.round(np.ptp(corners_out, axis=1) + 1.0)
    # "corner is the smallest, shift the corner to origin": the first corner whose distance to every
    # other corner is non-negative in all dimensions, with MONAI's AFFINE_TOL = 1e-3 taken as an
    # absolute tolerance in voxels (rotated affines from f32 NIfTI headers are sheared by ~1e-7)
    offset = None
    for i in range(corners_w.shape[1]):
        min_corner = np.min(all_dist - all_dist[:, i:i + 1], 1)
        if np.all(min_corner > -1e-3):
            offset = corners_w[:3, i]
            break
    assert offset is not None
    return out_shape.astype(int), offset


def ref_trilinear_border(vol: np.ndarray, xform: np.ndarray, out_shape) -> np.ndarray:
    """vol [C, d0, d1, d2] float32; xform 4x4 out index -> in index; border padding (coordinate
    clamped to [0, n-1]); corner accumulation order: bit d of the corner index = dimension d
    counted from the FASTEST axis (d2), weights multiplied d2, d1, d0 -- float64 throughout."""
    C = vol.shape[0]
    n = np.asarray(vol.shape[1:])
    idx = np.indices(tuple(int(v) for v in out_shape)).reshape(3, -1).astype(np.float64)
    src = xform[:3, :3] @ idx + xform[:3, 3:4]
    for d in range(3):
        src[d] = np.clip(src[d], 0.0, n[d] - 1.0)
    f0 = np.floor(src)
    fr = src - f0
    b = f0.astype(np.int64)
    lo = np.clip(b, 0, (n - 1)[:, None])
    hi = np.clip(b + 1, 0, (n - 1)[:, None])
    out = np.zeros((C, idx.shape[1]), np.float64)
    v64 = vol.astype(np.float64)
    for corner in range(8):
        # fastest axis (d2) is bit 0
        sel = [(corner >> (2 - d)) & 1 for d in range(3)]
        w = np.where(sel[2], fr[2], 1.0 - fr[2])
        w = w * np.where(sel[1], fr[1], 1.0 - fr[1])
        w = w * np.where(sel[0], fr[0], 1.0 - fr[0])
        ii = [hi[d] if sel[d] else lo[d] for d in range(3)]
        out = out + w[None] * v64[:, ii[0], ii[1], ii[2]]
    return out.reshape((C,) + tuple(int(v) for v in out_shape)).astype(np.float32)


def ref_nearest_border(vol: np.ndarray, xform: np.ndarray, out_shape) -> np.ndarray:
    """``mode="nearest"`` of the same resampler: torch ``grid_sample`` takes ``nearbyint`` of the
    (border-clamped) source coordinate -- round half to even, ``np.rint``."""
    n = np.asarray(vol.shape[1:])
    idx = np.indices(tuple(int(v) for v in out_shape)).reshape(3, -1).astype(np.float64)
    src = xform[:3, :3] @ idx + xform[:3, 3:4]
    ii = [np.clip(np.rint(np.clip(src[d], 0.0, n[d] - 1.0)).astype(np.int64), 0, n[d] - 1) for d in range(3)]
    return vol[:, ii[0], ii[1], ii[2]].reshape((vol.shape[0],) + tuple(int(v) for v in out_shape)).astype(np.float32)


def ref_spacing(vol: np.ndarray, affine: np.ndarray, pixdim: Sequence[float], nearest: bool = False):
    """MONAI ``Spacing(pixdim)`` forward: (resampled [C,...] float32, new affine)."""
    new_affine = ref_zoom_affine(affine, pixdim)
    out_shape, offset = ref_compute_shape_offset(vol.shape[1:], affine, new_affine)
    new_affine[:3, 3] = offset
    xform = np.linalg.solve(np.asarray(affine, np.float64), new_affine)
    fn = ref_nearest_border if nearest else ref_trilinear_border
    return fn(vol, xform, out_shape), new_affine


def ref_spacing_inverse(vol: np.ndarray, cur_affine: np.ndarray, orig_affine: np.ndarray, orig_shape):
    xform = np.linalg.solve(np.asarray(cur_affine, np.float64), np.asarray(orig_affine, np.float64))
    return ref_trilinear_border(vol, xform, orig_shape)


# ------------------------------------------------------------------------------- whole chain
def ref_preprocess(image: np.ndarray, affine: np.ndarray, spacing: Sequence[float] = (),
                   label: Optional[np.ndarray] = None, label_affine: Optional[np.ndarray] = None) -> Dict:
    """image / label [C, d0, d1, d2] in MONAI's LoadImage axis order (= NIfTI i, j, k)."""
    img, A, ornt = ref_to_ras(image.astype(np.float32), affine)
    full = img.shape[1:]
    img = ref_normalize(img)
    lab = None
    if label is not None:
        lab, _, _ = ref_to_ras(label.astype(np.float32), label_affine if label_affine is not None else affine)
    lo, hi = ref_foreground_bbox(lab if lab is not None else img)
    sl = (slice(None),) + tuple(slice(l, h) for l, h in zip(lo, hi))
    img = np.ascontiguousarray(img[sl])
    A_crop = A.copy()
    A_crop[:3, 3] = A[:3, 3] + A[:3, :3] @ np.asarray(lo, np.float64)
    rec = {"ornt": ornt, "crop": (lo, hi, tuple(full)), "affine_crop": A_crop,
           "shape_crop": tuple(img.shape[1:])}
    if lab is not None:
        lab = np.ascontiguousarray(lab[sl])
    if len(spacing):
        img, A2 = ref_spacing(img, A_crop, spacing)
        if lab is not None:
            lab, _ = ref_spacing(lab, A_crop, spacing)
        rec["affine"] = A2
    rec["image"] = img
    if lab is not None:
        rec["label"] = lab
    return rec


def ref_invert_and_discretize(logits: np.ndarray, rec: Dict) -> np.ndarray:
    """logits [K, ...] in the pre-processed grid -> label volume in the SOURCE grid (first max
    wins, ``torch.argmax`` / ``AsDiscrete(argmax=True)``)."""
    lg = logits.astype(np.float32)
    if "affine" in rec:
        lg = ref_spacing_inverse(lg, rec["affine"], rec["affine_crop"], rec["shape_crop"])
    lo, hi, full = rec["crop"]
    out = np.zeros((lg.shape[0],) + tuple(full), np.float32)
    out[(slice(None),) + tuple(slice(l, h) for l, h in zip(lo, hi))] = lg
    out = ref_from_ras(out, rec["ornt"])
    return np.argmax(out, axis=0)
