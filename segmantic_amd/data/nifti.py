"""Minimal NIfTI-1 (.nii / .nii.gz) reader / writer in numpy (nibabel / ITK are not available).

Replaces the ``LoadImaged(reader="ITKReader")`` / ``SaveImaged(writer="ITKWriter")`` ends of the
reference pipeline (``src/segmantic/seg/monai_unet.py:157-162, 600-608``) for the common case
of single-file NIfTI volumes.  Arrays are returned [z, y, x]; the 4x4 affine maps voxel index
(i=x, j=y, k=z) to RAS+ millimetres (sform if set, else qform, else pixdim scaling).
"""
from __future__ import annotations

import gzip
import struct
from pathlib import Path
from typing import Tuple

import numpy as np

_DT = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8,
       512: np.uint16, 768: np.uint32}
_CODE = {np.dtype(v).name: k for k, v in _DT.items()}


def _open(path: Path, mode: str):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quat_affine(b, c, d, qx, qy, qz, dx, dy, dz, qfac):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    A = np.eye(4)
    A[:3, :3] = R @ np.diag([dx, dy, dz * (qfac if qfac else 1.0)])
    A[:3, 3] = [qx, qy, qz]
    return A


def read_nifti(path) -> Tuple[np.ndarray, np.ndarray]:
    """-> (array [z,y,x] (or [t,z,y,x]), affine 4x4)"""
    path = Path(path)
    with _open(path, "rb") as f:
        raw = f.read()
    hdr = raw[:348]
    end = "<" if struct.unpack("<i", hdr[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", hdr[:4])[0] != 348:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    dim = struct.unpack(end + "8h", hdr[40:56])
    datatype, bitpix = struct.unpack(end + "hh", hdr[70:74])
    pixdim = struct.unpack(end + "8f", hdr[76:108])
    vox_offset = int(struct.unpack(end + "f", hdr[108:112])[0])
    slope, inter = struct.unpack(end + "ff", hdr[112:120])
    qform_code, sform_code = struct.unpack(end + "hh", hdr[252:256])
    quat = struct.unpack(end + "6f", hdr[256:280])
    srow = np.array(struct.unpack(end + "12f", hdr[280:328])).reshape(3, 4)
    if datatype not in _DT:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    nd = dim[0]
    shape = [int(d) for d in dim[1:1 + nd]]
    n = int(np.prod(shape))
    dt = np.dtype(_DT[datatype]).newbyteorder(end)
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=max(vox_offset, 352)).astype(_DT[datatype])
    arr = arr.reshape(shape[::-1])          # file order is x fastest -> [.., z, y, x]
    while arr.ndim > 3 and arr.shape[0] == 1:
        arr = arr[0]
    if slope not in (0.0, 1.0) or inter != 0.0:
        if slope != 0.0:
            arr = arr.astype(np.float32) * np.float32(slope) + np.float32(inter)
    if sform_code > 0:
        A = np.eye(4)
        A[:3, :] = srow
    elif qform_code > 0:
        A = _quat_affine(*quat, pixdim[1], pixdim[2], pixdim[3], pixdim[0])
    else:
        A = np.diag([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0, 1.0])
    return np.ascontiguousarray(arr), A


def write_nifti(path, arr: np.ndarray, affine: np.ndarray) -> None:
    """arr [z,y,x]; affine voxel(x,y,z) -> RAS mm."""
    path = Path(path)
    arr = np.ascontiguousarray(arr)
    if arr.dtype.name not in _CODE:
        arr = arr.astype(np.float32)
    shape = list(arr.shape[::-1])
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [len(shape)] + shape + [1] * (7 - len(shape))
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<hh", hdr, 70, _CODE[arr.dtype.name], arr.dtype.itemsize * 8)
    A = np.asarray(affine, np.float64)
    sp = np.sqrt((A[:3, :3] ** 2).sum(0))
    struct.pack_into("<8f", hdr, 76, 1.0, *[float(s) for s in sp], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<ff", hdr, 112, 1.0, 0.0)
    hdr[123] = 2  # xyzt_units: mm
    struct.pack_into("<hh", hdr, 252, 0, 2)  # qform 0, sform = aligned
    struct.pack_into("<12f", hdr, 280, *[float(v) for v in A[:3, :].reshape(-1)])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())
