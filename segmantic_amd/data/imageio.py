"""Volume file IO by suffix: NIfTI-1 (``.nii`` / ``.nii.gz``), MetaImage (``.mha`` / ``.mhd``) and NRRD
(``.nrrd``) -- the formats segmantic data sets come in.  The reference reads through
``LoadImaged(reader="ITKReader")`` (``src/segmantic/seg/monai_unet.py:157-162``), i.e. anything ITK
reads, and MONAI's ITKReader converts ITK's LPS geometry to a RAS affine; so does this module.

Arrays are returned [z, y, x]; the 4x4 affine maps voxel index (i = x, j = y, k = z) to RAS+ mm.
"""
from __future__ import annotations

import gzip
import zlib
from pathlib import Path
from typing import Tuple

import numpy as np

from .nifti import read_nifti, write_nifti

_LPS_TO_RAS = np.diag([-1.0, -1.0, 1.0, 1.0])

_MET = {"MET_UCHAR": np.uint8, "MET_CHAR": np.int8, "MET_SHORT": np.int16, "MET_USHORT": np.uint16,
        "MET_INT": np.int32, "MET_UINT": np.uint32, "MET_FLOAT": np.float32, "MET_DOUBLE": np.float64}
_MET_NAME = {np.dtype(v).name: k for k, v in _MET.items()}

_NRRD = {"uchar": np.uint8, "unsigned char": np.uint8, "uint8": np.uint8, "uint8_t": np.uint8,
         "signed char": np.int8, "int8": np.int8, "int8_t": np.int8,
         "short": np.int16, "int16": np.int16, "int16_t": np.int16, "short int": np.int16,
         "ushort": np.uint16, "unsigned short": np.uint16, "uint16": np.uint16, "uint16_t": np.uint16,
         "int": np.int32, "int32": np.int32, "int32_t": np.int32, "signed int": np.int32,
         "uint": np.uint32, "unsigned int": np.uint32, "uint32": np.uint32, "uint32_t": np.uint32,
         "float": np.float32, "double": np.float64}


def _geometry(origin, axes_times_spacing, lps: bool) -> np.ndarray:
    A = np.eye(4)
    A[:3, :3] = np.asarray(axes_times_spacing, np.float64).T     # column i = direction of voxel axis i * spacing
    A[:3, 3] = origin
    return _LPS_TO_RAS @ A if lps else A


# ------------------------------------------------------------------------------------ MetaImage
def _read_meta(path: Path) -> Tuple[np.ndarray, np.ndarray]:
    raw = path.read_bytes()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("latin1").strip()
        pos = end + 1
        if "=" in line:
            k, v = (t.strip() for t in line.split("=", 1))
            hdr[k] = v
            if k == "ElementDataFile":
                break
    nd = int(hdr.get("NDims", 3))
    if nd != 3 or int(hdr.get("ElementNumberOfChannels", 1)) != 1:
        raise ValueError(f"{path}: only single-channel 3-D MetaImages are supported")
    size = [int(v) for v in hdr["DimSize"].split()]
    sp = [float(v) for v in hdr.get("ElementSpacing", hdr.get("ElementSize", "1 1 1")).split()]
    org = [float(v) for v in hdr.get("Offset", hdr.get("Position", hdr.get("Origin", "0 0 0"))).split()]
    mat = [float(v) for v in hdr.get("TransformMatrix", hdr.get("Rotation", hdr.get("Orientation",
                                                                                   "1 0 0 0 1 0 0 0 1"))).split()]
    axes = np.asarray(mat, np.float64).reshape(3, 3) * np.asarray(sp, np.float64)[:, None]   # row i = axis i
    dt = np.dtype(_MET[hdr["ElementType"]])
    dt = dt.newbyteorder(">" if hdr.get("ElementByteOrderMSB", hdr.get("BinaryDataByteOrderMSB", "False")).lower() == "true" else "<")
    if hdr["ElementDataFile"] == "LOCAL":
        data = raw[pos:]
    else:
        data = (path.parent / hdr["ElementDataFile"]).read_bytes()
    if hdr.get("CompressedData", "False").lower() == "true":
        data = zlib.decompress(data)
    n = int(np.prod(size))
    arr = np.frombuffer(data, dtype=dt, count=n).astype(dt.newbyteorder("=")).reshape(size[::-1])
    return np.ascontiguousarray(arr), _geometry(org, axes, lps=True)


def _write_meta(path: Path, arr: np.ndarray, affine: np.ndarray) -> None:
    arr = np.ascontiguousarray(arr)
    if arr.dtype.name not in _MET_NAME:
        arr = arr.astype(np.float32)
    A = _LPS_TO_RAS @ np.asarray(affine, np.float64)
    sp = np.sqrt((A[:3, :3] ** 2).sum(0))
    sp[sp == 0] = 1.0
    dirs = (A[:3, :3] / sp).T                                   # row i = direction of voxel axis i
    fmt = lambda v: " ".join(repr(float(x)) for x in v)
    hdr = ["ObjectType = Image", "NDims = 3", "BinaryData = True", "BinaryDataByteOrderMSB = False",
           "CompressedData = False", f"TransformMatrix = {fmt(dirs.reshape(-1))}", f"Offset = {fmt(A[:3, 3])}",
           "CenterOfRotation = 0 0 0", "AnatomicalOrientation = RAI", f"ElementSpacing = {fmt(sp)}",
           f"DimSize = {arr.shape[2]} {arr.shape[1]} {arr.shape[0]}", f"ElementType = {_MET_NAME[arr.dtype.name]}",
           "ElementDataFile = LOCAL"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("latin1"))
        f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())


# ------------------------------------------------------------------------------------ NRRD
def _vec(s: str):
    return [float(v) for v in s.strip().strip("()").split(",")]


def _read_nrrd(path: Path) -> Tuple[np.ndarray, np.ndarray]:
    raw = path.read_bytes()
    if not raw.startswith(b"NRRD"):
        raise ValueError(f"{path}: not an NRRD file")
    # the header ends at the first blank line (LF or CRLF line ends)
    ends = [(raw.find(sep), len(sep)) for sep in (b"\n\n", b"\r\n\r\n") if raw.find(sep) >= 0]
    if not ends:
        raise ValueError(f"{path}: NRRD header has no terminating blank line")
    end, seplen = min(ends)
    hdr = {}
    for line in raw[:end].decode("latin1").splitlines()[1:]:
        if line.startswith("#") or ":" not in line:
            continue
        k, v = line.split(":", 1)
        hdr[k.strip().lower()] = v.lstrip("=").strip()
    if "data file" in hdr or "datafile" in hdr:
        raise ValueError(f"{path}: detached NRRD headers (data file: ...) are not supported")
    if int(hdr["dimension"]) != 3:
        raise ValueError(f"{path}: only 3-D NRRD volumes are supported")
    size = [int(v) for v in hdr["sizes"].split()]
    dt = np.dtype(_NRRD[hdr["type"].lower()])
    dt = dt.newbyteorder(">" if hdr.get("endian", "little").lower() == "big" else "<")
    if "space directions" in hdr:
        axes = [_vec(t) for t in hdr["space directions"].replace(") (", ")|(").split("|")]
    else:
        sp = [float(v) for v in hdr.get("spacings", "1 1 1").split()]
        axes = np.diag(sp).tolist()
    org = _vec(hdr["space origin"]) if "space origin" in hdr else [0.0, 0.0, 0.0]
    space = hdr.get("space", "left-posterior-superior").lower()
    if space in ("left-posterior-superior", "lps"):
        lps = True
    elif space in ("right-anterior-superior", "ras"):
        lps = False
    else:
        raise ValueError(f"{path}: NRRD space '{space}' is not supported (LPS / RAS)")
    data = raw[end + seplen:]
    enc = hdr.get("encoding", "raw").lower()
    if enc in ("gzip", "gz"):
        data = gzip.decompress(data)
    elif enc != "raw":
        raise ValueError(f"{path}: NRRD encoding '{enc}' is not supported (raw / gzip)")
    n = int(np.prod(size))
    arr = np.frombuffer(data, dtype=dt, count=n).astype(dt.newbyteorder("=")).reshape(size[::-1])
    return np.ascontiguousarray(arr), _geometry(org, axes, lps=lps)


def _write_nrrd(path: Path, arr: np.ndarray, affine: np.ndarray) -> None:
    arr = np.ascontiguousarray(arr)
    names = {np.dtype(v).name: k for k, v in (("uchar", np.uint8), ("signed char", np.int8), ("short", np.int16),
                                              ("ushort", np.uint16), ("int", np.int32), ("uint", np.uint32),
                                              ("float", np.float32), ("double", np.float64))}
    if arr.dtype.name not in names:
        arr = arr.astype(np.float32)
    A = _LPS_TO_RAS @ np.asarray(affine, np.float64)
    v = lambda x: "(" + ",".join(repr(float(t)) for t in x) + ")"
    hdr = ["NRRD0004", f"type: {names[arr.dtype.name]}", "dimension: 3", "space: left-posterior-superior",
           f"sizes: {arr.shape[2]} {arr.shape[1]} {arr.shape[0]}",
           "space directions: " + " ".join(v(A[:3, i]) for i in range(3)), "kinds: domain domain domain",
           "endian: little", "encoding: gzip", f"space origin: {v(A[:3, 3])}"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n\n").encode("latin1"))
        f.write(gzip.compress(arr.astype(arr.dtype.newbyteorder("<")).tobytes(), compresslevel=1))


# ------------------------------------------------------------------------------------ dispatch
def _kind(path: Path) -> str:
    name = path.name.lower()
    if name.endswith((".nii", ".nii.gz")):
        return "nifti"
    if name.endswith((".mha", ".mhd")):
        return "meta"
    if name.endswith(".nrrd"):
        return "nrrd"
    raise ValueError(f"{path}: unsupported image format (NIfTI .nii/.nii.gz, MetaImage .mha/.mhd, NRRD .nrrd)")


def read_image(path) -> Tuple[np.ndarray, np.ndarray]:
    """-> (array [z, y, x], RAS affine 4x4)"""
    path = Path(path)
    kind = _kind(path)
    return {"nifti": read_nifti, "meta": _read_meta, "nrrd": _read_nrrd}[kind](path)


def write_image(path, arr: np.ndarray, affine: np.ndarray) -> None:
    """arr [z, y, x]; affine voxel (x, y, z) -> RAS mm"""
    path = Path(path)
    kind = _kind(path)
    if kind == "meta" and path.name.lower().endswith(".mhd"):
        raise ValueError("write_image: write .mha (header + data in one file), not .mhd")
    {"nifti": write_nifti, "meta": _write_meta, "nrrd": _write_nrrd}[kind](path, arr, affine)


def strip_image_suffix(name: str) -> str:
    low = name.lower()
    for suf in (".nii.gz", ".nii", ".mha", ".mhd", ".nrrd"):
        if low.endswith(suf):
            return name[:-len(suf)]
    return Path(name).stem
