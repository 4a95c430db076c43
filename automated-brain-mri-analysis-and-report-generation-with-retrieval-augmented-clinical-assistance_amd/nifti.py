"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer.

nibabel and SimpleITK - what the reference uses for I/O (run_brats2021_inference_singlethread.py
:11,14; nnU-Net's SimpleITK export) - are not available, and the path needs only this much:
read four single-file NIfTI-1 volumes, write one uint8 label volume whose geometry header
(pixdim, qform/sform, units) is the input's, so downstream consumers see the same zooms, affine
and shape (feature_extraction/utils.py:117-124, evaluate_segmentation.py:78-81).

Array convention: ``NiftiImage.data`` is indexed (x, y, z) like nibabel; ``as_zyx()`` gives the
(z, y, x) view SimpleITK / nnU-Net work in.
"""
from __future__ import annotations

import gzip
import struct
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


@dataclass
class NiftiImage:
    data: np.ndarray          # (x, y, z), scaled (scl_slope/inter applied) if the header asks for it
    header: bytes             # the 348-byte NIfTI-1 header as read (template for writing)
    endian: str = "<"

    def _f(self, fmt, off):
        return struct.unpack_from(self.endian + fmt, self.header, off)

    @property
    def zooms(self) -> Tuple[float, float, float]:
        return tuple(float(v) for v in self._f("3f", 80))  # pixdim[1..3]

    @property
    def affine(self) -> np.ndarray:
        """sform if set, else qform, else pixdim scaling (nibabel's get_best_affine order)."""
        qform_code, sform_code = self._f("2h", 252)
        if sform_code > 0:
            rows = np.array(self._f("12f", 280), dtype=np.float64).reshape(3, 4)
            return np.vstack([rows, [0, 0, 0, 1]])
        pix = self._f("8f", 76)
        if qform_code > 0:
            b, c, d = self._f("3f", 256)
            ox, oy, oz = self._f("3f", 268)
            a2 = 1.0 - (b * b + c * c + d * d)
            a = np.sqrt(a2) if a2 > 0 else 0.0
            R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                          [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                          [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
            qfac = -1.0 if pix[0] < 0 else 1.0
            S = np.diag([pix[1], pix[2], pix[3] * qfac])
            M = np.eye(4)
            M[:3, :3] = R @ S
            M[:3, 3] = [ox, oy, oz]
            return M
        return np.diag([pix[1], pix[2], pix[3], 1.0])

    def as_zyx(self) -> np.ndarray:
        return np.ascontiguousarray(self.data.transpose(2, 1, 0))


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def load(path) -> NiftiImage:
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError(f"{path}: too short for a NIfTI-1 file")
    endian = "<"
    if struct.unpack_from("<i", raw, 0)[0] != 348:
        if struct.unpack_from(">i", raw, 0)[0] != 348:
            raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
        endian = ">"
    if raw[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError(f"{path}: bad NIfTI magic {raw[344:348]!r}")
    if raw[344:348] == b"ni1\0":
        raise ValueError(f"{path}: two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack_from(endian + "8h", raw, 40)
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError(f"{path}: bad dim[0]={ndim}")
    shape = tuple(int(v) for v in dim[1:1 + ndim])
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    if len(shape) != 3:
        raise ValueError(f"{path}: expected a 3-D volume, got shape {shape}")
    datatype, bitpix = struct.unpack_from(endian + "2h", raw, 70)
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    vox_offset = int(struct.unpack_from(endian + "f", raw, 108)[0])
    n = int(np.prod(shape))
    if len(raw) < vox_offset + n * dt.itemsize:
        raise ValueError(f"{path}: truncated image data")
    data = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(shape, order="F")
    slope, inter = struct.unpack_from(endian + "2f", raw, 112)
    if slope != 0 and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope) and np.isfinite(inter):
        data = data.astype(np.float64) * slope + inter
    return NiftiImage(data=np.asarray(data), header=bytes(raw[:348]), endian=endian)


def save_like(path, data_xyz: np.ndarray, like: NiftiImage) -> None:
    """Write ``data_xyz`` (x, y, z) with ``like``'s geometry (pixdim, qform, sform, units)."""
    data_xyz = np.asarray(data_xyz)
    if data_xyz.dtype.name not in _CODES:
        raise ValueError(f"unsupported dtype {data_xyz.dtype}")
    if data_xyz.ndim != 3:
        raise ValueError("expected a 3-D array")
    e = "<"
    hdr = bytearray(348)
    src = like.header
    if like.endian != "<":  # re-pack the geometric fields little-endian
        src = bytearray(src)
        for fmt, off in (("8f", 76), ("2h", 252), ("6f", 256), ("12f", 280), ("f", 108)):
            struct.pack_into("<" + fmt, src, off, *struct.unpack_from(">" + fmt, like.header, off))
        src = bytes(src)
    struct.pack_into(e + "i", hdr, 0, 348)
    hdr[39] = src[39]                                   # dim_info
    struct.pack_into(e + "8h", hdr, 40, 3, *data_xyz.shape, 1, 1, 1, 1)
    struct.pack_into(e + "2h", hdr, 70, _CODES[data_xyz.dtype.name], data_xyz.dtype.itemsize * 8)
    hdr[76:108] = src[76:108]                           # pixdim
    struct.pack_into(e + "f", hdr, 108, 352.0)          # vox_offset
    struct.pack_into(e + "2f", hdr, 112, 1.0, 0.0)      # scl_slope / scl_inter
    hdr[123] = src[123]                                 # xyzt_units
    hdr[252:328] = src[252:328]                         # qform/sform codes, quaternion, offsets, srow_*
    hdr[344:348] = b"n+1\0"
    payload = bytes(hdr) + b"\0\0\0\0" + np.asfortranarray(data_xyz.astype(data_xyz.dtype.newbyteorder("<"))).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.GzipFile(path, "wb", compresslevel=1, mtime=0) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


def make_header(shape_xyz, zooms=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), dtype=np.int16) -> NiftiImage:
    """A fresh axis-aligned header (tests / synthetic cases)."""
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, *shape_xyz, 1, 1, 1, 1)
    struct.pack_into("<2h", hdr, 70, _CODES[np.dtype(dtype).name], np.dtype(dtype).itemsize * 8)
    struct.pack_into("<8f", hdr, 76, 1.0, *zooms, 0.0, 0.0, 0.0, 0.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<2f", hdr, 112, 1.0, 0.0)
    hdr[123] = 2  # mm
    struct.pack_into("<2h", hdr, 252, 1, 1)
    struct.pack_into("<6f", hdr, 256, 0.0, 0.0, 0.0, *origin)
    struct.pack_into("<12f", hdr, 280, zooms[0], 0, 0, origin[0], 0, zooms[1], 0, origin[1], 0, 0, zooms[2], origin[2])
    hdr[344:348] = b"n+1\0"
    return NiftiImage(data=np.zeros(shape_xyz, dtype=dtype), header=bytes(hdr))
