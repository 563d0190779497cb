"""Particle-stack I/O for the command-line entry points.

The reference reads EMAN2 HDF / bdb stacks through EMAN2 (`EMData.read_images`,
test_mref_gpu_align.py:1358-1375) and writes class averages with `write_image`
(:564).  Without EMAN2 the engine accepts:
  * `.npy`            float32 [n][ny][nx]
  * `.mrc` / `.mrcs`  MRC2014 mode-2 stacks (EMAN2 / RELION read and write these)
  * `.hdf` / `.h5`    EMAN2 MDF layout `/MDF/images/<i>/image`, read and written by the dependency-free
                      HDF5 subset implementation in `mdfio.py`
Parameter rows follow the reference's text outputs: `idx angle_psi shift_x shift_y mirror class`
(notebook/03 cell 6) and `alpha sx sy mirror` for initial2Dparams.txt
(test_reffree_gpu_align.py:561-569).
"""
import os
import struct

import numpy as np


def _read_mrc(path, first=0, last=None):
    with open(path, "rb") as f:
        hdr = f.read(1024)
        nx, ny, nz, mode = struct.unpack("<4i", hdr[:16])
        nsymbt = struct.unpack("<i", hdr[92:96])[0]
        if mode != 2:
            raise ValueError("%s: only MRC mode 2 (float32) stacks are supported, got mode %d" % (path, mode))
        first = max(0, first); last = nz if last is None else min(last, nz)
        cnt = max(0, last - first)
        f.seek(1024 + nsymbt + 4 * nx * ny * first)
        data = np.fromfile(f, dtype="<f4", count=nx * ny * cnt)
    if data.size != nx * ny * cnt:
        raise ValueError("%s: truncated MRC file" % path)
    return data.reshape(cnt, ny, nx).astype(np.float32)


def _write_mrc(path, arr):
    arr = np.ascontiguousarray(arr, dtype="<f4")
    if arr.ndim == 2:
        arr = arr[None]
    nz, ny, nx = arr.shape
    hdr = bytearray(1024)
    struct.pack_into("<4i", hdr, 0, nx, ny, nz, 2)
    struct.pack_into("<3i", hdr, 28, nx, ny, nz)                      # mx my mz
    struct.pack_into("<3f", hdr, 40, float(nx), float(ny), float(nz))  # cell a b c
    struct.pack_into("<3f", hdr, 52, 90.0, 90.0, 90.0)
    struct.pack_into("<3i", hdr, 64, 1, 2, 3)                         # mapc mapr maps
    struct.pack_into("<3f", hdr, 76, float(arr.min()), float(arr.max()), float(arr.mean()))
    hdr[208:212] = b"MAP "
    hdr[212:216] = bytes([0x44, 0x44, 0, 0])
    struct.pack_into("<f", hdr, 216, float(arr.std()))
    with open(path, "wb") as f:
        f.write(hdr)
        arr.tofile(f)


def _read_hdf(path, first=0, last=None):
    from . import mdfio
    return mdfio.read_mdf_stack(path, first=first, last=last)


def _write_hdf(path, arr):
    from . import mdfio
    mdfio.write_mdf_stack(path, arr)


def stack_size(path):
    """number of images without reading the pixel data"""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path, mmap_mode="r")
        return 1 if a.ndim == 2 else a.shape[0]
    if ext in (".mrc", ".mrcs", ".st"):
        with open(path, "rb") as f:
            nx, ny, nz = np.frombuffer(f.read(12), "<i4")
        return int(nz)
    if ext in (".hdf", ".h5"):
        from . import mdfio
        return mdfio.mdf_image_count(path)
    raise ValueError("unsupported stack format: %s" % path)


def read_stack(path, first=0, last=None):
    """float32 [n][ny][nx]; first / last select images [first, last): a rank reads only its MPI_start_end slice
    (test_mref_gpu_align.py:1358-1375), through a memory map, not the whole file"""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path, mmap_mode="r")
        a = a[None] if a.ndim == 2 else a
        return np.ascontiguousarray(a[first:last], np.float32)
    if ext in (".mrc", ".mrcs", ".st"):
        return _read_mrc(path, first, last)
    if ext in (".hdf", ".h5"):
        return _read_hdf(path, first, last)
    raise ValueError("unsupported stack format: %s" % path)


def write_stack(path, arr):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        np.save(path, np.asarray(arr, np.float32))
    elif ext in (".mrc", ".mrcs", ".st"):
        _write_mrc(path, arr)
    elif ext in (".hdf", ".h5"):
        _write_hdf(path, arr)
    else:
        raise ValueError("unsupported stack format: %s" % path)


def write_text_rows(path, rows, fmt="%14.6f"):
    """sp_utilities.write_text_row analogue: one row per line, blank separated."""
    with open(path, "w") as f:
        for r in rows:
            f.write("  ".join((("%12d" % v) if isinstance(v, (int, np.integer)) else (fmt % v)) for v in r) + "\n")
