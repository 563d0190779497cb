"""Host-side geometry and parameter algebra of the 2-D alignment path (numpy only).

Mirrors the SPHIRE helpers the reference drivers import
(test_mref_gpu_align.py:228-235): Numrinit, ringwe, model_circle, search_range,
combine_params2, inverse_transform2, MPI_start_end and the AlignParam -> xform.align2d
conversion at test_mref_gpu_align.py:578-588.
"""
import math

import numpy as np


def numrinit(first_ring, last_ring, skip=1):
    """sp_alignment.Numrinit(first, last, skip, "F") (test_mref_gpu_align.py:348).
    Returns the flat list of (radius, 1-based offset, length) triplets."""
    MAXFFT = 32768
    dpi = 2 * math.pi
    numr = []
    lcirc = 1
    for k in range(first_ring, last_ring + 1, skip):
        numr.append(k)
        jp = int(dpi * k + 0.5)
        ip = 2 ** (int(math.floor(math.log2(jp))) + 1)
        if k + skip <= last_ring and jp > ip + ip // 2:
            ip = min(MAXFFT, 2 * ip)
        if k + skip > last_ring and jp > ip + ip // 5:
            ip = min(MAXFFT, 2 * ip)
        numr.append(lcirc)
        numr.append(ip)
        lcirc += ip
    return numr


def ringwe(numr):
    """sp_alignment.ringwe(numr, "F") (test_mref_gpu_align.py:349)."""
    dpi = 2 * math.pi
    nring = len(numr) // 3
    maxrin = float(numr[-1])
    return [numr[3 * i] * dpi / float(numr[3 * i + 2]) * maxrin / float(numr[3 * i + 2]) for i in range(nring)]


def model_circle(r, nx, ny):
    """sp_utilities.model_circle: 1 inside radius r about (nx//2, ny//2)."""
    y, x = np.mgrid[0:ny, 0:nx]
    x2 = (x.astype(np.float32) - nx // 2) ** 2 / np.float32(r * r)
    y2 = (y.astype(np.float32) - ny // 2) ** 2 / np.float32(r * r)
    return ((x2 + y2) <= 1.0).astype(np.float32)


def normalize_mask(img, mask, no_sigma):
    """processor normalize.mask (test_mref_gpu_align.py:336,342): subtract the mean under
    mask>0.5; when no_sigma != 0 also divide by the sample sigma under the mask."""
    sel = mask > 0.5
    v = img[sel].astype(np.float64)
    n = v.size
    mean = np.float32(v.sum() / n) if n else np.float32(0)
    sigma = np.float32(1.0)
    if no_sigma != 0:
        sigma = np.float32(math.sqrt(np.float32((np.dot(v, v) - v.sum() ** 2 / n) / (n - 1))))
    return ((img - mean) / sigma).astype(np.float32)


def search_range(n, radius, shift, rng):
    """sp_alignment.search_range, returned as the callers use it after their swap
    (test_mref_gpu_align.py:1035-1038): [left, right]."""
    cn = n // 2 + 1
    ql = max(cn + shift - radius - 2, 0)
    qe = max(n - cn - shift - radius, 0)
    return [min(ql, rng), min(qe, rng)]


def mpi_start_end(nima, nproc, myid):
    """sp_applications.MPI_start_end (test_mref_gpu_align.py:289)."""
    return int(round(float(nima) / nproc * myid)), int(round(float(nima) / nproc * (myid + 1)))


def _tf(alpha, tx, ty, mirror):
    a = math.radians(alpha)
    c, s = math.cos(a), math.sin(a)
    sg = -1.0 if mirror else 1.0
    return np.array([[sg * c, sg * s, sg * tx], [-s, c, ty], [0, 0, 1.0]])


def _tf_params(T):
    m = 1 if (T[0, 0] * T[1, 1] - T[0, 1] * T[1, 0]) < 0 else 0
    sg = -1.0 if m else 1.0
    alpha = math.degrees(math.atan2(sg * T[0, 1], sg * T[0, 0])) % 360.0
    return alpha, sg * T[0, 2], T[1, 2], m


def combine_params2(a1, sx1, sy1, m1, a2, sx2, sy2, m2):
    """sp_utilities.combine_params2: parameters of T2*T1 with v' = M(R(alpha) v + t)."""
    return _tf_params(_tf(a2, sx2, sy2, m2) @ _tf(a1, sx1, sy1, m1))


def inverse_transform2(alpha, tx=0.0, ty=0.0, mirror=0):
    """sp_utilities.inverse_transform2."""
    return _tf_params(np.linalg.inv(_tf(alpha, tx, ty, mirror)))


def alignparam_to_eman2(angle, shift_x, shift_y, mirror):
    """AlignParam -> (alpha, sx, sy, mirror) exactly as test_mref_gpu_align.py:578-588
    ("this is usually done in ormq()").  Vectorised over numpy arrays."""
    angle = np.asarray(angle, np.float64)
    sx_neg = -np.asarray(shift_x, np.float64)
    sy_neg = -np.asarray(shift_y, np.float64)
    c_ang = np.cos(np.radians(angle))
    s_ang = -np.sin(np.radians(angle))
    sx = sx_neg * c_ang - sy_neg * s_ang
    sy = sx_neg * s_ang + sy_neg * c_ang
    return angle, sx, sy, np.asarray(mirror).astype(np.int32)


def shift_list(xrng, yrng, step):
    """search offsets in EMAN2's loop order (y outer, x inner), Util.multiref_polar_ali_2d;
    returns float32 [S][2] = (ix, iy)."""
    kx = int(xrng / step)
    ky = int(yrng / step)
    out = []
    for i in range(-ky, ky + 1):
        for j in range(-kx, kx + 1):
            out.append((j * step, i * step))
    return np.array(out, np.float32)
