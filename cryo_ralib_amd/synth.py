"""Synthetic particle stacks with planted truth.

float32, numpy PCG64.  References: seed=1000, 12 anisotropic Gaussians each under a soft
circular mask of radius `ou`, normalised under model_circle(ou).  Particles: seed=2000+shard,
random class / angle / mirror / integer shift, rotate -> shift -> mirror-x with quadratic
interpolation (the rot_shift2D convention of test_mref_gpu_align.py:1055) plus N(0, sigma^2).
"""
import math

import numpy as np

from . import geometry


def rot_shift2d_np(img, ang, sx, sy, mirror):
    """numpy restatement of rot_shift2D(img, ang, sx, sy, mirror) with the SPHIRE defaults
    ("quadratic", "background"): rot_scale_trans2D_background then xform.mirror(x).
    Arithmetic in float32 like the reference's in-tree restatement
    (notebook/02_CuPy_Image_Processing_rot_shift2d.ipynb cell 2)."""
    f = np.float32
    img = np.ascontiguousarray(img, np.float32)
    ny, nx = img.shape
    a = f(ang) * f(math.pi) / f(180.0)
    xc, yc = nx // 2, ny // 2
    shiftxc, shiftyc = f(xc) + f(sx), f(yc) + f(sy)
    cang, sang = f(math.cos(float(a))), f(math.sin(float(a)))
    iy, ix = np.mgrid[0:ny, 0:nx]
    y = iy.astype(f) - shiftyc
    ycang = y * cang + f(yc)
    ysang = -y * sang + f(xc)
    x = ix.astype(f) - shiftxc
    xold = x * cang + ysang + f(1.0)
    yold = x * sang + ycang + f(1.0)
    out_of = (xold < 1.0) | (xold >= f(nx + 1)) | (yold < 1.0) | (yold >= f(ny + 1))
    xq = np.where(out_of, (ix + 1).astype(f), xold)
    yq = np.where(out_of, (iy + 1).astype(f), yold)
    i = xq.astype(np.int32); j = yq.astype(np.int32)
    dx0 = xq - i; dy0 = yq - j
    ip1 = i + 1; im1 = i - 1; jp1 = j + 1; jm1 = j - 1
    ip1 = np.where(ip1 > nx, ip1 - nx, ip1); im1 = np.where(im1 < 1, im1 + nx, im1)
    jp1 = np.where(jp1 > ny, jp1 - ny, jp1); jm1 = np.where(jm1 < 1, jm1 + ny, jm1)

    def fd(ii, jj):
        return img[jj - 1, ii - 1]
    f0 = fd(i, j)
    c1 = fd(ip1, j) - f0
    c2 = (c1 - f0 + fd(im1, j)) * f(0.5)
    c3 = fd(i, jp1) - f0
    c4 = (c3 - f0 + fd(i, jm1)) * f(0.5)
    dxb = dx0 - f(1); dyb = dy0 - f(1)
    hxc = np.where(dx0 >= 0, 1, -1); hyc = np.where(dy0 >= 0, 1, -1)
    ic = i + hxc; jc = j + hyc
    ic = np.where(ic > nx, ic - nx, np.where(ic < 1, ic + nx, ic))
    jc = np.where(jc > ny, jc - ny, np.where(jc < 1, jc + ny, jc))
    hx = hxc.astype(f); hy = hyc.astype(f)
    c5 = (fd(ic, jc) - f0 - hx * c1 - (hx * (hx - f(1))) * c2 - hy * c3 - (hy * (hy - f(1))) * c4) * (hx * hy)
    out = (f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4)).astype(f)
    if mirror:
        start = 1 - nx % 2
        out[:, start:] = out[:, start:][:, ::-1].copy()
    return out


def make_references(nref, nx, ou, seed=1000):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:nx, 0:nx].astype(np.float64)
    cx = cy = nx // 2
    r = np.hypot(xx - cx, yy - cy)
    soft = 0.5 * (1.0 - np.tanh((r - (ou - 2.0)) / 1.5))
    mask = geometry.model_circle(ou, nx, nx)
    refs = np.zeros((nref, nx, nx), np.float32)
    for k in range(nref):
        img = np.zeros((nx, nx), np.float64)
        for _ in range(12):
            s1, s2 = rng.uniform(2.0, 6.0, 2)
            rad = 0.6 * ou * math.sqrt(rng.uniform())
            phi = rng.uniform(0, 2 * math.pi)
            gx, gy = cx + rad * math.cos(phi), cy + rad * math.sin(phi)
            amp = rng.uniform(0.5, 1.5)
            th = rng.uniform(0, math.pi)
            u = (xx - gx) * math.cos(th) + (yy - gy) * math.sin(th)
            v = -(xx - gx) * math.sin(th) + (yy - gy) * math.cos(th)
            img += amp * np.exp(-0.5 * ((u / s1) ** 2 + (v / s2) ** 2))
        img *= soft
        refs[k] = geometry.normalize_mask(img.astype(np.float32), mask, 1)
    return refs


def plant_truth(nref, n, xr, yr, seed):
    """ground truth arrays (class, angle, mirror, sx, sy) for shard `seed`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cls = rng.integers(0, nref, n).astype(np.int32)
    ang = rng.uniform(0.0, 360.0, n).astype(np.float32)
    mir = rng.integers(0, 2, n).astype(np.int32)
    sx = rng.integers(-int(xr), int(xr) + 1, n).astype(np.float32)
    sy = rng.integers(-int(yr), int(yr) + 1, n).astype(np.float32)
    noise_seed = int(rng.integers(0, 2 ** 31 - 1))
    return dict(cls=cls, ang=ang, mir=mir, sx=sx, sy=sy, noise_seed=noise_seed)


def make_particles(refs, n, xr, yr, sigma_n, shard=0, ou=None):
    """numpy generator (tests / small stacks).  Returns (particles[n][nx][nx], truth)."""
    nref, nx = refs.shape[0], refs.shape[-1]
    truth = plant_truth(nref, n, xr, yr, 2000 + shard)
    rng = np.random.Generator(np.random.PCG64(truth["noise_seed"]))
    out = np.zeros((n, nx, nx), np.float32)
    for i in range(n):
        img = rot_shift2d_np(refs[truth["cls"][i]], truth["ang"][i], truth["sx"][i], truth["sy"][i],
                             truth["mir"][i])
        out[i] = img + rng.standard_normal((nx, nx), np.float32) * np.float32(sigma_n)
    if ou is not None:
        mask = geometry.model_circle(ou, nx, nx)
        for i in range(n):
            out[i] = geometry.normalize_mask(out[i], mask, 0)
    return out, truth
