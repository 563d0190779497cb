"""ctypes binding of the CPU oracle (oracle/ralign_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from cryo_ralib_amd/ (the product).
PARITY UNPINNED: see oracle/ralign_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libralign_oracle.so")

ORC_MAXRING = 512
INTERP_BILINEAR = 0
INTERP_QUADRI = 1


class Rings(ctypes.Structure):
    _fields_ = [("nring", ctypes.c_int), ("maxrin", ctypes.c_int), ("lcirc", ctypes.c_int),
                ("first_ring", ctypes.c_int), ("last_ring", ctypes.c_int), ("skip", ctypes.c_int),
                ("numr", ctypes.c_int * (3 * ORC_MAXRING)), ("wr", ctypes.c_float * ORC_MAXRING)]

    def numr_list(self):
        return list(self.numr[:3 * self.nring])

    def wr_list(self):
        return list(self.wr[:self.nring])


class SearchInfo(ctypes.Structure):
    _fields_ = [("ix", ctypes.c_float), ("iy", ctypes.c_float), ("jtot", ctypes.c_int), ("tot", ctypes.c_float)]


def build(force=False):
    """compile the oracle with the committed Makefile (gcc only)."""
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ralign_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        rp = ctypes.POINTER(Rings)
        sp = ctypes.POINTER(SearchInfo)
        L.orc_rings_init.argtypes = [rp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_rings_init.restype = ctypes.c_int
        L.orc_polar2dm.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, rp, fp, ctypes.c_int]
        L.orc_normalize_ring.argtypes = [fp, rp]
        L.orc_frngs.argtypes = [fp, rp]
        L.orc_applyws.argtypes = [fp, rp]
        L.orc_crosrng_ms.argtypes = [fp, fp, rp, dp, fp, dp, fp, ip, ip]
        L.orc_prb1d7.argtypes = [dp]
        L.orc_prb1d7.restype = ctypes.c_float
        L.orc_ang_n.argtypes = [ctypes.c_float, ctypes.c_int]
        L.orc_ang_n.restype = ctypes.c_float
        L.orc_multiref_polar_ali_2d.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, fp, fp,
                                                ctypes.c_float, rp, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_int, ctypes.c_int, fp, sp]
        L.orc_ormq.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, fp, ctypes.c_float, rp,
                               ctypes.c_float, ctypes.c_float, ctypes.c_int, dp, sp]
        L.orc_model_circle.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int]
        L.orc_normalize_mask.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int]
        L.orc_rot_shift2d.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_int]
        L.orc_combine_params2.argtypes = [ctypes.c_double] * 3 + [ctypes.c_int] + [ctypes.c_double] * 3 + [ctypes.c_int, dp]
        L.orc_inverse_transform2.argtypes = [ctypes.c_double] * 3 + [ctypes.c_int, dp]
        L.orc_state_from_params.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp]
        L.orc_search_range.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, fp]
        L.orc_prepare_refs.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, rp, ctypes.c_int, fp]
        L.orc_mref_iteration.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, rp,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                         fp, fp, sp, fp, ip, ctypes.c_int, ctypes.c_int]
        L.orc_reffree_iteration.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, rp,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                            fp, fp, fp, sp, fp, ctypes.c_int, ctypes.c_int, dp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def rings(first_ring, last_ring, skip=1):
    rg = Rings()
    n = lib().orc_rings_init(ctypes.byref(rg), first_ring, last_ring, skip)
    assert n > 0
    return rg


def polar2dm(img, cnx, cny, rg, interp=INTERP_BILINEAR):
    img = _f32(img)
    ny, nx = img.shape
    circ = np.zeros(rg.lcirc, np.float32)
    lib().orc_polar2dm(_f(img), nx, ny, cnx, cny, ctypes.byref(rg), _f(circ), interp)
    return circ


def normalize_ring(circ, rg):
    circ = _f32(circ).copy()
    lib().orc_normalize_ring(_f(circ), ctypes.byref(rg))
    return circ


def frngs(circ, rg):
    circ = _f32(circ).copy()
    lib().orc_frngs(_f(circ), ctypes.byref(rg))
    return circ


def applyws(circ, rg):
    circ = _f32(circ).copy()
    lib().orc_applyws(_f(circ), ctypes.byref(rg))
    return circ


def crosrng_ms(c1, c2, rg):
    c1 = _f32(c1); c2 = _f32(c2)
    qn = ctypes.c_double(); qm = ctypes.c_double()
    tot = ctypes.c_float(); tmt = ctypes.c_float()
    jn = ctypes.c_int(); jm = ctypes.c_int()
    lib().orc_crosrng_ms(_f(c1), _f(c2), ctypes.byref(rg), ctypes.byref(qn), ctypes.byref(tot),
                         ctypes.byref(qm), ctypes.byref(tmt), ctypes.byref(jn), ctypes.byref(jm))
    return dict(qn=qn.value, tot=tot.value, qm=qm.value, tmt=tmt.value, jn=jn.value, jm=jm.value)


def prb1d7(b):
    b = np.ascontiguousarray(b, np.float64)
    return lib().orc_prb1d7(_d(b))


def ang_n(tot, maxrin):
    return lib().orc_ang_n(tot, maxrin)


def multiref_polar_ali_2d(img, crefim, xrng, yrng, step, rg, cnx, cny,
                          interp=INTERP_BILINEAR, normalize=True):
    img = _f32(img); crefim = _f32(crefim)
    ny, nx = img.shape
    nref = crefim.shape[0]
    xr = np.array(xrng, np.float32); yr = np.array(yrng, np.float32)
    out = np.zeros(6, np.float32)
    info = SearchInfo()
    lib().orc_multiref_polar_ali_2d(_f(img), nx, ny, _f(crefim), nref, _f(xr), _f(yr), step,
                                    ctypes.byref(rg), cnx, cny, interp, int(normalize), _f(out),
                                    ctypes.byref(info))
    return out, info


def ormq(img, crefim, xrng, yrng, step, rg, cnx, cny, interp=INTERP_BILINEAR):
    img = _f32(img); crefim = _f32(crefim)
    ny, nx = img.shape
    xr = np.array(xrng, np.float32); yr = np.array(yrng, np.float32)
    out = np.zeros(5, np.float64)
    info = SearchInfo()
    lib().orc_ormq(_f(img), nx, ny, _f(crefim), _f(xr), _f(yr), step, ctypes.byref(rg), cnx, cny,
                   interp, _d(out), ctypes.byref(info))
    return out, info


def set_nomirror(flag):
    """ormq(..., nomirror): only the straight half of Crosrng_ms (Util.Crosrng_ns); process-wide switch"""
    lib().orc_set_nomirror(int(bool(flag)))


def set_ormq_normalize(flag):
    """ormq with Normalize_ring between Polar2Dm and Frngs (checker of the engine option normalize_ring = 1 in RA_MODE_REFFREE)"""
    lib().orc_set_ormq_normalize(int(bool(flag)))


def model_circle(r, nx, ny, edge_le=True):
    m = np.zeros((ny, nx), np.float32)
    lib().orc_model_circle(r, nx, ny, _f(m), int(edge_le))
    return m


def normalize_mask(img, mask, no_sigma):
    img = _f32(img).copy(); mask = _f32(mask)
    lib().orc_normalize_mask(_f(img), _f(mask), img.size, no_sigma)
    return img


def rot_shift2d(img, ang, sx, sy, mirror):
    img = _f32(img)
    ny, nx = img.shape
    out = np.zeros_like(img)
    lib().orc_rot_shift2d(_f(img), _f(out), nx, ny, ang, sx, sy, int(mirror))
    return out


def combine_params2(a1, sx1, sy1, m1, a2, sx2, sy2, m2):
    out = np.zeros(4, np.float64)
    lib().orc_combine_params2(a1, sx1, sy1, int(m1), a2, sx2, sy2, int(m2), _d(out))
    return out[0], out[1], out[2], int(out[3])


def inverse_transform2(alpha, tx=0.0, ty=0.0, mirror=0):
    out = np.zeros(4, np.float64)
    lib().orc_inverse_transform2(alpha, tx, ty, int(mirror), _d(out))
    return out[0], out[1], out[2], int(out[3])


def state_from_params(params, mode=0, cs=(0.0, 0.0)):
    """d [n][2] float32 the next search starts from, rebuilt from the float32 header values (alpha, sx, sy[, mirror]) as
    the reference's loops do (mode 0: mref_ali2d, inverse_transform2; mode 1: ali2d_single_iter, combine_params2 with the
    centre correction then inverse_transform2)"""
    params = np.ascontiguousarray(params, np.float32)
    n = params.shape[0]
    assert params.shape == (n, 6)
    d = np.zeros((n, 2), np.float32)
    csa = np.array(cs, np.float32)
    lib().orc_state_from_params(_f(params), n, int(mode), _f(csa), _f(d))
    return d


def search_range(n, radius, shift, rng):
    out = np.zeros(2, np.float32)
    lib().orc_search_range(n, radius, shift, rng, _f(out))
    return [float(out[0]), float(out[1])]


def prepare_refs(refs, mask, rg, interp=INTERP_BILINEAR):
    """returns (normalised refs, crefim[nref][lcirc])"""
    refs = _f32(refs).copy()
    nref, nx = refs.shape[0], refs.shape[-1]
    cref = np.zeros((nref, rg.lcirc), np.float32)
    mp = _f(_f32(mask)) if mask is not None else None
    lib().orc_prepare_refs(_f(refs), nref, nx, mp, ctypes.byref(rg), interp, _f(cref))
    return refs, cref


def mref_iteration(particles, crefim, rg, xrng, yrng, step, d, sums=None, counts=None,
                   index0=0, nthreads=1, interp=INTERP_BILINEAR, normalize=True):
    """one pass of the per-particle loop of mref_ali2d.  d [n][2] is updated in place.
    returns (params[n][6], infos, sums[nref][2][nx][nx], counts[nref])"""
    particles = _f32(particles); crefim = _f32(crefim)
    n, nx = particles.shape[0], particles.shape[-1]
    nref = crefim.shape[0]
    assert d.dtype == np.float32 and d.shape == (n, 2) and d.flags.c_contiguous
    params = np.zeros((n, 6), np.float32)
    infos = (SearchInfo * n)()
    if sums is None:
        sums = np.zeros((nref, 2, nx, nx), np.float32)
    if counts is None:
        counts = np.zeros(nref, np.int32)
    lib().orc_mref_iteration(_f(particles), n, nx, _f(crefim), nref, ctypes.byref(rg), xrng, yrng, step,
                             interp, int(normalize), _f(d), _f(params), infos, _f(sums), _i(counts),
                             index0, nthreads)
    return params, infos, sums, counts


def reffree_iteration(particles, crefim, rg, xrng, yrng, step, cs, d, params, sums=None,
                      index0=0, nthreads=1, interp=INTERP_BILINEAR):
    particles = _f32(particles); crefim = _f32(crefim)
    n, nx = particles.shape[0], particles.shape[-1]
    assert params.dtype == np.float32 and params.shape == (n, 6)
    infos = (SearchInfo * n)()
    if sums is None:
        sums = np.zeros((1, 2, nx, nx), np.float32)
    csa = np.array(cs, np.float32)
    ss = np.zeros(2, np.float64)
    lib().orc_reffree_iteration(_f(particles), n, nx, _f(crefim), ctypes.byref(rg), xrng, yrng, step, interp,
                                _f(csa), _f(d), _f(params), infos, _f(sums), index0, nthreads, _d(ss))
    return params, infos, sums, ss
