"""CPU restatement (numpy, double precision FFTs) of the per-iteration reference update of the
EMAN2/SPHIRE drivers: fsc / fsc_mask, fit_tanh (+ amoeba), the default user function ref_ali2d
(filt_tanl + center_2D method 1 = phase_cog + fshift) and normalize.mask.

TEST INFRASTRUCTURE ONLY (see oracle/ralign_oracle.h).  PARITY UNPINNED: the routines live in
EMAN2 2.31 / SPHIRE, which are not in the reference tree; this file restates their published
algorithms and follows the reference's call sites:
    test_mref_gpu_align.py:517-564   (fsc per class, averaged fsc, user_func, normalize.mask)
    test_reffree_gpu_align.py:374-429 (fsc_mask, user_func, fshift by the average centre)
The only in-tree pin is the tangent filter formula of cuda/gpu_aln_noref.cu:786-816 and the
"cut-off 0.120 / fall-off 0.200" lines of notebook/00 cell 3 (the clamps of ref_ali2d).
"""
import math

import numpy as np


def fsc(img1, img2, w=1.0):
    """sp_statistics.fsc -> EMData::calc_fourier_shell_correlation (2-D, square).
    Returns [freq, fsc, npoints] lists of length inc+1, inc = round(max(nx/2, ny/2) / w)."""
    ny, nx = img1.shape
    nx2, ny2 = nx // 2, ny // 2
    inc = int(round(max(nx2, ny2) / w))
    f = np.fft.rfft2(img1.astype(np.float64))
    g = np.fft.rfft2(img2.astype(np.float64))
    ky = np.arange(ny)
    ky = np.where(ky > ny2, ky - ny, ky).astype(np.float64)[:, None]
    kx = np.arange(nx // 2 + 1).astype(np.float64)[None, :]
    argx = 0.5 * np.sqrt(np.float32(ky * ky / (ny2 * ny2) + kx * kx / (nx2 * nx2)).astype(np.float64))
    r = np.floor(inc * 2 * argx + 0.5).astype(int)
    # the kx = 0 column holds both members of every conjugate pair: only ky >= 0 is counted
    use = (r <= inc) & ~((kx == 0) & (ky < 0))
    ret = np.zeros(inc + 1); n1 = np.zeros(inc + 1); n2 = np.zeros(inc + 1); lr = np.zeros(inc + 1)
    np.add.at(ret, r[use], (f * np.conj(g)).real[use])
    np.add.at(n1, r[use], (np.abs(f) ** 2)[use])
    np.add.at(n2, r[use], (np.abs(g) ** 2)[use])
    np.add.at(lr, r[use], 2.0)
    freq = [i / (2.0 * inc) for i in range(inc + 1)]
    val = [float(ret[i] / math.sqrt(n1[i] * n2[i])) if lr[i] > 0 and n1[i] > 0 and n2[i] > 0 else 0.0 for i in range(inc + 1)]
    return [freq, val, [float(x) for x in lr]]


def fsc_mask(img1, img2, mask, w=1.0):
    """sp_statistics.fsc_mask: subtract the mean under the (binarized) mask, multiply by the mask."""
    m = mask > 0.5
    s1 = img1[m].astype(np.float64).mean()
    s2 = img2[m].astype(np.float64).mean()
    return fsc(((img1 - np.float32(s1)) * mask).astype(np.float32), ((img2 - np.float32(s2)) * mask).astype(np.float32), w)


def amoeba(var, scale, func, ftolerance=1.e-4, xtolerance=1.e-4, itmax=500, data=None):
    """sp_utilities.amoeba: simplex MAXIMISATION (reflect / expand / shrink variant)."""
    nvar = len(var)
    nsimplex = nvar + 1
    simplex = [list(var) for _ in range(nsimplex)]
    for i in range(nvar):
        simplex[i + 1][i] += scale[i]
    fvalue = [func(simplex[i], data) for i in range(nsimplex)]
    iteration = 0
    while True:
        ssworst = ssbest = 0
        for i in range(nsimplex):
            if fvalue[i] > fvalue[ssbest]:
                ssbest = i
            if fvalue[i] < fvalue[ssworst]:
                ssworst = i
        pavg = [0.0] * nvar
        for i in range(nsimplex):
            if i != ssworst:
                for j in range(nvar):
                    pavg[j] += simplex[i][j]
        pavg = [p / nvar for p in pavg]
        simscale = sum(abs(pavg[i] - simplex[ssworst][i]) / scale[i] for i in range(nvar)) / nvar
        fscale = (abs(fvalue[ssbest]) + abs(fvalue[ssworst])) / 2.0
        frange = abs(fvalue[ssbest] - fvalue[ssworst]) / fscale if fscale != 0.0 else 0.0
        if (((ftolerance <= 0.0 or frange < ftolerance) and (xtolerance <= 0.0 or simscale < xtolerance))
                or (itmax and iteration >= itmax)):
            return simplex[ssbest], fvalue[ssbest], iteration
        pnew = [2.0 * pavg[i] - simplex[ssworst][i] for i in range(nvar)]
        fnew = func(pnew, data)
        if fnew <= fvalue[ssworst]:
            for i in range(nsimplex):
                if i != ssbest and i != ssworst:
                    for j in range(nvar):
                        simplex[i][j] = 0.5 * simplex[ssbest][j] + 0.5 * simplex[i][j]
                    fvalue[i] = func(simplex[i], data)
            pnew = [0.5 * simplex[ssbest][j] + 0.5 * simplex[ssworst][j] for j in range(nvar)]
            fnew = func(pnew, data)
        elif fnew >= fvalue[ssbest]:
            pnew2 = [3.0 * pavg[i] - 2.0 * simplex[ssworst][i] for i in range(nvar)]
            fnew2 = func(pnew2, data)
            if fnew2 > fnew:
                pnew, fnew = pnew2, fnew2
        simplex[ssworst] = list(pnew)
        fvalue[ssworst] = fnew
        iteration += 1


def _fit_tanh_func(args, data):
    v = 0.0
    if data[1][0] < 0.0:
        data[1][0] *= -1.0
    for i in range(len(data[0])):
        f = 2 * data[1][i] / (1.0 + data[1][i])
        if args[0] == 0 or args[1] == 0:
            qt = 0
        else:
            qt = f - 0.5 * (math.tanh(math.pi * (data[0][i] + args[0]) / 2.0 / args[1] / args[0]) -
                            math.tanh(math.pi * (data[0][i] - args[0]) / 2.0 / args[1] / args[0]))
        v -= qt * qt
    return v


def fit_tanh(dres, low=0.1):
    """sp_filter.fit_tanh: (cut-off frequency, fall-off) of the tangent filter fitted to the FSC
    (half-data FSC converted with 2r/(1+r); the curve is zeroed after its first drop below `low`).
    Modifies dres[1] in place, as the original does."""
    setzero = False
    for i in range(1, len(dres[0])):
        if not setzero:
            if 2 * dres[1][i] / (1.0 + dres[1][i]) < low:
                setzero = True
        if setzero:
            dres[1][i] = 0.0
    freq = -1.0
    for i in range(1, len(dres[0]) - 1):
        if 2 * dres[1][i] / (1.0 + dres[1][i]) < 0.5:
            freq = dres[0][i - 1]
            break
    if freq < 0.0:
        if dres[1][len(dres[1]) - 1] < 0.5:
            return 0.5, 0.2
        return 0.49, 0.1
    result = amoeba([freq, 0.1], [0.05, 0.05], _fit_tanh_func, data=dres)
    return result[0][0], result[0][1]


def tanl_filter_values(nx, fl, aa):
    """H on the half spectrum [ny][nx/2+1]; formula pinned by cuda/gpu_aln_noref.cu:799-814"""
    ky = np.arange(nx)
    ky = np.where(ky > nx // 2, ky - nx, ky).astype(np.float64)[:, None] / nx
    kx = np.arange(nx // 2 + 1).astype(np.float64)[None, :] / nx
    return tanl_profile(np.sqrt(kx * kx + ky * ky), fl, aa)


def tanl_profile(d, fl, aa):
    """tangent low-pass H(d) at absolute frequency d; pinned by tests/golden/tanl_ref.npz, which holds the
    output of the reference tree's own kernel text (cuda/gpu_aln_noref.cu:799-814)"""
    c = math.pi / (2.0 * aa * fl)
    return 0.5 * (np.tanh(c * (d + fl)) - np.tanh(c * (d - fl)))


def filt_tanl(img, fl, aa):
    """sp_filter.filt_tanl (no padding)"""
    nx = img.shape[0]
    return np.fft.irfft2(np.fft.rfft2(img.astype(np.float64)) * tanl_filter_values(nx, fl, aa), s=img.shape).astype(np.float32)


def phase_cog(img):
    """EMData::phase_cog (2-D): centre of gravity from the phase of the first Fourier harmonic of
    the row / column sums, relative to the image centre nx/2"""
    ny, nx = img.shape
    P = 2 * math.pi / nx
    X = img.astype(np.float64).sum(0)       # column sums, indexed by x
    T = img.astype(np.float64).sum(1)       # row sums, indexed by y
    j = np.arange(ny); i = np.arange(nx)
    C, S = float((np.cos(P * j) * T).sum()), float((np.sin(P * j) * T).sum())
    F1 = math.atan2(S, C)
    if F1 < 0.0:
        F1 += 2 * math.pi
    sny = F1 / P
    C, S = float((np.cos(P * i) * X).sum()), float((np.sin(P * i) * X).sum())
    F1 = math.atan2(S, C)
    if F1 < 0.0:
        F1 += 2 * math.pi
    snx = F1 / P
    return snx - nx // 2, sny - ny // 2


def fshift(img, delx, dely):
    """sp_fundamentals.fshift: Fourier shift of the image content by (+delx, +dely) pixels"""
    ny, nx = img.shape
    ky = np.arange(ny)
    ky = np.where(ky > ny // 2, ky - ny, ky).astype(np.float64)[:, None]
    kx = np.arange(nx // 2 + 1).astype(np.float64)[None, :]
    ph = np.exp(-2j * math.pi * (kx * delx / nx + ky * dely / ny))
    return np.fft.irfft2(np.fft.rfft2(img.astype(np.float64)) * ph, s=img.shape).astype(np.float32)


def ref_ali2d(mask, center, av, frsc):
    """sp_user_functions.ref_ali2d(ref_data = [mask, center, av, fsc]): tangent low-pass fitted to
    the FSC (clamped: fall-off <= 0.2, 0.12 <= cut-off <= 0.4), then center_2D(method `center`)"""
    fl, aa = fit_tanh(frsc)
    aa = min(aa, 0.2)
    fl = max(min(0.4, fl), 0.12)
    tavg = filt_tanl(av, fl, aa)
    cs = [0.0, 0.0]
    if center == 1:
        cs = list(phase_cog(tavg))
        tavg = fshift(tavg, -cs[0], -cs[1])
    return tavg, cs, fl, aa


def normalize_mask(img, mask):
    """normalize.mask(no_sigma=1): zero mean, unit sigma under the mask"""
    m = mask > 0.5
    v = img[m].astype(np.float64)
    n = v.size
    mean = np.float32(v.sum() / n)
    sigma = np.float32(math.sqrt((np.sum(v * v) - v.sum() ** 2 / n) / (n - 1)))
    return ((img - mean) / sigma).astype(np.float32)


def mref_reference_update(sums, counts, mask, center=1, min_count=4, replacements=None):
    """test_mref_gpu_align.py:517-564 for the default user function.  sums [R][2][nx][nx], counts [R];
    replacements: {class: image} for vanished classes (the reference draws a random particle).
    Returns (new references [R][nx][nx], averaged fsc [3][len], (fl, aa), cs [R][2])."""
    R = sums.shape[0]
    ave_fsc, c_fsc, frsc = None, 0, None
    refs = np.zeros((R,) + sums.shape[2:], np.float32)
    for j in range(R):
        if counts[j] < min_count:
            refs[j] = replacements[j]
        else:
            frsc = fsc(sums[j, 0], sums[j, 1], 1.0)
            refs[j] = (sums[j, 0] + sums[j, 1]) * np.float32(1.0 / float(np.float32(counts[j])))
            if ave_fsc is None:
                ave_fsc, c_fsc = list(frsc[1]), 1
            else:
                ave_fsc = [a + b for a, b in zip(ave_fsc, frsc[1])]
                c_fsc += 1
    if ave_fsc is not None and sum(ave_fsc) != 0:
        frsc[1] = [a / float(c_fsc) for a in ave_fsc]
    curve = [list(frsc[0]), list(frsc[1]), list(frsc[2])]
    out = np.zeros_like(refs)
    css = np.zeros((R, 2))
    fl = aa = None
    for j in range(R):
        # fit_tanh edits the curve in place; every class sees the same (already edited) curve
        tavg, cs, fl, aa = ref_ali2d(mask, center, refs[j], frsc)
        out[j] = normalize_mask(tavg, mask)
        css[j] = cs
    return out, curve, (fl, aa), css
