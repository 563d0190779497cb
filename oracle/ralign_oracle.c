/*
 * ralign_oracle.c -- CPU restatement of the EMAN2 2.31 / SPHIRE 2-D alignment
 * path called by the reference's CPU drivers.  See ralign_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from
 * the product path (cryo_ralib_amd/).  PARITY UNPINNED (see header).
 *
 * Every function cites the reference call site it restates (paths relative to
 * /root/reference) and the EMAN2 routine whose published algorithm it follows
 * (SURVEY.md Appendix A).  Data are float, accumulators follow EMAN2 (float in
 * Normalize_ring, double in Crosrng_ms / normalize.mask / Transform algebra).
 */
#include "ralign_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ rings */

static void build_twiddles(int maxn);
static int ilog2_floor(int n) { int l = -1; while (n > 0) { n >>= 1; l++; } return l; }

/* sp_alignment.Numrinit(first,last,skip,"F") + ringwe(numr,"F")
 * reference call sites: test_mref_gpu_align.py:984-985, 348-349 */
int orc_rings_init(orc_rings *rg, int first_ring, int last_ring, int skip)
{
    const int MAXFFT = 32768;
    const double dpi = 2.0 * M_PI;
    int lcirc = 1, nring = 0;
    memset(rg, 0, sizeof(*rg));
    rg->first_ring = first_ring; rg->last_ring = last_ring; rg->skip = skip;
    for (int k = first_ring; k <= last_ring; k += skip) {
        if (nring >= ORC_MAXRING) return -1;
        int jp = (int)(dpi * k + 0.5);
        int ip = 1 << (ilog2_floor(jp) + 1);
        if (k + skip <= last_ring && jp > ip + ip / 2) ip = (2 * ip < MAXFFT) ? 2 * ip : MAXFFT;
        if (k + skip >  last_ring && jp > ip + ip / 5) ip = (2 * ip < MAXFFT) ? 2 * ip : MAXFFT;
        rg->numr[3 * nring + 0] = k;
        rg->numr[3 * nring + 1] = lcirc;
        rg->numr[3 * nring + 2] = ip;
        lcirc += ip;
        nring++;
    }
    rg->nring = nring;
    rg->lcirc = lcirc - 1;
    rg->maxrin = rg->numr[3 * nring - 1];
    build_twiddles(rg->maxrin);
    /* ringwe: wr[i] = r * dpi / n * maxrin / n  (python doubles -> vector<float>) */
    for (int i = 0; i < nring; i++) {
        double n = (double)rg->numr[3 * i + 2];
        rg->wr[i] = (float)((double)rg->numr[3 * i] * dpi / n * (double)rg->maxrin / n);
    }
    return nring;
}

/* --------------------------------------------------------- interpolation */

/* Util::bilinear : 1-based coordinates, no bounds check, no wrap */
static inline float bilinear_1b(float xold, float yold, int nsam, int nrow, const float *xim)
{
    int ixold = (int)xold, iyold = (int)yold;
    float ydif = yold - iyold, xdif = xold - ixold;
    /* EMAN2 reads xim(ixold+1, .) / xim(., iyold+1) unchecked; search_range lets a sample land
     * exactly on the last column / row (xdif or ydif == 0), where that tap lies one past the
     * image and carries zero weight.  Clamp the tap instead of reading out of bounds. */
    int x0 = ixold < 1 ? 1 : (ixold > nsam ? nsam : ixold), x1 = x0 + 1 > nsam ? nsam : x0 + 1;
    int y0 = iyold < 1 ? 1 : (iyold > nrow ? nrow : iyold), y1 = y0 + 1 > nrow ? nrow : y0 + 1;
    float f00 = xim[(size_t)(y0 - 1) * nsam + (x0 - 1)], f10 = xim[(size_t)(y0 - 1) * nsam + (x1 - 1)];
    float f01 = xim[(size_t)(y1 - 1) * nsam + (x0 - 1)], f11 = xim[(size_t)(y1 - 1) * nsam + (x1 - 1)];
    return f00 + ydif * (f01 - f00) + xdif * (f10 - f00 + ydif * (f11 - f10 - f01 + f00));
}

/* Util::quadri : 1-based, circulant wrap (older alrl_ms; switchable) */
static inline float quadri_1b(float xx, float yy, int nxdata, int nydata, const float *fdata)
{
#define FD(i, j) fdata[(size_t)((j) - 1) * nxdata + ((i) - 1)]
    float x = xx, y = yy;
    while (x < 1.0f) x += nxdata;
    while (x >= (float)(nxdata + 1)) x -= nxdata;
    while (y < 1.0f) y += nydata;
    while (y >= (float)(nydata + 1)) y -= nydata;
    int i = (int)x, j = (int)y;
    float dx0 = x - i, dy0 = y - j;
    int ip1 = i + 1, im1 = i - 1, jp1 = j + 1, jm1 = j - 1;
    if (ip1 > nxdata) ip1 -= nxdata;
    if (im1 < 1) im1 += nxdata;
    if (jp1 > nydata) jp1 -= nydata;
    if (jm1 < 1) jm1 += nydata;
    float f0 = FD(i, j);
    float c1 = FD(ip1, j) - f0;
    float c2 = (c1 - f0 + FD(im1, j)) * 0.5f;
    float c3 = FD(i, jp1) - f0;
    float c4 = (c3 - f0 + FD(i, jm1)) * 0.5f;
    float dxb = dx0 - 1, dyb = dy0 - 1;
    int hxc = (dx0 >= 0) ? 1 : -1, hyc = (dy0 >= 0) ? 1 : -1;
    int ic = i + hxc, jc = j + hyc;
    if (ic > nxdata) ic -= nxdata; else if (ic < 1) ic += nxdata;
    if (jc > nydata) jc -= nydata; else if (jc < 1) jc += nydata;
    float c5 = ((FD(ic, jc) - f0 - hxc * c1 - (hxc * (hxc - 1.0f)) * c2
                 - hyc * c3 - (hyc * (hyc - 1.0f)) * c4) * (hxc * hyc));
    return f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
#undef FD
}

/* Util::quadri_background as restated in the reference tree
 * (notebook/02_CuPy_Image_Processing_rot_shift2d.ipynb cell 2) */
static inline float quadri_background_1b(float xx, float yy, int nxdata, int nydata,
                                         const float *fdata, int xnew, int ynew)
{
#define FD(i, j) fdata[(size_t)((j) - 1) * nxdata + ((i) - 1)]
    float x = xx, y = yy;
    if ((x < 1.0f) || (x >= (float)(nxdata + 1)) || (y < 1.0f) || (y >= (float)(nydata + 1))) {
        x = (float)xnew;
        y = (float)ynew;
    }
    int i = (int)x, j = (int)y;
    float dx0 = x - i, dy0 = y - j;
    int ip1 = i + 1, im1 = i - 1, jp1 = j + 1, jm1 = j - 1;
    if (ip1 > nxdata) ip1 -= nxdata;
    if (im1 < 1) im1 += nxdata;
    if (jp1 > nydata) jp1 -= nydata;
    if (jm1 < 1) jm1 += nydata;
    float f0 = FD(i, j);
    float c1 = FD(ip1, j) - f0;
    float c2 = (c1 - f0 + FD(im1, j)) * 0.5f;
    float c3 = FD(i, jp1) - f0;
    float c4 = (c3 - f0 + FD(i, jm1)) * 0.5f;
    float dxb = dx0 - 1, dyb = dy0 - 1;
    int hxc = (dx0 >= 0) ? 1 : -1, hyc = (dy0 >= 0) ? 1 : -1;
    int ic = i + hxc, jc = j + hyc;
    if (ic > nxdata) ic -= nxdata; else if (ic < 1) ic += nxdata;
    if (jc > nydata) jc -= nydata; else if (jc < 1) jc += nydata;
    float c5 = ((FD(ic, jc) - f0 - hxc * c1 - (hxc * (hxc - 1.0f)) * c2
                 - hyc * c3 - (hyc * (hyc - 1.0f)) * c4) * (hxc * hyc));
    return f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
#undef FD
}

static inline float interp_1b(float x, float y, int nx, int ny, const float *im, int interp)
{
    return interp == ORC_INTERP_QUADRI ? quadri_1b(x, y, nx, ny, im) : bilinear_1b(x, y, nx, ny, im);
}

/* ------------------------------------------------------------- Polar2Dm */

/* Util::Polar2Dm -> Util::alrl_ms, mode 'F'.  Sample j of a ring with radius r and
 * length n sits at (cnx + r sin(2 pi j/n), cny + r cos(2 pi j/n)); the routine walks the
 * first quadrant with a float angle and mirrors it into the other three.
 * reference call site: test_mref_gpu_align.py:1015 (refs) and inside
 * Util.multiref_polar_ali_2d (:1043). */
void orc_polar2dm(const float *img, int nx, int ny, float cnx, float cny,
                  const orc_rings *rg, float *circ, int interp)
{
    const double dpi = 2 * atan(1.0);
    for (int it = 0; it < rg->nring; it++) {
        int inr = rg->numr[3 * it], kcirc = rg->numr[3 * it + 1] - 1, l = rg->numr[3 * it + 2];
        int lt = l / 4, nsim = lt - 1;
        double dfi = dpi / (nsim + 1);
        float *c = circ + kcirc;
        c[0]      = interp_1b(0.0f + cnx, inr + cny, nx, ny, img, interp);
        c[lt]     = interp_1b(inr + cnx, 0.0f + cny, nx, ny, img, interp);
        c[2 * lt] = interp_1b(0.0f + cnx, -inr + cny, nx, ny, img, interp);
        c[3 * lt] = interp_1b(-inr + cnx, 0.0f + cny, nx, ny, img, interp);
        for (int jt = 1; jt <= nsim; jt++) {
            float fi = (float)(dfi * jt);
            float x = sinf(fi) * inr, y = cosf(fi) * inr;
            c[jt]          = interp_1b(x + cnx, y + cny, nx, ny, img, interp);
            c[jt + lt]     = interp_1b(y + cnx, -x + cny, nx, ny, img, interp);
            c[jt + 2 * lt] = interp_1b(-x + cnx, -y + cny, nx, ny, img, interp);
            c[jt + 3 * lt] = interp_1b(-y + cnx, x + cny, nx, ny, img, interp);
        }
    }
}

/* Util::Normalize_ring : float accumulators, arc-length weights */
void orc_normalize_ring(float *circ, const orc_rings *rg)
{
    float av = 0.0f, sq = 0.0f, nn = 0.0f;
    for (int i = 0; i < rg->nring; i++) {
        int n = rg->numr[3 * i + 2], o = rg->numr[3 * i + 1] - 1;
        float w = (float)(rg->numr[3 * i] * 2 * M_PI / (float)n);
        for (int j = 0; j < n; j++) {
            float v = circ[o + j];
            av += v * w;
            sq += v * v * w;
            nn += w;
        }
    }
    float avg = av / nn;
    float sgm = sqrtf((sq - av * av / nn) / nn);
    for (int i = 0; i < rg->lcirc; i++) {
        circ[i] -= avg;
        circ[i] /= sgm;
    }
}

/* ------------------------------------------------------------------ FFTs */
/* EMAN2's fftr_q / fftr_d are radix-2 real FFTs (n/2-point complex transform plus a
 * split step), unscaled forward and 1/n-scaled inverse, so that inverse(forward(x)) = x.
 * Only that pairing matters to Crosrng_ms; the restatement uses the same
 * structure and operation count with table twiddles. */

/* twiddle tables e^{-2 pi i k/n}, k < n/2, for n = 2^l (built once in orc_rings_init) */
#define ORC_MAXLOG 16
static double *tw_d[ORC_MAXLOG + 1];
static float  *tw_f[ORC_MAXLOG + 1];
static void build_twiddles(int maxn)
{
    for (int l = 1; l <= ORC_MAXLOG && (1 << l) <= maxn; l++) {
        if (tw_d[l]) continue;
        int n = 1 << l, h = n / 2;
        double *td = (double *)malloc(sizeof(double) * 2 * h);
        float *tf = (float *)malloc(sizeof(float) * 2 * h);
        for (int k = 0; k < h; k++) {
            double a = -2.0 * M_PI * k / n;
            td[2 * k] = cos(a); td[2 * k + 1] = sin(a);
            tf[2 * k] = (float)td[2 * k]; tf[2 * k + 1] = (float)td[2 * k + 1];
        }
        tw_f[l] = tf;
        tw_d[l] = td;
    }
}
static int ilog2_exact(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

#define DEF_FFT(NAME, T, TW)                                                               \
    /* in-place radix-2 complex FFT, n power of two, sign = -1 forward / +1 inverse */     \
    static void cfft_##NAME(T *re, T *im, int n, int sign)                                 \
    {                                                                                      \
        for (int i = 1, j = 0; i < n; i++) {                                               \
            int bit = n >> 1;                                                              \
            for (; j & bit; bit >>= 1) j ^= bit;                                           \
            j ^= bit;                                                                      \
            if (i < j) { T t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; } \
        }                                                                                  \
        int l = 1;                                                                         \
        for (int len = 2; len <= n; len <<= 1, l++) {                                      \
            int half = len >> 1;                                                           \
            const T *tw = TW[l];                                                           \
            for (int k = 0; k < half; k++) {                                               \
                T wr = tw[2 * k], wi = (sign < 0) ? tw[2 * k + 1] : -tw[2 * k + 1];        \
                for (int i = k; i < n; i += len) {                                         \
                    int j = i + half;                                                      \
                    T tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;          \
                    re[j] = re[i] - tr; im[j] = im[i] - ti;                                \
                    re[i] += tr;        im[i] += ti;                                       \
                }                                                                          \
            }                                                                              \
        }                                                                                  \
    }                                                                                      \
    /* forward real FFT of x[n] -> packed [X0, X(n/2), ReX1, ImX1, ...] in place */        \
    static void rfft_fwd_##NAME(T *x, int n, T *wre, T *wim)                               \
    {                                                                                      \
        int h = n / 2;                                                                     \
        const T *tw = TW[ilog2_exact(n)];                                                  \
        for (int i = 0; i < h; i++) { wre[i] = x[2 * i]; wim[i] = x[2 * i + 1]; }          \
        cfft_##NAME(wre, wim, h, -1);                                                      \
        x[0] = wre[0] + wim[0];                                                            \
        x[1] = wre[0] - wim[0];                                                            \
        for (int k = 1; k < h; k++) {                                                      \
            int m = h - k;                                                                 \
            T er = (T)0.5 * (wre[k] + wre[m]), ei = (T)0.5 * (wim[k] - wim[m]);            \
            T orr = (T)0.5 * (wim[k] + wim[m]), oi = (T)-0.5 * (wre[k] - wre[m]);          \
            T c = tw[2 * k], s = tw[2 * k + 1];                                            \
            x[2 * k]     = er + orr * c - oi * s;                                          \
            x[2 * k + 1] = ei + orr * s + oi * c;                                          \
        }                                                                                  \
    }                                                                                      \
    /* inverse of the above (1/n scaled): packed spectrum -> real x[n] in place */         \
    static void rfft_inv_##NAME(T *x, int n, T *wre, T *wim)                               \
    {                                                                                      \
        int h = n / 2;                                                                     \
        const T *tw = TW[ilog2_exact(n)];                                                  \
        /* Z_k = E_k + i O_k, E_k = (X_k + conj X_{h-k})/2, O_k = e^{+2 pi i k/n}(X_k - conj X_{h-k})/2 */ \
        for (int k = 0; k < h; k++) {                                                      \
            T xr, xi, yr, yi;                                                              \
            if (k == 0) { xr = x[0]; xi = 0; yr = x[1]; yi = 0; }                          \
            else { int m = h - k; xr = x[2 * k]; xi = x[2 * k + 1]; yr = x[2 * m]; yi = -x[2 * m + 1]; } \
            T er = (T)0.5 * (xr + yr), ei = (T)0.5 * (xi + yi);                            \
            T dr = (T)0.5 * (xr - yr), di = (T)0.5 * (xi - yi);                            \
            T c = tw[2 * k], s = -tw[2 * k + 1];                                           \
            T orr = dr * c - di * s, oi = dr * s + di * c;                                 \
            wre[k] = er - oi;                                                              \
            wim[k] = ei + orr;                                                             \
        }                                                                                  \
        cfft_##NAME(wre, wim, h, +1);                                                      \
        T sc = (T)1.0 / (T)h;                                                              \
        for (int i = 0; i < h; i++) { x[2 * i] = wre[i] * sc; x[2 * i + 1] = wim[i] * sc; }\
    }

DEF_FFT(f, float, tw_f)
DEF_FFT(d, double, tw_d)

/* Util::Frngs : forward real FFT (fftr_q) of every ring in place
 * reference call site: test_mref_gpu_align.py:1016 */
void orc_frngs(float *circ, const orc_rings *rg)
{
    float wre[ORC_MAXRING * 64], wim[ORC_MAXRING * 64]; /* maxrin/2 <= 16384 */
    for (int i = 0; i < rg->nring; i++) {
        int n = rg->numr[3 * i + 2], o = rg->numr[3 * i + 1] - 1;
        rfft_fwd_f(circ + o, n, wre, wim);
    }
}

/* Util::Applyws  (test_mref_gpu_align.py:1017) */
void orc_applyws(float *circ, const orc_rings *rg)
{
    for (int i = 0; i < rg->nring; i++) {
        int n = rg->numr[3 * i + 2], o = rg->numr[3 * i + 1] - 1;
        float w = rg->wr[i];
        circ[o] *= w;
        if (n == rg->maxrin) circ[o + 1] *= w;
        else                 circ[o + 1] *= 0.5f * w;
        for (int j = 2 + o; j < n + o; j++) circ[j] *= w;
    }
}

/* Util::prb1d, npoint == 7 (coefficients also at cuda/gpu_aln_noref.cu:1434-1442) */
float orc_prb1d7(const double *b)
{
    double c2 = 49. * b[0] + 6. * b[1] - 21. * b[2] - 32. * b[3] - 27. * b[4] - 6. * b[5] + 31. * b[6];
    double c3 = 5. * b[0] - 3. * b[2] - 4. * b[3] - 3. * b[4] + 5. * b[6];
    float pos = 0.0f;
    if (c3 != 0.0) pos = (float)(c2 / (2.0 * c3) - 4);
    return pos;
}

/* Util::ang_n, mode 'F' */
float orc_ang_n(float tot, int maxrin)
{
    return fmodf(((tot - 1.0f) / maxrin + 1.0f) * 360.0f, 360.0f);
}

/* Util::Crosrng_ms : straight (q) and mirrored (t) rotational CCF of two ring sets in
 * Fourier form; float products, double accumulation, double inverse FFT, ">=" scan so
 * the LAST maximum wins, 7-point parabolic refinement. */
void orc_crosrng_ms(const float *circ1, const float *circ2, const orc_rings *rg,
                    double *qn_, float *tot_, double *qm_, float *tmt_,
                    int *jtot_n, int *jtot_m)
{
    const int maxrin = rg->maxrin;
    double *q = (double *)calloc((size_t)maxrin, sizeof(double));
    double *t = (double *)calloc((size_t)maxrin, sizeof(double));
    double *wre = (double *)malloc(sizeof(double) * maxrin);
    double *wim = wre + maxrin / 2;
    for (int i = 0; i < rg->nring; i++) {
        int n = rg->numr[3 * i + 2], o = rg->numr[3 * i + 1] - 1;
        const float *c = circ1 + o, *d = circ2 + o;
        float t1 = c[0] * d[0];
        q[0] += t1; t[0] += t1;
        t1 = c[1] * d[1];
        if (n == maxrin) { q[1] += t1; t[1] += t1; }
        else             { q[n] += t1; t[n] += t1; }
        for (int j = 2; j < n; j += 2) {
            float c1 = c[j], c2 = c[j + 1], d1 = d[j], d2 = d[j + 1];
            float p1 = c1 * d1, p2 = c2 * d2, p3 = c1 * d2, p4 = c2 * d1;
            q[j]     += p1 + p2;
            q[j + 1] += -p3 + p4;
            t[j]     += p1 - p2;
            t[j + 1] += -p3 - p4;
        }
    }
    double t7[7];
    rfft_inv_d(q, maxrin, wre, wim);
    double qn = -1.0e20; int jtot = 0;
    for (int j = 0; j < maxrin; j++) if (q[j] >= qn) { qn = q[j]; jtot = j + 1; }
    for (int k = -3; k <= 3; k++) t7[k + 3] = q[(jtot + k + maxrin - 1) % maxrin];
    float pos = orc_prb1d7(t7);
    *qn_ = qn; *tot_ = (float)jtot + pos; if (jtot_n) *jtot_n = jtot;

    rfft_inv_d(t, maxrin, wre, wim);
    double qm = -1.0e20; jtot = 0;
    for (int j = 0; j < maxrin; j++) if (t[j] >= qm) { qm = t[j]; jtot = j + 1; }
    for (int k = -3; k <= 3; k++) t7[k + 3] = t[(jtot + k + maxrin - 1) % maxrin];
    pos = orc_prb1d7(t7);
    *qm_ = qm; *tmt_ = (float)jtot + pos; if (jtot_m) *jtot_m = jtot;
    free(q); free(t); free(wre);
}

/* ----------------------------------------------------------- the searches */

/* Util::multiref_polar_ali_2d  (reference call: test_mref_gpu_align.py:1043-1044, 771-772) */
void orc_multiref_polar_ali_2d(const float *img, int nx, int ny,
                               const float *crefim, int nref,
                               const float xrng[2], const float yrng[2], float step,
                               const orc_rings *rg, float cnx, float cny,
                               int interp, int normalize_ring,
                               float out[6], orc_search_info *info)
{
    int lkx = (int)(xrng[0] / step), rkx = (int)(xrng[1] / step);
    int lky = (int)(yrng[0] / step), rky = (int)(yrng[1] / step);
    float *cimage = (float *)malloc(sizeof(float) * rg->lcirc);
    int nrefw = 0, mirror = 0, jbest = 0;
    float sx = 0, sy = 0, peak = -1.0E23f, ang = 0.0f, totbest = 0.0f;
    for (int i = -lky; i <= rky; i++) {
        float iy = i * step;
        for (int j = -lkx; j <= rkx; j++) {
            float ix = j * step;
            orc_polar2dm(img, nx, ny, cnx + ix, cny + iy, rg, cimage, interp);
            if (normalize_ring) orc_normalize_ring(cimage, rg);
            orc_frngs(cimage, rg);
            for (int iref = 0; iref < nref; iref++) {
                double qn, qm; float tot, tmt; int jn, jm;
                orc_crosrng_ms(crefim + (size_t)iref * rg->lcirc, cimage, rg, &qn, &tot, &qm, &tmt, &jn, &jm);
                if (qn >= peak || qm >= peak) {
                    sx = -ix; sy = -iy; nrefw = iref;
                    if (qn >= qm) { ang = orc_ang_n(tot, rg->maxrin); peak = (float)qn; mirror = 0; jbest = jn; totbest = tot; }
                    else          { ang = orc_ang_n(tmt, rg->maxrin); peak = (float)qm; mirror = 1; jbest = jm; totbest = tmt; }
                }
            }
        }
    }
    float co = (float)cos(ang * M_PI / 180.0), so = (float)(-sin(ang * M_PI / 180.0));
    out[0] = ang;
    out[1] = sx * co - sy * so;
    out[2] = sx * so + sy * co;
    out[3] = (float)mirror;
    out[4] = (float)nrefw;
    out[5] = peak;
    if (info) { info->ix = -sx; info->iy = -sy; info->jtot = jbest; info->tot = totbest; }
    free(cimage);
}

/* ormq(..., nomirror): with nomirror SPHIRE calls Util.Crosrng_ns, the straight half of Crosrng_ms alone
 * (test_reffree_gpu_align.py:846 passes nomirror down to ali2d_single_iter -> ormq).  Process-wide switch of the checker. */
static int g_nomirror = 0;
void orc_set_nomirror(int flag) { g_nomirror = flag; }
/* hedge of include/ralign.h (ra_options.normalize_ring = 1 in RA_MODE_REFFREE): ormq with Util::Normalize_ring between Polar2Dm and
 * Frngs -- ormq's own rules otherwise (double peak, clamped shifts).  Process-wide switch of the checker, off by default. */
static int g_ormq_normalize = 0;
void orc_set_ormq_normalize(int flag) { g_ormq_normalize = flag; }

/* sp_alignment.ormq : single reference, python doubles, no Normalize_ring
 * (reference call: test_reffree_gpu_align.py:844-847 -> ali2d_single_iter -> ormq) */
void orc_ormq(const float *img, int nx, int ny, const float *crefim,
              const float xrng[2], const float yrng[2], float step,
              const orc_rings *rg, float cnx, float cny, int interp,
              double out[5], orc_search_info *info)
{
    int lkx = (int)(xrng[0] / step), rkx = (int)(xrng[1] / step);
    int lky = (int)(yrng[0] / step), rky = (int)(yrng[1] / step);
    float *cimage = (float *)malloc(sizeof(float) * rg->lcirc);
    double peak = -1.0E23, sx = 0, sy = 0, ang = 0;
    int mirror = 0, jbest = 0; float totbest = 0;
    for (int i = -lky; i <= rky; i++) {
        double iy = (double)i * step;
        for (int j = -lkx; j <= rkx; j++) {
            double ix = (double)j * step;
            orc_polar2dm(img, nx, ny, (float)(cnx + ix), (float)(cny + iy), rg, cimage, interp);
            if (g_ormq_normalize) orc_normalize_ring(cimage, rg);
            orc_frngs(cimage, rg);
            double qn, qm; float tot, tmt; int jn, jm;
            orc_crosrng_ms(crefim, cimage, rg, &qn, &tot, &qm, &tmt, &jn, &jm);
            if (g_nomirror) {
                if (qn >= peak) { sx = -ix; sy = -iy; ang = orc_ang_n(tot, rg->maxrin); peak = qn; mirror = 0; jbest = jn; totbest = tot; }
            } else if (qn >= peak || qm >= peak) {
                sx = -ix; sy = -iy;
                if (qn >= qm) { ang = orc_ang_n(tot, rg->maxrin); peak = qn; mirror = 0; jbest = jn; totbest = tot; }
                else          { ang = orc_ang_n(tmt, rg->maxrin); peak = qm; mirror = 1; jbest = jm; totbest = tmt; }
            }
        }
    }
    double co = cos(ang * M_PI / 180.0), so = -sin(ang * M_PI / 180.0);
    out[0] = ang;
    out[1] = sx * co - sy * so;
    out[2] = sx * so + sy * co;
    out[3] = mirror;
    out[4] = peak;
    if (info) { info->ix = (float)-sx; info->iy = (float)-sy; info->jtot = jbest; info->tot = totbest; }
    free(cimage);
}

/* ------------------------------------------------------ pre/post-processing */

/* sp_utilities.model_circle -> "testimage.circlesphere" {radius, fill=1} */
void orc_model_circle(float radius, int nx, int ny, float *mask, int edge_le)
{
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) {
            float x2 = ((float)i - nx / 2) * ((float)i - nx / 2) / (radius * radius);
            float y2 = ((float)j - ny / 2) * ((float)j - ny / 2) / (radius * radius);
            float r = x2 + y2;
            mask[(size_t)j * nx + i] = (edge_le ? (r <= 1.0f) : (r < 1.0f)) ? 1.0f : 0.0f;
        }
}

/* NormalizeMaskProcessor: mean over mask>0.5 (double sums); no_sigma==0 -> subtract the
 * mean only, otherwise also divide by the sample sigma under the mask
 * (reference call sites: test_mref_gpu_align.py:336, 342, 1014, 1028) */
void orc_normalize_mask(float *img, const float *mask, int npix, int no_sigma)
{
    double sum = 0, sq2 = 0; size_t n = 0;
    for (int i = 0; i < npix; i++)
        if (mask[i] > 0.5f) { sum += img[i]; sq2 += img[i] * (double)img[i]; n++; }
    float mean = (n == 0) ? 0.0f : (float)sum / n;
    float sigma = 1.0f;
    if (no_sigma != 0) sigma = sqrtf((float)((sq2 - sum * sum / n) / (n - 1)));
    for (int i = 0; i < npix; i++) img[i] = (img[i] - mean) / sigma;
}

static inline float restrict2(float x, int nx)
{
    while (x >= (float)nx) x -= nx;
    while (x <= -(float)nx) x += nx;
    return x;
}

/* sp_fundamentals.rot_shift2D(img, ang, sx, sy, mirror) with the defaults "quadratic" /
 * "background" = EMData::rot_scale_trans2D_background(ang,sx,sy,1) then xform.mirror(x);
 * restated after notebook/02_CuPy_Image_Processing_rot_shift2d.ipynb cell 2
 * (reference call site: test_mref_gpu_align.py:1055, 781) */
void orc_rot_shift2d(const float *in, float *out, int nx, int ny,
                     float ang_deg, float delx, float dely, int mirror)
{
    float ang = ang_deg * (float)M_PI / 180.0f;
    delx = restrict2(delx, nx);
    dely = restrict2(dely, ny);
    int xc = nx / 2, yc = ny / 2;
    float shiftxc = xc + delx, shiftyc = yc + dely;
    /* EMAN2: `float cang = cos(ang)` on a float argument; evaluated here in double and rounded,
     * which is what the unqualified C `cos` gives and what the device code can reproduce exactly */
    float cang = (float)cos((double)ang), sang = (float)sin((double)ang);
    for (int iy = 0; iy < ny; iy++) {
        float y = (float)iy - shiftyc;
        float ycang = y * cang + yc;
        float ysang = -y * sang + xc;
        for (int ix = 0; ix < nx; ix++) {
            float x = (float)ix - shiftxc;
            float xold = x * cang + ysang;
            float yold = x * sang + ycang;
            out[(size_t)iy * nx + ix] =
                quadri_background_1b(xold + 1.0f, yold + 1.0f, nx, ny, in, ix + 1, iy + 1);
        }
    }
    if (mirror) {
        /* xform.mirror axis x: columns [1 - nx%2, nx) are reversed (notebook 02 cell 2) */
        int start = 1 - nx % 2;
        for (int iy = 0; iy < ny; iy++) {
            float *row = out + (size_t)iy * nx;
            for (int a = start, b = nx - 1; a < b; a++, b--) { float t = row[a]; row[a] = row[b]; row[b] = t; }
        }
    }
}

/* EMAN2 Transform 2-D algebra: v' = M (R(alpha) v + t),  R = [[c, s], [-s, c]],
 * M = diag(-1, 1) when mirror.  3x3 homogeneous matrices in double. */
static void tf_build(double a_deg, double tx, double ty, int m, double T[3][3])
{
    double a = a_deg * M_PI / 180.0, c = cos(a), s = sin(a);
    double sgn = m ? -1.0 : 1.0;
    T[0][0] = sgn * c;  T[0][1] = sgn * s;  T[0][2] = sgn * tx;
    T[1][0] = -s;       T[1][1] = c;        T[1][2] = ty;
    T[2][0] = 0; T[2][1] = 0; T[2][2] = 1;
}
static void tf_params(const double T[3][3], double out[4])
{
    double det = T[0][0] * T[1][1] - T[0][1] * T[1][0];
    int m = det < 0;
    double sgn = m ? -1.0 : 1.0;
    double c = sgn * T[0][0], s = sgn * T[0][1];
    double alpha = atan2(s, c) * 180.0 / M_PI;
    alpha = fmod(alpha, 360.0);
    if (alpha < 0) alpha += 360.0;
    if (alpha >= 360.0) alpha -= 360.0;
    out[0] = alpha; out[1] = sgn * T[0][2]; out[2] = T[1][2]; out[3] = m;
}

/* sp_utilities.combine_params2 = parameters of T2 * T1 (known answer: cuda/EMAN2_test.ipynb 23-24) */
void orc_combine_params2(double a1, double sx1, double sy1, int m1,
                         double a2, double sx2, double sy2, int m2, double out[4])
{
    double A[3][3], B[3][3], C[3][3];
    tf_build(a1, sx1, sy1, m1, A);
    tf_build(a2, sx2, sy2, m2, B);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            C[i][j] = 0;
            for (int k = 0; k < 3; k++) C[i][j] += B[i][k] * A[k][j];
        }
    tf_params(C, out);
}

/* sp_utilities.inverse_transform2 (known answer: cuda/EMAN2_test.ipynb 25) */
void orc_inverse_transform2(double alpha, double tx, double ty, int mirror, double out[4])
{
    double A[3][3], I[3][3];
    tf_build(alpha, tx, ty, mirror, A);
    double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    I[0][0] = A[1][1] / det;  I[0][1] = -A[0][1] / det;
    I[1][0] = -A[1][0] / det; I[1][1] = A[0][0] / det;
    I[0][2] = -(I[0][0] * A[0][2] + I[0][1] * A[1][2]);
    I[1][2] = -(I[1][0] * A[0][2] + I[1][1] * A[1][2]);
    I[2][0] = 0; I[2][1] = 0; I[2][2] = 1;
    tf_params(I, out);
}

/* The state round trip of the reference's iteration loops: the shift d = (sxi, syi) a search starts from is rebuilt
 * every iteration from the header values (alpha, sx, sy[, mirror]) -- float32 here, as xform.align2d stores them --
 * mref_ali2d: inverse_transform2(alpha, sx, sy) (test_mref_gpu_align.py:1024-1026, mirror not passed);
 * ali2d_single_iter: combine_params2(alpha, sx, sy, mirror, 0, -cs[0], -cs[1], 0) then inverse_transform2
 * (test_reffree_gpu_align.py:844-847).  Algebraically d_new = d_old + (ix, iy); numerically it comes back as e.g.
 * -6.9999995, which can narrow an edge-limited search window by one offset (search_range -> int(range / step)) or
 * trip the |sxi| > mashi reset.  The reference keeps sxi as a Python float; here it is rounded to float32 once
 * (round-trip errors below half a float32 ulp of d are thereby snapped back). mode: 0 = mref, 1 = reference-free. */
void orc_state_from_params(const float *params, int n, int mode, const float cs[2], float *d)
{
    for (int im = 0; im < n; im++) {
        const float *p = params + 6 * (size_t)im;
        double cp[4], inv[4];
        if (mode == 0) {
            orc_inverse_transform2((double)p[0], (double)p[1], (double)p[2], 0, inv);
        } else {
            orc_combine_params2((double)p[0], (double)p[1], (double)p[2], (int)p[3], 0.0, -(double)cs[0], -(double)cs[1], 0, cp);
            orc_inverse_transform2(cp[0], cp[1], cp[2], 0, inv);
        }
        d[2 * im] = (float)inv[1];
        d[2 * im + 1] = (float)inv[2];
    }
}

/* sp_alignment.search_range, returned already swapped as every caller does
 * (test_mref_gpu_align.py:1035-1038): out = {left, right} = {min(ql,range), min(qe,range)} */
void orc_search_range(int n, float radius, float shift, float range, float out[2])
{
    int cn = n / 2 + 1;
    float ql = cn + shift - radius - 2;
    float qe = n - cn - shift - radius;
    if (ql < 0) ql = 0;
    if (qe < 0) qe = 0;
    out[0] = ql < range ? ql : range;
    out[1] = qe < range ? qe : range;
}

/* references: normalize.mask(no_sigma=1) -> Polar2Dm(cnx,cny) -> Frngs -> Applyws
 * (test_mref_gpu_align.py:1013-1018, 741-746) */
void orc_prepare_refs(float *refs, int nref, int nx, const float *mask,
                      const orc_rings *rg, int interp, float *crefim)
{
    float cnx = (float)(nx / 2 + 1), cny = cnx;
    for (int j = 0; j < nref; j++) {
        float *r = refs + (size_t)j * nx * nx;
        if (mask) orc_normalize_mask(r, mask, nx * nx, 1);
        float *c = crefim + (size_t)j * rg->lcirc;
        orc_polar2dm(r, nx, nx, cnx, cny, rg, c, interp);
        orc_frngs(c, rg);
        orc_applyws(c, rg);
    }
}

/* ---------------------------------------------------- iteration drivers */

static void add_img(float *dst, const float *src, int n)
{
    for (int i = 0; i < n; i++) dst[i] += src[i];
}

/* per-particle loop of mref_ali2d / mref_ali2d_MPI (test_mref_gpu_align.py:1023-1060,
 * 754-786).  The state carried per particle is d = (sxi, syi), the value
 * inverse_transform2(alpha, sx, sy) reconstructs from the stored xform.align2d
 * (algebraically d_new = sxi + ix, syi + iy; see DESIGN.md "state"). */
void orc_mref_iteration(const float *particles, int n, int nx,
                        const float *crefim, int nref, const orc_rings *rg,
                        float xrng, float yrng, float step, int interp,
                        int normalize_ring,
                        float *d, float *params, orc_search_info *infos,
                        float *sums, int *counts, int index0, int nthreads)
{
    const int npix = nx * nx;
    const int cnx = nx / 2 + 1, cny = cnx;
    /* the caller's last_ring PARAMETER (test_mref_gpu_align.py:740, 761-766: mashi and search_range take the argument of
     * mref_ali2d, not numr[-3]); with a ring step > 1 the outermost sampled ring can lie below it.  ali2d_single_iter, in
     * contrast, reads ou = numr[-3] (orc_reffree_iteration). */
    const int last_ring = rg->last_ring;
    const float mashi = (float)(cnx - last_ring - 2);
    int *iref_of = (int *)malloc(sizeof(int) * n);
    float *aligned = NULL;
    if (nthreads > 1) aligned = (float *)malloc(sizeof(float) * (size_t)n * npix);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int im = 0; im < n; im++) {
        const float *img = particles + (size_t)im * npix;
        float sxi = d[2 * im], syi = d[2 * im + 1];
        if (fabsf(sxi) > mashi || fabsf(syi) > mashi) { sxi = 0.0f; syi = 0.0f; }
        float txrng[2], tyrng[2], res[6];
        orc_search_range(nx, (float)last_ring, sxi, xrng, txrng);
        orc_search_range(nx, (float)last_ring, syi, yrng, tyrng);
        orc_search_info inf;
        orc_multiref_polar_ali_2d(img, nx, nx, crefim, nref, txrng, tyrng, step, rg,
                                  (float)(cnx + sxi), (float)(cny + syi), interp, normalize_ring,
                                  res, &inf);
        double cp[4];
        orc_combine_params2(0.0, -sxi, -syi, 0, res[0], res[1], res[2], (int)res[3], cp);
        float *p = params + 6 * (size_t)im;
        p[0] = (float)cp[0]; p[1] = (float)cp[1]; p[2] = (float)cp[2]; p[3] = (float)cp[3];
        p[4] = res[4]; p[5] = res[5];
        d[2 * im] = sxi + inf.ix;
        d[2 * im + 1] = syi + inf.iy;
        if (infos) infos[im] = inf;
        iref_of[im] = (int)res[4];
        if (nthreads > 1) {
            orc_rot_shift2d(img, aligned + (size_t)im * npix, nx, nx, p[0], p[1], p[2], (int)p[3]);
        } else {
            float *temp = (float *)malloc(sizeof(float) * npix);
            orc_rot_shift2d(img, temp, nx, nx, p[0], p[1], p[2], (int)p[3]);
            add_img(sums + ((size_t)iref_of[im] * 2 + ((index0 + im) % 2)) * npix, temp, npix);
            counts[iref_of[im]] += 1;
            free(temp);
        }
    }
    if (nthreads > 1) {
        for (int im = 0; im < n; im++) {   /* particle order, like the serial loop */
            add_img(sums + ((size_t)iref_of[im] * 2 + ((index0 + im) % 2)) * npix,
                    aligned + (size_t)im * npix, npix);
            counts[iref_of[im]] += 1;
        }
        free(aligned);
    }
    free(iref_of);
}

/* ali2d_single_iter (SPHIRE sp_alignment; reference call site
 * test_reffree_gpu_align.py:844-847): params[im] = {alpha, sx, sy, mirror, -, peak} is
 * in/out state, cs the average-centre correction folded in before the search,
 * shifts clamped (not reset) to +-mashi. */
void orc_reffree_iteration(const float *particles, int n, int nx,
                           const float *crefim, const orc_rings *rg,
                           float xrng, float yrng, float step, int interp,
                           const float cs[2],
                           float *d, float *params, orc_search_info *infos,
                           float *sums, int index0, int nthreads,
                           double sxsy_sum[2])
{
    const int npix = nx * nx;
    const int cnx = nx / 2 + 1, cny = cnx;
    const int ou = rg->numr[3 * (rg->nring - 1)];
    const float mashi = (float)(cnx - ou - 2);
    float *aligned = (float *)malloc(sizeof(float) * (size_t)n * npix);
    double sx_sum = 0, sy_sum = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int im = 0; im < n; im++) {
        const float *img = particles + (size_t)im * npix;
        float *p = params + 6 * (size_t)im;
        double cp[4], inv[4];
        orc_combine_params2(p[0], p[1], p[2], (int)p[3], 0.0, -cs[0], -cs[1], 0, cp);
        orc_inverse_transform2(cp[0], cp[1], cp[2], 0, inv);
        float sxi = (float)inv[1], syi = (float)inv[2];
        sxi = fminf(fmaxf(sxi, -mashi), mashi);
        syi = fminf(fmaxf(syi, -mashi), mashi);
        float txrng[2], tyrng[2];
        orc_search_range(nx, (float)ou, sxi, xrng, txrng);
        orc_search_range(nx, (float)ou, syi, yrng, tyrng);
        double res[5]; orc_search_info inf;
        orc_ormq(img, nx, nx, crefim, txrng, tyrng, step, rg, (float)(cnx + sxi), (float)(cny + syi),
                 interp, res, &inf);
        orc_combine_params2(0.0, -sxi, -syi, 0, res[0], res[1], res[2], (int)res[3], cp);
        p[0] = (float)cp[0]; p[1] = (float)cp[1]; p[2] = (float)cp[2]; p[3] = (float)cp[3];
        p[4] = 0.0f; p[5] = (float)res[4];
        d[2 * im] = sxi + inf.ix;
        d[2 * im + 1] = syi + inf.iy;
        if (infos) infos[im] = inf;
        orc_rot_shift2d(img, aligned + (size_t)im * npix, nx, nx, p[0], p[1], p[2], (int)p[3]);
    }
    for (int im = 0; im < n; im++) {
        const float *p = params + 6 * (size_t)im;
        if ((int)p[3] == 0) sx_sum += p[1]; else sx_sum -= p[1];
        sy_sum += p[2];
        add_img(sums + (size_t)((index0 + im) % 2) * npix, aligned + (size_t)im * npix, npix);
    }
    if (sxsy_sum) { sxsy_sum[0] = sx_sum; sxsy_sum[1] = sy_sum; }
    free(aligned);
}
