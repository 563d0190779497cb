// Sub-bin angle of the winner, re-evaluated the way the CPU path evaluates it.
//
// The search kernels find the winning (reference, offset, mirror, angular bin) on f32 data: f32 ring FFTs of their own
// factorisation, f32 accumulation of the ring products on the matrix cores, f32 inverse FFT.  The integer winner agrees
// with the CPU path (audited float ties aside), but the SUB-BIN angle comes from Util::prb1d on the 7 CCF samples
// around the peak, pos = c2 / (2 c3) - 4 with c3 = 5 b1 - 3 b3 - 4 b4 - 3 b5 + 5 b7 -- a second difference that cancels to
// almost nothing on a flat peak (reference-free alignment against the blurred average of an unaligned stack), where it
// turns the 1e-6 relative difference between an f32 and an f64 neighbourhood into degrees.
//
// refine_winner_kernel recomputes that neighbourhood for ONE (offset, reference, mirror) per particle with the CPU
// path's own arithmetic (oracle/ralign_oracle.c restates the same routines):
//   Polar2Dm   bilinear samples in Util::bilinear's operation order (bit-identical positions and values)
//   Normalize_ring (multi-reference mode)  (v - avg) / sigma per sample in f32; avg and sigma from f64 sums -- a uniform
//              shift / scale cancels in prb1d (both coefficient sets sum to zero, pos is a ratio), only the rounding of the
//              individual samples matters
//   Frngs      the radix-2 real FFT of fftr_q in f32, table twiddles, no contraction: bit-identical spectra
//   Crosrng_ms f32 products, f64 accumulation over the rings in ring order, against reference spectra prepared with the
//              same three routines + Applyws (refspec_exact_kernel)
//   inverse    the 7 samples q(jtot - 3 .. jtot + 3) by direct f64 summation of the inverse real DFT (fftr_d computes all
//              maxrin of them; the values agree to f64 rounding)
// One wave per particle; every ring is transformed serially by one lane (the price of the CPU's operation order), so the
// kernel runs for the particles finalize_kernel flags as ill-conditioned (|c3| < refine_thr x max |b|), or for all of
// them when the threshold is infinite.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq -> Crosrng_ms -> prb1d (test_mref_gpu_align.py:1043-1044,
// test_reffree_gpu_align.py:844-847).
#pragma once

#include "ralign_kernels.h"

namespace ralign {

// in-place radix-2 complex FFT of fftr_q / cfft_f (oracle/ralign_oracle.c), forward (sign -1), n a power of two;
// tw + twoff[l] = e^{-2 pi i k / 2^l}, k < 2^(l-1), (float) of the double-precision cos / sin
__device__ __forceinline__ void exact_cfft_fwd(float *re, float *im, int n, const float *__restrict__ tw, const int *__restrict__ twoff)
{
#pragma clang fp contract(off)
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    int l = 1;
    for (int len = 2; len <= n; len <<= 1, l++) {
        const int half = len >> 1;
        const float *t = tw + twoff[l];
        for (int k = 0; k < half; k++) {
            const float wr = t[2 * k], wi = t[2 * k + 1];
            for (int i = k; i < n; i += len) {
                const int j = i + half;
                const float tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] += tr; im[i] += ti;
            }
        }
    }
}

// rfft_fwd_f: x[n] -> packed [X0, X(n/2), Re X1, Im X1, ...] in place; wre / wim: n/2 floats each
__device__ __forceinline__ void exact_rfft_fwd(float *x, int n, float *wre, float *wim, const float *__restrict__ tw,
                                               const int *__restrict__ twoff)
{
#pragma clang fp contract(off)
    const int h = n / 2;
    const float *t = tw + twoff[31 - __clz(n)];
    for (int i = 0; i < h; i++) { wre[i] = x[2 * i]; wim[i] = x[2 * i + 1]; }
    exact_cfft_fwd(wre, wim, h, tw, twoff);
    x[0] = wre[0] + wim[0];
    x[1] = wre[0] - wim[0];
    for (int k = 1; k < h; k++) {
        const int m = h - k;
        const float er = 0.5f * (wre[k] + wre[m]), ei = 0.5f * (wim[k] - wim[m]);
        const float orr = 0.5f * (wim[k] + wim[m]), oi = -0.5f * (wre[k] - wre[m]);
        const float c = t[2 * k], s = t[2 * k + 1];
        x[2 * k] = er + orr * c - oi * s;
        x[2 * k + 1] = ei + orr * s + oi * c;
    }
}

// references: Polar2Dm(cnx, cny) -> Frngs -> Applyws with the CPU path's arithmetic, natural (EMAN2) ring layout [lcirc]
__global__ __launch_bounds__(64) void refspec_exact_kernel(DevGeom g, const int *__restrict__ numr, const float *__restrict__ wr,
                                                           const float *__restrict__ tw, const int *__restrict__ twoff,
                                                           const float *__restrict__ refs, int nref, float *__restrict__ out)
{
#pragma clang fp contract(off)
    extern __shared__ float lds[];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= nref) return;
    float *circ = lds, *work = lds + g.lcirc;
    const float *img = refs + (size_t)r * g.nx * g.nx;
    const float c = (float)g.cnx;
    for (int i = lane; i < g.lcirc; i += 64) circ[i] = bilinear_1b(img, g.nx, g.samp_dx[i] + c, g.samp_dy[i] + c);
    __syncthreads();
    for (int i = lane; i < g.nring; i += 64) {
        const int n = numr[3 * i + 2], o = numr[3 * i + 1] - 1;
        exact_rfft_fwd(circ + o, n, work + o, work + o + n / 2, tw, twoff);
        const float w = wr[i];
        circ[o] *= w;
        if (n == g.maxrin) circ[o + 1] *= w;
        else circ[o + 1] *= 0.5f * w;
        for (int j = 2 + o; j < n + o; j++) circ[j] *= w;
    }
    __syncthreads();
    for (int i = lane; i < g.lcirc; i += 64) out[(size_t)r * g.lcirc + i] = circ[i];
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// tail of finalize_kernel for a given sub-bin position: Util::ang_n (mode F), the ormq shift rotation, combine_params2
__device__ __forceinline__ void finish_params(const DevGeom &g, float sxi, float syi, int jtot, float pos, int bs,
                                              float *alpha_out, float *sx_out, float *sy_out)
{
    const float tot = (float)jtot + pos;
    const float ang = fmodf(((tot - 1.0f) / g.maxrin + 1.0f) * 360.0f, 360.0f);
    const float ixw = g.shift_x[bs], iyw = g.shift_y[bs];
    const float sx = -ixw, sy = -iyw;
    const float co = (float)cos((double)ang * M_PI / 180.0), so = (float)(-sin((double)ang * M_PI / 180.0));
    const float sxs = sx * co - sy * so, sys = sx * so + sy * co;
    const double a = (double)ang * M_PI / 180.0, c = cos(a), s = sin(a);
    const double tx = c * (double)(-sxi) + s * (double)(-syi) + (double)sxs;
    const double ty = -s * (double)(-sxi) + c * (double)(-syi) + (double)sys;
    double alpha = atan2(s, c) * 180.0 / M_PI;
    alpha = fmod(alpha, 360.0);
    if (alpha < 0) alpha += 360.0;
    if (alpha >= 360.0) alpha -= 360.0;
    *alpha_out = (float)alpha; *sx_out = (float)tx; *sy_out = (float)ty;
}

// refx: exact reference spectra [nref][lcirc] (refspec_exact_kernel); res, particles, cls: of the chunk (indexed by rec.p)
__global__ __launch_bounds__(64) void refine_winner_kernel(DevGeom g, const int *__restrict__ numr, const float *__restrict__ tw,
                                                           const int *__restrict__ twoff, const float *__restrict__ particles,
                                                           const float *__restrict__ refx, const RefineRec *__restrict__ list,
                                                           const int *__restrict__ count, ra_result *__restrict__ res,
                                                           const int *__restrict__ cls)
{
#pragma clang fp contract(off)
    if ((int)blockIdx.x >= *count) return;
    extern __shared__ float lds[];
    const RefineRec rec = list[blockIdx.x];
    const int lane = threadIdx.x;
    float *circ = lds, *work = lds + g.lcirc;
    const float *img = particles + (size_t)rec.p * g.nx * g.nx;
    const float cx = ((float)g.cnx + rec.sxi) + g.shift_x[rec.bs], cy = ((float)g.cnx + rec.syi) + g.shift_y[rec.bs];
    double av = 0.0, sq = 0.0;
    for (int i = lane; i < g.lcirc; i += 64) {
        const float v = bilinear_1b(img, g.nx, g.samp_dx[i] + cx, g.samp_dy[i] + cy);
        circ[i] = v;
        const float w = g.samp_w[i];
        av += (double)(v * w); sq += (double)(v * v * w);
    }
    if (g.mode == RA_MODE_MREF) {
        av = wave_sum_f64(av); sq = wave_sum_f64(sq);
        const float nn = g.nn_weight, avf = (float)av, sqf = (float)sq;
        const float avg = avf / nn;
        const float sgm = sqrtf((sqf - avf * avf / nn) / nn);
        __syncthreads();
        for (int i = lane; i < g.lcirc; i += 64) { float v = circ[i]; v -= avg; v /= sgm; circ[i] = v; }
    }
    __syncthreads();
    for (int i = lane; i < g.nring; i += 64) {
        const int n = numr[3 * i + 2], o = numr[3 * i + 1] - 1;
        exact_rfft_fwd(circ + o, n, work + o, work + o + n / 2, tw, twoff);
    }
    __syncthreads();
    // Crosrng_ms: q (straight) or t (mirrored) spectrum of the winner, f32 products, f64 sums over the rings in ring order
    const int N = g.maxrin;
    double *spec = reinterpret_cast<double *>(work);          // [N] doubles (the FFT work space is free again: N <= lcirc / 2)
    const float *c1 = refx + (size_t)(cls ? cls[rec.p] : rec.ref) * g.lcirc;      // class-resident mode: the particle's own class
    const bool mir = rec.mirror != 0;
    for (int j = 2 * lane; j < N; j += 128) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < g.nring; i++) {
            const int n = numr[3 * i + 2], o = numr[3 * i + 1] - 1;
            const float *c = c1 + o, *d = circ + o;
            if (j == 0) {
                s0 += (double)(c[0] * d[0]);
                if (n == N) s1 += (double)(c[1] * d[1]);
            } else if (j < n) {
                const float a1 = c[j], a2 = c[j + 1], d1 = d[j], d2 = d[j + 1];
                const float p1 = a1 * d1, p2 = a2 * d2, p3 = a1 * d2, p4 = a2 * d1;
                if (mir) { s0 += (double)(p1 - p2); s1 += (double)(-p3 - p4); }
                else { s0 += (double)(p1 + p2); s1 += (double)(-p3 + p4); }
            } else if (j == n) {
                s0 += (double)(c[1] * d[1]);          // Nyquist coefficient of a ring shorter than maxrin: real, at index n
            }
        }
        spec[j] = s0; spec[j + 1] = s1;
    }
    __syncthreads();
    // 7 samples of the inverse real transform around the peak: x[m] = (X0 + (-1)^m X_{N/2} + 2 sum_k Re(X_k e^{+2 pi i k m / N})) / N
    double b[7];
#pragma unroll
    for (int t = 0; t < 7; t++) b[t] = 0.0;
    const int m0 = rec.jtot - 1;
    for (int k = 1 + lane; k < N / 2; k += 64) {
        const double xr = spec[2 * k], xi = spec[2 * k + 1];
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int m = (m0 + t - 3 + N) & (N - 1);
            const int km = (int)(((long long)k * m) & (N - 1));
            double sn, cs;
            sincospi(2.0 * (double)km / (double)N, &sn, &cs);
            b[t] += xr * cs - xi * sn;
        }
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        b[t] = wave_sum_f64(b[t]);
        const int m = (m0 + t - 3 + N) & (N - 1);
        b[t] = (spec[0] + ((m & 1) ? -spec[1] : spec[1]) + 2.0 * b[t]) / (double)N;
    }
    if (lane == 0) {
        const double c2 = 49. * b[0] + 6. * b[1] - 21. * b[2] - 32. * b[3] - 27. * b[4] - 6. * b[5] + 31. * b[6];
        const double c3 = 5. * b[0] - 3. * b[2] - 4. * b[3] - 3. * b[4] + 5. * b[6];
        const float pos = (c3 != 0.0) ? (float)(c2 / (2.0 * c3) - 4) : 0.f;
        float alpha, sx, sy;
        finish_params(g, rec.sxi, rec.syi, rec.jtot, pos, rec.bs, &alpha, &sx, &sy);
        res[rec.p].alpha = alpha; res[rec.p].sx = sx; res[rec.p].sy = sy;
    }
}

}  // namespace ralign
