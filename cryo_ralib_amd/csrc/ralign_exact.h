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
// path's own arithmetic:
//   Polar2Dm   bilinear samples in Util::bilinear's operation order (bit-identical positions and values)
//   Normalize_ring (multi-reference mode)  (v - avg) / sigma per sample in f32; avg and sigma from FLOAT sums taken sample after
//              sample in ring order, as Util::Normalize_ring takes them (their rounding walk is ~1e-6 of sigma and differs from
//              offset to offset: it cancels in one candidate's prb1d but orders two offsets whose maxima are closer than that)
//   Frngs      the radix-2 real FFT of fftr_q in f32, table twiddles, no contraction: bit-identical spectra
//   Crosrng_ms f32 products, f64 accumulation over the rings in ring order, against reference spectra prepared with the
//              same three routines + Applyws (refspec_exact_kernel)
//   inverse    all maxrin samples of q (or t) by direct f64 summation of the inverse real DFT (fftr_d computes the same values to
//              f64 rounding), scanned with the CPU path's ">=" for the last maximum: the winner's bin is the f64 CCF's own,
//              wherever the f32 transform put it
// One wave per particle (the butterflies of a radix-2 stage are independent, so they are dealt to the lanes without
// changing an operation); the kernel runs for the particles finalize_kernel flags as ill-conditioned
// (|c3| < refine_thr x max |b|), or for all of them when the threshold is negative.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq -> Crosrng_ms -> prb1d (test_mref_gpu_align.py:1043-1044,
// test_reffree_gpu_align.py:844-847).
#pragma once

#include "ralign_kernels.h"

namespace ralign {

// Frngs with fftr_q's arithmetic (radix-2, bit-reversal first, table
// twiddles, split step), every ring in place in `circ`, `work` = lcirc floats of scratch.  The butterflies of one stage
// are independent of each other, so they are dealt to the lanes without changing a single operation: each of the four
// 16-lane groups of the wave transforms one ring at a time (rings dealt longest first, round robin), one butterfly per
// lane and step.  tw + twoff[l] = e^{-2 pi i k / 2^l}, k < 2^(l-1), (float) of the double-precision cos / sin.
// RA_EXACT_THREADS threads (4 waves) work on one image: 16 groups of 16 lanes, one ring per group and step.
// GM: the ring buffers live in global memory (boxes whose rings exceed the LDS: 271 KB per offset at 256 x 256 / ou = 120); the
// workgroup's threads then hand data to each other through the CU's vector cache.
#define RA_EXACT_THREADS 256
#define RA_TIE_EXP 64            // entries of the replayed scan: candidates x copies of their references (more: the first 64)
#define RA_EXACT_TABLE_BYTES(maxrin) ((size_t)24 * (maxrin))      // refine_winner_kernel: [maxrin] double2 twiddles + [maxrin] double samples
template <bool GM = false> __device__ __forceinline__ void exact_lds_sync()
{
    if constexpr (GM) __threadfence_block();
    __syncthreads();
}

// sum over the workgroup in a fixed order: butterfly inside a wave, the waves in wave order (red: LDS scratch of 4 doubles)
__device__ __forceinline__ double block_sum_f64(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

// maximum over the workgroup with the CPU scan's ">=" rule: of equal values the LARGEST index wins (the scan runs over ascending
// bins and keeps the last maximum); red / redi: LDS scratch of 4 doubles / 4 ints
__device__ __forceinline__ void block_argmax_f64(double &v, int &idx, double *red, int *redi)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi > idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    v = red[0]; idx = redi[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
        if (red[w] > v || (red[w] == v && redi[w] > idx)) { v = red[w]; idx = redi[w]; }
}

template <bool GM = false>
__device__ __forceinline__ void exact_frngs(float *circ, float *work, const DevGeom &g, const int *__restrict__ numr,
                                            const float *__restrict__ tw, const int *__restrict__ twoff, int lane,
                                            const float *__restrict__ wr /* Applyws weights or null */)
{
#pragma clang fp contract(off)
    const int l16 = lane & 15, grp = lane >> 4;          // lane = thread of the workgroup: 16 ring groups
    for (int s0 = 0; s0 < g.nring; s0 += RA_EXACT_THREADS / 16) {
        const int i = g.nring - 1 - (s0 + grp);                 // this group's ring of the slot (rings are sorted by length)
        const bool have = i >= 0;
        const int n = have ? numr[3 * i + 2] : 2, o = have ? numr[3 * i + 1] - 1 : 0, h = n >> 1;
        const int lgh = 31 - __clz(h);
        const int nmax = numr[3 * (g.nring - 1 - s0) + 2];       // the longest ring of the slot bounds the uniform loops
        const int lgmax = 31 - __clz(nmax >> 1);
        float *x = circ + o, *wre = work + o, *wim = work + o + h;
        if (have)
            for (int q = l16; q < h; q += 16) {                   // deinterleave into bit-reversed order: out[rev(q)] = in[q]
                const int r = lgh ? (int)(__brev((unsigned)q) >> (32 - lgh)) : 0;
                wre[r] = x[2 * q]; wim[r] = x[2 * q + 1];
            }
        exact_lds_sync<GM>();
        for (int l = 1; l <= lgmax; l++) {
            if (have && l <= lgh) {
                const int len = 1 << l, half = len >> 1;
                const float *t = tw + twoff[l];
                for (int bfly = l16; bfly < (h >> 1); bfly += 16) {
                    const int k = bfly & (half - 1), a = ((bfly >> (l - 1)) << l) + k, c = a + half;
                    const float w_r = t[2 * k], w_i = t[2 * k + 1];
                    const float tr = wre[c] * w_r - wim[c] * w_i, ti = wre[c] * w_i + wim[c] * w_r;
                    wre[c] = wre[a] - tr; wim[c] = wim[a] - ti;
                    wre[a] += tr; wim[a] += ti;
                }
            }
            exact_lds_sync<GM>();
        }
        if (have) {
            const float *t = tw + twoff[lgh + 1];
            const float w = wr ? wr[i] : 1.0f;
            for (int k = l16; k < h; k += 16) {
                float x0, x1;
                if (k == 0) {
                    x0 = wre[0] + wim[0];
                    x1 = wre[0] - wim[0];
                    if (wr) { x0 *= w; x1 *= (n == g.maxrin) ? w : 0.5f * w; }
                } else {
                    const int m = h - k;
                    const float er = 0.5f * (wre[k] + wre[m]), ei = 0.5f * (wim[k] - wim[m]);
                    const float orr = 0.5f * (wim[k] + wim[m]), oi = -0.5f * (wre[k] - wre[m]);
                    const float c = t[2 * k], sn = t[2 * k + 1];
                    x0 = er + orr * c - oi * sn;
                    x1 = ei + orr * sn + oi * c;
                    if (wr) { x0 *= w; x1 *= w; }
                }
                x[2 * k] = x0; x[2 * k + 1] = x1;
            }
        }
        exact_lds_sync<GM>();
    }
}

// references: Polar2Dm(cnx, cny) -> Frngs -> Applyws with the CPU path's arithmetic, natural (EMAN2) ring layout [lcirc]
// gscr (GM): [gridDim.x][2 lcirc] floats of global scratch
template <bool GM>
__global__ __launch_bounds__(RA_EXACT_THREADS) void refspec_exact_kernel(DevGeom g, const int *__restrict__ numr, const float *__restrict__ wr,
                                                           const float *__restrict__ tw, const int *__restrict__ twoff,
                                                           const float *__restrict__ refs, int nref, float *__restrict__ out,
                                                           float *__restrict__ gscr)
{
#pragma clang fp contract(off)
    extern __shared__ float lds[];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= nref) return;
    float *circ, *work;
    if constexpr (GM) { circ = gscr + (size_t)blockIdx.x * 2 * g.lcirc; work = circ + g.lcirc; }
    else { circ = lds; work = lds + g.lcirc; }
    const float *img = refs + (size_t)r * g.nx * g.nx;
    const float c = (float)g.cnx;
    for (int i = lane; i < g.lcirc; i += RA_EXACT_THREADS) circ[i] = g.interp ? quadri_1b(img, g.nx, g.nx, g.samp_dx[i] + c, g.samp_dy[i] + c) : bilinear_1b(img, g.nx, g.samp_dx[i] + c, g.samp_dy[i] + c);
    exact_lds_sync<GM>();
    exact_frngs<GM>(circ, work, g, numr, tw, twoff, lane, wr);          // Frngs, then Applyws on the way out of the split step
    exact_lds_sync<GM>();
    for (int i = lane; i < g.lcirc; i += RA_EXACT_THREADS) out[(size_t)r * g.lcirc + i] = circ[i];
}

// tail of finalize_kernel for a given sub-bin position: Util::ang_n (mode F), the ormq shift rotation, combine_params2
__device__ __forceinline__ void finish_params(const DevGeom &g, float sxi, float syi, int jtot, float pos, int bs,
                                              float *alpha_out, float *sx_out, float *sy_out)
{
#pragma clang fp contract(off)
    const float tot = (float)jtot + pos;
    const float ang = fmodf(((tot - 1.0f) / g.maxrin + 1.0f) * 360.0f, 360.0f);
    const float ixw = g.shift_x[bs], iyw = g.shift_y[bs];
    const float sx = -ixw, sy = -iyw;
    // (float tail in Util::multiref_polar_ali_2d, Python doubles in sp_alignment.ormq: finalize_tail)
    const double a = (double)ang * M_PI / 180.0, c = cos(a), s = sin(a);
    double sxs, sys;
    if (g.mode == RA_MODE_MREF) {
        const float co = (float)c, so = (float)(-s);
        sxs = (double)(sx * co - sy * so); sys = (double)(sx * so + sy * co);
    } else {
        const double co = c, so = -s;
        sxs = (double)sx * co - (double)sy * so; sys = (double)sx * so + (double)sy * co;
    }
    const double tx = c * (double)(-sxi) + s * (double)(-syi) + sxs;
    const double ty = -s * (double)(-sxi) + c * (double)(-syi) + sys;
    double alpha = atan2(s, c) * 180.0 / M_PI;
    alpha = fmod(alpha, 360.0);
    if (alpha < 0) alpha += 360.0;
    if (alpha >= 360.0) alpha -= 360.0;
    *alpha_out = (float)alpha; *sx_out = (float)tx; *sy_out = (float)ty;
}

// One candidate (search offset bs, reference spectrum c1, orientation mir) of particle image `img`, evaluated as the CPU path
// evaluates it: samples, Normalize_ring, Frngs, the q or t spectrum, the whole CCF in f64 and its maximum (">=": the last of
// equal values, as the CPU scan); jtot <- that bin, b <- the 7 samples around it.  Uniform results.
template <bool GM>
__device__ __forceinline__ void exact_candidate(const DevGeom &g, const int *__restrict__ numr, const float *__restrict__ tw,
                                                const int *__restrict__ twoff, const float *__restrict__ img,
                                                const float *__restrict__ c1, int bs, float sxi, float syi, bool mir, int &jtot,
                                                double (&b)[7], float *circ, float *work, int lane, double *red, int *redi,
                                                const double2 *twd, double *xs)
{
#pragma clang fp contract(off)
    const float cx = ((float)g.cnx + sxi) + g.shift_x[bs], cy = ((float)g.cnx + syi) + g.shift_y[bs];
    exact_lds_sync<GM>();
    constexpr int T = RA_EXACT_THREADS;
    for (int i0 = lane; i0 < g.lcirc; i0 += 4 * T) {        // four samples per trip: their 16 image taps are in flight together
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + T * u, g.lcirc - 1);
            v[u] = g.interp ? quadri_1b(img, g.nx, g.nx, g.samp_dx[i] + cx, g.samp_dy[i] + cy) : bilinear_1b(img, g.nx, g.samp_dx[i] + cx, g.samp_dy[i] + cy);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + T * u < g.lcirc) circ[i0 + T * u] = v[u];
    }
    if (g.norm_ring) {
        // Util::Normalize_ring sums av += v w and sq += v v w in FLOAT, sample after sample in ring order: a rounding random walk of
        // ~1e-6 of sq over the 5816 samples of the headline geometry, different for every search offset -- and sigma scales the
        // peak.  Two OFFSETS whose CCF maxima lie within 1e-6 of each other (half-pixel steps: a ridge in (shift, angle)) are thus
        // ordered by that very walk, so the re-evaluation repeats it: wave 0 forms the products 64 at a time, and every lane adds
        // them in sample order (v_readlane: one chain per wave, uniform over its lanes).  Until round 6 the sums were taken in double --
        // the same sigma to 1e-6, which decides nothing about one candidate's sub-bin angle but does decide between two offsets.
        exact_lds_sync<GM>();
        if (lane < 128) {          // wave 0 walks av, wave 1 walks sq: two independent chains, side by side on two SIMDs
            const bool sq_wave = lane >= 64;
            const int l = lane & 63;
            float acc = 0.f;
            for (int i0 = 0; i0 < g.lcirc; i0 += 64) {
                const int i = i0 + l;
                const float v = i < g.lcirc ? circ[i] : 0.f, w = i < g.lcirc ? g.samp_w[i] : 0.f;
                const float p = sq_wave ? v * v * w : v * w;          // (padding lanes add +0: no change)
#pragma unroll
                for (int k = 0; k < 64; k++)
                    acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), k));
            }
            if (l == 0) red[sq_wave ? 1 : 0] = (double)acc;
        }
        exact_lds_sync<GM>();
        const float nn = g.nn_weight, avf = (float)red[0], sqf = (float)red[1];
        const float avg = avf / nn;
        const float sgm = sqrtf((sqf - avf * avf / nn) / nn);
        exact_lds_sync<GM>();
        for (int i = lane; i < g.lcirc; i += T) { float v = circ[i]; v -= avg; v /= sgm; circ[i] = v; }
    }
    exact_lds_sync<GM>();
    exact_frngs<GM>(circ, work, g, numr, tw, twoff, lane, nullptr);
    exact_lds_sync<GM>();
    // Crosrng_ms: q (straight) or t (mirrored) spectrum, f32 products, f64 sums over the rings in ring order
    const int N = g.maxrin;
    double *spec = reinterpret_cast<double *>(work);          // [N] doubles (the FFT work space is free again: N <= lcirc / 2)
    for (int j = 2 * lane; j < N; j += 2 * T) {
        double s0 = 0.0, s1 = 0.0;
        if (j == 0) {
            for (int i = 0; i < g.nring; i++) {
                const int n = numr[3 * i + 2], o = numr[3 * i + 1] - 1;
                s0 += (double)(c1[o] * circ[o]);
                if (n == N) s1 += (double)(c1[o + 1] * circ[o + 1]);
            }
        } else {
            // rings are sorted by length: those of length j contribute their (real) Nyquist coefficient at index j, the longer
            // ones a regular complex product -- in ring order, four rings' operands in flight
            int i = 0;
            while (i < g.nring && numr[3 * i + 2] < j) i++;
            for (; i < g.nring && numr[3 * i + 2] == j; i++) {
                const int o = numr[3 * i + 1] - 1;
                s0 += (double)(c1[o + 1] * circ[o + 1]);
            }
            for (; i < g.nring; i += 4) {
                float a1[4], a2[4], d1[4], d2[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int o = numr[3 * min(i + u, g.nring - 1) + 1] - 1;
                    a1[u] = c1[o + j]; a2[u] = c1[o + j + 1]; d1[u] = circ[o + j]; d2[u] = circ[o + j + 1];
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i + u < g.nring) {
                        const float p1 = a1[u] * d1[u], p2 = a2[u] * d2[u], p3 = a1[u] * d2[u], p4 = a2[u] * d1[u];
                        if (mir) { s0 += (double)(p1 - p2); s1 += (double)(-p3 - p4); }
                        else { s0 += (double)(p1 + p2); s1 += (double)(-p3 + p4); }
                    }
            }
        }
        spec[j] = s0; spec[j + 1] = s1;
    }
    exact_lds_sync<GM>();
    // the whole inverse real transform in f64, x[m] = (X0 + (-1)^m X_{N/2} + 2 sum_k Re(X_k e^{+2 pi i k m / N})) / N by direct
    // summation (twd[j] = (cos, sin)(2 pi j / N)), thread m -> samples m, m + T, ..: fftr_d computes the same N values (they agree
    // to f64 rounding), and Crosrng_ms scans ALL of them.  Until round 5 only the 7 samples around the f32 winner were evaluated
    // (re-centred on their own maximum): two SEPARATED maxima of one flat CCF within 3e-6 of each other -- the first reference-free
    // iteration against the blob average of an unaligned stack -- were then decided by the f32 transform (13 of 65 536 particles
    // disagreed with the CPU path, profiles/r05_parity_audit_large.json).
    for (int m = lane; m < N; m += T) {
        double acc = 0.0;
        for (int k = 1; k < N / 2; k++) {
            const double2 w = twd[(k * m) & (N - 1)];
            acc += spec[2 * k] * w.x - spec[2 * k + 1] * w.y;
        }
        xs[m] = (spec[0] + ((m & 1) ? -spec[1] : spec[1]) + 2.0 * acc) / (double)N;
    }
    exact_lds_sync<GM>();
    // the CPU scan runs over j = 1 .. N with ">=": of equal values the largest bin wins
    double bm = -1.0e300;
    int jm = 0;
    for (int m = lane; m < N; m += T) {
        const double v = xs[m];
        if (v >= bm) { bm = v; jm = m + 1; }
    }
    block_argmax_f64(bm, jm, red, redi);
    jtot = jm;
#pragma unroll
    for (int t = 0; t < 7; t++) b[t] = xs[(jm - 1 + t - 3 + N) & (N - 1)];
}

// Copies of a reference inside the stack.  hash[r]: 64 bits over the words of reference r's exact spectrum; ref_groups_kernel then
// compares the spectra of equal hashes word for word and writes DevGeom::ref_dup (first copy, next copy).  One launch each per
// ra_set_references, nothing goes through the host.
__global__ __launch_bounds__(256) void ref_hash_kernel(const float *__restrict__ refx, int lcirc, unsigned long long *__restrict__ hash)
{
    __shared__ unsigned long long part[4];
    const unsigned *w = reinterpret_cast<const unsigned *>(refx + (size_t)blockIdx.x * lcirc);
    unsigned long long h = 0;
    for (int i = threadIdx.x; i < lcirc; i += 256) {
        unsigned long long x = ((unsigned long long)w[i] << 32 | (unsigned)(i + 1)) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        h += x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) hash[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void ref_groups_kernel(const float *__restrict__ refx, int lcirc, int nref,
                                                         const unsigned long long *__restrict__ hash, int *__restrict__ dup)
{
    for (int r = threadIdx.x; r < nref; r += 256) {
        const unsigned *a = reinterpret_cast<const unsigned *>(refx + (size_t)r * lcirc);
        int first = r, next = -1;
        for (int r2 = 0; r2 < nref; r2++) {
            if (r2 == r || hash[r2] != hash[r] || (r2 > r && next >= 0)) continue;
            if (r2 < r && first < r) continue;          // the first copy is already known
            const unsigned *b = reinterpret_cast<const unsigned *>(refx + (size_t)r2 * lcirc);
            bool eq = true;
            for (int i = 0; i < lcirc && eq; i++) eq = a[i] == b[i];
            if (!eq) continue;
            if (r2 < r) first = r2; else next = r2;
        }
        dup[2 * r] = first; dup[2 * r + 1] = next;
    }
}

// refx: exact reference spectra [nref][lcirc] (refspec_exact_kernel); res, particles, cls, state: of the chunk (indexed by rec.p)
// GM: the flagged particles are dealt to gridDim.x waves (each with 2 lcirc floats of global scratch in gscr)
template <bool GM>
__global__ __launch_bounds__(RA_EXACT_THREADS) void refine_winner_kernel(DevGeom g, const int *__restrict__ numr, const float *__restrict__ tw,
                                                           const int *__restrict__ twoff, const float *__restrict__ particles,
                                                           const float *__restrict__ refx, const RefineRec *__restrict__ list,
                                                           const int *__restrict__ count, ra_result *__restrict__ res,
                                                           const int *__restrict__ cls, float *__restrict__ state,
                                                           float *__restrict__ gscr)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) float lds[];
    __shared__ double red[4];
    __shared__ int redi[4];
    __shared__ int cand_inf[RA_TIE_ALTS + 1][4];             // the candidates of a float tie: (reference, mirror, bin, offset),
                                                             // the 7 CCF samples around the maximum (double)
    __shared__ double cand_b[RA_TIE_ALTS + 1][7];
    __shared__ int cand_win;
    __shared__ long long ex_key[RA_TIE_EXP];                 // the replay list: the candidates and the copies of their references
    __shared__ unsigned char ex_ci[RA_TIE_EXP], ex_mir[RA_TIE_EXP];
    __shared__ int ex_ref, ex_bs;
    const int lane = threadIdx.x;
    float *circ, *work;
    if constexpr (GM) { circ = gscr + (size_t)blockIdx.x * 2 * g.lcirc; work = circ + g.lcirc; }
    else { circ = lds; work = lds + g.lcirc; }
    // behind the ring buffers (GM: at the start of the dynamic LDS): [maxrin] (cos, sin)(2 pi j / maxrin) and the maxrin samples of
    // a candidate's CCF, both in double (RA_EXACT_TABLE_BYTES)
    // (the spectrum of a candidate -- maxrin doubles at `work` -- ends at lcirc + 2 maxrin floats, beyond 2 lcirc for very short ring
    // sets: the tables start behind whichever is larger, as setup_refine sizes the allocation)
    double2 *twd = reinterpret_cast<double2 *>(GM ? lds : lds + (((2 * g.lcirc > g.lcirc + 2 * g.maxrin ? 2 * g.lcirc : g.lcirc + 2 * g.maxrin) + 3) & ~3));
    double *xs = reinterpret_cast<double *>(twd + g.maxrin);
    for (int j = lane; j < g.maxrin; j += RA_EXACT_THREADS) {
        double sn, cs;
        sincospi(2.0 * (double)j / (double)g.maxrin, &sn, &cs);
        twd[j] = make_double2(cs, sn);
    }
    __syncthreads();
    for (int item = blockIdx.x; item < *count; item += gridDim.x) {
    const RefineRec rec = list[item];
    const float *img = particles + (size_t)rec.p * g.nx * g.nx;
    const float *cbase = refx + (cls ? (size_t)cls[rec.p] * g.lcirc : (size_t)0);      // class-resident mode: the particle's own class
    int ref = rec.ref, mirror = rec.mirror, jtot = rec.jtot, bs = rec.bs;
    double b[7];
    exact_candidate<GM>(g, numr, tw, twoff, img, cbase + (cls ? (size_t)0 : (size_t)ref * g.lcirc), bs, rec.sxi, rec.syi, mirror != 0, jtot, b,
                        circ, work, lane, red, redi, twd, xs);
    const bool copies = g.ref_dup && !cls && (g.ref_dup[2 * ref] != ref || g.ref_dup[2 * ref + 1] >= 0);
    if (rec.nalt > 0 || copies) {
        // the other candidates within the tie tolerance of the winner: every peak in the CPU path's arithmetic, then the CPU path's
        // own scan over them -- offsets, then references, ascending; per (offset, reference) "if (qn >= peak || qm >= peak)
        // { if (qn >= qm) straight else mirrored; peak = that }" -- replayed literally, because it is NOT a total order in the
        // multi-reference entry point: Util::multiref_polar_ali_2d keeps `peak` as a FLOAT (peak = static_cast<float>(qn)) and
        // compares the next double against the rounded value, so of two offsets whose maxima round to the same float the earlier
        // one stays when its value was rounded up (108 / 46 / ts 0.5: both 6459.770508).  ormq keeps a double.
        // Candidates that are not in the list lie below the tolerance and cannot win.  COPIES of a candidate's reference
        // (DevGeom::ref_dup) take part with the candidate's own values -- their spectra are equal to the bit --: of k copies the
        // float rule keeps the first when its peak was rounded up and takes the last otherwise.
        const int ncand = 1 + rec.nalt;
        if (lane == 0) {
            cand_inf[0][0] = ref; cand_inf[0][1] = mirror; cand_inf[0][2] = jtot; cand_inf[0][3] = bs;
            for (int t = 0; t < 7; t++) cand_b[0][t] = b[t];
        }
        for (int ai = 0; ai < rec.nalt; ai++) {
            const RefineAlt alt = rec.alt[ai];
            const int ref2 = alt.refmir & 0xffff, mirror2 = alt.refmir >> 16;
            int jt2 = 1;
            double b2[7];
            exact_candidate<GM>(g, numr, tw, twoff, img, cbase + (cls ? (size_t)0 : (size_t)ref2 * g.lcirc), alt.bs, rec.sxi, rec.syi,
                                mirror2 != 0, jt2, b2, circ, work, lane, red, redi, twd, xs);
            if (lane == 0) {
                cand_inf[ai + 1][0] = ref2; cand_inf[ai + 1][1] = mirror2; cand_inf[ai + 1][2] = jt2; cand_inf[ai + 1][3] = alt.bs;
                for (int t = 0; t < 7; t++) cand_b[ai + 1][t] = b2[t];
            }
        }
        __syncthreads();
        if (lane == 0) {
            int nex = 0;
            for (int ci = 0; ci < ncand; ci++) {
                const int r0 = cand_inf[ci][0];
                const long long hi = (long long)cand_inf[ci][3] << 32;
                if (g.ref_dup && !cls) {
                    for (int r2 = g.ref_dup[2 * r0]; r2 >= 0 && nex < RA_TIE_EXP; r2 = g.ref_dup[2 * r2 + 1]) {
                        ex_key[nex] = hi | (unsigned)r2; ex_ci[nex] = (unsigned char)ci; ex_mir[nex] = (unsigned char)cand_inf[ci][1]; nex++;
                    }
                } else if (nex < RA_TIE_EXP) {
                    ex_key[nex] = hi | (unsigned)r0; ex_ci[nex] = (unsigned char)ci; ex_mir[nex] = (unsigned char)cand_inf[ci][1]; nex++;
                }
            }
            const bool float_peak = g.mode == RA_MODE_MREF;
            double peak = -1.0e23;
            int cur = -1;
            unsigned long long used = 0;
            for (int step = 0; step < nex; step++) {
                int c = -1;          // the next (offset, reference) in scan order
                for (int k = 0; k < nex; k++)
                    if (!((used >> k) & 1) && (c < 0 || ex_key[k] < ex_key[c])) c = k;
                if (c < 0) break;
                int cs = -1, cm = -1;          // its straight and mirrored entries (the first of each, duplicates are dropped)
                for (int k = 0; k < nex; k++)
                    if (!((used >> k) & 1) && ex_key[k] == ex_key[c]) {
                        used |= 1ull << k;
                        if (ex_mir[k]) { if (cm < 0) cm = k; } else { if (cs < 0) cs = k; }
                    }
                const double qn = cs >= 0 ? cand_b[ex_ci[cs]][3] : -1.0e300, qm = cm >= 0 ? cand_b[ex_ci[cm]][3] : -1.0e300;
                if (qn >= peak || qm >= peak) {
                    cur = qn >= qm ? cs : cm;
                    const double v = qn >= qm ? qn : qm;
                    peak = float_peak ? (double)(float)v : v;
                }
            }
            if (cur < 0) cur = 0;
            cand_win = ex_ci[cur];
            ex_ref = (int)(ex_key[cur] & 0xffffffffll); ex_bs = (int)(ex_key[cur] >> 32);
        }
        __syncthreads();
        const int wsel = cand_win;
        ref = ex_ref; mirror = cand_inf[wsel][1]; jtot = cand_inf[wsel][2]; bs = ex_bs;
#pragma unroll
        for (int t = 0; t < 7; t++) b[t] = cand_b[wsel][t];
        __syncthreads();
    }
    if (lane == 0) {
        const double c2 = 49. * b[0] + 6. * b[1] - 21. * b[2] - 32. * b[3] - 27. * b[4] - 6. * b[5] + 31. * b[6];
        const double c3 = 5. * b[0] - 3. * b[2] - 4. * b[3] - 3. * b[4] + 5. * b[6];
        const float pos = (c3 != 0.0) ? (float)(c2 / (2.0 * c3) - 4) : 0.f;
        float alpha, sx, sy;
        finish_params(g, rec.sxi, rec.syi, jtot, pos, bs, &alpha, &sx, &sy);
        ra_result r = res[rec.p];
        r.alpha = alpha; r.sx = sx; r.sy = sy;
        if (jtot != rec.jtot || bs != rec.bs || ref != rec.ref || mirror != rec.mirror) {
            // the f64 CCF orders a float tie differently: the integer assignment follows it
            r.angle_bin = jtot; r.shift_idx = bs; r.ref_id = cls ? r.ref_id : ref; r.mirror = mirror; r.peak = (float)b[3];
            state[2 * rec.p] = rec.sxi + g.shift_x[bs];
            state[2 * rec.p + 1] = rec.syi + g.shift_y[bs];
        }
        res[rec.p] = r;
    }
    }
}

}  // namespace ralign
