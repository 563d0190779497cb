// Size-generic kernels of the alignment path for geometries the LDS-resident kernels of
// ralign_kernels.h cannot hold: images larger than one CU's LDS (e.g. 256 x 256), more than 48
// rings, rings longer than 256 samples (maxrin 512 ... 4096).  Same arithmetic and the same
// operand layout (bin-major A panels, B tiles, CandT records) as the specialised kernels, so the
// two paths are interchangeable stage by stage and are tested against each other.
//
//   polar_generic_kernel<false>  Polar2Dm + Normalize_ring + Frngs: one wave per search offset
//                                (4 offsets = one A block per workgroup), image sampled from
//                                global memory / L2, one ring at a time through a wave-private
//                                LDS buffer, spectra scattered straight into the A panels.
//   polar_generic_kernel<true>   the reference side (Polar2Dm + Frngs, natural layout).
//   ccf_generic_kernel           Crosrng_ms: the same 16x16x4 f32 MFMA contraction per Fourier
//                                bin; the CCF spectra of the 8 x 8 tile go through an L2-resident
//                                scratch (they do not fit LDS at maxrin >= 512), are inverse
//                                transformed in LDS a few pairs at a time and reduced by a
//                                wavefront argmax.
//   transform_generic_kernel     rot_shift2D reading the image from global memory.
#pragma once

#include <type_traits>

#include "ralign_kernels.h"

namespace ralign {

#define RA_GEN_THREADS 256

__device__ __forceinline__ float rf4(const float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
#define RA_GCCF_THREADS 512

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Wave-level Stockham FFT of h complex points (h a power of two >= 2) between two LDS buffers:
// one radix-2 stage when log2 h is odd, radix-4 stages otherwise; natural-order output.
// tw[m] = e^{-2 pi i m / T}, T a power of two >= h.  SIGN = -1 forward, +1 inverse (unscaled).
// Returns the buffer that holds the result.
template <int SIGN>
__device__ __forceinline__ float2 *wave_fft(float2 *x, float2 *y, int h, const float2 *__restrict__ tw, int T, int lane)
{
    int Ns = 1;
    const int lg = 31 - __clz(h);
    if (lg & 1) {
        const int q = h >> 1;
        for (int j = lane; j < q; j += 64) {
            const float2 a = x[j], b = x[j + q];
            y[2 * j] = cadd(a, b);
            y[2 * j + 1] = csub(a, b);
        }
        wave_lds_sync();
        float2 *t = x; x = y; y = t;
        Ns = 2;
    }
    for (; Ns < h; Ns <<= 2) {
        const int q = h >> 2, tstep = T / (4 * Ns);
        for (int j = lane; j < q; j += 64) {
            const int k = j & (Ns - 1);
            float2 v0 = x[j], v1 = x[j + q], v2 = x[j + 2 * q], v3 = x[j + 3 * q];
            if (k) {
                float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                if (SIGN > 0) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
            }
            dft4<SIGN>(v0, v1, v2, v3);
            const int j0 = ((j - k) << 2) + k;
            y[j0] = v0; y[j0 + Ns] = v1; y[j0 + 2 * Ns] = v2; y[j0 + 3 * Ns] = v3;
        }
        wave_lds_sync();
        float2 *t = x; x = y; y = t;
    }
    return x;
}

// LDS slot of element i of a wave_fft_dif buffer: an XOR swizzle inside aligned blocks of 32 that makes the stride-4,
// stride-16 and stride-64 butterflies of the last radix-4 stages bank-conflict free for ds_read_b64 (a 32-lane group maps
// one-to-one onto the 32 complex slots of a bank row; without it those stages run 4-way conflicted)
__device__ __forceinline__ int dif_slot(int i) { return i ^ (((i >> 5) & 3) * 5) ^ (((i >> 6) & 1) << 4); }

// Wave-level in-place FFT of h complex points (h a power of two >= 4), decimation in frequency: one radix-2 stage when
// log2 h is odd, then radix-4 stages; a single buffer (half the LDS of the Stockham form above).  The output is digit
// reversed: position p = d1 (h/r1) + d2 (h/(r1 r2)) + ... holds X[d1 + r1 d2 + r1 r2 d3 + ...] for the stage radices
// r1, r2, ... = [2,] 4, 4, ... (dif_index_of_pos / dif_pos_of_index), and element i lives in LDS slot dif_slot(i).  tw[m] = e^{-2 pi i m / T}, T >= h.
template <int SIGN>
__device__ __forceinline__ void wave_fft_dif(float2 *x, int h, const float2 *__restrict__ tw, int T, int lane)
{
    const int lg = 31 - __clz(h);
    int L = h;
    if (lg & 1) {
        const int q = h >> 1, ts = T / h;
        for (int j = lane; j < q; j += 64) {
            const int s0 = dif_slot(j), s1 = dif_slot(j + q);
            const float2 a = x[s0], b = x[s1];
            float2 w = tw[j * ts];
            if (SIGN > 0) w.y = -w.y;
            x[s0] = cadd(a, b);
            x[s1] = cmul(csub(a, b), w);
        }
        wave_lds_sync();
        L = q;
    }
    for (; L >= 4; L >>= 2) {
        const int q = L >> 2, ts = T / L;
        // four butterflies per lane at a time: their 16 reads (and 12 twiddles) are in flight together
        for (int idx0 = lane; idx0 < (h >> 2); idx0 += 256) {
            int sl[4][4], jj[4];
            float2 v[4][4], w[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int idx = min(idx0 + 64 * u, (h >> 2) - 1);       // a transform shorter than 1024: duplicates, stored once
                const int j = idx & (q - 1), base = ((idx - j) << 2) + j;
                jj[u] = j;
#pragma unroll
                for (int m = 0; m < 4; m++) { sl[u][m] = dif_slot(base + m * q); v[u][m] = x[sl[u][m]]; }
#pragma unroll
                for (int m = 1; m < 4; m++) w[u][m - 1] = tw[m * j * ts];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                dft4<SIGN>(v[u][0], v[u][1], v[u][2], v[u][3]);
#pragma unroll
                for (int m = 1; m < 4; m++) {
                    float2 ww = w[u][m - 1];
                    if (SIGN > 0) ww.y = -ww.y;
                    v[u][m] = cmul(v[u][m], ww);          // j = 0: ww = 1
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (idx0 + 64 * u < (h >> 2)) {
#pragma unroll
                    for (int m = 0; m < 4; m++) x[sl[u][m]] = v[u][m];
                }
            (void)jj;
        }
        wave_lds_sync();
    }
}
// Inverse FFT of N = R12 * R12 * R3 complex points (1024 = 16 x 16 x 4) by one wave, three register stages with one LDS
// transpose and one quad exchange in between (wave_fft_dif runs five radix-4 stages through the LDS, each with a wave hand-off):
//   n = R12 R3 n1 + R3 n2 + n3,  k = k1 + R12 k2 + R12^2 k3
//   A [k1; n2 n3] = sum_n1 x W_R12^{n1 k1},           lane = (n2, n3):   x[64 n1 + lane], then * W_{R12^2}^{n2 k1}
//   B [k1 k2; n3] = sum_n2 A W_R12^{n2 k2},           lane = (k1, n3):   then * W_N^{n3 (k1 + R12 k2)}
//   X [k]         = sum_n3 B W_R3^{n3 k3},            across the R3 = 4 lanes (k1, 0..3) of a quad (DPP), stored once
// x: the pair's LDS image, rows of 64 complex padded by R3 (row stride 64 + R3: the stride-R3 reads of stage B spread over
// the banks); the transform ends in the same image, element k in slot ifft3_slot(k) (the 7-point neighbourhood of the peak is
// read from it).  tw[m] = e^{-2 pi i m / N} (conjugated here: inverse, unscaled).  Running maxima of Re (q) and Im (t) with the
// largest index winning ties, as ccf_generic_kernel's scan.
// twa [R12][R12] = conj W_N^{R3 n2 k1} at [k1 * R12 + n2], twb [R12][64] = conj W_N^{n3 (k1 + R12 k2)} at [k2 * 64 + lane]
// (ifft3_twiddles): consecutive lanes read consecutive entries -- out of the plain table W_N^m the stage-A reads of k1 = 8 hit one
// bank 16 times.
template <int R12, int R3>
__device__ __forceinline__ void ifft3_twiddles(const float2 *__restrict__ tw, float2 *twa, float2 *twb, int tid, int nthreads)
{
    constexpr int N = R12 * R12 * R3, LG3 = R3 == 4 ? 2 : 3;
    for (int i = tid; i < R12 * R12; i += nthreads) {
        const float2 w = tw[(R3 * (i % R12) * (i / R12)) & (N - 1)];
        twa[i] = make_float2(w.x, -w.y);
    }
    for (int i = tid; i < R12 * 64; i += nthreads) {
        const int k2 = i >> 6, ln = i & 63, k1 = ln >> LG3, n3 = ln & (R3 - 1);
        const float2 w = tw[(n3 * (k1 + R12 * k2)) & (N - 1)];
        twb[i] = make_float2(w.x, -w.y);
    }
}
template <int R12, int R3>
__device__ __forceinline__ void wave_ifft3_argmax(float2 *x, const float2 *__restrict__ twa, const float2 *__restrict__ twb, int lane,
                                                  float &bq, int &iq, float &bt, int &it)
{
    constexpr int N = R12 * R12 * R3, RS = 64 + R3, LG3 = R3 == 4 ? 2 : 3;
    static_assert(R12 * R3 == 64, "one wave: R12 * R3 lanes per stage");
    float2 v[R12];
    // ---- stage A: DFT-R12 over n1, twiddle W_{R12^2}^{n2 k1} = W_N^{R3 n2 k1}
#pragma unroll
    for (int n1 = 0; n1 < R12; n1++) v[n1] = x[n1 * RS + lane];
    Dft<1, R12>::run(v);
    {
        const int n2 = lane >> LG3;
#pragma unroll
        for (int k1 = 1; k1 < R12; k1++) v[k1] = cmul(v[k1], twa[k1 * R12 + n2]);
    }
    wave_lds_sync();
#pragma unroll
    for (int k1 = 0; k1 < R12; k1++) x[k1 * RS + lane] = v[k1];
    wave_lds_sync();
    // ---- stage B: lane = (k1, n3): DFT-R12 over n2, twiddle W_N^{n3 (k1 + R12 k2)}
    {
        const int k1 = lane >> LG3, n3 = lane & (R3 - 1);
#pragma unroll
        for (int n2 = 0; n2 < R12; n2++) v[n2] = x[k1 * RS + n2 * R3 + n3];
        Dft<1, R12>::run(v);
#pragma unroll
        for (int k2 = 0; k2 < R12; k2++) v[k2] = cmul(v[k2], twb[k2 * 64 + lane]);       // n3 = 0: w = 1
        // ---- stage C: DFT-4 over n3 = the four lanes of a quad, through DPP instead of a third pass over the image: the
        // butterflies of vdft4 (a0 +- a2, a1 +- a3, the odd difference turned by +i, then sums and differences of neighbours),
        // the same additions in the same pairs, so the values are those of Dft<1, 4> bit for bit; lane n3 ends with
        // X[k1 + R12 k2 + R12^2 s(n3)], s = 0 2 1 3, and stores it in the slot of B[k1 k2; s(n3)] (ifft3_slot)
        static_assert(R3 == 4, "stage C is written for quads");
        const float sg2 = (lane & 2) ? -1.0f : 1.0f, sg1 = (lane & 1) ? -1.0f : 1.0f;
        const bool l3 = n3 == 3;
        const int s3 = ((n3 & 1) << 1) | (n3 >> 1);
        bq = -1.0e20f; bt = -1.0e20f; iq = -1; it = -1;
        wave_lds_sync();
#pragma unroll
        for (int k2 = 0; k2 < R12; k2++) {
            const float2 a = v[k2];
            const float tx = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), 0x4E, 0xF, 0xF, true));      // lane ^ 2
            const float ty = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.y), 0x4E, 0xF, 0xF, true));
            const float bx = __builtin_fmaf(sg2, a.x, tx), by = __builtin_fmaf(sg2, a.y, ty);         // a + t | t - a (exact products)
            const float cx = l3 ? -by : bx, cy = l3 ? bx : by;                                         // lane 3: (a1 - a3) * i
            const float ux = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cx), 0xB1, 0xF, 0xF, true));       // lane ^ 1
            const float uy = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cy), 0xB1, 0xF, 0xF, true));
            const float2 z = make_float2(__builtin_fmaf(sg1, cx, ux), __builtin_fmaf(sg1, cy, uy));
            const int k = k1 + R12 * k2 + R12 * R12 * s3;
            if (z.x > bq || (z.x == bq && k > iq)) { bq = z.x; iq = k; }
            if (z.y > bt || (z.y == bt && k > it)) { bt = z.y; it = k; }
            x[k1 * RS + k2 * R3 + s3] = z;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float oq = __shfl_xor(bq, o); const int oiq = __shfl_xor(iq, o);
        const float ot = __shfl_xor(bt, o); const int oit = __shfl_xor(it, o);
        if (oq > bq || (oq == bq && oiq > iq)) { bq = oq; iq = oiq; }
        if (ot > bt || (ot == bt && oit > it)) { bt = ot; it = oit; }
    }
    wave_lds_sync();
}
// LDS slot of output element k of wave_ifft3_argmax<16, 4> (N = 1024); complex slots per pair of its image
__device__ __forceinline__ int ifft3_slot(int k) { return (k & 15) * 68 + ((k >> 4) & 15) * 4 + (k >> 8); }
#define RA_IFFT3_PSTRIDE (16 * 68 + 4)       // + 4: the 8 images of a batch start 8 banks apart (the scratch -> LDS writes of gccf_ifft_kernel)

// frequency index held at position p of a wave_fft_dif output of h points, and its inverse
__device__ __forceinline__ int dif_index_of_pos(int p, int h)
{
    int bits = 31 - __clz(h), k = 0, sh = 0;
    if (bits & 1) { bits--; k = (p >> bits) & 1; sh = 1; }
    while (bits > 0) { bits -= 2; k |= ((p >> bits) & 3) << sh; sh += 2; }
    return k;
}
__device__ __forceinline__ int dif_pos_of_index(int k, int h)
{
    int bits = 31 - __clz(h), p = 0;
    if (bits & 1) { bits--; p = (k & 1) << bits; k >>= 1; }
    while (bits > 0) { bits -= 2; p |= (k & 3) << bits; k >>= 2; }
    return p;
}
// real-FFT split step: bin k (0..h) of the n = 2h point real transform from the h-point complex
// transform Z of the packed sequence z_m = x_2m + i x_2m+1
__device__ __forceinline__ float2 split_bin(const float2 *Z, int k, int h, float2 w)
{
    const float2 zk = Z[k & (h - 1)], zm = Z[(h - k) & (h - 1)];
    const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
    const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
    const float tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
    if (k == 0 || k == h) return make_float2(er + tr, 0.f);
    return make_float2(er + tr, ei + ti);
}

// sample j of a ring of 4 lt = 4 << lgl positions and radius fr, relative to the centre: alrl_ms's table position from the first-quadrant
// (sinf, cosf) table of the ring length in LDS (qt), mirrored into the quadrant of j -- the floats of samp_dx / samp_dy
// (ralign_geom.h: sinf(fi) * inr, quadrants by sign and swap) without a trip to memory in front of the taps
__device__ __forceinline__ float2 ring_pos(const float2 *qt, int lt, int lgl, float fr, int j)
{
    const int q = j >> lgl;
    const float2 sc = qt[j & (lt - 1)];
    const float x = __fmul_rn(sc.x, fr), y = __fmul_rn(sc.y, fr);
    const float a = (q & 1) ? y : x, b = (q & 1) ? x : y;
    return make_float2((q & 2) ? -a : a, ((q + 1) & 2) ? -b : b);
}

// Polar2Dm (bilinear) [+ Normalize_ring] + Frngs for images of any size.
//   REFS = false: particles; workgroup = (particle, group of 4 search offsets), wave = offset slot;
//                 output = the A block of that group (layout of ralign_geom.h: ent_apos).
//   REFS = true : references; workgroup = reference, waves share the rings; output natural
//                 padded ring layout [lring] (what ref_polar_fft_kernel writes).
// (reference call sites: test_mref_gpu_align.py:1015-1016, 1043-1044)
// Normalize_ring is linear, and at these sizes a second sampling pass for its statistics costs as much as the
// transform itself: the particle spectra are written RAW, the statistics {avg, 1/sigma} of every (particle, offset) go to
// `stats`, and the contraction applies them (DC bin: a -= avg * sum_r n_r C_r(0); peak records: * 1/sigma).
template <bool REFS, bool QUADRI = false>
__global__ __launch_bounds__(RA_GEN_THREADS, 3) void polar_generic_kernel(DevGeom g, const float *__restrict__ images,
                                                                      const float *__restrict__ state, int n,
                                                                      float *__restrict__ out, float2 *__restrict__ stats)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *bx = reinterpret_cast<float2 *>(lds) + (size_t)wave * g.maxrin;     // two buffers of maxrin/2 complex
    float2 *by = bx + g.maxrin / 2;
    float2 *tw_s = reinterpret_cast<float2 *>(lds) + (size_t)(RA_GEN_THREADS / 64) * g.maxrin;      // twiddle table in LDS
    float2 *qt_s = tw_s + g.maxrin;                                                                  // (sinf, cosf) tables of the ring lengths
    for (int i = tid; i < g.maxrin; i += RA_GEN_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RA_GEN_THREADS) qt_s[i] = g.qtab[i];
    __syncthreads();
    const int npix = g.nx * g.nx;

    if (REFS) {
        const int r = blockIdx.x;
        if (r >= n) return;
        const float *img = images + (size_t)r * npix;
        const float c = (float)g.cnx;
        for (int i = wave; i < g.nring; i += RA_GEN_THREADS / 64) {
            const int4 ri = g.ringinfo[i];
            const int nlen = ri.z, h = nlen >> 1, lt = nlen >> 2, lgl = 31 - __clz(lt);
            const float fr = (float)ri.y;
            float *xr = reinterpret_cast<float *>(bx);
            for (int j = lane; j < nlen; j += 64) {
                const float2 d = ring_pos(qt_s + ri.w, lt, lgl, fr, j);
                xr[j] = polar_sample_1b<QUADRI>(img, g.nx, d.x + c, d.y + c);
            }
            wave_lds_sync();
            const float2 *Z = (h >= 2) ? wave_fft<-1>(bx, by, h, tw_s, g.maxrin, lane) : bx;
            float *dst = out + (size_t)r * g.lring + ri.x;
            for (int k = lane; k <= h; k += 64) {
                const float2 X = split_bin(Z, k, h, tw_s[k * (g.maxrin / nlen)]);
                dst[2 * k] = X.x; dst[2 * k + 1] = X.y;
            }
            wave_lds_sync();
        }
        return;
    }

    // entry of this wave: block blockIdx.x holds entries 4 blockIdx.x .. + 3, entry e = particle e / ent_stride, offset e % ent_stride
    // (ent_stride = nshift: dense, a block may hold offsets of two particles; an offset index beyond nshift -- padded stride, or the
    // tail of the chunk -- repeats the last offset)
    const long long ent = (long long)blockIdx.x * 4 + wave;
    int p = (int)min((long long)n - 1, ent / g.ent_stride);
    const int slot = wave;
    int si = min((int)(ent - (long long)p * g.ent_stride), g.nshift - 1);
    if (g.ent_base) {          // live-offset lists: entry -> (particle, j-th offset of its window); entries past the end repeat the last one
        const int el = (int)min(ent, (long long)*g.ent_total - 1);
        int lo = 0, hi = n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (g.ent_base[mid] <= el) lo = mid; else hi = mid; }
        p = lo;
        si = el - g.ent_base[p];          // position in the window: turned into the list index below
    }
    const float *img = images + (size_t)p * npix;
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    if (g.ent_base) si = live_shift(g, w, si);
    const float cx = ((float)g.cnx + w.sxi) + g.shift_x[si], cy = ((float)g.cnx + w.syi) + g.shift_y[si];

    // one sampling pass: ring FFTs and, in multi-reference mode, the Normalize_ring partial sums (added in ring order)
    float av = 0.f, sq = 0.f;
    float *blk = out + (size_t)blockIdx.x * g.a_blk;
    if (g.quad_aligned) {
        // Panels with ring quads aligned across bins (maxrin <= 1024): the spectra of rings 4c .. 4c+3 stay in registers
        // (bin k = lane + 64 t, t < 9) and leave as whole float4 panel pieces, Re row and Im row of this offset slot --
        // 16-byte stores instead of the 4-byte scattered ones of the loop below.
        constexpr int T = 9;
        int base[T], i0al[T];
        float rns[T];
        int nsk[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int k = min(lane + 64 * t, g.nbins - 1);
            nsk[t] = (g.bin_offp[k + 1] - g.bin_offp[k]) >> 2;
            rns[t] = 1.0f / (float)nsk[t];
            base[t] = g.bin_offp[k] * 8 + 8 * slot;
            i0al[t] = g.bin_first[k] & ~3;
        }
        for (int c4 = 0; c4 < g.nring; c4 += 4) {
            float2 xq[4][T];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = c4 + r;
                if (i < g.nring) {        // uniform
                    const int4 ri = g.ringinfo[i];
                    const int nlen = ri.z, h = nlen >> 1, lt = nlen >> 2, lgl = 31 - __clz(lt);
                    const float wt = g.ringw[i], fr = (float)ri.y;
                    const float2 *qt = qt_s + ri.w;
                    float *xr = reinterpret_cast<float *>(bx);
                    float a = 0.f, q = 0.f;
                    // four samples per trip: their 16 image taps (L2) are in flight together; the Normalize_ring partial
                    // sums keep their order
                    for (int j0 = lane; j0 < nlen; j0 += 256) {
                        float sv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int j = min(j0 + 64 * u, nlen - 1);
                            const float2 d = ring_pos(qt, lt, lgl, fr, j);
                            sv[u] = RA_DBG(g, 512) ? (float)j : polar_sample_1b<QUADRI>(img, g.nx, d.x + cx, d.y + cy);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (j0 + 64 * u < nlen) {
                                xr[j0 + 64 * u] = sv[u];
                                a += sv[u] * wt; q += sv[u] * sv[u] * wt;
                            }
                    }
                    av += a; sq += q;          // per-lane partial sums over the rings, reduced once below
                    wave_lds_sync();
                    const float2 *Z = (h >= 2 && !RA_DBG(g, 256)) ? wave_fft<-1>(bx, by, h, tw_s, g.maxrin, lane) : bx;
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const int k = lane + 64 * t;
                        xq[r][t] = k <= h ? split_bin(Z, k, h, tw_s[k * (g.maxrin / nlen)]) : make_float2(0.f, 0.f);
                    }
                    wave_lds_sync();
                } else {
#pragma unroll
                    for (int t = 0; t < T; t++) xq[r][t] = make_float2(0.f, 0.f);
                }
            }
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int k = lane + 64 * t;
                if (k < g.nbins && c4 >= i0al[t]) {
                    const int j0 = c4 - i0al[t];
                    const int kk = (int)(((float)j0 + 0.5f) * rns[t]), s4 = j0 - kk * nsk[t];      // j0 / ns, exact (see align_ring_quads)
                    float *dst = blk + base[t] + (s4 >> 2) * 128 + kk * 32;
                    if (RA_DBG(g, 1024) && xq[0][t].x != 1.2345f) continue;          // profiling: no panel stores
                    *reinterpret_cast<float4 *>(dst) = make_float4(xq[0][t].x, xq[1][t].x, xq[2][t].x, xq[3][t].x);
                    *reinterpret_cast<float4 *>(dst + 4) = make_float4(xq[0][t].y, xq[1][t].y, xq[2][t].y, xq[3][t].y);
                }
            }
        }
    } else
    for (int i = 0; i < g.nring; i++) {
        const int4 ri = g.ringinfo[i];
        const int nlen = ri.z, h = nlen >> 1, lt = nlen >> 2, lgl = 31 - __clz(lt);
        const float wt = g.ringw[i], fr = (float)ri.y;
        float *xr = reinterpret_cast<float *>(bx);
        float a = 0.f, q = 0.f;
        for (int j = lane; j < nlen; j += 64) {
            const float2 d = ring_pos(qt_s + ri.w, lt, lgl, fr, j);
            const float sv = polar_sample_1b<QUADRI>(img, g.nx, d.x + cx, d.y + cy);
            xr[j] = sv;
            a += sv * wt; q += sv * sv * wt;
        }
        av += a; sq += q;          // per-lane partial sums over the rings, reduced once below
        wave_lds_sync();
        const float2 *Z = (h >= 2) ? wave_fft<-1>(bx, by, h, tw_s, g.maxrin, lane) : bx;
        for (int k = lane; k <= h; k += 64) {
            const float2 X = split_bin(Z, k, h, tw_s[k * (g.maxrin / nlen)]);
            const int2 ap = g.ent_apos[g.bin_off[k] + (i - g.bin_first[k])];
            blk[ap.x + (2 * slot) * ap.y] = X.x;
            blk[ap.x + (2 * slot + 1) * ap.y] = X.y;
        }
        wave_lds_sync();
    }
    av = wave_sum_dpp(av); sq = wave_sum_dpp(sq);      // fixed order: reproducible
    if (lane == 0) {
        float avg = 0.f, rsg = 1.f;
        if (g.norm_ring) {
            const float nn = g.nn_weight;
            avg = av / nn;
            rsg = 1.0f / sqrtf((sq - av * av / nn) / nn);
        }
        stats[(size_t)blockIdx.x * 4 + slot] = make_float2(avg, rsg);
    }
}

// Crosrng_ms at any maxrin.  A workgroup contracts a TM x TR block of 8 (particle-offset) x 8 (reference) tiles at a time:
// every A operand it fetches feeds TR and every B operand TM 16x16x4 MFMA tiles (the kernel is bound by the operand stream from HBM /
// Infinity Cache, so the block size sets its speed); persistent workgroups walk the blocks.
// zscr: [gridDim.x][N][64 TM TR] complex scratch for the CCF spectra of the block (they do not fit LDS at maxrin >= 512);
// P = pairs transformed per LDS batch (power of two, P * (N + 1) complex fit the dynamic LDS), in place.
// stats [n_mtile * 8] {avg, 1/sigma} of every particle-offset (polar_generic_kernel), cdc [nref] = sum_r n_r C_r(0).
// Block shapes (TM tiles of 8 particle-offsets x TR tiles of 8 references): 2 x 2 by default; 1 x 7 when the reference
// tiles come in (nearly) whole sevens -- 13 tiles at 100 references: the A operands are read twice instead of seven times
// (45.2 -> 44.2 ms per chunk).  The scratch holds the largest shape.
#define RA_GCCF_ZPAIRS_MAX (64 * 14)
inline bool gccf_wide_blocks(int nrtile) { return nrtile >= 6 && (nrtile + 6) / 7 * 7 - nrtile <= 1; }
// particle-offset tiles per block of the x 7 shape: 4 at maxrin 1024 (split kernels; the contraction waits on its operand stream -- 8
// requests of 1 KB per 7 matrix instructions at 1 x 7, 11 per 28 at 4 x 7, 214 registers, one workgroup per CU: contraction +
// transforms 36.3 / 31.9 / 30.2 ms per chunk at TM = 1 / 2 / 4), 1 elsewhere; RALIGN_GCCF_TM = 1 | 2 | 4 overrides (4: split kernels only)
// blocks a contraction workgroup walks per slice of the split kernels: 2 (the scratch holds a slice: 7.5 GB at configs[4];
// fewer launches and tails: 27.9 / 27.6 / 27.1 ms per chunk at 1 / 2 / 5 blocks); RALIGN_GCCF_BPW = 1 .. 8 overrides
inline int gccf_blocks_per_wg()
{
    const char *ev = RA_EXP_ENV("RALIGN_GCCF_BPW");
    const int v = ev ? atoi(ev) : 0;
    return v >= 1 && v <= 8 ? v : 2;
}
inline int gccf_tm(int nrtile, int maxrin)
{
    if (!gccf_wide_blocks(nrtile)) return 1;
    const char *ev = getenv("RALIGN_GCCF_TM");
    const int v = ev ? atoi(ev) : 0;
    if (v == 1 || v == 2 || (v == 4 && maxrin == 1024)) return v;
    return maxrin == 1024 ? 4 : 1;
}
// SPLIT (maxrin 1024): the kernel only contracts -- block task0 + blockIdx.x of the slice [task0, task0 + ntask) of the
// (mt2, rt2) blocks, spectra to zscr[blockIdx.x] -- and gccf_ifft_kernel transforms the slice afterwards: two kernels with
// their own register budgets and occupancies instead of two phases that share 128 registers and a barrier.
template <int TM, int TR, bool SPLIT = false>
__global__ __launch_bounds__(RA_GCCF_THREADS, TM >= 4 ? 2 : 4) void ccf_generic_kernel(DevGeom g, const float *__restrict__ A,
                                                                      const float *__restrict__ B, int n_mtile, int nrtile,
                                                                      int nref, CandT *__restrict__ cand,
                                                                      float2 *__restrict__ zscr, int P,
                                                                      const float2 *__restrict__ stats, const float *__restrict__ cdc,
                                                                      int task0 = 0, int ntask = 0)
{
    extern __shared__ __align__(16) float lds[];
    __shared__ CandT pc[64];
    const int N = g.maxrin;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = RA_GCCF_THREADS / 64;
    constexpr int ZPAIRS = 64 * TM * TR;
    float2 *xb = reinterpret_cast<float2 *>(lds);
    const int pstride = N + 1;              // complex slots per pair: one N-point buffer (in-place transform) + 1 (bank skew)
    float2 *tw_s = xb + (size_t)P * pstride;          // twiddles of the inverse transforms, in LDS
    if constexpr (!SPLIT) {
        for (int i = tid; i < N; i += RA_GCCF_THREADS) tw_s[i] = g.tw[i];
        __syncthreads();
    }

    // live-offset lists: the number of entries of the chunk is known on the device only -- tiles and (SPLIT) the tasks of this slice follow it
    if (g.ent_total) n_mtile = (*g.ent_total + 7) >> 3;
    const int n_mt2 = (n_mtile + TM - 1) / TM, n_rt2 = (nrtile + TR - 1) / TR;
    if (g.ent_total && SPLIT) ntask = max(0, min(ntask, n_mt2 * n_rt2 - task0));
    // SPLIT: the workgroup takes blocks task0 + blockIdx.x, + gridDim.x, .. of the slice; block tl leaves its spectra in zscr[tl]
    // (N/2 + 1 bins of 16 bytes per pair)
    for (int tl = SPLIT ? (int)blockIdx.x : 0; tl < (SPLIT ? ntask : 1); tl += SPLIT ? (int)gridDim.x : 1) {
    const int task = task0 + tl;
    float2 *zs = zscr + (size_t)(SPLIT ? tl : (int)blockIdx.x) * ZPAIRS * (SPLIT ? N + 2 : N);
    for (int mt2 = SPLIT ? task / n_rt2 : (int)blockIdx.x; mt2 < n_mt2; mt2 += SPLIT ? n_mt2 : (int)gridDim.x) {
        for (int rt2 = SPLIT ? task % n_rt2 : 0; rt2 < n_rt2; rt2 += SPLIT ? n_rt2 : 1) {
            // ---- phase 1: contraction per Fourier bin (operand layout of ccf_kernel), TM x TR tiles per operand fetch
            if (!RA_DBG(g, 2)) {
                const int r16 = lane & 15, kk = lane >> 4, odd = lane & 1;
                const int pair = (2 * (lane >> 4) + odd) * 8 + ((lane & 15) >> 1);      // offset-in-tile * 8 + reference slot
                const int la = kk * 8 + (r16 & 7), lb = kk * 16 + r16;
                const float *Ab[TM], *Bb[TR];
#pragma unroll
                for (int ai = 0; ai < TM; ai++)          // a tile past the end repeats the last one: its spectra are never transformed
                    Ab[ai] = A + (size_t)(2 * min(TM * mt2 + ai, n_mtile - 1) + (r16 >> 3)) * g.a_blk;
#pragma unroll
                for (int bi = 0; bi < TR; bi++) Bb[bi] = B + (size_t)min(TR * rt2 + bi, nrtile - 1) * g.LBP * 16;
                // the last block of reference tiles may hold TR - 1 of them (13 tiles at 100 references = 7 + 6): its own instantiation of
                // the bin loop, so that the missing tile costs neither operand requests, matrix instructions nor scratch stores (it used to
                // repeat the tile before it: 1 / 14 of the contraction at configs[4])
                auto bin_loop = [&](auto tre_c) {
                constexpr int TRE = decltype(tre_c)::value;
                for (int k = wave; k < g.nbins; k += (SPLIT ? (int)(blockDim.x >> 6) : NW)) {      // (SPLIT: profiling builds may launch fewer waves, scripts/dev/gccf_waves.sh)
                    const int e0 = g.bin_offp[k], ns = (g.bin_offp[k + 1] - e0) >> 2;
                    const float *pa[TM], *pb[TR];
#pragma unroll
                    for (int ai = 0; ai < TM; ai++) pa[ai] = Ab[ai] + (size_t)e0 * 8;
#pragma unroll
                    for (int bi = 0; bi < TRE; bi++) pb[bi] = Bb[bi] + (size_t)e0 * 16;
                    f32x4 acc[TM][TR];
#pragma unroll
                    for (int ai = 0; ai < TM; ai++)
#pragma unroll
                        for (int bi = 0; bi < TRE; bi++) acc[ai][bi] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    int oa = 0, ob = 0;
                    const int nq = ns >> 2;
                    if (nq > 0) {      // chunks of 4 ring steps, the next chunk's operands in flight while this one multiplies
                        float4 va[TM], vb[TR], na[TM], nb[TR];
#pragma unroll
                        for (int ai = 0; ai < TM; ai++) va[ai] = *reinterpret_cast<const float4 *>(pa[ai] + la * 4);
#pragma unroll
                        for (int bi = 0; bi < TRE; bi++) vb[bi] = *reinterpret_cast<const float4 *>(pb[bi] + lb * 4);
                        for (int q = 0; q < nq; q++) {
                            const int step = q + 1 < nq ? 1 : 0;       // the last chunk re-requests itself (discarded)
#pragma unroll
                            for (int ai = 0; ai < TM; ai++) na[ai] = *reinterpret_cast<const float4 *>(pa[ai] + oa + step * 128 + la * 4);
#pragma unroll
                            for (int bi = 0; bi < TRE; bi++) nb[bi] = *reinterpret_cast<const float4 *>(pb[bi] + ob + step * 256 + lb * 4);
#pragma unroll
                            for (int c = 0; c < 4; c++)
#pragma unroll
                                for (int ai = 0; ai < TM; ai++)
#pragma unroll
                                    for (int bi = 0; bi < TRE; bi++)
                                        acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf4(va[ai], c), rf4(vb[bi], c), acc[ai][bi], 0, 0, 0);
#pragma unroll
                            for (int ai = 0; ai < TM; ai++) va[ai] = na[ai];
#pragma unroll
                            for (int bi = 0; bi < TRE; bi++) vb[bi] = nb[bi];
                            oa += 128; ob += 256;
                        }
                    }
                    if (ns & 2) {
                        float2 wa[TM], wb[TR];
#pragma unroll
                        for (int ai = 0; ai < TM; ai++) wa[ai] = *reinterpret_cast<const float2 *>(pa[ai] + oa + la * 2);
#pragma unroll
                        for (int bi = 0; bi < TRE; bi++) wb[bi] = *reinterpret_cast<const float2 *>(pb[bi] + ob + lb * 2);
#pragma unroll
                        for (int ai = 0; ai < TM; ai++)
#pragma unroll
                            for (int bi = 0; bi < TRE; bi++) {
                                acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ai].x, wb[bi].x, acc[ai][bi], 0, 0, 0);
                                acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ai].y, wb[bi].y, acc[ai][bi], 0, 0, 0);
                            }
                        oa += 64; ob += 128;
                    }
                    if (ns & 1) {
                        float sa[TM], sb[TR];
#pragma unroll
                        for (int ai = 0; ai < TM; ai++) sa[ai] = pa[ai][oa + la];
#pragma unroll
                        for (int bi = 0; bi < TRE; bi++) sb[bi] = pb[bi][ob + lb];
#pragma unroll
                        for (int ai = 0; ai < TM; ai++)
#pragma unroll
                            for (int bi = 0; bi < TRE; bi++)
                                acc[ai][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[ai], sb[bi], acc[ai][bi], 0, 0, 0);
                    }
                    // a=c1d1 b=c1d2 c=c2d1 d=c2d2 after the 2x2 exchange between the Re/Im column lanes
#pragma unroll
                    for (int ai = 0; ai < TM; ai++)
#pragma unroll
                        for (int bi = 0; bi < TRE; bi++) {
                            const f32x4 c4 = acc[ai][bi];
                            const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                            const float r0 = swap_lane_pair(s0), r1 = swap_lane_pair(s1);
                            // bin 0 only: Normalize_ring mean of this lane's particle-offset times the DC weight of its reference
                            float dcw = 0.f;
                            if (k == 0)
                                dcw = stats[(size_t)min(TM * mt2 + ai, n_mtile - 1) * 8 + (pair >> 3)].x *
                                      cdc[min(min(TR * rt2 + bi, nrtile - 1) * g.rpt + (pair & 7), nref - 1)];
                            const float ca = (odd ? r0 : c4[0]) - dcw, cb = odd ? r1 : c4[1];
                            const float cc = odd ? c4[2] : r0, cd = odd ? c4[3] : r1;
                            const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                            const int zp = (TR * ai + bi) * 64 + pair;
                            if (RA_DBG(g, 8) && apd != 1.2345f) continue;                 // profiling: no stores
                            const int kw = RA_DBG(g, 4) ? (k & 7) : k;                    // profiling: stores to a cache-resident piece
                            if constexpr (SPLIT) {
                                // [bin 0 .. N/2][pair]{Z_k, Z_{N-k}}: one 16-byte store per lane, 1 KB per instruction
                                reinterpret_cast<float4 *>(zs)[(size_t)kw * ZPAIRS + zp] = make_float4(apd + bpc, cmb + amd, apd - bpc, amd - cmb);
                            } else {
                                zs[(size_t)kw * ZPAIRS + zp] = make_float2(apd + bpc, cmb + amd);
                                zs[(size_t)((N - kw) & (N - 1)) * ZPAIRS + zp] = make_float2(apd - bpc, amd - cmb);
                            }
                        }
                }
                };
                if (TR > 1 && nrtile - TR * rt2 == TR - 1) bin_loop(std::integral_constant<int, (TR > 1 ? TR - 1 : 1)>{});
                else bin_loop(std::integral_constant<int, TR>{});
            }
            if constexpr (SPLIT) continue;
            __syncthreads();
            // ---- phase 2: inverse FFT + argmax tile by tile, P pairs per batch, one wave per pair
            const int lgp = 31 - __clz(P);
            for (int sub = 0; sub < TM * TR; sub++) {
                const int mtile = TM * mt2 + sub / TR, rtile = TR * rt2 + sub % TR;
                if (mtile >= n_mtile || rtile >= nrtile) continue;          // uniform over the workgroup
                const int ref0 = rtile * g.rpt;
                const int nvalid = min(g.rpt, nref - ref0);
                if (RA_DBG(g, ~0) && tid < 64) {   // profiling builds that skip a phase still emit in-range records
                    pc[tid].val = 0.f; pc[tid].jtot = 1; pc[tid].refmir = min(ref0 + (tid & 7), nref - 1);
                    for (int k = 0; k < 7; k++) pc[tid].t7[k] = 0.f;
                }
                if (!RA_DBG(g, 1))
                for (int base = 0; base < 64; base += P) {
                    // scratch [k][pairs] -> LDS [pair][k]: 8 independent loads per thread in flight (P is a power of two)
                    const float2 *src = zs + sub * 64 + base;
                    for (int idx0 = tid; idx0 < P * N; idx0 += 8 * RA_GCCF_THREADS) {
                        float2 t[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int idx = idx0 + u * RA_GCCF_THREADS;
                            if (idx < P * N) t[u] = src[(size_t)(idx >> lgp) * ZPAIRS + (idx & (P - 1))];
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int idx = idx0 + u * RA_GCCF_THREADS;
                            if (idx < P * N) xb[(size_t)(idx & (P - 1)) * pstride + dif_slot(idx >> lgp)] = t[u];
                        }
                    }
                    __syncthreads();
                    for (int pp = wave; pp < P; pp += NW) {
                        const int pair = base + pp, slot = pair & 7;
                        if (slot >= nvalid) continue;           // wave-uniform
                        float2 *x = xb + (size_t)pp * pstride;
                        wave_fft_dif<1>(x, N, tw_s, N, lane);
                        // position p = 64 i + lane holds angle index khi(i) + (N / 64) * klane (N >= 64: the low six bits of
                        // p are three radix-4 digits); the scan order is arbitrary, so ties go by the index: the last wins
                        const int klane = ((lane >> 4) & 3) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
                        float bq = -1.0e20f, bt = -1.0e20f;
                        int iq = -1, it = -1;
                        for (int p0 = 0; p0 < N; p0 += 64) {
                            if (p0 + lane < N) {
                                const float2 v = x[dif_slot(p0 + lane)];
                                const int kx = N >= 64 ? dif_index_of_pos(p0, N) + (N >> 6) * klane : dif_index_of_pos(p0 + lane, N);
                                if (v.x > bq || (v.x == bq && kx > iq)) { bq = v.x; iq = kx; }
                                if (v.y > bt || (v.y == bt && kx > it)) { bt = v.y; it = kx; }
                            }
                        }
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const float oq = __shfl_xor(bq, o); const int oiq = __shfl_xor(iq, o);
                            const float ot = __shfl_xor(bt, o); const int oit = __shfl_xor(it, o);
                            if (oq > bq || (oq == bq && oiq > iq)) { bq = oq; iq = oiq; }
                            if (ot > bt || (ot == bt && oit > it)) { bt = ot; it = oit; }
                        }
                        {
                            const bool mir = !g.nomirror && !(bq >= bt);        // qn >= qm keeps the straight match; nomirror: straight only
                            const int jt = mir ? it : iq;
                            CandT *dst = pc + pair;              // lanes 0..6 store the 7-point neighbourhood, lane 0 the rest
                            if (lane < 7) {
                                const float2 zz = x[dif_slot(dif_pos_of_index((jt + lane - 3 + N) & (N - 1), N))];
                                dst->t7[lane] = mir ? zz.y : zz.x;
                            }
                            if (lane == 0) {
                                dst->val = mir ? bt : bq; dst->jtot = jt + 1; dst->refmir = ((mir ? 1 : 0) << 16) | (ref0 + slot);
                            }
                        }
                    }
                    __syncthreads();
                }
                // ---- best reference of the tile per particle-offset (ascending ref, ">=": later wins), dword-wise copy
                if (tid < 8 * (int)(sizeof(CandT) / 4)) {
                    constexpr int W = sizeof(CandT) / 4;
                    const int o = tid / W, wd = tid - o * W;
                    float bv = pc[o * 8].val; int br = 0;
                    for (int rr = 1; rr < nvalid; rr++) {
                        const float v = pc[o * 8 + rr].val;
                        if (v >= bv) { bv = v; br = rr; }
                    }
                    int word = reinterpret_cast<const int *>(pc + o * 8 + br)[wd];
                    if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * stats[(size_t)mtile * 8 + o].y);     // val, t7[] * 1/sigma
                    reinterpret_cast<int *>(cand + ((size_t)mtile * 8 + o) * nrtile + rtile)[wd] = word;
                }
                __syncthreads();
            }
        }
    }
    }
}

// Second half of the split contraction (maxrin 1024): inverse FFT + argmax + best reference per particle-offset for the blocks
// [task0, task0 + ntask) whose CCF spectra gccf's SPLIT instantiation left in zscr[task - task0][N/2 + 1][64 TM TR]{Z_k, Z_{N-k}}.  A
// batch = the 8 references of one particle-offset of one 8 x 8 tile (8 consecutive pairs: one 128-byte piece per bin of the scratch):
// scratch -> LDS (row-padded images of wave_ifft3_argmax), one transform per wave in registers, the batch's best reference
// (ascending, ">=": later wins) scaled by 1/sigma to `cand`.  Persistent workgroups walk the batches; two per CU.
template <int TM, int TR>
__global__ __launch_bounds__(RA_GCCF_THREADS, 4) void gccf_ifft_kernel(DevGeom g, int n_mtile, int nrtile, int nref, CandT *__restrict__ cand,
                                                                    const float2 *__restrict__ zscr, const float2 *__restrict__ stats,
                                                                    int task0, int ntask)
{
    extern __shared__ __align__(16) float lds[];
    __shared__ CandT pc[8];
    constexpr int N = 1024, ZPAIRS = 64 * TM * TR, NW = RA_GCCF_THREADS / 64, PS = RA_IFFT3_PSTRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *xb = reinterpret_cast<float2 *>(lds);             // [8 pairs][PS]
    float2 *twa_s = xb + (size_t)NW * PS, *twb_s = twa_s + 16 * 16;
    ifft3_twiddles<16, 4>(g.tw, twa_s, twb_s, tid, RA_GCCF_THREADS);
    const int n_rt2 = (nrtile + TR - 1) / TR;
    if (g.ent_total) {          // live-offset lists: the chunk's entry count lives on the device
        n_mtile = (*g.ent_total + 7) >> 3;
        ntask = max(0, min(ntask, ((n_mtile + TM - 1) / TM) * n_rt2 - task0));
    }
    const int nbatch = ntask * TM * TR * 8;
    for (int bt_ = blockIdx.x; bt_ < nbatch; bt_ += gridDim.x) {
        const int tl = bt_ / (TM * TR * 8), rem = bt_ - tl * (TM * TR * 8), sub = rem >> 3, o = rem & 7;
        const int task = task0 + tl, mt2 = task / n_rt2, rt2 = task - mt2 * n_rt2;
        const int mtile = TM * mt2 + sub / TR, rtile = TR * rt2 + sub % TR;
        if (mtile >= n_mtile || rtile >= nrtile) continue;          // uniform over the workgroup
        const int ref0 = rtile * g.rpt, nvalid = min(g.rpt, nref - ref0);
        // scratch [bin 0 .. N/2][pairs]{Z_k, Z_{N-k}} -> LDS [pair][k]: pairs sub * 64 + 8 o .. + 7 are 128 contiguous bytes per bin;
        // bins 0 and N/2 are their own partners: the second value is the one that counts (the order of the unsplit kernel's stores)
        const float4 *src = reinterpret_cast<const float4 *>(zscr + (size_t)tl * ZPAIRS * (N + 2)) + sub * 64 + 8 * o;
        // (no barrier here: behind the one in front of the record reduction nobody reads the images any more, and the records are
        // written again only behind the next one)
        {
            constexpr int NT = 8 * (N / 2) / RA_GCCF_THREADS;
            float4 t[NT], tn = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const int idx = u * RA_GCCF_THREADS + tid;
                if (RA_DBG(g, 32)) t[u] = make_float4((float)idx, 1.f, 0.f, (float)u);      // profiling: no reads at all
                else
                t[u] = src[(size_t)(RA_DBG(g, 16) ? (idx >> 3) & 7 : idx >> 3) * ZPAIRS + (idx & 7)];      // profiling: cache-resident reads
            }
            if (tid < 8 && !RA_DBG(g, 32)) tn = src[(size_t)(N / 2) * ZPAIRS + tid];
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const int idx = u * RA_GCCF_THREADS + tid, kq = idx >> 3, km = (N - kq) & (N - 1);
                float2 *img = xb + (size_t)(idx & 7) * PS;
                if (kq) img[(kq >> 6) * 68 + (kq & 63)] = make_float2(t[u].x, t[u].y);
                img[(km >> 6) * 68 + (km & 63)] = make_float2(t[u].z, t[u].w);
            }
            if (tid < 8) xb[(size_t)tid * PS + (N / 2 >> 6) * 68] = make_float2(tn.z, tn.w);
        }
        __syncthreads();
        if (wave < nvalid) {
            float2 *x = xb + (size_t)wave * PS;
            float bq, bt; int iq, it;
            wave_ifft3_argmax<16, 4>(x, twa_s, twb_s, lane, bq, iq, bt, it);
            const bool mir = !g.nomirror && !(bq >= bt);        // qn >= qm keeps the straight match; nomirror: straight only
            const int jt = mir ? it : iq;
            CandT *dst = pc + wave;              // lanes 0..6 store the 7-point neighbourhood, lane 0 the rest
            if (lane < 7) {
                const float2 zz = x[ifft3_slot((jt + lane - 3 + N) & (N - 1))];
                dst->t7[lane] = mir ? zz.y : zz.x;
            }
            if (lane == 0) { dst->val = mir ? bt : bq; dst->jtot = jt + 1; dst->refmir = ((mir ? 1 : 0) << 16) | (ref0 + wave); }
        }
        __syncthreads();
        // best reference of the batch (ascending reference, ">=": later wins), dword-wise copy, * 1/sigma
        if (tid < (int)(sizeof(CandT) / 4)) {
            float bv = pc[0].val; int br = 0;
            for (int rr = 1; rr < nvalid; rr++) {
                const float v = pc[rr].val;
                if (v >= bv) { bv = v; br = rr; }
            }
            int word = reinterpret_cast<const int *>(pc + br)[tid];
            if (tid == 0 || tid >= 3) word = __float_as_int(__int_as_float(word) * stats[(size_t)mtile * 8 + o].y);
            reinterpret_cast<int *>(cand + ((size_t)mtile * 8 + o) * nrtile + rtile)[tid] = word;
        }
    }
}

// DC weights of the references: cdc[ref] = sum over rings of n_r * (prepared reference, bin 0) -- what subtracting the
// Normalize_ring mean from every sample of the particle does to the CCF spectrum
__global__ void ref_dc_weights_kernel(DevGeom g, const float *__restrict__ refspec, int nref, float *__restrict__ cdc)
{
    const int ref = blockIdx.x * blockDim.x + threadIdx.x;
    if (ref >= nref) return;
    const float inv = 1.0f / (float)g.maxrin;
    float s = 0.f;
    for (int r = 0; r < g.nring; r++) {
        const int4 ri = g.ringinfo[r];
        const int e = g.bin_off[0] + (r - g.bin_first[0]);
        s += (float)ri.z * refspec[(size_t)ref * g.lring + ri.x] * g.ent_wgt[e] * inv;
    }
    cdc[ref] = s;
}

// rot_shift2D for images that do not fit LDS: quadri_background reads global memory
// (test_mref_gpu_align.py:1055; arithmetic of transform_kernel)
__global__ __launch_bounds__(256) void transform_generic_kernel(int nx, const float *__restrict__ particles, int n,
                                                                int index0, const ra_result *__restrict__ res,
                                                                float *__restrict__ aligned, float *__restrict__ sums,
                                                                int *__restrict__ counts)
{
#pragma clang fp contract(off)
    const int p = blockIdx.x, tid = threadIdx.x, npix = nx * nx;
    if (p >= n) return;
    const float *img = particles + (size_t)p * npix;
    const ra_result r = res[p];
    const float ang = r.alpha * (float)M_PI / 180.0f;
    float delx = r.sx, dely = r.sy;
    while (delx >= (float)nx) delx -= nx;
    while (delx <= -(float)nx) delx += nx;
    while (dely >= (float)nx) dely -= nx;
    while (dely <= -(float)nx) dely += nx;
    const int xc = nx / 2, yc = nx / 2;
    const float shiftxc = xc + delx, shiftyc = yc + dely;
    const float cang = (float)cos((double)ang), sang = (float)sin((double)ang);
    float *dsum = sums ? sums + ((size_t)r.ref_id * 2 + ((index0 + p) & 1)) * npix : nullptr;
    float *dal = aligned ? aligned + (size_t)p * npix : nullptr;
    const int mstart = 1 - nx % 2;
    for (int i = blockIdx.y * blockDim.x + tid; i < npix; i += gridDim.y * blockDim.x) {
        const int iy = i / nx, ix = i - iy * nx;
        float y = (float)iy - shiftyc;
        float ycang = y * cang + yc;
        float ysang = -y * sang + xc;
        float x = (float)ix - shiftxc;
        float xold = x * cang + ysang;
        float yold = x * sang + ycang;
        float v = quadri_background_1b(img, nx, nx, xold + 1.0f, yold + 1.0f, ix + 1, iy + 1);
        int ox = ix;
        if (r.mirror && ix >= mstart) ox = mstart + (nx - 1) - ix;
        const int o = iy * nx + ox;
        if (dal) dal[o] = v;
        if (dsum) atomicAdd(dsum + o, v);
    }
    if (counts && tid == 0 && blockIdx.y == 0) atomicAdd(counts + r.ref_id, 1);
}

}  // namespace ralign
