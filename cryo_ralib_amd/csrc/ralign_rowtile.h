// "Row tile" variant of the contraction for 9 ... ~14 references: one workgroup contracts ONE A block that
// holds a whole row of the search grid (OB = 2*nkx+1 <= 8 offsets, 16 operand rows) against TWO reference
// tiles at once (two accumulator chains sharing the A fragment), so that the particle spectra are read from
// HBM once per iteration instead of once per reference tile.  With the 8-offset tiles of ccf_kernel the LDS
// image of the CCF spectra (2184 B per pair) caps a tile at 64 pairs, i.e. 8 references per sweep, and 10
// references cost two sweeps over the 1.3 MB of spectra of every particle -- measured, the operand loads
// alone (3.2 ms per 7143 particles, 5.3 TB/s) set the pace of that kernel.  7 offsets x 10 references = 70
// pairs fit.
//   A block (row layout): [bin panel][kk][16 rows][w] floats, row = 2 * (offset in row) + (Re | Im), LBP*16 + 64
//   floats; written by polar_fft_kernel through the (bin, kk, chunk) table rt_went / rt_wmeta.
//   cand: [(block * 8 + offset in row)][sweep]
#pragma once

#include "ralign_kernels.h"

namespace ralign {

#define RA_ROW_MAXPAIRS 72      // pairs whose CCF spectra fit LDS beside the records and the twiddle table

template <int NS> struct Operands3 { float a[NS], b0[NS], b1[NS]; };

template <int NS>
__device__ __forceinline__ void load_operands3(Operands3<NS> &o, const float *__restrict__ pa, const float *__restrict__ pb0,
                                               const float *__restrict__ pb1, int l16)
{
    constexpr int N4 = NS >> 2;
#pragma unroll
    for (int q = 0; q < N4; q++) {
        const float4 va = *reinterpret_cast<const float4 *>(pa + q * 256 + l16 * 4);
        const float4 v0 = *reinterpret_cast<const float4 *>(pb0 + q * 256 + l16 * 4);
        const float4 v1 = *reinterpret_cast<const float4 *>(pb1 + q * 256 + l16 * 4);
        o.a[4 * q] = va.x; o.a[4 * q + 1] = va.y; o.a[4 * q + 2] = va.z; o.a[4 * q + 3] = va.w;
        o.b0[4 * q] = v0.x; o.b0[4 * q + 1] = v0.y; o.b0[4 * q + 2] = v0.z; o.b0[4 * q + 3] = v0.w;
        o.b1[4 * q] = v1.x; o.b1[4 * q + 1] = v1.y; o.b1[4 * q + 2] = v1.z; o.b1[4 * q + 3] = v1.w;
    }
    int off = N4 * 256;
    if constexpr ((NS & 2) != 0) {
        const float2 va = *reinterpret_cast<const float2 *>(pa + off + l16 * 2);
        const float2 v0 = *reinterpret_cast<const float2 *>(pb0 + off + l16 * 2);
        const float2 v1 = *reinterpret_cast<const float2 *>(pb1 + off + l16 * 2);
        o.a[4 * N4] = va.x; o.a[4 * N4 + 1] = va.y;
        o.b0[4 * N4] = v0.x; o.b0[4 * N4 + 1] = v0.y;
        o.b1[4 * N4] = v1.x; o.b1[4 * N4 + 1] = v1.y;
        off += 128;
    }
    if constexpr ((NS & 1) != 0) { o.a[NS - 1] = pa[off + l16]; o.b0[NS - 1] = pb0[off + l16]; o.b1[NS - 1] = pb1[off + l16]; }
}

// one accumulator's epilogue: 2x2 exchange between the Re/Im column lanes, Z_k and Z_{N-k} into LDS
template <int N>
__device__ __forceinline__ void store_pair_spectrum(const f32x4 acc, float *Z, int pair, int odd, int k, bool live)
{
    typedef ZLayout<N> ZL;
    const float s0 = odd ? acc[0] : acc[2], s1 = odd ? acc[1] : acc[3];
    const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
    const float ca = odd ? r0 : acc[0], cb = odd ? r1 : acc[1];
    const float cc = odd ? acc[2] : r0, cd = odd ? acc[3] : r1;
    const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
    if (live) {
        *reinterpret_cast<float2 *>(Z + ZL::addr(pair, k)) = make_float2(apd + bpc, cmb + amd);
        *reinterpret_cast<float2 *>(Z + ZL::addr(pair, (N - k) & (N - 1))) = make_float2(apd - bpc, amd - cmb);
    }
}

template <int N, int NS>
__device__ __forceinline__ void contract_bin2(const Operands3<NS> &o, float *Z, int pair0, int pair1, bool live0, bool live1,
                                              int odd, int k)
{
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; s++) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s], o.b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s], o.b1[s], acc1, 0, 0, 0);
    }
    store_pair_spectrum<N>(acc0, Z, pair0, odd, k, live0);
    store_pair_spectrum<N>(acc1, Z, pair1, odd, k, live1);
}

template <int N, int NS, int NW>
__device__ __forceinline__ void contract_class2(const float *__restrict__ Ablk, const float *__restrict__ B0,
                                                const float *__restrict__ B1, float *Z, int kbeg, int kend, int p0_class,
                                                int wave, int l16, int pair0, int pair1, bool live0, bool live1, int odd)
{
    int k = kbeg + wave;
    if (k >= kend) return;
    Operands3<NS> cur, nxt;
    size_t e = (size_t)(p0_class + (k - kbeg) * 4 * NS) * 16;
    load_operands3<NS>(cur, Ablk + e, B0 + e, B1 + e, l16);
    while (true) {
        const int k1 = k + NW;
        if (k1 < kend) {
            e = (size_t)(p0_class + (k1 - kbeg) * 4 * NS) * 16;
            load_operands3<NS>(nxt, Ablk + e, B0 + e, B1 + e, l16);
        }
        contract_bin2<N, NS>(cur, Z, pair0, pair1, live0, live1, odd, k);
        if (k1 >= kend) break;
        const int k2 = k1 + NW;
        if (k2 < kend) {
            e = (size_t)(p0_class + (k2 - kbeg) * 4 * NS) * 16;
            load_operands3<NS>(cur, Ablk + e, B0 + e, B1 + e, l16);
        }
        contract_bin2<N, NS>(nxt, Z, pair0, pair1, live0, live1, odd, k1);
        if (k2 >= kend) break;
        k = k2;
    }
}

template <int N>
__global__ __launch_bounds__(RA_CCF_THREADS, RA_CCF_THREADS >= 1024 ? 4 : 2) void ccf_row_kernel(DevGeom g, const float *__restrict__ A,
                                                                                           const float *__restrict__ B, int nblocks,
                                                                                           int nrtile, int nref,
                                                                                           CandT *__restrict__ cand)
{
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    typedef ZLayout<N> ZL;
    extern __shared__ __align__(16) float Z[];
    __shared__ CandT pc[RA_ROW_MAXPAIRS];
    __shared__ float2 tws[R1 * R2];
    const int blk = blockIdx.x;
    if (blk >= nblocks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = RA_CCF_THREADS / 64;
    const int OB = g.rt_ob, rpt = g.rpt, RT2 = 2 * rpt, npairs = OB * RT2;

    for (int i = tid; i < R1 * R2; i += RA_CCF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2) * (g.maxrin / N)) & (g.maxrin - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    const float2 *twl = tws + (lane & 15);

    for (int sweep = 0; sweep < g.rt_nsweep; sweep++) {
        const int ref0 = sweep * RT2;
        const int nv = min(RT2, nref - ref0);                 // live references of this sweep (tiles are contiguous)
        const int nv0 = min(rpt, nv), nv1 = nv - nv0;
        // ---- phase 1
        {
            const int l16 = lane, odd = lane & 1;              // operand lane = kk * 16 + row / column
            const int o = 2 * (lane >> 4) + odd, rs = (lane & 15) >> 1;
            const int pair0 = o * RT2 + rs, pair1 = pair0 + rpt;
            const bool live0 = o < OB && rs < nv0, live1 = o < OB && rs < nv1;
            const float *Ablk = A + (size_t)blk * g.rt_ablk;
            const float *B0 = B + (size_t)(2 * sweep) * g.LBP * 16;
            const float *B1 = (2 * sweep + 1 < nrtile) ? B0 + (size_t)g.LBP * 16 : B0;
            int p0 = 0;
            for (int c = 0; c < g.n_class; c++) {
                const int kb = g.class_k0[c], ke = (c + 1 < g.n_class) ? g.class_k0[c + 1] : g.class_k0_end, ns = g.class_ns[c];
                switch (ns) {
#define RA_CASE(NSV) case NSV: contract_class2<N, NSV, NW>(Ablk, B0, B1, Z, kb, ke, p0, (wave - g.class_rot[c] + NW) % NW, l16, pair0, pair1, live0, live1, odd); break;
                    RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6)
                    RA_CASE(7) RA_CASE(8) RA_CASE(9)
#undef RA_CASE
                default: break;
                }
                p0 += (ke - kb) * 4 * ns;
            }
        }
        __syncthreads();
        // ---- phase 2: inverse FFT + argmax of every pair, 4 pairs per wave and round
        {
            const int j = lane & 15, sub = lane >> 4;
            const int nround = (npairs + 3) >> 2;
            for (int r = wave; r < nround; r += NW) {
                const int praw = 4 * r + sub, pair = min(praw, npairs - 1);      // a short last round repeats the last pair
                CandT c;
                ifft_argmax_core<ZL, N, 1, R2>(Z, twl, pair, pair, j, c);
                if (j == 0 && praw < npairs) {
                    c.refmir |= ref0 + pair % RT2;
                    pc[pair] = c;
                }
            }
        }
        __syncthreads();
        // ---- best reference of the sweep per offset (ascending ref, ">=": later wins)
        if (tid < OB) {
            CandT best = pc[tid * RT2];
            for (int rr = 1; rr < nv; rr++) {
                const CandT c = pc[tid * RT2 + rr];
                if (c.val >= best.val) best = c;
            }
            cand[((size_t)blk * 8 + tid) * g.rt_nsweep + sweep] = best;
        }
        __syncthreads();
    }
}

// polar write-out into the row layout: thread = (table entry, source row of the pass).  The entry names the
// <= 4 consecutive rings of one chunk of bin k; the 8 source rows of the pass (4 offsets x Re/Im) are rows
// row0 .. row0+7 of the block, so the 8 threads of an entry write one 128-byte line (rows of offsets beyond
// `nlive` are written as zeros).
__device__ __forceinline__ void write_rows(const DevGeom &g, const float *bufs, const float *rsg, float *__restrict__ Ablk,
                                           int row0, int nlive, int tid, int nthreads)
{
    const int total = g.rt_nwent * 8;
    for (int idx = tid; idx < total; idx += nthreads) {
        const int ent = idx >> 3, r8 = idx & 7, slot = r8 >> 1, comp = r8 & 1;
        const int4 src = g.rt_went[ent];
        const int2 meta = g.rt_wmeta[ent];
        const int w = meta.y;                      // entries are sorted by width: a wave sees one width (two at a seam)
        const float rs = slot < nlive ? rsg[slot] : 0.f;
        const float *sb = bufs + slot * g.sbuf + comp;
        float *dst = Ablk + meta.x + (row0 + r8) * w;
        const float v0 = src.x >= 0 ? sb[src.x] * rs : 0.f;
        if (w == 4) {
            const float v1 = src.y >= 0 ? sb[src.y] * rs : 0.f, v2 = src.z >= 0 ? sb[src.z] * rs : 0.f;
            const float v3 = src.w >= 0 ? sb[src.w] * rs : 0.f;
            *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
        } else if (w == 2) {
            const float v1 = src.y >= 0 ? sb[src.y] * rs : 0.f;
            *reinterpret_cast<float2 *>(dst) = make_float2(v0, v1);
        } else {
            *dst = v0;
        }
    }
}

}  // namespace ralign
