// Particle-resident search kernel for gfx950 (MI355X): the whole search of one particle -- polar resampling,
// Normalize_ring, ring FFTs, the reference x particle contraction (Crosrng_ms), inverse FFTs and argmax -- runs inside
// one workgroup (16 waves); the particle spectra never leave the CU.  HBM sees the image once (32 KB at 90 x 90) and one
// candidate record per search offset; the prepared references (the MFMA B operand) stream from L2.
//
// A pass handles 4 consecutive search offsets, exactly like a pass of polar_fft_kernel (same wave-jobs, same LDS ring
// buffers): Polar2Dm + Normalize_ring statistics + Frngs of all rings into 4 ring buffers.  Instead of writing operand
// panels to HBM, the pass goes on inside the CU:
//   contraction   v_mfma_f32_4x4x1_16b_f32: one instruction = 16 Fourier bins x [4 rows = 2 offsets x (Re, Im) D] x
//                 [4 columns = 2 references x (Re, Im) C] for ONE ring -- no padding of the reference count (10
//                 references = 5 pairs) and none of the ring count.  The A operand is read straight from the LDS ring
//                 buffers (one ds_read per MFMA), the B operand (4 rings per 16-byte load) from L2.  A unit =
//                 (16-bin group, reference pair, offset pair) accumulates over the rings that have those bins in
//                 4 VGPRs; the accumulators exist only between the ring jobs and the inverse FFT of the same pass,
//                 so they never compete with the ring jobs for registers.
//   spectra       Z_k = Q_k + i T_k, Z_{N-k} = conj Q_k + i conj T_k go to LDS (the ring-buffer space, idle by then),
//                 one N-point complex inverse FFT per (offset, reference) pair yields q and t together, wavefront
//                 argmax with EMAN2's tie rules (ifft_argmax of ralign_kernels.h); the best reference per offset
//                 leaves the CU as one 40-byte record.
// Normalize_ring: subtracting the mean only moves the DC coefficients, so it is applied to the contracted DC term
// (a -= avg * sum_r n_r B_r(0)); the scale 1/sigma multiplies the peak record instead of every spectrum element.
// Synchronisation per pass: workgroup barriers (LDS traffic only) after the ring jobs, the contraction and the spectra
// store; the end of the inverse FFTs is an arrival counter that the NEXT pass's ring jobs wait for between their sampling
// and their first write to the ring buffers (PassSync), so waves that are through with their transforms sample meanwhile.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq as called from test_mref_gpu_align.py:1043-1044 and
// sp_alignment.ali2d_single_iter (test_reffree_gpu_align.py:844-847); SURVEY.md Appendix A.3-A.9.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ralign_geom.h"
#include "ralign_kernels.h"

namespace ralign {

constexpr int RF_WAVES = 16;
constexpr int RF_THREADS = RF_WAVES * 64;
constexpr int RF_MAXREF = 16;     // references the accumulators of one pass hold (<= 4 reference pairs x 2 offset pairs per wave)

struct FusedGeom {
    int on;
    int ng;                        // 16-bin groups: maxrin / 32 (the Nyquist bin of full-length rings rides in bin 0)
    int wpg;                       // waves per group = 16 / ng
    int nrp;                       // reference pairs
    int nrpw;                      // reference pairs per wave = ceil(nrp / wpg): template parameter of the kernel
    int rz, nzr, rz_inv;           // references per inverse-FFT round, rounds, ceil(2^16 / rz) (division by multiply-shift)
    int b_floats;
    int grp_ring0[16];             // first ring that has bins of group m (rings are sorted by length)
    int s_crop, s_cropm;           // solo / duo / pair kernels: side of the LDS image when it is a crop around the particle's sampling centre (0: the whole image), and
                                   // the distance of its first column from that centre (crop_plan)
    int pack;                      // search_tiled_kernel: dense offset stream over the workgroup's particles (set at launch)
    int grp_boff[16];              // float offset of group m's B block: [pair][ring quad][lane][4 rings]
    int grp_nq[16];                // ring quads of group m
    const int *bsrc;               // [b_floats] (entry << 5 | reference << 1 | imaginary part), -1 = 0
    int roff[68];                  // ring offsets in a ring buffer (padded with the last ring's)
    int gstr;                      // ints per group in the LDS table of ring offsets (quads of 4, padded by one quad)
    const float *cdc_w;            // [nref] DC weights of the references: sum over rings of n_r * B_r(bin 0) (ref_dc_weights_kernel)
    int wmap[16];                  // contraction role of wave w: bin group | share of the reference pairs << 8 (balance_waves)
    int stat_wave[4];              // the wave that reduces the Normalize_ring partials of offset slot s
    int ctr_wave;                  // the wave that writes the next pass's sampling centres (slack behind its contraction, no statistics)
    int ntile, nh;                 // search_tiled_kernel (ralign_tiled.h): reference tiles per pass, reference pairs per tile
    int ifft_full;                 // inverse-FFT slots 32 .. go to waves 8, 9, .. in FULL calls (4 transforms each) instead of half-filled ones
    // search_solo_kernel (ralign_solo.h): one search offset resident per pass (maxrin 512)
    int s_pst, s_rows;             // row stride and rows of the LDS image (no search-range border: out-of-window offsets are skipped)
    int s_sbuf;                    // floats of the ring buffer | CCF spectra of a reference tile
    int s_rank[16];                // wave w runs ring jobs s_rank[w], s_rank[w] + 16, ..; the highest ranks have none
    int s_call[16];                // the transform (reference of the tile) wave w carries in every tile, -1: none
    int s_stat, s_ctr, s_rec;      // waves that reduce the Normalize_ring partials, write the next centre, merge the records
    int s_r2;                      // search_duo_kernel: a second round of ring jobs starts at rank 16 - s_r2
};

struct FusedPlanHost {
    FusedGeom f{};
    std::vector<int> bsrc;
    std::vector<float> cdc_w;
    size_t lds_bytes = 0;
};

// B-stream source of (bin k, ring r, reference ref, Re/Im part): entry code or -1 (zero).  The Nyquist coefficient of
// a full-length ring rides in the imaginary slot of bin 0 (EMAN2's packing); bins a ring does not have are zero.
inline int rf_bsrc_code(const Geometry &g, int k, int r, int ref, int part)
{
    if (r >= g.nring) return -1;
    const int n = g.numr[3 * r + 2];
    int ks = k;
    if (k == 0 && part == 1) {
        if (n != g.maxrin) return -1;
        ks = n / 2; part = 0;
    } else if (k > n / 2 || (k == n / 2 && n == g.maxrin)) return -1;
    else if (k == n / 2 && part == 1) return -1;        // Nyquist of a shorter ring is real
    const int e = g.bin_off[ks] + (r - g.bin_first[ks]);
    return (e << 8) | (ref << 1) | part;
}

// B-stream layout shared by search_fused_kernel and search_tiled_kernel: [bin group][reference pair][ring quad][lane][4 rings]
inline void rf_layout_b(const Geometry &g, int nref, FusedGeom &f, std::vector<int> &bsrc)
{
    int boff = 0;
    for (int m = 0; m < f.ng; m++) {
        int r0 = 0;
        while (r0 < g.nring) {      // first ring with a bin >= 16 m (bins 0 .. n/2, the full-length Nyquist merged into bin 0)
            const int n = g.numr[3 * r0 + 2], nbin = (n == g.maxrin) ? n / 2 : n / 2 + 1;
            if (16 * m < nbin) break;
            r0++;
        }
        f.grp_ring0[m] = r0;
        f.grp_nq[m] = (g.nring - r0 + 3) / 4;
        f.grp_boff[m] = boff;
        boff += f.nrp * f.grp_nq[m] * 256;
    }
    f.b_floats = boff;
    bsrc.assign(boff, -1);
    for (int m = 0; m < f.ng; m++)
        for (int rp = 0; rp < f.nrp; rp++)
            for (int rq = 0; rq < f.grp_nq[m]; rq++)
                for (int lane = 0; lane < 64; lane++)
                    for (int c = 0; c < 4; c++) {
                        const int r = f.grp_ring0[m] + 4 * rq + c, k = 16 * m + (lane >> 2), j = lane & 3;
                        const int ref = 2 * rp + (j >> 1);
                        if (ref >= nref) continue;
                        bsrc[f.grp_boff[m] + ((rp * f.grp_nq[m] + rq) * 64 + lane) * 4 + c] = rf_bsrc_code(g, k, r, ref, j & 1);
                    }
}

// lds_polar_floats: LDS floats polar_fft_kernel needs for this geometry (image, 4 ring buffers, tables, reductions)
inline bool build_fused_plan(const Geometry &g, int nref, int sbuf, size_t lds_polar_floats, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.bsrc.clear(); out.cdc_w.clear();
    if (nref > RF_MAXREF || !(g.maxrin == 256 || g.maxrin == 128) || g.numr[2] < 8) return false;
    f.ng = g.maxrin / 32; f.wpg = RF_WAVES / f.ng;
    f.nrp = (nref + 1) / 2;
    f.nrpw = (f.nrp + f.wpg - 1) / f.wpg;
    if (f.nrpw > 4) return false;
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    int rzmax = std::min(64 / 4, (4 * sbuf) / (4 * zstride));
    rzmax = std::max(1, rzmax);
    f.nzr = (nref + rzmax - 1) / rzmax;
    f.rz = (nref + f.nzr - 1) / f.nzr;
    f.rz_inv = (65536 + f.rz - 1) / f.rz;
    rf_layout_b(g, nref, f, out.bsrc);
    // Contraction roles.  Waves w, w + 4, w + 8, w + 12 of a workgroup share a SIMD, and the matrix pipe of a SIMD is what the
    // contraction phase runs on (wave timeline: ~8 matrix cycles x 8 NH instructions per ring quad and wave): deal the
    // (bin group, share of the reference pairs) items to the waves so that the four SIMDs get equal numbers of matrix
    // instructions, the heavier items to the older waves (the SIMDs issue oldest first).
    {
        struct Item { int m, s, cost; };
        std::vector<Item> items;
        for (int m = 0; m < f.ng; m++)
            for (int sh = 0; sh < f.wpg; sh++) {
                const int pairs = std::max(0, std::min(f.nrpw, f.nrp - sh * f.nrpw));
                items.push_back({m, sh, f.grp_nq[m] * pairs});
            }
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
        int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
        int light[4] = {0, 1, 2, 3};
        // (counting the inverse-FFT calls of a SIMD's waves into its load before the deal -- 8 .. 19 units per call -- measured
        // within noise, more than that slower)
        f.ifft_full = (RA_EXP_ENV("RALIGN_IFFT_FULL") && ra_atoi(RA_EXP_ENV("RALIGN_IFFT_FULL")) == 0) ? 0 : 1;
        for (const Item &it : items) {
            int c = -1;
            for (int q = 0; q < 4; q++)
                if (used[q] < 4 && (c < 0 || load[q] < load[c])) c = q;
            const int w = c + 4 * used[c];
            f.wmap[w] = it.m | (it.s << 8);
            load[c] += it.cost; used[c]++;
            light[c] = w;            // items arrive heaviest first: the last one of a class is its lightest
        }
        for (int q = 0; q < 4; q++) f.stat_wave[q] = light[q];
        // the next pass's sampling centres (two dependent global reads in front of the wave's contraction): the third wave of SIMD 0
        // (wave 8: 3.8 k cycles of slack at the barrier behind the contraction in the headline geometry; wave 4, which ended that
        // phase, did it until round 4: 29.6 -> 29.35 ms per 50 k), or the next wave that reduces no statistics
        f.ctr_wave = 8;
        for (int w : {8, 9, 10, 11, 4, 5, 6, 7})
            if (w != light[0] && w != light[1] && w != light[2] && w != light[3]) { f.ctr_wave = w; break; }
    }
    out.cdc_w.assign(g.nring, 0.f);
    if (g.nring > 64) return false;
    for (int r = 0; r < 68; r++) f.roff[r] = g.ring_off[std::min(r, g.nring - 1)];
    f.gstr = 4 * ((g.nring + 3) / 4) + 8;      // two quads of slack: the contraction reads the offsets of up to two quads past a group's last
    const size_t fl = lds_polar_floats + 2 * g.maxrin + (4 * RF_MAXREF + 4) * (sizeof(CandT) / 4) + 8 * (g.nring + 12) + RF_MAXREF + 64;
    out.lds_bytes = fl * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024 && 4 * f.rz * zstride <= 4 * sbuf;
    return f.on != 0;
}

// ------------------------------------------------------------------------------------------
// device

// prepared references -> B stream of the fused kernel (Applyws weights and 1/maxrin folded in)
__global__ void pack_refs_fused_kernel(DevGeom g, FusedGeom f, const float *__restrict__ refspec, int nref,
                                       float *__restrict__ Bf)
{
    // class-resident mode: gridDim.y classes of nref (= 1) references each, one B stream per class
    refspec += (size_t)blockIdx.y * nref * g.lring;
    Bf += (size_t)blockIdx.y * f.b_floats;
    const float inv = 1.0f / (float)g.maxrin;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < f.b_floats; idx += gridDim.x * blockDim.x) {
        const int code = f.bsrc[idx];
        float v = 0.f;
        if (code >= 0) {
            const int e = code >> 8, ref = (code >> 1) & 127;
            if (ref < nref) v = refspec[(size_t)ref * g.lring + g.ent_src[e] + (code & 1)] * g.ent_wgt[e] * inv;
        }
        Bf[idx] = v;
    }
}

__device__ __forceinline__ float rf_f4(const float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// Z_k = Q_k + i T_k and Z_{N-k} = conj Q_k + i conj T_k from the four products a = c1 d1, b = c1 d2, c = c2 d1,
// d = c2 d2 summed over rings (Util::Crosrng_ms: Q = (a + d) + i (c - b), T = (a - d) - i (b + c)); bin 0 carries the
// DC term in a and the Nyquist term of the full-length rings in d
template <int N>
__device__ __forceinline__ void rf_store_z(float *Z, int zslot, int k, float ca, float cb, float cc, float cd)
{
    typedef ZLayout<N> ZL;
    if (k == 0) {
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, 0)) = make_float2(ca, ca);
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, N / 2)) = make_float2(cd, cd);
    } else {
        const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, k)) = make_float2(apd + bpc, cmb + amd);
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, N - k)) = make_float2(apd - bpc, amd - cmb);
    }
}

// contraction of one pass for one wave: NH live reference pairs rp0 .. rp0 + NH - 1 of bin group xm, both offset pairs;
// unit (h, op) accumulates in acc[2 h + op]
// OP1 = false (last pass with at most two live offsets): the second offset pair is padding, its A reads and multiplies are skipped
template <int NRPW, int NH, bool OP1>
__device__ __forceinline__ void rf_contract(const DevGeom &g, const FusedGeom &f, const float *bufs, const int *goff_s,
                                            const float *__restrict__ Bf, int xm, int rp0, int ln, f32x4 (&acc)[2 * NRPW],
                                            const float4 (&b0)[NRPW], int4 oA, int4 oB, bool tl = false, int tgrp = 0, int twave = 0)
{
    const int xb = ln >> 2, xj = ln & 3;
    const int nq = f.grp_nq[xm];
    // B operands: one wave-uniform row base per reference pair (scalar registers) + the lane's 16 bytes; the quad index moves
    // the scalar part only (global_load ... v_lane, s[base]: no vector instruction on the request path).  The schedule
    // requests up to one quad past the end of a row (never multiplied): the stream is allocated with one quad of slack.
    const char *bu[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) bu[h] = reinterpret_cast<const char *>(Bf + f.grp_boff[xm] + (size_t)((rp0 + h) * nq) * 256);
    const unsigned loff = (unsigned)ln * 16u;
    // A element of (offset pair op, ring r): bufs[(2 op + (xj >> 1)) * sbuf + roff[r] + 2 (16 m + b) + (xj & 1)]
    const char *abase = reinterpret_cast<const char *>(bufs + (xj >> 1) * g.sbuf + 2 * (16 * xm + xb) + (xj & 1));
    const int a1off = 8 * g.sbuf;
    // ring offsets of the quads: LDS table row of the group, walked with a pointer kept in a vector register (one v_add per
    // trip, immediate offsets in between); rows are padded with copies of the last ring's offset: no clamps
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) i32x4 *lds_quad_p;
    lds_quad_p gq = (lds_quad_p)(goff_s + xm * f.gstr);
    asm volatile("" : "+v"(gq));
    auto quad_offsets = [&](int i) { const i32x4 q = gq[i]; return make_int4(q.x, q.y, q.z, q.w); };
    float4 bA[NH], bB[NH];
    float aA[8], aB[8];
#ifdef RALIGN_PROFILE_SWITCHES
    for (int h = 0; h < NH; h++) bB[h] = b0[h];          // defined operands for the switches that skip the requests
    for (int c = 0; c < 8; c++) aA[c] = aB[c] = (float)(ln + c);
#endif
    // one running 32-bit offset in a vector register (lane part + quad part, advanced once per trip) on top of the scalar row
    // bases: global_load ... v_off, s[base] offset:imm -- no 64-bit pointer arithmetic per request
    unsigned voff = loff;                  // + 1024 per quad consumed
    auto load_b = [&](int ahead, float4 (&b)[NH]) {      // B of the quad `ahead` quads past the running offset
        if (RA_DBG(g, 512)) return;       // profiling: no B requests in the loop
#pragma unroll
        for (int h = 0; h < NH; h++)
            b[h] = *reinterpret_cast<const float4 *>(bu[h] + (size_t)(RA_DBG(g, 32) ? loff : voff) + (size_t)ahead * 1024);
    };
    auto read_a = [&](int4 o, float (&a)[8]) {     // 4 rings x 2 offset pairs (padded ring slots have zero B)
        if (RA_DBG(g, 64)) o = make_int4(0, 0, 0, 0);
        if (RA_DBG(g, 2048)) return;      // profiling: no A reads in the loop
        a[0] = *reinterpret_cast<const float *>(abase + o.x); a[2] = *reinterpret_cast<const float *>(abase + o.y);
        a[4] = *reinterpret_cast<const float *>(abase + o.z); a[6] = *reinterpret_cast<const float *>(abase + o.w);
        if constexpr (OP1) {
            a[1] = *reinterpret_cast<const float *>(abase + o.x + a1off); a[3] = *reinterpret_cast<const float *>(abase + o.y + a1off);
            a[5] = *reinterpret_cast<const float *>(abase + o.z + a1off); a[7] = *reinterpret_cast<const float *>(abase + o.w + a1off);
        }
    };
    auto mul_rq = [&](const float (&a)[8], const float4 (&b)[NH]) {
        if (RA_DBG(g, 128)) {        // profiling: operands stay live, no matrix instructions
#pragma unroll
            for (int c = 0; c < 8; c++) asm volatile("" :: "v"(a[c]));
#pragma unroll
            for (int h = 0; h < NH; h++) asm volatile("" :: "v"(b[h].x), "v"(b[h].y), "v"(b[h].z), "v"(b[h].w));
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int h = 0; h < NH; h++) {
                acc[2 * h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2 * c], rf_f4(b[h], c), acc[2 * h], 0, 0, 0);
                if constexpr (OP1) acc[2 * h + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2 * c + 1], rf_f4(b[h], c), acc[2 * h + 1], 0, 0, 0);
            }
    };
    // modulo schedule over the ring quads: while quad rq is multiplied, the A operands of quad rq + 1 are on
    // their way from LDS, its B operands from L2, and the ring offsets of quad rq + 2 from the LDS table
    // (sched_barrier: hipcc otherwise sinks the requests to just in front of their first use).  The wave timeline
    // (scripts/fused_timeline.sh) shows 800 - 1000 cycles per ring quad against 200 cycles of matrix time; neither the B
    // stream (three buffers: slower; every request to one L1-resident line: unchanged) nor the A reads are what it waits
    // for -- the per-quad chain of scalar index arithmetic, LDS round trips and the matrix burst is.
    // quad 0 (its B operands were requested before the barrier that ends the ring jobs) starts the accumulators: C = 0
    // is an inline constant of the matrix instruction, so nothing is cleared
    auto mul_first = [&](const float (&a)[8], const float4 (&b)[NRPW]) {
        if (RA_DBG(g, 128)) {
#pragma unroll
            for (int i = 0; i < 2 * NH; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            return;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NH; h++) {
            acc[2 * h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[h].x, zero, 0, 0, 0);
            if constexpr (OP1) acc[2 * h + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[h].x, zero, 0, 0, 0);
            else acc[2 * h + 1] = zero;
        }
#pragma unroll
        for (int c = 1; c < 4; c++)
#pragma unroll
            for (int h = 0; h < NH; h++) {
                acc[2 * h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2 * c], rf_f4(b[h], c), acc[2 * h], 0, 0, 0);
                if constexpr (OP1) acc[2 * h + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2 * c + 1], rf_f4(b[h], c), acc[2 * h + 1], 0, 0, 0);
            }
    };
#pragma unroll
    for (int i = 2 * NH; i < 2 * NRPW; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};      // a dummy pair: defined, never stored
    const int ql = nq - 1;
    read_a(oA, aA);
    load_b(1, bB);
    read_a(oB, aB);
    oA = quad_offsets(2);
    gq += 1;                     // gq[i] is now quad rq + i of the trip that starts at rq = 1
    __builtin_amdgcn_sched_barrier(0);
    mul_first(aA, b0);
    __builtin_amdgcn_sched_barrier(0);
    // Quads 1 .. nq - 1, two per trip; at the head of a trip (aB, bB) hold quad rq and oA the ring offsets of quad rq + 1.
    // Both multiplies of a trip are unconditional: with the second one under `if (rq + 1 < nq)` the compiler sank the
    // requests of its B operands into that block, right in front of their use (one exposed memory latency per trip); a
    // last single quad is multiplied after the loop.
#pragma unroll 1
    for (int rq = 1; rq + 1 < nq; rq += 2) {
        RA_STAMP(g, tl && rq < 8, tgrp, twave, 9 + (rq >> 1));       // profiling builds: iteration starts (stamps 9 .. 12)
        load_b(2, bA);
        read_a(oA, aA);
        oB = quad_offsets(2);
        __builtin_amdgcn_sched_barrier(0);
        mul_rq(aB, bB);
        __builtin_amdgcn_sched_barrier(0);
        load_b(3, bB);
        voff += 2048;
        read_a(oB, aB);
        oA = quad_offsets(3);
        gq += 2;
        __builtin_amdgcn_sched_barrier(0);
        mul_rq(aA, bA);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!(nq & 1)) mul_rq(aB, bB);
}

// workgroup barrier that orders LDS traffic only: global requests (the B stream, the record stores) stay in flight
// across it (__syncthreads() drains them with s_waitcnt vmcnt(0))
#define RF_LDS_BARRIER()                                                   \
    do {                                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    \
        __builtin_amdgcn_s_barrier();                                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");    \
    } while (0)

// ring_job's `sync`: the point where the previous pass's inverse FFTs must be over (their spectra occupy the ring buffers),
// between the sampling and the first write to the ring buffers.  Not an s_barrier: a wave that finishes its transforms adds
// one to an LDS counter and walks on into its sampling; here a wave only waits until the counter says that all 16 are done
// (a barrier at this point would also wait for the SAMPLING of the waves that transform last).  on / target: wave-uniform.
struct PassSync {
    bool on;
    const int *done;        // LDS: number of (wave, pass) inverse-FFT phases finished by this workgroup
    int target;
#ifdef RALIGN_PROFILE_SWITCHES
    unsigned long long *tl; // wave timeline (profiling builds): stamps 13 / 14 of this wave's row before / after the wait
#endif
    __device__ __forceinline__ void operator()() const
    {
        if (!on) return;
#ifdef RALIGN_PROFILE_SWITCHES
        if (tl && threadIdx.x % 64 == 0) tl[13] = clock64();
#endif
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#ifdef RALIGN_PROFILE_SWITCHES
        if (tl && threadIdx.x % 64 == 0) tl[14] = clock64();
#endif
    }
};

// ONE: all references of a pass fit one store / inverse-FFT round (f.nzr == 1); the two-round code stays out of that kernel
// (the instruction cache holds 64 KB for two CUs: code that is never run still spreads the code that is)
// SB: ring-buffer stride known at compile time (RF_SBUF_FIXED floats; the engine pads the stride up to it when the LDS
// allows) or 0 = g.sbuf.  With a constant stride the second offset pair of every A operand is an immediate offset of its
// read (ds_read2st64_b32: both offset pairs in one instruction): 4 instead of 8 address and LDS instructions per ring quad.
#define RF_SBUF_FIXED 6432       // >= the stride at ou = 36 (6416); 8 x 6432 bytes is a multiple of 256
// PACK: the offsets of consecutive particles of a workgroup fill the passes without padding (see the pass loop)
// CROP: the LDS image is a crop around the particle's sampling centre (engines of the size-generic class, tcrop_wanted); a template
// parameter because the run-time branch in load_image cost the 90 x 90 instantiations 0.7 % (two more registers, nine more spilled scalars)
template <int N, int NRPW, bool ONE, int SB, bool PACK = false, bool CROP = false>
__global__ __launch_bounds__(RF_THREADS) void search_fused_kernel(DevGeom g_in, FusedGeom f, const float *__restrict__ particles,
                                                                  const float *__restrict__ state, int n,
                                                                  const float *__restrict__ Bf0, int nref,
                                                                  CandT *__restrict__ cand, const int *__restrict__ cls)
{
    // cls != null (class-resident mode, one reference): particle p is aligned to reference cls[p], whose B stream is
    // Bf0 + cls[p] * b_floats (pack_refs_fused_kernel with blockIdx.y = class)
    DevGeom g = g_in;
    g.maxrin = N; g.lg_maxrin = __builtin_ctz(N);      // the kernel is instantiated per maxrin: constants, not kernel arguments
    if constexpr (SB != 0) g.sbuf = SB;
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    extern __shared__ __align__(16) float lds[];
    // LDS plan of polar_fft_kernel, then the extras of this kernel
    const int npad = g.pst * g.pst;
    float *img = lds;
    float *bufs = lds + ((npad + 3) & ~3);                             // [4][sbuf] ring buffers | CCF spectra
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + 4 * g.sbuf);     // [maxrin]
    float2 *qt_s = tw_s + g.maxrin;                                    // [n_qtab]
    int4 *inst_s = reinterpret_cast<int4 *>(qt_s + g.n_qtab + (g.n_qtab & 1));   // [n_inst]
    int4 *jobs_s = inst_s + g.n_inst;                                  // [n_job]
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job);      // [n_inst]
    float *red = instw_s + g.n_inst;    // [8] -, [8] avg / rsigma, [8] centres, [4*nring*2] ring partials
    float2 *tws = reinterpret_cast<float2 *>(red + 24 + 8 * g.nring + ((g.n_inst + 8 * g.nring) & 1));   // [R1*R2] inverse-FFT twiddles
    CandT *pc = reinterpret_cast<CandT *>(tws + R1 * R2);              // [4][nref] records of the pass
    int *goff_s = reinterpret_cast<int *>(pc + 4 * RF_MAXREF + 4);     // [ng][gstr] ring offsets (bytes) of every group's ring quads
    float *cdc_s = reinterpret_cast<float *>(goff_s + f.ng * f.gstr);      // [RF_MAXREF] DC weights of the references
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tables and the ring-buffer slack are set up once per workgroup; the workgroup then walks over its particles
    for (int i = tid; i < g.maxrin; i += RF_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RF_THREADS) qt_s[i] = g.qtab[i];
    for (int i = tid; i < g.n_inst; i += RF_THREADS) { inst_s[i] = g.inst[i]; instw_s[i] = g.instw[i]; }
    for (int i = tid; i < g.n_job; i += RF_THREADS) jobs_s[i] = g.jobs[i];
    for (int i = tid; i < R1 * R2; i += RF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2) * (g.maxrin / N)) & (g.maxrin - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    for (int i = tid; i < 4 * g.sbuf; i += RF_THREADS) bufs[i] = 0.f;      // slack between rings must hold finite values
    for (int i = tid; i < g.pst * g.pst; i += RF_THREADS) img[i] = 0.f;    // zero border of the padded image (load_image writes the pixels only)
    for (int i = tid; i < RF_MAXREF; i += RF_THREADS) cdc_s[i] = i < nref ? f.cdc_w[i] : 0.f;
    for (int i = tid; i < f.ng * f.gstr; i += RF_THREADS) {
        const int m = i / f.gstr, j = i - m * f.gstr;
        goff_s[i] = 4 * f.roff[min(f.grp_ring0[m] + j, g.nring - 1)];
    }
    const float *imgb = img + (g.bd - 1) * g.pst + (g.bd - 1);          // (f.s_crop: re-based with every image, load_image)
    int *ifft_done = reinterpret_cast<int *>(red + 6);      // arrival counter of the inverse-FFT phases (PassSync)
    if (tid == 0) *ifft_done = 0;
    int done_target = 0;

    // contraction roles: group m = 16 bins (block b = lane >> 2 is bin 16 m + b); a wave takes NRPW reference pairs
    // rp0 .. rp0 + NRPW - 1 of its group and both offset pairs: unit (h, op) accumulates in acc[2 h + op].
    // A row = lane & 3 = (offset in pair, Re/Im), C column = lane & 3 = (reference in pair, Re/Im); after the 2x2
    // exchange the even lane keeps the first offset of the pair, the odd lane the second.
    // (The lane arithmetic of the contraction and of the spectra rounds is rebuilt in every pass from a zero the
    // compiler cannot see through -- red[7], rewritten per pass -- so that it is not hoisted out of the pass loop and
    // kept alive across the ring jobs, which need every register they can get.)
    constexpr int NU = 2 * NRPW;
    const int xm = f.wmap[wave] & 255, rp0 = (f.wmap[wave] >> 8) * NRPW;

    // The workgroup's particles p(i) = blockIdx.x + i gridDim.x form one stream of search offsets, SPP slots per particle, cut into
    // passes of 4.  PACK = false: SPP = nshift_pad, every particle starts a pass and the last pass of a particle carries padding
    // (49 offsets = 12 passes + 1/4: the 13th costs as much as a full one -- every phase has a latency floor -- 7.3 % of the time for
    // 2 % of the work).  PACK = true: SPP = nshift, the stream is dense -- 4 particles x 49 offsets = 49 full passes -- and a pass
    // may hold the last offsets of one particle and the first of the next: its ring jobs then run twice, over the slots of the
    // resident image, and after the next image has replaced it over the rest.  Everything behind the sampling is per slot
    // (centres, Normalize_ring statistics, spectra, records), so the results are the same to the bit.
    const int SPP = PACK ? g.nshift : g.nshift_pad;
    const int npw = (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nslots = npw * SPP, npass = (nslots + 3) >> 2;
    auto particle_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    // the particle's pixels into the padded LDS image (its zero border is written once, above): a wave per row, every request of
    // the wave in flight at once -- as a loop over rows with the zero padding inside, each wave made its ~14 round trips to HBM one
    // after the other.
    auto load_image = [&](int i) {
        const float *src = particles + (size_t)particle_of(i) * g.nx * g.nx;
        if constexpr (CROP) {
            // a box far larger than the rings (engines of the size-generic class, tcrop_wanted): the LDS holds a CROP of f.s_crop rows
            // and columns whose origin follows the particle's sampling centre, not clamped to the box (search_tiled_kernel: load_image)
            const int pk = particle_of(i);
            const Window wk = particle_window(g, state[2 * pk], state[2 * pk + 1]);
            const int ox0 = __builtin_amdgcn_readfirstlane((int)floorf((float)g.cnx + wk.sxi) - 1 - f.s_cropm);
            const int oy0 = __builtin_amdgcn_readfirstlane((int)floorf((float)g.cnx + wk.syi) - 1 - f.s_cropm);
#pragma unroll 1
            for (int y = wave; y < f.s_crop; y += RF_WAVES) {
                const int sy = oy0 + y;
                if (sy < 0 || sy >= g.nx) continue;
                const float *row = src + sy * g.nx + ox0;
                float *dst = img + y * g.pst;
#pragma unroll 1
                for (int c0 = 0; c0 < f.s_crop; c0 += 64) {
                    const int sx = ox0 + c0 + lane;
                    if (c0 + lane < f.s_crop && sx >= 0 && sx < g.nx)
                        __builtin_amdgcn_global_load_lds(row + c0 + lane, (__attribute__((address_space(3))) void *)(dst + c0), 4, 0, 0);
                }
            }
            imgb = img - g.pst - 1 - (oy0 * g.pst + ox0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        // global -> LDS without a stop in registers (global_load_lds_dword: lane l of a request writes LDS dword base + l, i.e. a
        // piece of 64 pixels of one image row).  A rolled loop: the requests carry no registers, so nothing in it waits, and the
        // kernel's code stays small -- unrolled over 8 rows at both call sites it grew by 3.9 KB and every pass slowed down by 4 %.
#pragma unroll 1
        for (int y = wave; y < g.nx; y += RF_WAVES) {
            const float *row = src + y * g.nx;
            float *dst = img + (y + g.bd) * g.pst + g.bd;
#pragma unroll 1
            for (int c0 = 0; c0 < g.nx; c0 += 64)
                if (c0 + lane < g.nx)
                    __builtin_amdgcn_global_load_lds(row + c0 + lane, (__attribute__((address_space(3))) void *)(dst + c0), 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // sampling centre of slot k of the pass that starts at (particle i0, offset s0): lanes 0 .. 3 of one wave
    auto write_centre = [&](int i0, int s0, int k) {
        int i = i0, sk = s0 + k;
        if (sk >= SPP) { sk -= SPP; i++; }
        i = min(i, npw - 1);
        const int pk = particle_of(i);
        const Window wk = particle_window(g, state[2 * pk], state[2 * pk + 1]);
        const int si = min(sk, g.nshift - 1);
        red[16 + 2 * k] = ((float)g.cnx + wk.sxi) + g.shift_x[si];
        red[17 + 2 * k] = ((float)g.cnx + wk.syi) + g.shift_y[si];
        red[7] = 0.f;
    };
    // this wave's job of round 0 is the same in every pass
    if (tid < 4 && npw > 0) write_centre(0, 0, (int)tid);          // sampling centres of the stream's first pass
    RF_LDS_BARRIER();
    const int4 jd0 = jobs_s[min((int)wave, g.n_job - 1)];
    constexpr bool defer = ONE;         // one store / inverse-FFT round per pass: its end is the arrival counter, not a barrier
    // best reference per offset of the pass that started at (i0r, s0r) (ascending reference, ">=": later wins), scaled by
    // 1/sigma; by wave 2: an old wave -- it gets through its ring job first -- that is not the one with the extra short job
    auto reduce_records = [&](int i0r, int s0r, int nl) {
        constexpr int W = sizeof(CandT) / 4;
        if (wave == 2 && lane < nl * W) {
            const int o = lane / W, wd = lane - o * W;
            float bv = pc[o * nref].val, sv = -3.0e38f; int br = 0, sr = 0;
            for (int q3 = 1; q3 < nref; q3++) {
                const float v = pc[o * nref + q3].val;
                if (v >= bv) { sv = bv; sr = br; bv = v; br = q3; }
                else if (v >= sv) { sv = v; sr = q3; }
            }
            int word = reinterpret_cast<const int *>(pc + o * nref + br)[wd];
            if (wd == 1 && sv >= bv - RA_TIE_RTOL * fabsf(bv)) word = cand_pack_runner(word, pc[o * nref + sr]);      // float tie between references
            if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * red[12 + o]);     // val, t7[]
            int io = i0r, so = s0r + o;
            if (so >= SPP) { so -= SPP; io++; }
            reinterpret_cast<int *>(cand + (size_t)particle_of(io) * g.ent_stride + so)[wd] = word;
        }
    };
    int i0 = 0, s0 = 0, img_i = -1;             // first slot of the current pass: particle index in the stream, offset; resident image
    int i0p = 0, s0p = 0, nlp = 4;              // ... and live slots of the previous pass (its records are reduced inside this one)
#pragma unroll 1
    for (int grp = 0; grp < npass; grp++) {
        const bool tl = blockIdx.x == 0 && grp < 64;      // timeline: first passes of workgroup 0
        RA_STAMP(g, tl, grp, wave, 0);
        // live slots of the pass (leading), slots of the first particle, and whether a second particle starts inside it
        const int nlive = PACK ? min(4, nslots - 4 * grp) : min(4, g.nshift - s0);
        const int a = min(4, SPP - s0);
        const bool split = PACK && a < nlive;
        const float *Bf = cls ? Bf0 + (size_t)cls[particle_of(i0)] * f.b_floats : Bf0;
        // ---- ring jobs: sampling + Normalize_ring partial sums + ring FFT of the 4 offsets (as polar_fft_kernel).
        // The spectra of the previous pass occupy the ring buffers until its inverse FFTs are done; the wait for that
        // (PassSync: an arrival counter) sits INSIDE this pass's first ring job, after the sampling (which touches only the
        // image and the tables): a wave that is through with its inverse FFTs -- or had none -- samples while the others
        // still transform.
        const bool pend = defer && grp > 0;
        if (!RA_DBG(g, 16)) {
            // one copy of the job code (the instruction cache holds 64 KB for two CUs): a pass that holds the offsets of two
            // particles runs it twice, over slots [0, a) of the resident image and -- behind the exchange of the image between
            // two barriers -- over [a, nlive)
#pragma unroll 1
            for (int ph = 0; ph < (split ? 2 : 1); ph++) {
                if (img_i != i0 + ph) {
                    // the next particle's image (one copy of the load): at the top of a pass every wave is through with the sampling
                    // of the previous pass (barrier 1 of that pass), so the image may go; inside a pass the first job round has to end
                    if (ph) RF_LDS_BARRIER();
                    load_image(i0 + ph);
                    img_i = i0 + ph;
                    RF_LDS_BARRIER();
                }
                const int slo = ph ? a : 0, shi = (split && !ph) ? a : nlive;
                const bool wait = pend && !ph;
#pragma unroll 1
                for (int jr = 0; jr * RF_WAVES < g.n_job; jr++) {
                    // jobs sorted longest first, dealt round robin: the SIMDs issue oldest-wave-first, so the first waves finish
                    // their long jobs earliest (wave timeline, scripts/fused_timeline.sh) and take the short jobs of round 1
                    const int job = jr * RF_WAVES + wave;
                    if (job >= g.n_job) continue;
                    const int4 jd = jr == 0 ? jd0 : jobs_s[job];
#ifdef RALIGN_PROFILE_SWITCHES
                    const PassSync ps = {wait && jr == 0, ifft_done, done_target, g.timeline && tl && grp < 64 ? g.timeline + (grp * 16 + wave) * 16 : nullptr};
#else
                    const PassSync ps = {wait && jr == 0, ifft_done, done_target};
#endif
                    switch (__builtin_amdgcn_readfirstlane(jd.x)) {
                    case 1: ring_job<8, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                    case 6: ring_job<16, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                    case 7: ring_job<8, 4, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                    case 9: ring_job_mix<true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                    default: break;      // the fused job table holds codes 1, 6, 7 and 9 only (make_jobs with 4 offset slots)
                    }
                }
                if (wait && wave >= g.n_job) { const PassSync ps = {true, ifft_done, done_target}; ps(); }      // a wave without a job in round 0
            }
        } else if (pend) {
            const PassSync ps = {true, ifft_done, done_target};
            ps();
        }
        if (pend) reduce_records(i0p, s0p, nlp);
        RA_STAMP(g, tl, grp, wave, 1);
        // lane roles of the contraction (see above), and its first B quad: requested here, it travels while the last ring
        // jobs finish and the barrier is crossed (the barrier does not drain global requests)
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, __float_as_int(red[7])));
        float4 b0[NRPW];
        {
            const float *bp = Bf + f.grp_boff[xm] + ln * 4;
            const int nq = f.grp_nq[xm];
#pragma unroll
            for (int h = 0; h < NRPW; h++)
                b0[h] = *reinterpret_cast<const float4 *>(bp + (RA_DBG(g, 32) ? 0 : min(rp0 + h, f.nrp - 1) * nq * 256));
        }
        // ... and so do the ring offsets of its first two quads (a static table)
        const int4 o0 = reinterpret_cast<const int4 *>(goff_s + xm * f.gstr)[0];
        const int4 o1 = reinterpret_cast<const int4 *>(goff_s + xm * f.gstr)[min(1, f.grp_nq[xm] - 1)];
        f32x4 acc[NU];
        RF_LDS_BARRIER();
        RA_STAMP(g, tl, grp, wave, 2);
        // Normalize_ring: avg = av/nn, sigma = sqrt((sq - av^2/nn)/nn).  One wave per offset (the lightest contraction role of each SIMD) reduces its ring partials
        // with a fixed butterfly (reproducible) on its way into the contraction; nobody waits for it: subtracting avg from
        // every sample only moves the DC coefficients, so the correction is applied to the contracted DC term
        // (a -= avg * sum_r n_r B_r(0), store_round) and 1/sigma to the peak record.
        const int os = wave == f.stat_wave[0] ? 0 : wave == f.stat_wave[1] ? 1 : wave == f.stat_wave[2] ? 2 : wave == f.stat_wave[3] ? 3 : -1;
        if (os >= 0) {                           // the wave with the lightest contraction role of each SIMD
            float a = 0.f, q = 0.f;
            for (int i = lane; i < g.nring; i += 64) { a += red[24 + 2 * (os * g.nring + i)]; q += red[25 + 2 * (os * g.nring + i)]; }
            a = wave_sum_dpp(a); q = wave_sum_dpp(q);
            float avg = 0.f, rsg = 1.f;
            if (g.norm_ring) {
                // avg = av / nn, 1 / sigma = 1 / sqrt((sq - av^2 / nn) / nn) with the host's 1 / nn and v_rsq_f32 (1 ulp):
                // four IEEE divisions and a square root are ~100 instructions on the wave that ends the phase
                avg = a * g.inv_nn_weight;
                rsg = __builtin_amdgcn_rsqf((q - a * avg) * g.inv_nn_weight);
            }
            if (lane == 0) { red[8 + os] = avg; red[12 + os] = rsg; }
        } else if (wave == f.ctr_wave && lane < 4 && grp + 1 < npass) {
            // sampling centres of the next pass: this pass's ring jobs are done with the current ones (rf_plan: ctr_wave)
            int in = i0, sn = s0 + 4;
            if (sn >= SPP) { sn -= SPP; in++; }
            write_centre(in, sn, (int)lane);
        }
        // ---- contraction: accumulate this wave's units over the rings that have bins of its group
        const int xb = ln >> 2, xj = ln & 3, odd = ln & 1;
        if (!RA_DBG(g, 2) && rp0 < f.nrp) {
            // the last wave of a group may hold one reference pair less: no requests or matrix instructions for a dummy pair
            constexpr int NHL = NRPW > 1 ? NRPW - 1 : 1;
            const bool lastshare = NRPW > 1 && f.nrp - rp0 == NRPW - 1;
            if (nlive > 2) {
                if (lastshare) rf_contract<NRPW, NHL, true>(g, f, bufs, goff_s, Bf, xm, rp0, ln, acc, b0, o0, o1, tl, grp, wave);
                else rf_contract<NRPW, NRPW, true>(g, f, bufs, goff_s, Bf, xm, rp0, ln, acc, b0, o0, o1, tl, grp, wave);
            } else {
                if (lastshare) rf_contract<NRPW, NHL, false>(g, f, bufs, goff_s, Bf, xm, rp0, ln, acc, b0, o0, o1, tl, grp, wave);
                else rf_contract<NRPW, NRPW, false>(g, f, bufs, goff_s, Bf, xm, rp0, ln, acc, b0, o0, o1, tl, grp, wave);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NU; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        RA_STAMP(g, tl, grp, wave, 3);
        RF_LDS_BARRIER();                       // every wave has finished reading the ring buffers
        RA_STAMP(g, tl, grp, wave, 4);
        // ---- CCF spectra -> LDS (over the ring buffers), inverse FFT, argmax: rounds of f.rz references
        if (RA_DBG(g, ~0) && tid < 4 * nref) {   // profiling builds that skip a phase still emit in-range records
            pc[tid].val = 0.f; pc[tid].jtot = 1; pc[tid].refmir = tid % nref;
            for (int k7 = 0; k7 < 7; k7++) pc[tid].t7[k7] = 0.f;
        }
        auto store_round = [&](int ref_lo, int nrz) {
            // Z_k = Q_k + i T_k and Z_{N-k} of rf_store_z for this lane's bin k of every unit; the lane's part of the addresses
            // (its bin, its offset parity, its first reference) is formed once, a unit adds a wave-uniform stride
            typedef ZLayout<N> ZL;
            const int k = 16 * xm + xb, km = k ? N - k : N / 2;          // bin 0: DC term -> slot 0, Nyquist term -> slot N/2
            const int ref_b = 2 * rp0 + (xj >> 1);
            float *zk = bufs + (odd * f.rz + ref_b - ref_lo) * ZL::kPairStride + 2 * (k + (k >> 4));
            const int dkm = 2 * (km + (km >> 4)) - 2 * (k + (k >> 4));
            // Normalize_ring mean x DC weight of every unit (bin group 0 only), read up front: under the unit's own condition
            // each of the reads is a dependent LDS round trip of its own
            float dcv[NU];
            if (xm == 0) {
                const float av0 = red[8 + odd], av1 = red[10 + odd];
#pragma unroll
                for (int h = 0; h < NRPW; h++) {
                    const float w = cdc_s[min(ref_b + 2 * h, nref - 1)];
                    dcv[2 * h] = av0 * w; dcv[2 * h + 1] = av1 * w;
                }
            }
#pragma unroll
            for (int i = 0; i < NU; i++) {
                const int h = i >> 1, op = i & 1, ref = ref_b + 2 * h, o = 2 * op + odd, rr = ref - ref_lo;
                const f32x4 c4 = acc[i];
                const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                const float r0x = swap_lane_pair(s0), r1x = swap_lane_pair(s1);
                float ca = odd ? r0x : c4[0];
                const float cb = odd ? r1x : c4[1], cc = odd ? c4[2] : r0x, cd = odd ? c4[3] : r1x;
                const bool live = ref < nref && rr >= 0 && rr < nrz && o < nlive;      // ref < nref implies a real reference pair
                float2 vk, vm;
                if (xm == 0) {                                            // wave-uniform: the group that holds bin 0
                    if (xb == 0 && live) ca -= dcv[i];                    // Normalize_ring mean: the DC term only
                    const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                    vk = xb == 0 ? make_float2(ca, ca) : make_float2(apd + bpc, cmb + amd);
                    vm = xb == 0 ? make_float2(cd, cd) : make_float2(apd - bpc, amd - cmb);
                } else {
                    const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                    vk = make_float2(apd + bpc, cmb + amd);
                    vm = make_float2(apd - bpc, amd - cmb);
                }
                if (live) {
                    float *z = zk + (2 * op * f.rz + 2 * h) * ZL::kPairStride;
                    *reinterpret_cast<float2 *>(z) = vk;
                    *reinterpret_cast<float2 *>(z + dkm) = vm;
                }
            }
        };
        auto ifft_round = [&](int ref_lo, int nrz) {
            // lane groups (sub, sub ^ 1) of a half-wave take spectrum slots zs and zs + 16: their LDS images are 32 banks apart
            const int j = ln & 15, sub = ln >> 4, uu = 2 * wave + (sub >> 1);
            // slots 32 .. 4 rz - 1: two per wave 8 - 15 (half-filled calls, 32 banks apart), or -- ifft_full, the default -- four per
            // wave 8, 9, .. (full calls with 2-way bank conflicts on their 35 LDS instructions: two calls of 309 vector instructions
            // less per pass at 10 references; the phase is bound by instruction issue, not by the LDS: 4.45 -> 4.40 ms)
            const int zs = (wave >= 8 && f.ifft_full) ? 32 + 4 * (wave - 8) + sub : (uu & 15) + 16 * (sub & 1) + 32 * (uu >> 4);
            const int o = __mul24(zs, f.rz_inv) >> 16, rr = zs - __mul24(o, f.rz);      // zs / rz, zs % rz (zs < 64, rz <= 16: exact)
            if (zs < 4 * f.rz && rr < nrz && o < nlive && !RA_DBG(g, 1))      // uniform over the 16-lane group
                ifft_argmax<N, 1, 0>(bufs, pc + (o * nref + ref_lo + rr) - zs, tws + j, zs, zs, j, ref_lo + rr, g.nomirror != 0);
        };
        if (!RA_DBG(g, 4)) {
            if constexpr (ONE) {               // the accumulators die before the inverse FFT: no register pressure from them
                store_round(0, nref);
                RA_STAMP(g, tl, grp, wave, 5);
                RF_LDS_BARRIER();
                RA_STAMP(g, tl, grp, wave, 6);
                ifft_round(0, nref);
                RA_STAMP(g, tl, grp, wave, 7);
            } else {
                for (int zr = 0; zr < f.nzr; zr++) {
                    const int ref_lo = zr * f.rz, nrz = min(f.rz, nref - ref_lo);
                    store_round(ref_lo, nrz);
                    RF_LDS_BARRIER();
                    ifft_round(ref_lo, nrz);
                    RF_LDS_BARRIER();
                }
            }
        }
        if (defer) {
            if (grp + 1 < npass) {
                // this wave's transforms (if it had any) are over: count it; the next pass's ring jobs wait for all 16 before
                // they write to the ring buffers (PassSync).  The LDS array serves requests in order, so the reads above
                // precede the increment and the increment precedes whatever a wave that has seen it writes.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_fetch_add(ifft_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                done_target += RF_WAVES;
            } else {
                RF_LDS_BARRIER();
            }
        }
        RA_STAMP(g, tl, grp, wave, 8);
        if (!defer || grp + 1 == npass) reduce_records(i0, s0, nlive);
        i0p = i0; s0p = s0; nlp = nlive;
        s0 += 4;
        if (s0 >= SPP) { s0 -= SPP; i0++; }
        // no barrier here: the next pass's ring jobs sample first (image and tables only) and wait for the counter before
        // they touch the ring buffers; the ring partials were read before the contraction barrier, the centres written after
        // the first barrier of this pass; `pc` and red[12..15] are next written two and one barriers into the next pass
    }
}

}  // namespace ralign
