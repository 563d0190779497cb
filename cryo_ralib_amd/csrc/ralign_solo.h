// Particle-resident search kernel for rings of up to 512 samples (maxrin 512: ou = 41 .. ~62 in boxes of 100 .. 136 pixels,
// e.g. the reference's own documented run, notebook/00_Multireference_Alignment.ipynb cell 3: 130 x 130, ou = 52, nref = 50).
//
// At these sizes the padded image (68 KB at 130 x 130) and the ring spectra of ONE search offset (52 - 72 KB) fill a CU's LDS, so a
// pass of this kernel handles one offset ("solo") where search_fused_kernel / search_tiled_kernel (ralign_fused.h, ralign_tiled.h)
// handle four; everything else keeps their shape:
//   ring jobs     the same wave jobs (ring_job / ring_job_mix of ralign_kernels.h) plus one for 512-sample rings (R1 = LR = 16):
//                 bilinear taps from the LDS image, in-register DFTs, LDS transposes, Normalize_ring partial sums.
//   A slice       wave role = 16-bin group (16 groups = 16 waves at maxrin 512).  After the ring jobs a wave loads ITS slice of the
//                 spectra -- its 16 bins x (Re, Im) x every ring that has them, one float per ring and lane, <= 64 VGPRs -- and the
//                 ring buffer is dead: the CCF spectra of every reference tile use its space.
//   contraction   v_mfma_f32_4x4x1_16b_f32 in one straight line per wave (A in registers, B = the prepared references streamed from
//                 L2 with scalar row offsets), tiles of 2 NH <= 10 references; rows 2 and 3 of the 4 x 4 blocks (the second offset
//                 of the four-offset kernels) idle: with one resident offset there is nothing to put there.
//   spectra       Z_k = Q_k + i T_k and Z_{N-k} to LDS, one 512-point complex inverse FFT per reference and WAVE (8 x 8 x 8 in
//                 registers with two transposes through the transform's own LDS image; ifft512_wave_argmax), argmax with the CPU
//                 scan's ">=" rules, best reference per offset with the runner-up inside AND across tiles for the float-tie
//                 re-evaluation (ralign_exact.h).
// Search offsets outside a particle's window (search_range; finalize_kernel ignores them) are skipped altogether, so no tap ever
// leaves the image and the LDS copy needs no search-range border: rows and columns 1 .. nx + 1 (the last one zero: the zero-weight
// tap of a sample that lands exactly on the last row or column).
// The end of a pass's inverse FFTs is an arrival counter (PassSync): the next pass's ring jobs sample first and wait for it before
// they write to the ring buffer; the waves with the lightest ring jobs (or none) carry the transforms.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq (test_mref_gpu_align.py:1043-1044, test_reffree_gpu_align.py:
// 844-847); replaces cu_resample_to_polar + cuFFT + cu_ccf_mult_m + CcfResultTable + cu_max_idx_batch
// (cuda/gpu_aln_noref.cu:818-879, 1009-1143, 2095-2206, 1289-1346) for this geometry class.
#pragma once

#include <type_traits>

#include "ralign_tiled.h"

namespace ralign {

constexpr int RS_NQ = 16;          // ring quads of a wave's A slice (<= 64 rings)
constexpr int RS_MAXNH = 5;        // reference pairs per tile
constexpr int RS_GSTR = 4 * RS_NQ; // ints per group in the LDS table of ring offsets

// LDS plan of search_solo_kernel (floats); every region starts on a multiple of 4 floats
struct SoloLds {
    int img, bufs, tw, qt, inst, jobs, instw, red, tws, pc, goff, cdc, total;
};
inline int rs_up4(int v) { return (v + 3) & ~3; }
inline SoloLds solo_lds_plan(int N, int rows, int pst, int sbuf, int n_qtab, int n_inst, int n_job, int nring, int nref)
{
    SoloLds L;
    int o = 0;
    L.img = o; o += rs_up4(rows * pst);
    L.bufs = o; o += rs_up4(sbuf);
    L.tw = o; o += 2 * N;
    L.qt = o; o += rs_up4(2 * n_qtab);
    L.inst = o; o += 4 * n_inst;
    L.jobs = o; o += 4 * n_job;
    L.instw = o; o += rs_up4(n_inst);
    L.red = o; o += rs_up4(24 + 2 * nring);
    L.tws = o; o += 2 * N + 2 * 64;                                            // ifft512_twiddles: [8][64] + [8][8]
    L.pc = o; o += rs_up4((2 * RS_MAXNH + 4) * (int)(sizeof(CandT) / 4));      // [2 RS_MAXNH] records of a tile, [2] best / runner-up over the tiles
    L.goff = o; o += 16 * RS_GSTR;
    L.cdc = o; o += rs_up4(nref);
    L.total = o;
    return L;
}

// LDS image of the solo / duo / pair kernels.  The rings of every in-window offset of a particle touch the pixels within
// ou + (largest search shift) + 1 of its sampling centre only, so a box much larger than the rings (256 x 256 around ou = 40) need
// not be resident: the image is a CROP of 2 (ou + S) + 5 columns and rows whose origin follows the particle's centre (clamped to
// the box), plus the never-written zero row / column the tap of weight 0 at the box's edge may read.  Returns the resident columns.
inline int crop_plan(const Geometry &g, FusedGeom &f)
{
    const int S = (int)std::ceil(std::max(g.nkx, g.nky) * g.step - 1e-6);
    const int side = 2 * (S + g.last_ring) + 5;
    const bool crop = side < g.nx && !(getenv("RALIGN_CROP") && atoi(getenv("RALIGN_CROP")) == 0);
    f.s_crop = crop ? side : 0;
    f.s_cropm = S + g.last_ring + 1;
    const int cols = crop ? side : g.nx;
    f.s_rows = cols + 1;
    f.s_pst = cols + 1;
    while (!((f.s_pst & 1) && ((f.s_pst - 1) & 7) && ((f.s_pst + 1) & 7))) f.s_pst++;
    return cols;
}

// first column / row (0-based) of a particle's crop: its sampling centre (1-based, float) minus s_cropm, inside the box
__device__ __forceinline__ int crop_origin(const FusedGeom &f, int nx, float centre)
{
    return min(max((int)floorf(centre) - 1 - f.s_cropm, 0), nx - f.s_crop);
}

// n_qtab, n_inst, n_job: sizes of the job tables build_device_geometry made for ONE offset slot
inline bool build_solo_plan(const Geometry &g, int nref, int n_qtab, int n_inst, int n_job, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.bsrc.clear(); out.cdc_w.clear();
    if (g.maxrin != 512 || g.numr[2] < 8 || g.nring > 4 * RS_NQ || nref > 127 || nref < 1) return false;
    f.ng = g.maxrin / 32; f.wpg = 1;
    f.nrp = (nref + 1) / 2;
    f.ntile = (f.nrp + RS_MAXNH - 1) / RS_MAXNH;
    f.nh = (f.nrp + f.ntile - 1) / f.ntile;
    f.nrpw = f.nh;
    f.rz = 2 * f.nh; f.nzr = f.ntile;
    f.rz_inv = (65536 + f.rz - 1) / f.rz;
    rf_layout_b(g, nref, f, out.bsrc);
    // image without a search-range border; the row stride keeps vertical and diagonal neighbours out of one LDS bank
    // (as build_device_geometry does for the bordered image)
    crop_plan(g, f);
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    f.s_sbuf = std::max((g.lring + 31) / 32 * 32 + 16, f.rz * zstride);
    // ring jobs: longest first, wave w runs jobs rank[w], rank[w] + 16, ..; the waves with the highest ranks have no job in a
    // geometry with fewer than 16 jobs; the transforms of a tile (one per wave) go to the highest ranks.  RALIGN_SOLO_ORDER=1 gives
    // the jobs to the youngest waves and the transforms to the oldest (the SIMDs issue oldest-first).
    const bool rev = RA_EXP_ENV("RALIGN_SOLO_ORDER") && ra_atoi(RA_EXP_ENV("RALIGN_SOLO_ORDER")) == 1;
    for (int w = 0; w < 16; w++) {
        f.s_rank[w] = rev ? 15 - w : w;
        const int c = 15 - f.s_rank[w];
        f.s_call[w] = c < f.rz ? c : -1;          // one 512-point transform per wave and tile (ifft512_wave_argmax)
    }
    // wave of rank r takes bin group r: the groups of the high bins (only the longest rings reach them: few ring quads) go to
    // the waves that carry the transforms
    for (int w = 0; w < 16; w++) f.wmap[w] = f.s_rank[w];
    auto wave_of_rank = [&](int r) { for (int w = 0; w < 16; w++) if (f.s_rank[w] == r) return w; return 0; };
    f.s_rec = wave_of_rank(15);       // merges the records behind the counter wait: a wave without a ring job
    f.s_stat = wave_of_rank(14);
    f.s_ctr = wave_of_rank(13);
    out.cdc_w.assign(g.nring, 0.f);
    for (int r = 0; r < 68; r++) f.roff[r] = g.ring_off[std::min(r, g.nring - 1)];
    f.gstr = RS_GSTR;
    const SoloLds L = solo_lds_plan(g.maxrin, f.s_rows, f.s_pst, f.s_sbuf, n_qtab, n_inst, n_job, g.nring, nref);
    out.lds_bytes = (size_t)L.total * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024;
    return f.on != 0;
}

// contraction of one chunk of a tile for one wave (see rt_contract, ralign_tiled.h): NH reference pairs of the wave's bin group
// against the A slice in registers, NS ring-quad slots in one straight line; the slice is right-aligned in a[]
template <int NH, int NS, int NQT>
__device__ __forceinline__ void rs_contract(const float (&a)[4 * NQT], __amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                            const unsigned (&row)[NH], unsigned pad0, f32x4 (&acc)[NH])
{
    constexpr int A0 = 4 * (NQT - NS);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // one operand buffer per reference pair, refilled for the next slot right after its four multiplies.  (Two buffers -- the rows of
    // slot sl + 2 requested while slot sl + 1 multiplies -- cost 12 more registers: 1 - 4 % slower on both maxrin-512 workloads with
    // the spills that brought, scripts/dev/pf_runs.sh.)
    float4 bc[NH];
    // pad0 != 0: slot 0 is padding (zero A); its B requests are sent out of the buffer's range and return zeros without a fetch
#pragma unroll
    for (int h = 0; h < NH; h++) bc[h] = rt_load_b(rsrc, voff, row[h] | pad0);
#pragma unroll
    for (int sl = 0; sl < NS; sl++) {
#pragma unroll
        for (int h0 = 0; h0 < NH; h0 += 2) {
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int h = h0; h < h0 + 2 && h < NH; h++)
                    acc[h] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[A0 + 4 * sl + c], rf_f4(bc[h], c), (sl == 0 && c == 0) ? zero : acc[h], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (sl + 1 < NS) {
#pragma unroll
                for (int h = h0; h < h0 + 2 && h < NH; h++) bc[h] = rt_load_b(rsrc, voff, row[h] + (sl + 1) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// dbg_spec != null (tests: the polar stage bin for bin): the ring buffer after the ring jobs and {avg, 1 / sigma} of every
// (particle, in-window offset) go to dbg_spec[(p nshift + s) (lring + 2)]; nothing else runs
// ONE: a single reference tile (nref <= 2 NH): the slice is dead before the inverse FFTs
template <int N, int NH, bool ONE>
__global__ __launch_bounds__(RF_THREADS) void search_solo_kernel(DevGeom g_in, FusedGeom f, const float *__restrict__ particles,
                                                                 const float *__restrict__ state, int n,
                                                                 const float *__restrict__ Bf, int nref,
                                                                 CandT *__restrict__ cand, float *__restrict__ dbg_spec)
{
    DevGeom g = g_in;
    g.maxrin = N; g.lg_maxrin = __builtin_ctz(N);
    g.sbuf = f.s_sbuf; g.pst = f.s_pst;
    constexpr int RZ = 2 * NH;                                         // references per tile
    extern __shared__ __align__(16) float lds[];
    const int o_bufs = (f.s_rows * f.s_pst + 3) & ~3;
    float *img = lds;
    float *bufs = lds + o_bufs;                                        // ring buffer | CCF spectra of a tile
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + ((f.s_sbuf + 3) & ~3));
    float2 *qt_s = tw_s + N;
    int4 *inst_s = reinterpret_cast<int4 *>(reinterpret_cast<float *>(qt_s) + ((2 * g.n_qtab + 3) & ~3));
    int4 *jobs_s = inst_s + g.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job + g.n_job_b);
    float *red = instw_s + ((g.n_inst + 3) & ~3);      // [6] counter, [7] zero, [8] avg, [12] 1 / sigma, [16] centre, [24 ..] ring partials
    float2 *tws = reinterpret_cast<float2 *>(red + ((24 + 2 * g.nring + 3) & ~3));      // inverse-FFT twiddles: [8][64], then [8][8]
    CandT *pc = reinterpret_cast<CandT *>(tws + N + 64);               // [RZ] records of the tile
    CandT *pbest = pc + 2 * RS_MAXNH;                                  // [2] best record and runner-up over the tiles so far
    int *goff_s = reinterpret_cast<int *>(reinterpret_cast<float *>(pc) + (((2 * RS_MAXNH + 4) * (int)(sizeof(CandT) / 4) + 3) & ~3));
    float *cdc_s = reinterpret_cast<float *>(goff_s + 16 * RS_GSTR);   // [nref] DC weights of the references
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < N; i += RF_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RF_THREADS) qt_s[i] = g.qtab[i];
    for (int i = tid; i < g.n_inst; i += RF_THREADS) { inst_s[i] = g.inst[i]; instw_s[i] = g.instw[i]; }
    for (int i = tid; i < g.n_job + g.n_job_b; i += RF_THREADS) jobs_s[i] = g.jobs[i];
    static_assert(N == 512, "ifft512_wave_argmax");
    ifft512_twiddles(g.tw, tws, tws + N, tid, RF_THREADS);
    for (int i = tid; i < f.s_sbuf; i += RF_THREADS) bufs[i] = 0.f;       // slack between rings must hold finite values
    for (int i = tid; i < o_bufs; i += RF_THREADS) img[i] = 0.f;          // row and column nx + 1 stay zero
    for (int i = tid; i < nref; i += RF_THREADS) cdc_s[i] = f.cdc_w[i];
    for (int i = tid; i < 16 * RS_GSTR; i += RF_THREADS) {
        const int m = i / RS_GSTR, j = i - m * RS_GSTR;
        goff_s[i] = 4 * f.roff[min(f.grp_ring0[m] + j, g.nring - 1)];
    }
    // 1-based coordinates (ix, iy) -> img[(iy - 1) pst + ix - 1]
    const float *imgb = img - g.pst - 1;          // (re-based per particle when the image is a crop)
    int *ifft_done = reinterpret_cast<int *>(red + 6);
    if (tid == 0) { *ifft_done = 0; red[7] = 0.f; }
    int done_target = 0;

    const int xm = f.wmap[wave];                                       // this wave's bin group
    const int nq = f.grp_nq[xm];
    const int rank = f.s_rank[wave], call = f.s_call[wave];
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bf), 0, (f.b_floats + 256) * 4, 0x00020000);
    const int ntile = f.ntile, nx1 = 2 * g.nkx + 1;
    const int4 jd0 = g.jobs[min(rank, g.n_job - 1)];

    bool pend = false;                     // a pass whose last inverse FFTs and records are outstanding
    int p_prev = 0, s_prev = 0;
    int ipass = 0;                         // passes of this workgroup so far (profiling builds: the wave timeline covers the first 64)
    // records of tile t (ascending reference, ">=": later wins) against the best of the earlier tiles (a later tile wins ties, as a
    // later reference does).  The runner-up -- the largest peak among the references that lost, inside this tile or in an earlier
    // one -- travels along; the last tile of a pass scales by 1 / sigma and writes the record out, with the runner-up in spare bits
    // of the jtot word when it lies within RA_TIE_RTOL of the winner (finalize_kernel hands both to refine_winner_kernel)
    auto merge_records = [&](int t, bool last, int pw, int sw) {
        constexpr int W = sizeof(CandT) / 4;
        if (wave != f.s_rec) return;
        const int wd = rf_own_lane(lane);
        if (wd < W) {
            const int nrz = min(RZ, nref - t * RZ);
            float bv = pc[0].val, sv = -3.0e38f; int br = 0, sr = 0;
            for (int q3 = 1; q3 < nrz; q3++) {
                const float v = pc[q3].val;
                if (v >= bv) { sv = bv; sr = br; bv = v; br = q3; }
                else if (v >= sv) { sv = v; sr = q3; }
            }
            const CandT *win = pc + br, *run = pc + sr;
            float lv = sv;
            if (t > 0) {
                const float pv = pbest[0].val, l2 = pbest[1].val;
                if (!(bv >= pv)) {             // the best of the earlier tiles stays
                    if (bv >= l2) { run = pc + br; lv = bv; } else { run = pbest + 1; lv = l2; }
                    win = pbest; bv = pv;
                } else if (pv >= lv && pv >= l2) { run = pbest; lv = pv; }
                else if (l2 > lv) { run = pbest + 1; lv = l2; }
            }
            int word = reinterpret_cast<const int *>(win)[wd];
            int rword = reinterpret_cast<const int *>(run)[wd];
            if (last) {
                if (wd == 1 && lv >= bv - RA_TIE_RTOL * fabsf(bv)) word = cand_pack_runner(cand_jtot(word), *run);
                if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * red[12]);     // val, t7[]
                reinterpret_cast<int *>(cand + (size_t)pw * g.ent_stride + sw)[wd] = word;
            } else {
                if (wd == 0) rword = __float_as_int(lv);
                reinterpret_cast<int *>(pbest)[wd] = word;
                reinterpret_cast<int *>(pbest + 1)[wd] = rword;
            }
        }
    };

    // One pass behind the ring jobs: this wave's slice of the spectra -- bins 16 xm .. 16 xm + 15, every ring that has them, right-
    // aligned in a[4 RS_NQ] -- and the tiles of RZ references (contraction, spectra store, inverse FFTs)
    auto tile_loop = [&](int ln, int p, int s, bool tl) {
        constexpr int NQT = RS_NQ;
        const int xb = ln >> 2, xj = ln & 3, odd = ln & 1;
        float a[4 * NQT];
        {
            const char *abase = reinterpret_cast<const char *>(bufs + 2 * (16 * xm + xb) + (xj & 1));
            const int4 *gq = reinterpret_cast<const int4 *>(goff_s + xm * RS_GSTR);
#pragma unroll
            for (int sl = 0; sl < NQT; sl++) {
                if (sl >= NQT - nq) {
                    const int4 o = gq[sl - (NQT - nq)];
                    a[4 * sl] = *reinterpret_cast<const float *>(abase + o.x); a[4 * sl + 1] = *reinterpret_cast<const float *>(abase + o.y);
                    a[4 * sl + 2] = *reinterpret_cast<const float *>(abase + o.z); a[4 * sl + 3] = *reinterpret_cast<const float *>(abase + o.w);
                } else {
                    a[4 * sl] = a[4 * sl + 1] = a[4 * sl + 2] = a[4 * sl + 3] = 0.f;
                }
            }
        }
        RA_STAMP(g, tl, ipass, wave, 3);
#pragma unroll 1
        for (int t = 0; t < (ONE ? 1 : ntile); t++) {
            const int ref_lo = t * RZ, nrz = min(RZ, nref - ref_lo);
            // the tile's reference pairs in chunks of at most three: the slice, 12 accumulator and 12 operand registers fit a
            // wave's 128 next to the addresses, five pairs at once (104 + addresses) spilled 43 registers into the contraction
            constexpr int NC = NH > 3 ? 2 : 1, NH0 = NH > 3 ? (NH + 1) / 2 : NH, NH1 = NH - NH0;
            const int ns = (nq + 1) & ~1;          // straight-line instantiations for even slot counts; an odd group pads its first slot
            const unsigned voff = (unsigned)ln * 16u;
            auto chunk = [&](auto nhc_c, int h0) {
                constexpr int NHC = decltype(nhc_c)::value;
                f32x4 acc[NHC];
                {
                    // row offsets (bytes) of the reference pairs in the wave's group block, moved back by the slot the wave pads (its
                    // requests go out of the buffer's range: zeros, no memory traffic); a pair past the last one (final tile)
                    // re-reads the last pair's rows and is never stored
                    unsigned row[NHC];
                    const unsigned pad = ns > nq ? 0x80000000u : 0u;
#pragma unroll
                    for (int h = 0; h < NHC; h++)
                        row[h] = (unsigned)(f.grp_boff[xm] + min(t * NH + h0 + h, f.nrp - 1) * nq * 256) * 4u - (unsigned)((ns - nq) * 1024);
                    switch (ns) {
                    case 2: rs_contract<NHC, 2, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 4: rs_contract<NHC, 4, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 6: rs_contract<NHC, 6, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 8: rs_contract<NHC, 8, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 10: rs_contract<NHC, 10, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 12: rs_contract<NHC, 12, NQT>(a, brsrc, voff, row, pad, acc); break;
                    case 14: rs_contract<NHC, 14, NQT>(a, brsrc, voff, row, pad, acc); break;
                    default: rs_contract<NHC, 16, NQT>(a, brsrc, voff, row, pad, acc); break;
                    }
                }
                if (h0 == 0) {
                    RA_STAMP(g, tl && t == 0, ipass, wave, 4);
                    RF_LDS_BARRIER();         // t = 0: every slice is in registers; t > 0: the inverse FFTs of tile t - 1 are over
                    RA_STAMP(g, tl && t == 0, ipass, wave, 5);
                    if (t > 0) merge_records(t - 1, false, p, s);
                }
                // Z_k = Q_k + i T_k (even lane) and Z_{N-k} = conj Q_k + i conj T_k (odd lane) of this lane pair's bin for every
                // reference pair of the chunk, from the four products a = c1 d1, b = c1 d2 (even lane: the reference's real part
                // times the image's (Re, Im)), c = c2 d1, d = c2 d2 (odd lane) summed over the rings (Util::Crosrng_ms:
                // Q = (a + d) + i (c - b), T = (a - d) - i (b + c)); bin 0 carries the DC term in a and the Nyquist term of the
                // full-length rings in d
                typedef ZLayout<N> ZL;
                const int k = 16 * xm + xb, km = k ? N - k : N / 2;
                const int ks = odd ? km : k;
                const int ref_b = ref_lo + 2 * h0 + (xj >> 1);
                float *zk = bufs + (2 * h0 + (xj >> 1)) * ZL::kPairStride + 2 * (ks + (ks >> 4));
                float dcv[NHC];
                if (xm == 0) {
                    const float av = red[8];
#pragma unroll
                    for (int h = 0; h < NHC; h++) dcv[h] = av * cdc_s[min(ref_b + 2 * h, nref - 1)];
                }
#pragma unroll
                for (int h = 0; h < NHC; h++) {
                    const int ref = ref_b + 2 * h;
                    const f32x4 c4 = acc[h];
                    const float x0 = swap_lane_pair(c4[0]), x1 = swap_lane_pair(c4[1]);
                    float ca = odd ? x0 : c4[0];
                    const float cb = odd ? x1 : c4[1], cc = odd ? c4[0] : x0, cd = odd ? c4[1] : x1;
                    float2 v;
                    if (xm == 0) {
                        if (xb == 0) ca -= dcv[h];                       // Normalize_ring mean: the DC term only
                        const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                        v = xb == 0 ? (odd ? make_float2(cd, cd) : make_float2(ca, ca))
                                    : (odd ? make_float2(apd - bpc, amd - cmb) : make_float2(apd + bpc, cmb + amd));
                    } else {
                        const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                        v = odd ? make_float2(apd - bpc, amd - cmb) : make_float2(apd + bpc, cmb + amd);
                    }
                    if (ref < nref) *reinterpret_cast<float2 *>(zk + 2 * h * ZL::kPairStride) = v;
                }
            };
            chunk(std::integral_constant<int, NH0>{}, 0);
            if constexpr (NC > 1) chunk(std::integral_constant<int, NH1>{}, NH0);
            RA_STAMP(g, tl && t == 0, ipass, wave, 6);
            RF_LDS_BARRIER();         // the spectra of the tile are complete
            RA_STAMP(g, tl && t == 0, ipass, wave, 7);
            if (call >= 0 && call < nrz) ifft512_wave_argmax(bufs, pc + call, tws, tws + N, call, rf_own_lane(ln), ref_lo + call, g.nomirror != 0);
            RA_STAMP(g, tl && t == 0, ipass, wave, 8);
        }
    };

#pragma unroll 1
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
        // every wave is past the barrier behind the ring jobs of the previous particle's last pass: its image may go.
        // Pixels only (the zero row / column were written once, above), global -> LDS without a stop in registers
        Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
        {
            const int side = f.s_crop ? f.s_crop : g.nx;
            const int ox0 = f.s_crop ? __builtin_amdgcn_readfirstlane(crop_origin(f, g.nx, (float)g.cnx + w.sxi)) : 0;
            const int oy0 = f.s_crop ? __builtin_amdgcn_readfirstlane(crop_origin(f, g.nx, (float)g.cnx + w.syi)) : 0;
            const float *src = particles + (size_t)p * g.nx * g.nx + oy0 * g.nx + ox0;
#pragma unroll 1
            for (int y = wave; y < side; y += RF_WAVES) {
                const float *row = src + y * g.nx;
                float *dst = img + y * g.pst;
#pragma unroll 1
                for (int c0 = 0; c0 < side; c0 += 64)
                    if (c0 + lane < side)
                        __builtin_amdgcn_global_load_lds(row + c0 + lane, (__attribute__((address_space(3))) void *)(dst + c0), 4, 0, 0);
            }
            imgb = img - g.pst - 1 - (oy0 * g.pst + ox0);      // 1-based (ix, iy) of the BOX -> img[(iy - 1 - oy0) pst + ix - 1 - ox0]
        }
        // wave-uniform values: keep them in scalar registers
        w.lkx = __builtin_amdgcn_readfirstlane(w.lkx); w.rkx = __builtin_amdgcn_readfirstlane(w.rkx);
        w.lky = __builtin_amdgcn_readfirstlane(w.lky); w.rky = __builtin_amdgcn_readfirstlane(w.rky);
        const float cxf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)g.cnx + w.sxi)));
        const float cyf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)g.cnx + w.syi)));
        auto in_window = [&](int s) {
            const int iy = s / nx1 - g.nky, ix = s - (s / nx1) * nx1 - g.nkx;
            return ix >= -w.lkx && ix <= w.rkx && iy >= -w.lky && iy <= w.rky;
        };
        auto next_live = [&](int s) { while (s < g.nshift && !in_window(s)) s++; return s; };
        int s = next_live(0);
        if (wave == f.s_ctr && lane == 0 && s < g.nshift) {
            red[16] = cxf + g.shift_x[s];
            red[17] = cyf + g.shift_y[s];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RF_LDS_BARRIER();
#pragma unroll 1
        for (; s < g.nshift; ) {
            const int sn = next_live(s + 1);
            // profiling builds: wave timeline of workgroup 0 (stamps: 0 pass start, 13 / 14 around the counter wait inside the ring
            // job, 1 ring jobs done, 2 behind their barrier, 3 slice in registers, 4 contraction of tile 0 done, 5 behind barrier A,
            // 6 spectra stored, 7 behind barrier B, 8 transforms of tile 0 done, 9 end of the pass)
            const bool tl = blockIdx.x == 0 && ipass < 64;
            RA_STAMP(g, tl, ipass, wave, 0);
            // ---- ring jobs: sampling + Normalize_ring partial sums + ring FFTs of this offset.  The previous pass's last inverse
            // FFTs are awaited inside the job, between its sampling and its first write to the ring buffer
#pragma unroll 1
            for (int jr = 0; jr * RF_WAVES < g.n_job; jr++) {
                const int job = jr * RF_WAVES + rank;
                if (job >= g.n_job) continue;
                const int4 jd = jr == 0 ? jd0 : jobs_s[job];
#ifdef RALIGN_PROFILE_SWITCHES
                const PassSync ps = {pend && jr == 0, ifft_done, done_target, g.timeline && tl ? g.timeline + (ipass * 16 + wave) * 16 : nullptr};
#else
                const PassSync ps = {pend && jr == 0, ifft_done, done_target};
#endif
                switch (__builtin_amdgcn_readfirstlane(jd.x)) {
                case 11: ring_job512<true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 0: ring_job<8, 16, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 10: ring_job<16, 16, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 6: ring_job<16, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 1: ring_job<8, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 7: ring_job<8, 4, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                case 9: ring_job_mix<true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
                default: break;
                }
            }
            if (pend && rank >= g.n_job) {
#ifdef RALIGN_PROFILE_SWITCHES
                const PassSync ps = {true, ifft_done, done_target, nullptr};
#else
                const PassSync ps = {true, ifft_done, done_target};
#endif
                ps();
            }
            if (pend) merge_records(ntile - 1, true, p_prev, s_prev);
            RA_STAMP(g, tl, ipass, wave, 1);
            RF_LDS_BARRIER();
            RA_STAMP(g, tl, ipass, wave, 2);
            const int ln = rf_own_lane(lane);         // nothing derived from the lane index lives through the ring jobs
            // Normalize_ring statistics of the offset (fixed order: reproducible) and the next pass's sampling centre
            if (wave == f.s_stat) {
                float a = 0.f, q = 0.f;
                for (int i = ln; i < g.nring; i += 64) { a += red[24 + 2 * i]; q += red[25 + 2 * i]; }
                a = wave_sum_dpp(a); q = wave_sum_dpp(q);
                float avg = 0.f, rsg = 1.f;
                if (g.norm_ring) {
                    avg = a * g.inv_nn_weight;
                    rsg = __builtin_amdgcn_rsqf((q - a * avg) * g.inv_nn_weight);
                }
                if (ln == 0) { red[8] = avg; red[12] = rsg; }
            }
            if (wave == f.s_ctr && ln == 0 && sn < g.nshift) {
                red[16] = cxf + g.shift_x[sn];
                red[17] = cyf + g.shift_y[sn];
            }
            if (dbg_spec) {           // tests only
                RF_LDS_BARRIER();
                float *dst = dbg_spec + ((size_t)p * g.nshift + s) * (g.lring + 2);
                for (int i = tid; i < g.lring; i += RF_THREADS) dst[i] = bufs[i];
                if (tid == 0) { dst[g.lring] = red[8]; dst[g.lring + 1] = red[12]; }
                RF_LDS_BARRIER();
                s = sn;
                continue;
            }
            // ---- this wave's slice of the spectra and the reference tiles
            tile_loop(ln, p, s, tl);
            RA_STAMP(g, tl, ipass, wave, 9);
            ipass++;
            // this wave's transforms (if it had any) are over: count it; the next pass's ring jobs wait for all 16 before they write
            // to the ring buffer (PassSync)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_fetch_add(ifft_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            done_target += RF_WAVES;
            pend = true; p_prev = p; s_prev = s;
            s = sn;
        }
    }
    if (pend) {
#ifdef RALIGN_PROFILE_SWITCHES
        const PassSync ps = {true, ifft_done, done_target, nullptr};
#else
        const PassSync ps = {true, ifft_done, done_target};
#endif
        ps();
        merge_records(ntile - 1, true, p_prev, s_prev);
    }
}

// dbg_spec of search_solo_kernel -> ring spectra in EMAN2 packing [n][nshift][lcirc] with Normalize_ring applied (what Frngs leaves
// in `cimage` inside Util.multiref_polar_ali_2d); offsets outside a particle's window were not computed and read zero
__global__ void unpack_solo_spectra_kernel(DevGeom g, const float *__restrict__ raw, int n, const int *__restrict__ numr,
                                           const int *__restrict__ ring_off, float *__restrict__ out)
{
    const int m = blockIdx.x;                  // particle * nshift + shift
    if (m >= n * g.nshift) return;
    const float *src = raw + (size_t)m * (g.lring + 2);
    const float avg = src[g.lring], rsg = src[g.lring + 1];
    float *dst = out + (size_t)m * g.lcirc;
    for (int i = 0; i < g.nring; i++) {
        const int nlen = numr[3 * i + 2], o = numr[3 * i + 1] - 1, ro = ring_off[i];
        for (int j = threadIdx.x; j < nlen; j += blockDim.x) {
            // packed slot j: 0 -> Re X0, 1 -> Re X(n/2) (full-length rings: the imaginary slot of bin 0), 2k -> Re Xk, 2k+1 -> Im Xk
            float v;
            if (j == 0) v = src[ro] - avg * (float)nlen;
            else if (j == 1) v = nlen == g.maxrin ? src[ro + 1] : src[ro + nlen];
            else v = src[ro + j];
            dst[o + j] = v * rsg;
        }
    }
}

}  // namespace ralign
