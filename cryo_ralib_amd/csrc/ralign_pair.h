// search_pair_kernel: the particle-resident search for rings of up to 256 samples in boxes that are too large for the four ring
// buffers of search_fused_kernel / search_tiled_kernel (ralign_fused.h, ralign_tiled.h) -- e.g. 100 x 100 or 128 x 128 pixels at
// ou = 37 .. 40, where the bordered image (48 - 76 KB) plus 4 x 30 KB of ring buffers exceed a CU's 160 KB and the search fell to the
// size-generic kernels (213 k particles/s at 100 x 100 / ou = 40 / nref = 10 against 1.64 M at 90 x 90 / ou = 36).
//
// TWO search offsets per pass in two LDS ring buffers; otherwise the structure of search_tiled_kernel and of the kernels for rings
// of 512 samples (ralign_solo.h, ralign_duo.h), whose pieces it shares:
//   image         without a search-range border: offsets outside a particle's window are skipped (search_solo_kernel)
//   ring jobs     the wave jobs of ralign_kernels.h (codes 6, 1, 7, 9) for both offsets in one round
//   A slice       wave role = (16-bin group, half of the tile's reference pairs): 8 groups x 2 = 16 waves; a wave loads its group's
//                 slice of both offsets' spectra (block rows 0, 1 <- offset A, rows 2, 3 <- offset B), <= 40 registers
//   contraction   v_mfma_f32_4x4x1 in one straight line per wave, full blocks, B from L2 with scalar row offsets (rs_contract)
//   spectra       4 NHW references x 2 offsets per tile in the ring-buffer space, 256-point inverse FFTs on 16 lanes each
//                 (ifft_argmax), best reference per offset with the runner-up inside and across tiles
// A particle's odd last in-window offset runs as a pass of one.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq (test_mref_gpu_align.py:1043-1044, test_reffree_gpu_align.py:
// 844-847); replaces cuda/gpu_aln_noref.cu:818-879, 1009-1143, 2095-2206, 1289-1346 for this geometry class.
#pragma once

#include "ralign_solo.h"

namespace ralign {

constexpr int RP_NQ = 12;          // ring quads of a wave's A slice (<= 48 rings)
constexpr int RP_MAXNHW = 3;       // reference pairs per wave and tile: a tile holds 4 NHW references
constexpr int RP_GSTR = 4 * RP_NQ; // ints per group in the LDS table of ring offsets
constexpr int RP_MAXRZ = 4 * RP_MAXNHW;

inline SoloLds pair_lds_plan(int N, int rows, int pst, int sbuf, int n_qtab, int n_inst, int n_job, int nring, int nref)
{
    SoloLds L;
    int o = 0;
    L.img = o; o += rs_up4(rows * pst);
    L.bufs = o; o += rs_up4(2 * sbuf);
    L.tw = o; o += 2 * N;
    L.qt = o; o += rs_up4(2 * n_qtab);
    L.inst = o; o += 4 * n_inst;
    L.jobs = o; o += 4 * n_job;
    L.instw = o; o += rs_up4(n_inst);
    L.red = o; o += rs_up4(24 + 4 * nring);
    L.tws = o; o += 2 * N;                                                     // [R1][R2] inverse-FFT twiddles
    L.pc = o; o += rs_up4((2 * RP_MAXRZ + 4) * (int)(sizeof(CandT) / 4));       // [2][RZ] records of a tile, [2][2] best / runner-up
    L.goff = o; o += 8 * RP_GSTR;
    L.cdc = o; o += rs_up4(nref);
    L.total = o;
    return L;
}

// n_qtab, n_inst, n_job: sizes of the job tables build_device_geometry made for TWO offset slots
inline bool build_pair_plan(const Geometry &g, int nref, int n_qtab, int n_inst, int n_job, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.bsrc.clear(); out.cdc_w.clear();
    if (g.maxrin != 256 || g.numr[2] < 8 || g.nring > 4 * RP_NQ || nref > 127 || nref < 1) return false;
    f.ng = g.maxrin / 32; f.wpg = 2;
    f.nrp = (nref + 1) / 2;
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    crop_plan(g, f);
    // references per tile: 2 offsets x RZ spectra in the two ring buffers, which grow past the rings' own length when the LDS has the
    // room (ou 36: 12 spectra need 6552 floats, the rings 5.7 k -- without the slack 96 x 96 / ou 36 fell back to the generic kernels)
    // row stride of a crop: 101 words (the 90 x 90 headline geometry's, as for the engines that run its kernels over a crop: 35.9 ->
    // 35.3 ms per 32 768 at 100 x 100 / ou = 40) when the plan fits with it, else the narrowest conflict-poor one
    const int pst_narrow = f.s_pst;
    const bool try_wide = f.s_crop && f.s_pst < 101 && !(RA_EXP_ENV("RALIGN_CROP_PST101") && ra_atoi(RA_EXP_ENV("RALIGN_CROP_PST101")) == 0);
    bool found = false;
    auto plan_tiles = [&]() {
        for (int nhmax = RP_MAXRZ / 2; nhmax >= 1 && !found; nhmax--) {

            f.ntile = (f.nrp + nhmax - 1) / nhmax;
            f.nh = (f.nrp + f.ntile - 1) / f.ntile;
            f.nrpw = (f.nh + 1) / 2;                 // pairs per wave (template parameter): the two waves of a group share a tile's pairs
            f.nh = 2 * f.nrpw;
            f.rz = 2 * f.nh; f.nzr = f.ntile;
            if (f.rz > RP_MAXRZ || f.nrpw > RP_MAXNHW) continue;
            f.s_sbuf = (std::max(g.lring, f.rz * zstride) + 31) / 32 * 32 + 16;
            const SoloLds L = pair_lds_plan(g.maxrin, f.s_rows, f.s_pst, f.s_sbuf, n_qtab, n_inst, n_job, g.nring, nref);
            found = (size_t)L.total * sizeof(float) <= 160 * 1024;
        }
    };
    plan_tiles();
    if (found && try_wide) {          // the wide stride must not cost a tile
        const int ntile_narrow = f.ntile;
        f.s_pst = 101; found = false;
        plan_tiles();
        if (!found || f.ntile > ntile_narrow) { f.s_pst = pst_narrow; found = false; plan_tiles(); }
    }
    if (!found) return false;
    rf_layout_b(g, nref, f, out.bsrc);
    // wave w: ring jobs rank w (longest first); bin group w & 7, share w >> 3 of the tile's pairs; the inverse-FFT calls of a tile
    // (4 transforms each) from the highest rank down
    const int ncall = (2 * f.rz + 3) / 4;
    for (int w = 0; w < 16; w++) {
        f.s_rank[w] = w;
        f.wmap[w] = (w & 7) | ((w >> 3) << 8);
        const int c = 15 - w;
        f.s_call[w] = c < ncall ? c : -1;
    }
    f.s_rec = 15; f.s_stat = 14; f.s_ctr = 13;
    out.cdc_w.assign(g.nring, 0.f);
    for (int r = 0; r < 68; r++) f.roff[r] = g.ring_off[std::min(r, g.nring - 1)];
    f.gstr = RP_GSTR;
    const SoloLds L = pair_lds_plan(g.maxrin, f.s_rows, f.s_pst, f.s_sbuf, n_qtab, n_inst, n_job, g.nring, nref);
    out.lds_bytes = (size_t)L.total * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024;
    return f.on != 0;
}

template <int N, int NHW>
__global__ __launch_bounds__(RF_THREADS) void search_pair_kernel(DevGeom g_in, FusedGeom f, const float *__restrict__ particles,
                                                                 const float *__restrict__ state, int n,
                                                                 const float *__restrict__ Bf, int nref,
                                                                 CandT *__restrict__ cand, float *__restrict__ dbg_spec)
{
    DevGeom g = g_in;
    g.maxrin = N; g.lg_maxrin = __builtin_ctz(N);
    g.sbuf = f.s_sbuf; g.pst = f.s_pst;
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    constexpr int NH = 2 * NHW, RZ = 2 * NH;                           // reference pairs, references per tile
    extern __shared__ __align__(16) float lds[];
    const int o_bufs = (f.s_rows * f.s_pst + 3) & ~3;
    float *img = lds;
    float *bufs = lds + o_bufs;                                        // [2][sbuf] ring buffers | CCF spectra of a tile, [offset][reference]
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + ((2 * f.s_sbuf + 3) & ~3));
    float2 *qt_s = tw_s + N;
    int4 *inst_s = reinterpret_cast<int4 *>(reinterpret_cast<float *>(qt_s) + ((2 * g.n_qtab + 3) & ~3));
    int4 *jobs_s = inst_s + g.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job);
    // [6] counter, [7] zero, [8 + o] avg, [12 + o] 1 / sigma, [16 + 2 o] centre, [24 + 2 (o nring + ring)] ring partials
    float *red = instw_s + ((g.n_inst + 3) & ~3);
    float2 *tws = reinterpret_cast<float2 *>(red + ((24 + 4 * g.nring + 3) & ~3));
    CandT *pc = reinterpret_cast<CandT *>(tws + N);                    // [2][RZ] records of the tile
    CandT *pbest = pc + 2 * RP_MAXRZ;                                  // [2][2] best record and runner-up of an offset over the tiles so far
    int *goff_s = reinterpret_cast<int *>(reinterpret_cast<float *>(pc) + (((2 * RP_MAXRZ + 4) * (int)(sizeof(CandT) / 4) + 3) & ~3));
    float *cdc_s = reinterpret_cast<float *>(goff_s + 8 * RP_GSTR);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < N; i += RF_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RF_THREADS) qt_s[i] = g.qtab[i];
    for (int i = tid; i < g.n_inst; i += RF_THREADS) { inst_s[i] = g.inst[i]; instw_s[i] = g.instw[i]; }
    for (int i = tid; i < g.n_job; i += RF_THREADS) jobs_s[i] = g.jobs[i];
    for (int i = tid; i < R1 * R2; i += RF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2)) & (N - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    for (int i = tid; i < 2 * f.s_sbuf; i += RF_THREADS) bufs[i] = 0.f;   // slack between rings must hold finite values
    for (int i = tid; i < o_bufs; i += RF_THREADS) img[i] = 0.f;          // row and column nx + 1 stay zero
    for (int i = tid; i < nref; i += RF_THREADS) cdc_s[i] = f.cdc_w[i];
    for (int i = tid; i < 8 * RP_GSTR; i += RF_THREADS) {
        const int m = i / RP_GSTR, j = i - m * RP_GSTR;
        goff_s[i] = 4 * f.roff[min(f.grp_ring0[m] + j, g.nring - 1)];
    }
    const float *imgb = img - g.pst - 1;                               // 1-based coordinates (ix, iy) -> img[(iy - 1) pst + ix - 1]
    int *ifft_done = reinterpret_cast<int *>(red + 6);
    if (tid == 0) { *ifft_done = 0; red[7] = 0.f; }
    int done_target = 0;

    const int xm = f.wmap[wave] & 255, share = f.wmap[wave] >> 8;     // this wave's bin group and its half of a tile's pairs
    const int nq = f.grp_nq[xm];
    const int rank = f.s_rank[wave], call = f.s_call[wave];
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bf), 0, (f.b_floats + 256) * 4, 0x00020000);
    const int ntile = f.ntile, nx1 = 2 * g.nkx + 1;
    const int4 jd0 = g.jobs[min(rank, g.n_job - 1)];

    bool pend = false;
    int p_prev = 0, s_prev0 = 0, s_prev1 = -1;
    int ipass = 0;                         // profiling builds: the wave timeline covers the first 64 passes of workgroup 0 (stamps as search_solo_kernel)
    // records of tile t for offset slot o: best reference with the runner-up inside and across tiles (search_solo_kernel)
    auto merge_records = [&](int t, bool last, int pw, int sw, int o) {
        constexpr int W = sizeof(CandT) / 4;
        if (wave != f.s_rec) return;
        const int wd = rf_own_lane(lane) - 16 * o;
        if (wd >= 0 && wd < W) {
            const CandT *pco = pc + o * RZ;
            CandT *pb = pbest + 2 * o;
            const int nrz = min(RZ, nref - t * RZ);
            float bv = pco[0].val, sv = -3.0e38f; int br = 0, sr = 0;
            for (int q3 = 1; q3 < nrz; q3++) {
                const float v = pco[q3].val;
                if (v >= bv) { sv = bv; sr = br; bv = v; br = q3; }
                else if (v >= sv) { sv = v; sr = q3; }
            }
            const CandT *win = pco + br, *run = pco + sr;
            float lv = sv;
            if (t > 0) {
                const float pv = pb[0].val, l2 = pb[1].val;
                if (!(bv >= pv)) {
                    if (bv >= l2) { run = pco + br; lv = bv; } else { run = pb + 1; lv = l2; }
                    win = pb; bv = pv;
                } else if (pv >= lv && pv >= l2) { run = pb; lv = pv; }
                else if (l2 > lv) { run = pb + 1; lv = l2; }
            }
            int word = reinterpret_cast<const int *>(win)[wd];
            int rword = reinterpret_cast<const int *>(run)[wd];
            if (last) {
                if (wd == 1 && lv >= bv - RA_TIE_RTOL * fabsf(bv)) word = cand_pack_runner(cand_jtot(word), *run);
                if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * red[12 + o]);     // val, t7[]
                reinterpret_cast<int *>(cand + (size_t)pw * g.ent_stride + sw)[wd] = word;
            } else {
                if (wd == 0) rword = __float_as_int(lv);
                reinterpret_cast<int *>(pb)[wd] = word;
                reinterpret_cast<int *>(pb + 1)[wd] = rword;
            }
        }
    };

#pragma unroll 1
    for (int p = blockIdx.x; p < n; p += gridDim.x) {
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
        w.lkx = __builtin_amdgcn_readfirstlane(w.lkx); w.rkx = __builtin_amdgcn_readfirstlane(w.rkx);
        w.lky = __builtin_amdgcn_readfirstlane(w.lky); w.rky = __builtin_amdgcn_readfirstlane(w.rky);
        const float cxf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)g.cnx + w.sxi)));
        const float cyf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)g.cnx + w.syi)));
        auto in_window = [&](int s) {
            const int iy = s / nx1 - g.nky, ix = s - (s / nx1) * nx1 - g.nkx;
            return ix >= -w.lkx && ix <= w.rkx && iy >= -w.lky && iy <= w.rky;
        };
        auto next_live = [&](int s) { while (s < g.nshift && !in_window(s)) s++; return s; };
        auto write_centres = [&](int s0, int s1) {
            if (wave != f.s_ctr) return;
            const int cl = rf_own_lane(lane);
            if (cl < 2) {
                const int s = cl ? s1 : s0;
                if (s < g.nshift) { red[16 + 2 * cl] = cxf + g.shift_x[s]; red[17 + 2 * cl] = cyf + g.shift_y[s]; }
            }
        };
        int s0 = next_live(0);
        int s1 = next_live(s0 + 1);
        write_centres(s0, s1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RF_LDS_BARRIER();
#pragma unroll 1
        while (s0 < g.nshift) {
            const bool two = s1 < g.nshift;
            const int nlive = two ? 2 : 1;
            const int n0 = next_live(s1 + 1), n1 = next_live(n0 + 1);
            const bool tl = blockIdx.x == 0 && ipass < 64;
            RA_STAMP(g, tl, ipass, wave, 0);
            // ---- ring jobs of both offsets (slots 0, 1 of the job table) into the two ring buffers; the previous pass's last inverse
            // FFTs are awaited inside the job, between its sampling and its first write to the ring buffers
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
                case 6: ring_job<16, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, nlive, ps); break;
                case 1: ring_job<8, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, nlive, ps); break;
                case 7: ring_job<8, 4, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, nlive, ps); break;
                case 9: ring_job_mix<true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, nlive, ps); break;
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
            if (pend) {
                merge_records(ntile - 1, true, p_prev, s_prev0, 0);
                if (s_prev1 >= 0) merge_records(ntile - 1, true, p_prev, s_prev1, 1);
            }
            RA_STAMP(g, tl, ipass, wave, 1);
            RF_LDS_BARRIER();
            RA_STAMP(g, tl, ipass, wave, 2);
            const int ln = rf_own_lane(lane);         // nothing derived from the lane index lives through the ring jobs
            // Normalize_ring statistics (fixed order: reproducible), one wave per offset, and the next pass's centres
            if (wave == f.s_stat || (wave == f.s_rec && two)) {
                const int o = wave == f.s_stat ? 0 : 1;
                float a = 0.f, q = 0.f;
                for (int i = ln; i < g.nring; i += 64) { a += red[24 + 2 * (o * g.nring + i)]; q += red[25 + 2 * (o * g.nring + i)]; }
                a = wave_sum_dpp(a); q = wave_sum_dpp(q);
                float avg = 0.f, rsg = 1.f;
                if (g.norm_ring) {
                    avg = a * g.inv_nn_weight;
                    rsg = __builtin_amdgcn_rsqf((q - a * avg) * g.inv_nn_weight);
                }
                if (ln == 0) { red[8 + o] = avg; red[12 + o] = rsg; }
            }
            write_centres(n0, n1);
            if (dbg_spec) {           // tests only: both ring buffers and the statistics of the pass's offsets
                RF_LDS_BARRIER();
                for (int o = 0; o < nlive; o++) {
                    float *dst = dbg_spec + ((size_t)p * g.nshift + (o ? s1 : s0)) * (g.lring + 2);
                    for (int i = tid; i < g.lring; i += RF_THREADS) dst[i] = bufs[o * g.sbuf + i];
                    if (tid == 0) { dst[g.lring] = red[8 + o]; dst[g.lring + 1] = red[12 + o]; }
                }
                RF_LDS_BARRIER();
                s0 = n0; s1 = n1;
                continue;
            }
            // ---- this wave's slice of the spectra of both offsets: bins 16 xm .. 16 xm + 15, every ring that has them
            const int xb = ln >> 2, xj = ln & 3, odd = ln & 1;
            float a[4 * RP_NQ];
            {
                const char *abase = reinterpret_cast<const char *>(bufs + (xj >> 1) * g.sbuf + 2 * (16 * xm + xb) + (xj & 1));
                const int4 *gq = reinterpret_cast<const int4 *>(goff_s + xm * RP_GSTR);
#pragma unroll
                for (int sl = 0; sl < RP_NQ; sl++) {
                    if (sl >= RP_NQ - nq) {
                        const int4 o = gq[sl - (RP_NQ - nq)];
                        a[4 * sl] = *reinterpret_cast<const float *>(abase + o.x); a[4 * sl + 1] = *reinterpret_cast<const float *>(abase + o.y);
                        a[4 * sl + 2] = *reinterpret_cast<const float *>(abase + o.z); a[4 * sl + 3] = *reinterpret_cast<const float *>(abase + o.w);
                    } else {
                        a[4 * sl] = a[4 * sl + 1] = a[4 * sl + 2] = a[4 * sl + 3] = 0.f;
                    }
                }
            }
            RA_STAMP(g, tl, ipass, wave, 3);
            // ---- tiles of RZ references: this wave contracts pairs share * NHW .. + NHW - 1 of the tile
#pragma unroll 1
            for (int t = 0; t < ntile; t++) {
                const int ref_lo = t * RZ, nrz = min(RZ, nref - ref_lo);
                const int ns = (nq + 1) & ~1;
                const unsigned voff = (unsigned)ln * 16u;
                const int h0 = share * NHW;
                f32x4 acc[NHW];
                {
                    unsigned row[NHW];
                    const unsigned pad = ns > nq ? 0x80000000u : 0u;
#pragma unroll
                    for (int h = 0; h < NHW; h++)
                        row[h] = (unsigned)(f.grp_boff[xm] + min(t * NH + h0 + h, f.nrp - 1) * nq * 256) * 4u - (unsigned)((ns - nq) * 1024);
                    switch (ns) {
                    case 2: rs_contract<NHW, 2, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    case 4: rs_contract<NHW, 4, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    case 6: rs_contract<NHW, 6, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    case 8: rs_contract<NHW, 8, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    case 10: rs_contract<NHW, 10, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    default: rs_contract<NHW, 12, RP_NQ>(a, brsrc, voff, row, pad, acc); break;
                    }
                }
                RA_STAMP(g, tl && t == 0, ipass, wave, 4);
                RF_LDS_BARRIER();         // t = 0: every slice is in registers; t > 0: the inverse FFTs of tile t - 1 are over
                RA_STAMP(g, tl && t == 0, ipass, wave, 5);
                if (t > 0) { merge_records(t - 1, false, p, s0, 0); if (two) merge_records(t - 1, false, p, s1, 1); }
                {
                    // Z_k = Q_k + i T_k and Z_{N-k} = conj Q_k + i conj T_k for this lane's bin of the wave's reference pairs: the even
                    // lane of a pair keeps offset A (rows 0, 1), the odd lane offset B (rows 2, 3), after the 2 x 2 exchange of
                    // search_fused_kernel's store (Util::Crosrng_ms: Q = (a + d) + i (c - b), T = (a - d) - i (b + c))
                    typedef ZLayout<N> ZL;
                    const int k = 16 * xm + xb, km = k ? N - k : N / 2;
                    const int ref_b = ref_lo + 2 * h0 + (xj >> 1);
                    float *zk = bufs + (odd * RZ + 2 * h0 + (xj >> 1)) * ZL::kPairStride + 2 * (k + (k >> 4));
                    const int dkm = 2 * (km + (km >> 4)) - 2 * (k + (k >> 4));
                    float dcv[NHW];
                    if (xm == 0) {
                        const float av = red[8 + odd];
#pragma unroll
                        for (int h = 0; h < NHW; h++) dcv[h] = av * cdc_s[min(ref_b + 2 * h, nref - 1)];
                    }
#pragma unroll
                    for (int h = 0; h < NHW; h++) {
                        const int ref = ref_b + 2 * h;
                        const f32x4 c4 = acc[h];
                        const float s0v = odd ? c4[0] : c4[2], s1v = odd ? c4[1] : c4[3];
                        const float r0x = swap_lane_pair(s0v), r1x = swap_lane_pair(s1v);
                        float ca = odd ? r0x : c4[0];
                        const float cb = odd ? r1x : c4[1], cc = odd ? c4[2] : r0x, cd = odd ? c4[3] : r1x;
                        const bool live = ref < nref && (two || !odd);
                        float2 vk, vm;
                        if (xm == 0) {
                            if (xb == 0) ca -= dcv[h];                        // Normalize_ring mean: the DC term only
                            const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                            vk = xb == 0 ? make_float2(ca, ca) : make_float2(apd + bpc, cmb + amd);
                            vm = xb == 0 ? make_float2(cd, cd) : make_float2(apd - bpc, amd - cmb);
                        } else {
                            const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                            vk = make_float2(apd + bpc, cmb + amd);
                            vm = make_float2(apd - bpc, amd - cmb);
                        }
                        if (live) {
                            float *z = zk + 2 * h * ZL::kPairStride;
                            *reinterpret_cast<float2 *>(z) = vk;
                            *reinterpret_cast<float2 *>(z + dkm) = vm;
                        }
                    }
                }
                RA_STAMP(g, tl && t == 0, ipass, wave, 6);
                RF_LDS_BARRIER();         // the spectra of the tile are complete
                RA_STAMP(g, tl && t == 0, ipass, wave, 7);
                if (call >= 0) {          // four transforms per call, 16 lanes each: slot zs = offset * RZ + reference
                    const int j = ln & 15, zs = 4 * call + (ln >> 4);
                    const int o = zs >= RZ ? 1 : 0, rr = zs - o * RZ;
                    if (zs < 2 * RZ && rr < nrz && (two || !o))
                        ifft_argmax<N, 1, 0>(bufs, pc, tws + j, zs, zs, j, ref_lo + rr, g.nomirror != 0);
                }
                RA_STAMP(g, tl && t == 0, ipass, wave, 8);
            }
            RA_STAMP(g, tl, ipass, wave, 9);
            ipass++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_fetch_add(ifft_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            done_target += RF_WAVES;
            pend = true; p_prev = p; s_prev0 = s0; s_prev1 = two ? s1 : -1;
            s0 = n0; s1 = n1;
        }
    }
    if (pend) {
#ifdef RALIGN_PROFILE_SWITCHES
        const PassSync ps = {true, ifft_done, done_target, nullptr};
#else
        const PassSync ps = {true, ifft_done, done_target};
#endif
        ps();
        merge_records(ntile - 1, true, p_prev, s_prev0, 0);
        if (s_prev1 >= 0) merge_records(ntile - 1, true, p_prev, s_prev1, 1);
    }
}

}  // namespace ralign
