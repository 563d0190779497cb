// search_duo_kernel: search_solo_kernel (ralign_solo.h) with TWO search offsets per pass, although only one offset's ring spectra
// fit the LDS next to the image.
//
// What bounds search_solo_kernel is the stream of prepared references (the MFMA B operand): 683 KB per offset at 128 x 128 / ou = 60 /
// nref = 10 through a CU's vector memory path (64 B per clock: 13 - 14 k of a pass's 38 k cycles, profiles/r05_solo_wave_timeline.txt),
// used by HALF of every 4 x 4 x 1 block -- rows 2 and 3, the second offset of the four-offset kernels, are empty.  Here a pass takes
// two offsets through the ONE ring buffer one after the other:
//   ring jobs of offset A -> every wave loads its slice of A's spectra into the lanes that hold block rows 0 and 1
//   ring jobs of offset B (the slice of A stays in registers: the register-light jobs -- ring_job512, 8 sample pairs per lane --
//                          leave room for it) -> the lanes of rows 2 and 3 load their slice of B
//   contraction of both offsets at once: one B stream, full blocks; tiles of 2 NH references, the CCF spectra of both offsets
//   (4 NH transforms, one per wave) in the ring-buffer space
// A particle's odd last offset runs as a pass of one.  Everything else -- jobs, statistics, records with runner-up, window skip, image
// without border, arrival counter -- is search_solo_kernel's.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq (test_mref_gpu_align.py:1043-1044, test_reffree_gpu_align.py:
// 844-847); replaces cuda/gpu_aln_noref.cu:818-879, 1009-1143, 2095-2206, 1289-1346 for rings of 512 samples.
#pragma once

#include "ralign_solo.h"

namespace ralign {

constexpr int RD_MAXNH = 4;        // reference pairs per tile: 4 NH spectrum slots (two offsets) must fit the ring buffer

// LDS plan (floats): as solo_lds_plan with ring partials and records for two offsets
inline SoloLds duo_lds_plan(int N, int rows, int pst, int sbuf, int n_qtab, int n_inst, int n_job, int nring, int nref)
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
    L.red = o; o += rs_up4(24 + 4 * nring);
    L.tws = o; o += 2 * N + 2 * 64;
    L.pc = o; o += rs_up4((4 * RD_MAXNH + 4) * (int)(sizeof(CandT) / 4));      // [2][2 RD_MAXNH] records of a tile, [2][2] best / runner-up
    L.goff = o; o += 16 * RS_GSTR;
    L.cdc = o; o += rs_up4(nref);
    L.total = o;
    return L;
}

// plan: build_solo_plan's, with the tile size cut to what the ring buffer holds for two offsets
inline bool build_duo_plan(const Geometry &g, int nref, int n_qtab, int n_inst, int n_job, FusedPlanHost &out)
{
    if (!build_solo_plan(g, nref, n_qtab, n_inst, n_job, out)) return false;
    FusedGeom &f = out.f;
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    const int lring_pad = (g.lring + 31) / 32 * 32 + 16;
    int nhmax = std::min(RD_MAXNH, lring_pad / (4 * zstride));          // 2 offsets x 2 NH references
    if (nhmax < 1) return false;
    f.ntile = (f.nrp + nhmax - 1) / nhmax;
    f.nh = (f.nrp + f.ntile - 1) / f.ntile;
    f.nrpw = f.nh; f.rz = 2 * f.nh; f.nzr = f.ntile;
    f.s_sbuf = std::max(lring_pad, 2 * f.rz * zstride);
    f.s_r2 = RA_EXP_ENV("RALIGN_DUO_R2") ? ra_atoi(RA_EXP_ENV("RALIGN_DUO_R2")) & 15 : 0;
    // transforms of a tile: slot z = offset * rz + reference, one per wave, dealt from the highest rank down
    for (int w = 0; w < 16; w++) {
        const int c = 15 - f.s_rank[w];
        f.s_call[w] = c < 2 * f.rz ? c : -1;
    }
    const SoloLds L = duo_lds_plan(g.maxrin, f.s_rows, f.s_pst, f.s_sbuf, n_qtab, n_inst, n_job, g.nring, nref);
    out.lds_bytes = (size_t)L.total * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024;
    return f.on != 0;
}

// NQT: ring quads the A slice holds (>= the largest group's; the slice lives through the second offset's ring jobs: every register counts)
template <int N, int NH, int NQT>
__global__ __launch_bounds__(RF_THREADS) void search_duo_kernel(DevGeom g_in, FusedGeom f, const float *__restrict__ particles,
                                                                const float *__restrict__ state, int n,
                                                                const float *__restrict__ Bf, int nref,
                                                                CandT *__restrict__ cand, float *__restrict__ /*dbg_spec*/)
{
    DevGeom g = g_in;
    g.maxrin = N; g.lg_maxrin = __builtin_ctz(N);
    g.sbuf = f.s_sbuf; g.pst = f.s_pst;
    constexpr int RZ = 2 * NH;                                         // references per tile
    extern __shared__ __align__(16) float lds[];
    const int o_bufs = (f.s_rows * f.s_pst + 3) & ~3;
    float *img = lds;
    float *bufs = lds + o_bufs;                                        // ring buffer | CCF spectra of a tile, [offset][reference]
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + ((f.s_sbuf + 3) & ~3));
    float2 *qt_s = tw_s + N;
    int4 *inst_s = reinterpret_cast<int4 *>(reinterpret_cast<float *>(qt_s) + ((2 * g.n_qtab + 3) & ~3));
    int4 *jobs_s = inst_s + g.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job + g.n_job_b);
    // [6] counter, [7] zero, [8 + o] avg, [12 + o] 1 / sigma, [16 + 2 o] centre, [24 + 2 nring o ..] ring partials of offset o
    float *red = instw_s + ((g.n_inst + 3) & ~3);
    float2 *tws = reinterpret_cast<float2 *>(red + ((24 + 4 * g.nring + 3) & ~3));      // inverse-FFT twiddles: [8][64], then [8][8]
    CandT *pc = reinterpret_cast<CandT *>(tws + N + 64);               // [2][RZ] records of the tile
    CandT *pbest = pc + 4 * RD_MAXNH;                                  // [2][2] best record and runner-up of an offset over the tiles so far
    int *goff_s = reinterpret_cast<int *>(reinterpret_cast<float *>(pc) + (((4 * RD_MAXNH + 4) * (int)(sizeof(CandT) / 4) + 3) & ~3));
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
    const float *imgb = img - g.pst - 1;                               // 1-based coordinates (ix, iy) -> img[(iy - 1) pst + ix - 1]
    int *ifft_done = reinterpret_cast<int *>(red + 6);
    int *slice_done = reinterpret_cast<int *>(red + 5);      // arrival counter: waves that hold their slice of offset A (the ring buffer may take B)
    if (tid == 0) { *ifft_done = 0; *slice_done = 0; red[7] = 0.f; }
    int done_target = 0, slice_target = 0;

    const int xm = f.wmap[wave];                                       // this wave's bin group
    const int nq = f.grp_nq[xm];
    const int rank = f.s_rank[wave], call = f.s_call[wave];
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bf), 0, (f.b_floats + 256) * 4, 0x00020000);
    const int ntile = f.ntile, nx1 = 2 * g.nkx + 1;
    // job tables: [0, n_job) for the first offset of a pass (any job: no slice is live yet), [n_job, n_job + n_job_b) -- the
    // register-light jobs -- for the second; n_job_b = 0: one table for both
    const int jb0 = g.n_job_b ? g.n_job : 0, jbn = g.n_job_b ? g.n_job_b : g.n_job;
    const int4 jd0 = g.jobs[min(rank, g.n_job - 1)], jd1 = g.jobs[jb0 + min(rank, jbn - 1)];

    bool pend = false;                     // a pass whose last inverse FFTs and records are outstanding
    int p_prev = 0, s_prev0 = 0, s_prev1 = -1;
    int ipass = 0;                         // profiling builds: the wave timeline covers the first 64 passes of workgroup 0
    // records of tile t for offset slot o (see search_solo_kernel): best reference with the runner-up inside and across tiles
    auto merge_records = [&](int t, bool last, int pw, int sw, int o) {
        constexpr int W = sizeof(CandT) / 4;
        if (wave != f.s_rec) return;
        const int wd = rf_own_lane(lane) - 16 * o;                         // lanes 0 .. 9: offset 0, lanes 16 .. 25: offset 1
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
                if (!(bv >= pv)) {             // the best of the earlier tiles stays
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
    // ring jobs of one offset (slot o of the pass: centre red[16 + 2 o], partials red[24 + 2 nring o]); the first of a pass waits for
    // the previous pass's inverse FFTs inside the job, between its sampling and its first write to the ring buffer
    // (a second round of jobs -- more than 16 -- starts at rank 16 - s_r2)
    auto ring_jobs = [&](auto oc, bool wait, const int *counter, int target) {
        constexpr int o = decltype(oc)::value;
        const int base = o ? jb0 : 0, njob = o ? jbn : g.n_job;
#pragma unroll 1
        for (int jr = 0; jr * RF_WAVES < njob; jr++) {
            const int job = jr == 0 ? rank : jr * RF_WAVES + ((rank + f.s_r2) & (RF_WAVES - 1));
            if (job >= njob) continue;
            const int4 jd = jr == 0 ? (o ? jd1 : jd0) : jobs_s[base + job];
#ifdef RALIGN_PROFILE_SWITCHES
            const PassSync ps = {wait && jr == 0, counter, target, nullptr};
#else
            const PassSync ps = {wait && jr == 0, counter, target};
#endif
            const float *ctr = red + 16 + 2 * o;
            float *part = red + 24 + 2 * g.nring * o;
            switch (__builtin_amdgcn_readfirstlane(jd.x)) {
            case 10: if constexpr (o == 0) ring_job<16, 16, true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 6: if constexpr (o == 0) ring_job<16, 8, true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 11: ring_job512<true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 0: ring_job<8, 16, true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 1: ring_job<8, 8, true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 7: ring_job<8, 4, true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            case 9: ring_job_mix<true>(g, imgb, bufs, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, 1, ps); break;
            default: break;
            }
        }
        if (wait && rank >= njob) {
#ifdef RALIGN_PROFILE_SWITCHES
            const PassSync ps = {true, counter, target, nullptr};
#else
            const PassSync ps = {true, counter, target};
#endif
            ps();
        }
    };
    // Normalize_ring statistics of offset slot o (fixed order: reproducible)
    auto statistics = [&](int o) {
        float a = 0.f, q = 0.f;
        const float *part = red + 24 + 2 * g.nring * o;
        const int l0 = rf_own_lane(lane);
        for (int i = l0; i < g.nring; i += 64) { a += part[2 * i]; q += part[2 * i + 1]; }
        a = wave_sum_dpp(a); q = wave_sum_dpp(q);
        float avg = 0.f, rsg = 1.f;
        if (g.norm_ring) {
            avg = a * g.inv_nn_weight;
            rsg = __builtin_amdgcn_rsqf((q - a * avg) * g.inv_nn_weight);
        }
        if (l0 == 0) { red[8 + o] = avg; red[12 + o] = rsg; }
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
        auto write_centres = [&](int s0, int s1) {          // two lanes of the centre wave: the pass's offsets
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
            const int n0 = next_live(s1 + 1), n1 = next_live(n0 + 1);      // the next pass's offsets
            // the lane's block (bin of the group) and row (offset, Re / Im) in the multiplies; taken anew after each round of ring jobs
            // (rf_own_lane): nothing derived from it lives through the jobs
            int ln, xb, xj, odd;
            auto take_lane = [&]() { ln = rf_own_lane(lane); xb = ln >> 2; xj = ln & 3; odd = ln & 1; };
            // profiling builds, stamps: 0 pass start, 1 ring jobs of A done, 2 behind their barrier, 3 slice of A loaded, 4 behind the barrier
            // that frees the ring buffer, 5 ring jobs of B done, 6 behind their barrier, 7 slice of B loaded, 8 contraction of tile 0
            // done, 9 behind barrier A, 10 spectra stored, 11 behind barrier B, 12 transforms of tile 0 done, 15 end of the pass
            const bool tl = blockIdx.x == 0 && ipass < 64;
            RA_STAMP(g, tl, ipass, wave, 0);
            // ---- offset A: ring jobs, statistics, the slice of its spectra into the lanes of block rows 0 and 1 (all lanes load: the
            // lanes of rows 2 and 3 are overwritten below, or multiply into rows nobody stores)
            ring_jobs(std::integral_constant<int, 0>{}, pend, ifft_done, done_target);
            if (pend) {
                merge_records(ntile - 1, true, p_prev, s_prev0, 0);
                if (s_prev1 >= 0) merge_records(ntile - 1, true, p_prev, s_prev1, 1);
            }
            RA_STAMP(g, tl, ipass, wave, 1);
            RF_LDS_BARRIER();
            RA_STAMP(g, tl, ipass, wave, 2);
            if (wave == f.s_stat) statistics(0);
            take_lane();
            float a[4 * NQT];
            const char *abase = reinterpret_cast<const char *>(bufs + 2 * (16 * xm + xb) + (xj & 1));      // the lane's bin in ring 0's place
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
            RA_STAMP(g, tl, ipass, wave, 3);
            if (two) {
                // this wave holds its slice of A: count it.  The ring jobs of B sample first (image and tables only) and wait for all 16
                // before their first write to the ring buffer
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_fetch_add(slice_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                slice_target += RF_WAVES;
                RA_STAMP(g, tl, ipass, wave, 4);
                // ---- offset B: the same through the same buffer; the lanes of rows 2 and 3 replace their part of the slice
                ring_jobs(std::integral_constant<int, 1>{}, true, slice_done, slice_target);
                RA_STAMP(g, tl, ipass, wave, 5);
                RF_LDS_BARRIER();
                RA_STAMP(g, tl, ipass, wave, 6);
                if (wave == f.s_stat) statistics(1);
                take_lane();
                abase = reinterpret_cast<const char *>(bufs + 2 * (16 * xm + xb) + (xj & 1));
                if (xj & 2) {
#pragma unroll
                    for (int sl = 0; sl < NQT; sl++) {
                        if (sl >= NQT - nq) {
                            const int4 o = gq[sl - (NQT - nq)];
                            a[4 * sl] = *reinterpret_cast<const float *>(abase + o.x); a[4 * sl + 1] = *reinterpret_cast<const float *>(abase + o.y);
                            a[4 * sl + 2] = *reinterpret_cast<const float *>(abase + o.z); a[4 * sl + 3] = *reinterpret_cast<const float *>(abase + o.w);
                        }
                    }
                }
            }
            RA_STAMP(g, tl, ipass, wave, 7);
            write_centres(n0, n1);          // the next pass's centres: this pass's ring jobs are over
            // ---- tiles of RZ references, both offsets at once
#pragma unroll 1
            for (int t = 0; t < ntile; t++) {
                const int ref_lo = t * RZ, nrz = min(RZ, nref - ref_lo);
                const int ns = (nq + 1) & ~1;
                const unsigned voff = (unsigned)ln * 16u;
                // the tile's reference pairs in chunks of at most two or three (NH = 4: 2 + 2): slice, accumulators and operands of
                // four pairs at once spilled 75 registers
                constexpr int NH0 = NH > 3 ? (NH + 1) / 2 : NH, NH1 = NH - NH0;
                auto chunk = [&](auto nhc_c, int h0) {
                    constexpr int NHC = decltype(nhc_c)::value;
                    f32x4 acc[NHC];
                    {
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
                        case 14: if constexpr (NQT >= 14) rs_contract<NHC, 14, NQT>(a, brsrc, voff, row, pad, acc); break;
                        default: if constexpr (NQT >= 16) rs_contract<NHC, 16, NQT>(a, brsrc, voff, row, pad, acc); break;
                        }
                    }
                    if (h0 == 0) {
                        RA_STAMP(g, tl && t == 0, ipass, wave, 8);
                        RF_LDS_BARRIER();         // t = 0: every slice is in registers; t > 0: the inverse FFTs of tile t - 1 are over
                        RA_STAMP(g, tl && t == 0, ipass, wave, 9);
                        if (t > 0) { merge_records(t - 1, false, p, s0, 0); if (two) merge_records(t - 1, false, p, s1, 1); }
                    }
                    // Z_k = Q_k + i T_k and Z_{N-k} = conj Q_k + i conj T_k for this lane's bin of every reference pair: the even lane
                    // of a pair keeps offset A (rows 0, 1), the odd lane offset B (rows 2, 3), after the 2 x 2 exchange of
                    // search_fused_kernel's store (Util::Crosrng_ms: Q = (a + d) + i (c - b), T = (a - d) - i (b + c))
                    typedef ZLayout<N> ZL;
                    const int k = 16 * xm + xb, km = k ? N - k : N / 2;
                    const int ref_b = ref_lo + 2 * h0 + (xj >> 1);
                    float *zk = bufs + (odd * RZ + 2 * h0 + (xj >> 1)) * ZL::kPairStride + 2 * (k + (k >> 4));
                    const int dkm = 2 * (km + (km >> 4)) - 2 * (k + (k >> 4));
                    float dcv[NHC];
                    if (xm == 0) {
                        const float av = red[8 + odd];
#pragma unroll
                        for (int h = 0; h < NHC; h++) dcv[h] = av * cdc_s[min(ref_b + 2 * h, nref - 1)];
                    }
#pragma unroll
                    for (int h = 0; h < NHC; h++) {
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
                };
                chunk(std::integral_constant<int, NH0>{}, 0);
                if constexpr (NH1 > 0) chunk(std::integral_constant<int, NH1>{}, NH0);
                RA_STAMP(g, tl && t == 0, ipass, wave, 10);
                RF_LDS_BARRIER();         // the spectra of the tile are complete
                RA_STAMP(g, tl && t == 0, ipass, wave, 11);
                if (call >= 0) {          // slot z = offset * RZ + reference
                    const int o = call >= RZ ? 1 : 0, rr = call - o * RZ;
                    if (rr < nrz && (two || !o))
                        {
                        // NQT 16 (ou 57 .. 62): the lane's twiddles and addresses of the transform are rebuilt per tile -- held over the tile
                        // loop they pushed 18 registers to scratch, reloaded inside the contraction (box128: 533 k -> 540 k particles/s);
                        // NQT 14 has room for them (nb00, 9 tiles per pass: 141 k -> 148 k with them held)
                        const int il = NQT > 14 ? rf_own_lane(lane) : ln;
                        ifft512_wave_argmax(bufs, pc + call, tws, tws + N, call, il, ref_lo + rr, g.nomirror != 0);
                    }
                }
                RA_STAMP(g, tl && t == 0, ipass, wave, 12);
            }
            RA_STAMP(g, tl, ipass, wave, 15);
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
