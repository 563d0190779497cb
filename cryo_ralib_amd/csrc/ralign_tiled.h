// Particle-resident search kernel for MORE references than one pass of search_fused_kernel can accumulate
// (BASELINE configs[3]: 90 x 90, nref = 50; any nref at maxrin 256 and <= 36 rings).
//
// Same pass structure as search_fused_kernel (ralign_fused.h): 4 search offsets per pass, ring jobs into the 4 LDS ring
// buffers, contraction on v_mfma_f32_4x4x1_16b_f32, CCF spectra in LDS, inverse FFT + argmax there.  What changes is
// where the A operand (the particle spectra) lives while the references are walked in TILES of 2 NH <= 10:
//   * after the ring jobs every wave loads ITS slice of the spectra -- the 16 Fourier bins of its bin group x the
//     (Re, Im) of one offset pair x every ring that has those bins: one float per ring and lane, <= 36 VGPRs -- into
//     registers.  From then on the ring buffers are dead, so the CCF spectra of every tile can use their space (47 slots
//     of N complex points = 4 offsets x 10 references + slack) while the A operand stays available for the next tile.
//   * wave role = (bin group, offset pair): 8 groups x 2 offset pairs = 16 waves; a tile accumulates NH reference
//     pairs in NH x 4 VGPRs.  The contraction of a tile touches no LDS at all (A in registers, B from L2): the waves
//     that are through with the inverse FFTs of tile t run the contraction of tile t + 1 on the matrix pipe while the
//     others still transform on the vector pipe.
//   * two workgroup barriers per tile (before the spectra store: every inverse FFT of the previous tile is over;
//     after it: the spectra are complete); the best record per offset is carried across tiles in LDS with the same
//     ascending-reference ">=" rule reduce_records applies within a tile.
// HBM sees the image once and one record per search offset; the B stream (1.3 MB at 50 references) comes from L2.
//
// Reference call site restated: Util.multiref_polar_ali_2d (test_mref_gpu_align.py:1043-1044); replaces what
// cu_ccf_mult_m + CcfResultTable do at cuda/gpu_aln_noref.cu:1009-1143, 2095-2206.
#pragma once

#include "ralign_fused.h"

namespace ralign {

constexpr int RT_MAXNH = 5;        // reference pairs per tile (4 offsets x 10 references = 40 of the 47 spectrum slots)
constexpr int RT_NQ = 9;           // ring quads of a wave's A slice (<= 36 rings)
constexpr int RT_MINREF = 15;      // from here on the tiled kernel beats search_fused_kernel's two spectra rounds per pass (measured: 12 - 14
                                   // references 2 % slower, 15 and 16 4 % faster)

inline bool build_tiled_plan(const Geometry &g, int nref, int sbuf, size_t lds_polar_floats, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.bsrc.clear(); out.cdc_w.clear();
    if (g.maxrin != 256 || g.numr[2] < 8 || g.nring > 4 * RT_NQ || nref > 127) return false;
    f.ng = 8; f.wpg = 2;
    f.nrp = (nref + 1) / 2;
    f.ntile = (f.nrp + RT_MAXNH - 1) / RT_MAXNH;
    f.nh = (f.nrp + f.ntile - 1) / f.ntile;
    f.nrpw = f.nh;
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    f.rz = 2 * f.nh; f.nzr = f.ntile;
    f.rz_inv = (65536 + f.rz - 1) / f.rz;
    if (4 * f.rz * zstride > 4 * sbuf) return false;
    rf_layout_b(g, nref, f, out.bsrc);
    // wave roles (bin group m | offset pair << 8).  Waves w, w + 4, w + 8, w + 12 share a SIMD, whose issue port the vector
    // and the matrix instructions of its waves share (wave timeline, scripts/tiled_timeline.sh: a tile costs a SIMD the sum
    // of both).  The inverse-FFT round of a tile is one call (4 transforms) for each of the waves 0 .. 7 and, at rz = 10, a
    // second one for waves 0 and 1, so two SIMDs carry three calls and two carry two: the contraction items (cost = ring quads of the group) are dealt
    // heaviest-first to the SIMD with the least work so far, a call counted as RT_CALL_QUADS ring quads, and inside a SIMD
    // the heaviest item goes to the youngest wave (no call, or the call that starts last).
    {
        struct Item { int m, op, cost; };
        std::vector<Item> items;
        for (int m = 0; m < f.ng; m++)
            for (int op = 0; op < 2; op++) items.push_back({m, op, f.grp_nq[m]});
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
        constexpr int RT_CALL_QUADS = 10;      // ~390 vector instructions x 4 cycles against 20 matrix instructions x 8 cycles per ring quad
        int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
        for (int w = 0; w < 8; w++) load[w & 3] += RT_CALL_QUADS;              // ifft calls: waves 0 - 7, and a second one of
        if (4 * f.rz > 32) { load[0] += RT_CALL_QUADS; load[1] += RT_CALL_QUADS; }      // waves 0 and 1 (search_tiled_kernel)
        for (const Item &it : items) {
            int c = -1;
            for (int q = 0; q < 4; q++)
                if (used[q] < 4 && (c < 0 || load[q] < load[c])) c = q;
            f.wmap[c + 4 * (3 - used[c])] = it.m | (it.op << 8);
            load[c] += it.cost; used[c]++;
        }
        for (int q = 0; q < 4; q++) f.stat_wave[q] = q;           // the oldest waves: lightest contraction roles
    }
    out.cdc_w.assign(g.nring, 0.f);
    for (int r = 0; r < 68; r++) f.roff[r] = g.ring_off[std::min(r, g.nring - 1)];
    f.gstr = 4 * RT_NQ + 4;
    const size_t fl = lds_polar_floats + 2 * g.maxrin + (4 * 2 * RT_MAXNH + 8) * (sizeof(CandT) / 4) + f.ng * f.gstr + ((nref + 3) & ~3) + 64;
    out.lds_bytes = fl * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024;
    return f.on != 0;
}

// contraction of one tile for one wave: NH reference pairs of the wave's bin group against the A slice in registers (one
// offset pair); acc[h] = the 4 x 4 block (2 offsets x (Re, Im)) x (2 references x (Re, Im)) per bin.
// NS ring-quad slots in one straight line -- no branch between the B requests and their multiplies, so the compiler
// counts the outstanding requests exactly (s_waitcnt vmcnt(n)) instead of draining them at block boundaries.  The slice
// is RIGHT-aligned in a[]: a wave whose group has fewer than NS ring quads multiplies zeros in its first slots (the B
// rows it reads there belong to the preceding rows of the stream: finite values).  Two instantiations per kernel:
// NS = RT_NQ for the groups of the low bins, NS = 4 for the groups only the full-length rings reach.
// One B buffer per reference pair, refilled for the next slot as soon as its four multiplies have been issued (the
// matrix instruction reads its operands at issue); the pairs are multiplied two at a time, interleaved, so that
// consecutive matrix instructions are independent.  B requests are buffer loads: descriptor and row offsets in scalar
// registers, one lane offset in a vector register.
typedef int rt_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rt_load_b(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff)
{
    const rt_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

template <int NH, int NS>
__device__ __forceinline__ void rt_contract(const float (&a)[4 * RT_NQ], __amdgpu_buffer_rsrc_t rsrc, unsigned voff,
                                            const unsigned (&row)[NH], f32x4 (&acc)[NH])
{
    constexpr int A0 = 4 * (RT_NQ - NS);
    float4 bc[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) bc[h] = rt_load_b(rsrc, voff, row[h]);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
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

template <int N, int NH, int SB>
__global__ __launch_bounds__(RF_THREADS) void search_tiled_kernel(DevGeom g_in, FusedGeom f, const float *__restrict__ particles,
                                                                  const float *__restrict__ state, int n,
                                                                  const float *__restrict__ Bf, int nref,
                                                                  CandT *__restrict__ cand, const int *__restrict__ /*cls*/)
{
    DevGeom g = g_in;
    g.maxrin = N; g.lg_maxrin = __builtin_ctz(N);
    if constexpr (SB != 0) g.sbuf = SB;
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    constexpr int RZ = 2 * NH;                                         // references per tile
    extern __shared__ __align__(16) float lds[];
    // LDS plan of polar_fft_kernel / search_fused_kernel, then the extras of this kernel
    const int npad = g.pst * g.pst;
    float *img = lds;
    float *bufs = lds + ((npad + 3) & ~3);                             // [4][sbuf] ring buffers | CCF spectra of a tile
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + 4 * g.sbuf);
    float2 *qt_s = tw_s + g.maxrin;
    int4 *inst_s = reinterpret_cast<int4 *>(qt_s + g.n_qtab + (g.n_qtab & 1));
    int4 *jobs_s = inst_s + g.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job);
    float *red = instw_s + g.n_inst;
    float2 *tws = reinterpret_cast<float2 *>(red + 24 + 8 * g.nring + ((g.n_inst + 8 * g.nring) & 1));
    CandT *pc = reinterpret_cast<CandT *>(tws + R1 * R2);              // [4][RZ] records of the tile
    CandT *pbest = pc + 4 * 2 * RT_MAXNH;                              // [4] best record per offset over the tiles so far
    int *goff_s = reinterpret_cast<int *>(pbest + 8);                  // [ng][gstr] ring offsets (bytes) of every group's ring quads
    float *cdc_s = reinterpret_cast<float *>(goff_s + f.ng * f.gstr);  // [nref] DC weights of the references
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < g.maxrin; i += RF_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RF_THREADS) qt_s[i] = g.qtab[i];
    for (int i = tid; i < g.n_inst; i += RF_THREADS) { inst_s[i] = g.inst[i]; instw_s[i] = g.instw[i]; }
    for (int i = tid; i < g.n_job; i += RF_THREADS) jobs_s[i] = g.jobs[i];
    for (int i = tid; i < R1 * R2; i += RF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2) * (g.maxrin / N)) & (g.maxrin - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    for (int i = tid; i < 4 * g.sbuf; i += RF_THREADS) bufs[i] = 0.f;
    for (int i = tid; i < npad; i += RF_THREADS) img[i] = 0.f;            // zero border of the padded image
    for (int i = tid; i < nref; i += RF_THREADS) cdc_s[i] = f.cdc_w[i];
    for (int i = tid; i < f.ng * f.gstr; i += RF_THREADS) {
        const int m = i / f.gstr, j = i - m * f.gstr;
        goff_s[i] = 4 * f.roff[min(f.grp_ring0[m] + j, g.nring - 1)];
    }
    const float *imgb = img + (g.bd - 1) * g.pst + (g.bd - 1);          // (f.s_crop: re-based with every image, load_image)
    int *ifft_done = reinterpret_cast<int *>(red + 6);
    if (tid == 0) *ifft_done = 0;
    int done_target = 0;

    const int xm = f.wmap[wave] & 255, op = f.wmap[wave] >> 8;       // this wave's bin group and offset pair
    const int nq = f.grp_nq[xm];
    // the B stream as a buffer resource (scalar descriptor; requests carry a scalar row offset and the lane's 16 bytes)
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Bf), 0, (f.b_floats + 256) * 4, 0x00020000);
    const int ntile = f.ntile;
    // The workgroup's particles p(i) = blockIdx.x + i gridDim.x form one stream of search offsets cut into passes of 4, as in
    // search_fused_kernel: dense (f.pack: nshift slots per particle -- 4 particles x 49 offsets = 49 full passes, and a pass may hold
    // the last offsets of one particle and the first of the next: its ring jobs then run twice, over the slots of the resident image
    // and, after the next image has replaced it, over the rest) or padded (nshift_pad slots: every particle starts a pass; 49 offsets
    // = 13 passes, the last one with a single live offset -- 5.8 % of the passes for 2 % of the work).  Everything behind the sampling
    // is per slot (centres, statistics, spectra, records): the results are the same to the bit.
    const int SPP = f.pack ? g.nshift : g.nshift_pad;
    const int npw = (n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nslots = npw * SPP, npass = (nslots + 3) >> 2;
    auto particle_of = [&](int i) { return (int)blockIdx.x + i * (int)gridDim.x; };
    // pixels only (the zero border is written once, above), global -> LDS without a stop in registers, every request of the wave in
    // flight at once (search_fused_kernel: load_image)
    auto load_image = [&](int i) {
        const float *src = particles + (size_t)particle_of(i) * g.nx * g.nx;
        if (f.s_crop) {
            // a box far larger than the rings (engines of the size-generic class): the LDS holds a CROP of f.s_crop rows and columns
            // whose origin follows the particle's sampling centre (crop_plan, ralign_solo.h) -- NOT clamped to the box: this kernel
            // samples every offset of the window, and the taps of the offsets search_range excludes (their records are never read)
            // may lie outside the box, where the crop keeps whatever finite values it held
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
            imgb = img - g.pst - 1 - (oy0 * g.pst + ox0);      // 1-based (ix, iy) of the BOX -> img[(iy - 1 - oy0) pst + ix - 1 - ox0]
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
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
    if (tid < 4 && npw > 0) write_centre(0, 0, (int)tid);          // sampling centres of the stream's first pass
    RF_LDS_BARRIER();
    const int4 jd0 = jobs_s[min((int)wave, g.n_job - 1)];
    // the one-wave tasks of a pass (records, statistics, centres) start from an opaque copy of the lane index (rf_own_lane)
    auto own_lane = [](int v) { return rf_own_lane(v); };
    // records of tile t (ascending reference, ">=": later wins) against the best of the earlier tiles (a later tile wins
    // ties, as a later reference does); the last tile of a pass scales by 1/sigma and writes the records of the pass that started at
    // (particle i0r, offset s0r) out
    auto merge_records = [&](int t, bool last, int i0r, int s0r, int nl) {
        constexpr int W = sizeof(CandT) / 4;
        if (wave != 2) return;
        const int ml = own_lane(lane);
        if (ml < nl * W) {
            const int o = ml / W, wd = ml - o * W;
            const int nrz = min(RZ, nref - t * RZ);
            float bv = pc[o * RZ].val; int br = 0;
            for (int q3 = 1; q3 < nrz; q3++) {
                const float v = pc[o * RZ + q3].val;
                if (v >= bv) { bv = v; br = q3; }
            }
            const CandT *srcr = pc + o * RZ + br, *runr = pbest + o;
            float lv = -3.0e38f;                      // the loser of (this tile's best, best of the earlier tiles)
            if (t > 0) {
                const float pv = pbest[o].val;
                if (!(bv >= pv)) { runr = srcr; srcr = pbest + o; lv = bv; bv = pv; }
                else lv = pv;
            }
            int word = reinterpret_cast<const int *>(srcr)[wd];
            // float tie between references of different tiles: the loser travels in the jtot word (cand_pack_runner; ties inside a
            // tile would cost a second running maximum in this loop -- measured 3.9 % of the kernel -- and are left to the audit)
            if (wd == 1 && lv >= bv - RA_TIE_RTOL * fabsf(bv)) word = cand_pack_runner(cand_jtot(word), *runr);
            if (last) {
                if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * red[12 + o]);     // val, t7[]
                int io = i0r, so = s0r + o;
                if (so >= SPP) { so -= SPP; io++; }
                reinterpret_cast<int *>(cand + (size_t)particle_of(io) * g.ent_stride + so)[wd] = word;
            } else {
                reinterpret_cast<int *>(pbest + o)[wd] = word;
            }
        }
    };
    int i0 = 0, s0 = 0, img_i = -1;             // first slot of the current pass: particle index in the stream, offset; resident image
    int i0p = 0, s0p = 0, nlp = 4;              // ... and live slots of the previous pass (its last records are merged inside this one)
#pragma unroll 1
    for (int grp = 0; grp < npass; grp++) {
        // live slots of the pass (leading), slots of the first particle, and whether a second particle starts inside it
        const int nlive = f.pack ? min(4, nslots - 4 * grp) : min(4, g.nshift - s0);
        const int na = min(4, SPP - s0);
        const bool split = f.pack && na < nlive;
        const bool pend = grp > 0;
        // profiling builds: wave timeline of workgroup 0's first passes (stamps: 0 pass start, 1 ring jobs done, 2 behind their
        // barrier, 3 slice in registers; tile t < 2: 4 + 5 t contraction done, + 1 behind barrier A, + 2 spectra stored, + 3 behind
        // barrier B, + 4 transforms done; 15 end of the pass)
        const bool tl = blockIdx.x == 0 && grp < 64;
        RA_STAMP(g, tl, grp, wave, 0);
        // ---- ring jobs (as search_fused_kernel): the previous pass's last inverse FFTs are awaited inside the first job,
        // between its sampling and its first write to the ring buffers; a pass that holds the offsets of two particles runs them
        // twice, over slots [0, na) of the resident image and -- behind the exchange of the image between two barriers -- over [na, nlive)
#pragma unroll 1
        for (int ph = 0; ph < (split ? 2 : 1); ph++) {
            if (img_i != i0 + ph) {
                // at the top of a pass every wave is through with the sampling of the previous pass (its barrier 1): the image may go
                if (ph) RF_LDS_BARRIER();
                load_image(i0 + ph);
                img_i = i0 + ph;
                RF_LDS_BARRIER();
            }
            const int slo = ph ? na : 0, shi = (split && !ph) ? na : nlive;
            const bool wait = pend && !ph;
#pragma unroll 1
            for (int jr = 0; jr * RF_WAVES < g.n_job; jr++) {
                const int job = jr * RF_WAVES + wave;
                if (job >= g.n_job) continue;
                const int4 jd = jr == 0 ? jd0 : jobs_s[job];
#ifdef RALIGN_PROFILE_SWITCHES
                const PassSync ps = {wait && jr == 0, ifft_done, done_target, nullptr};
#else
                const PassSync ps = {wait && jr == 0, ifft_done, done_target};
#endif
                switch (__builtin_amdgcn_readfirstlane(jd.x)) {
                case 1: ring_job<8, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                case 6: ring_job<16, 8, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                case 7: ring_job<8, 4, true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                case 9: ring_job_mix<true>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf, shi, ps, slo); break;
                default: break;
                }
            }
            if (wait && wave >= g.n_job) {
#ifdef RALIGN_PROFILE_SWITCHES
                const PassSync ps = {true, ifft_done, done_target, nullptr};
#else
                const PassSync ps = {true, ifft_done, done_target};
#endif
                ps();
            }
        }
        if (pend) merge_records(ntile - 1, true, i0p, s0p, nlp);
        RA_STAMP(g, tl, grp, wave, 1);
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, __float_as_int(red[7])));
        RF_LDS_BARRIER();
        RA_STAMP(g, tl, grp, wave, 2);
        // Normalize_ring statistics of the 4 offsets and the next pass's sampling centres (as search_fused_kernel)
        const int os = wave == f.stat_wave[0] ? 0 : wave == f.stat_wave[1] ? 1 : wave == f.stat_wave[2] ? 2 : wave == f.stat_wave[3] ? 3 : -1;
        if (os >= 0) {
            float a = 0.f, q = 0.f;
            const int sl0 = own_lane(lane);
            for (int i = sl0; i < g.nring; i += 64) { a += red[24 + 2 * (os * g.nring + i)]; q += red[25 + 2 * (os * g.nring + i)]; }
            a = wave_sum_dpp(a); q = wave_sum_dpp(q);
            float avg = 0.f, rsg = 1.f;
            if (g.norm_ring) {
                avg = a * g.inv_nn_weight;
                rsg = __builtin_amdgcn_rsqf((q - a * avg) * g.inv_nn_weight);
            }
            if (sl0 == 0) { red[8 + os] = avg; red[12 + os] = rsg; }
        }
        // first slot of the next pass
        int in = i0, sn = s0 + 4;
        if (sn >= SPP) { sn -= SPP; in++; }
        if (wave == 5 && grp + 1 < npass) {
            const int cl = own_lane(lane);
            if (cl < 4) write_centre(in, sn, cl);
        }
        // ---- this wave's slice of the spectra: bins 16 xm .. 16 xm + 15 of its offset pair, every ring that has them
        const int xb = ln >> 2, xj = ln & 3, odd = ln & 1;
        const bool live_op = 2 * op < nlive;           // the last pass may leave an offset pair without a live offset
        float a[4 * RT_NQ];
        {
            const char *abase = reinterpret_cast<const char *>(bufs + (2 * op + (xj >> 1)) * g.sbuf + 2 * (16 * xm + xb) + (xj & 1));
            const int4 *gq = reinterpret_cast<const int4 *>(goff_s + xm * f.gstr);
#pragma unroll
            for (int sl = 0; sl < RT_NQ; sl++) {       // right-aligned: slot sl holds ring quad sl - (RT_NQ - nq)
                if (sl >= RT_NQ - nq && live_op) {
                    const int4 o = gq[sl - (RT_NQ - nq)];
                    a[4 * sl] = *reinterpret_cast<const float *>(abase + o.x); a[4 * sl + 1] = *reinterpret_cast<const float *>(abase + o.y);
                    a[4 * sl + 2] = *reinterpret_cast<const float *>(abase + o.z); a[4 * sl + 3] = *reinterpret_cast<const float *>(abase + o.w);
                } else {
                    a[4 * sl] = a[4 * sl + 1] = a[4 * sl + 2] = a[4 * sl + 3] = 0.f;
                }
            }
        }
        RA_STAMP(g, tl, grp, wave, 3);
        // ---- tiles of RZ references
#pragma unroll 1
        for (int t = 0; t < ntile; t++) {
            const int ref_lo = t * RZ, nrz = min(RZ, nref - ref_lo);
            f32x4 acc[NH];
            if (live_op) {
                // row offsets (bytes) of the tile's reference pairs in the wave's group block, moved back by the slots the
                // wave pads; a pair past the last one (final tile) re-reads the last pair's rows and is never stored
                unsigned row[NH];
                const int ns = nq <= 4 ? 4 : nq <= 7 ? 7 : nq == 8 ? 8 : RT_NQ;      // slots of the instantiation that runs (slice right-aligned in them)
#pragma unroll
                for (int h = 0; h < NH; h++)
                    row[h] = (unsigned)(f.grp_boff[xm] + min(t * NH + h, f.nrp - 1) * nq * 256) * 4u - (unsigned)((ns - nq) * 1024);
                switch (ns) {          // straight-line instantiations for the ring-quad counts of the headline geometry (9, 8, 7, 4)
                case 4: rt_contract<NH, 4>(a, brsrc, (unsigned)ln * 16u, row, acc); break;
                case 7: rt_contract<NH, 7>(a, brsrc, (unsigned)ln * 16u, row, acc); break;
                case 8: rt_contract<NH, 8>(a, brsrc, (unsigned)ln * 16u, row, acc); break;
                default: rt_contract<NH, RT_NQ>(a, brsrc, (unsigned)ln * 16u, row, acc); break;
                }
            } else {
#pragma unroll
                for (int h = 0; h < NH; h++) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            RA_STAMP(g, tl && t < 2, grp, wave, 4 + 5 * t);
            RF_LDS_BARRIER();         // t = 0: every slice is in registers; t > 0: the inverse FFTs of tile t - 1 are over
            RA_STAMP(g, tl && t < 2, grp, wave, 5 + 5 * t);
            if (t > 0) merge_records(t - 1, false, i0, s0, nlive);
            {
                // Z_k = Q_k + i T_k and Z_{N-k} (rf_store_z) for this lane's bin of every reference pair of the tile
                typedef ZLayout<N> ZL;
                const int k = 16 * xm + xb, km = k ? N - k : N / 2;
                const int ref_b = ref_lo + (xj >> 1), o = 2 * op + odd;
                float *zk = bufs + (o * RZ + (xj >> 1)) * ZL::kPairStride + 2 * (k + (k >> 4));
                const int dkm = 2 * (km + (km >> 4)) - 2 * (k + (k >> 4));
                float dcv[NH];
                if (xm == 0) {
                    const float av = red[8 + o];
#pragma unroll
                    for (int h = 0; h < NH; h++) dcv[h] = av * cdc_s[min(ref_b + 2 * h, nref - 1)];
                }
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const int ref = ref_b + 2 * h;
                    const f32x4 c4 = acc[h];
                    const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                    const float r0x = swap_lane_pair(s0), r1x = swap_lane_pair(s1);
                    float ca = odd ? r0x : c4[0];
                    const float cb = odd ? r1x : c4[1], cc = odd ? c4[2] : r0x, cd = odd ? c4[3] : r1x;
                    const bool live = ref < nref && o < nlive;
                    float2 vk, vm;
                    if (xm == 0) {
                        if (xb == 0 && live) ca -= dcv[h];
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
            RA_STAMP(g, tl && t < 2, grp, wave, 6 + 5 * t);
            RF_LDS_BARRIER();         // the spectra of the tile are complete
            RA_STAMP(g, tl && t < 2, grp, wave, 7 + 5 * t);
            // 4 RZ transforms, 16 lanes each, four per call: waves 0 - 7 take slots 0 .. 31 (lane groups (sub, sub ^ 1) of a
            // half-wave take slots zs and zs + 16, whose LDS images are 32 banks apart); slots 32 .. 4 RZ - 1 are a second call
            // of the OLDEST waves 0 and 1, whose first call is over first (the SIMDs issue oldest-first; full calls with 2-way
            // bank conflicts on their 35 LDS instructions rather than four half-filled calls of 390 vector instructions each:
            // the phase is bound by vector issue, not by the LDS)
#pragma unroll 1
            for (int call = 0; call < 2; call++) {
                if (wave >= (call ? 2 : 8) || (call == 1 && 4 * RZ <= 32)) break;
                const int j = ln & 15, sub = ln >> 4, uu = 2 * wave + (sub >> 1);
                const int zs = call == 0 ? (uu & 15) + 16 * (sub & 1) : 32 + 4 * wave + sub;
                const int o = __mul24(zs, f.rz_inv) >> 16, rr = zs - __mul24(o, RZ);
                if (zs < 4 * RZ && rr < nrz && o < nlive)
                    ifft_argmax<N, 1, 0>(bufs, pc + (o * RZ + rr) - zs, tws + j, zs, zs, j, ref_lo + rr, g.nomirror != 0);
            }
            RA_STAMP(g, tl && t < 2, grp, wave, 8 + 5 * t);
        }
        RA_STAMP(g, tl, grp, wave, 15);
        // end of the pass: its last inverse FFTs are counted, not awaited -- the next pass's first ring job (or the end of the stream)
        // waits for them, and merges the last tile's records
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) __hip_atomic_fetch_add(ifft_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        done_target += RF_WAVES;
        i0p = i0; s0p = s0; nlp = nlive;
        i0 = in; s0 = sn;
    }
    if (npass > 0) {
#ifdef RALIGN_PROFILE_SWITCHES
        const PassSync ps = {true, ifft_done, done_target, nullptr};
#else
        const PassSync ps = {true, ifft_done, done_target};
#endif
        ps();
        merge_records(ntile - 1, true, i0p, s0p, nlp);
    }
}

}  // namespace ralign
