// Polar stage of the size-generic class with the image in LDS (round 6): polar_zone_kernel.
//
// polar_generic_kernel (ralign_generic.h) takes the four taps of every polar sample from global memory, because an image of
// 256 x 256 floats (256 KB) does not fit a CU's LDS (160 KB) -- and is bound by the throughput of those scattered 4-byte
// requests (57 % of its time at BASELINE configs[4]: 2.7 G samples x 4 taps per chunk of 330 particles).  What a GROUP OF RINGS
// touches is much less than the image: the samples of rings r_lo .. r_hi under every search offset of a particle lie in an
// annulus of radii r_lo - S - 2 .. r_hi + S + 3 around the particle's centre (S = largest search shift).  The rings are
// therefore cut into ZONES of consecutive ring quads whose annulus fits the LDS next to the transform buffers, stored
// compactly row by row (per row a left and a right segment around the hole, addressed through a two-entry row table), and a
// workgroup = (particle, zone, quarter of the search offsets) fills its annulus once with coalesced loads and takes every tap
// from LDS.  Sample positions, tap values and the interpolation are those of polar_generic_kernel bit for bit (Util::bilinear's
// operation order, no contraction); ring FFTs, split step and the operand-panel stores are the same code.
//
//   work item of a wave = (search offset, ring quad): 4 rings sampled, transformed and stored as whole 16-byte panel pieces;
//   the Normalize_ring partial sums of an item go to stats_part[entry][quad] and polar_stats_kernel adds them in quad order
//   (fixed association: reproducible) into the {avg, 1/sigma} the contraction applies.
//
// Reference: what Polar2Dm + Normalize_ring + Frngs do inside Util.multiref_polar_ali_2d (test_mref_gpu_align.py:1043-1044);
// replaces cu_resample_to_polar + cuFFT R2C (cuda/gpu_aln_noref.cu:818-879, 1816-1820) for large boxes.
#pragma once

#include "ralign_generic.h"

namespace ralign {

struct ZoneDesc {
    int ring0;        // first ring of the zone (a multiple of 4)
    int nquad;        // ring quads in the zone
    int quad0;        // index of its first quad among all quads (stats_part column)
    int npix;         // pixels stored
    int pix_off;      // offset of its pixel list in ZonePlanDev::pixtab
    int row_off;      // offset of its row table in ZonePlanDev::rowtab
    int nrow;         // rows stored (2 H + 1)
    int H;            // rows -H .. H around the crop centre
};

struct ZonePlanDev {
    const ZoneDesc *zones;
    const int2 *rowtab;       // per zone [nrow + 1] {offL, offR}: LDS float index of pixel (x, row) = (x < 0 ? offL : offR) + x
    const int *pixtab;        // per zone [npix] (dy << 16) | (dx & 0xffff): source pixel of every stored float, relative to the crop centre
    int nzone, nchunk, nquad_total, max_rows;
};

struct ZonePlanHost {
    std::vector<ZoneDesc> zones;
    std::vector<int2> rowtab;
    std::vector<int> pixtab;
    int S = 0, nw = 0, nchunk = 4, nquad_total = 0, max_rows = 0, max_pix = 0;
    size_t lds_bytes = 0;
    bool ok = false;
};

inline size_t zone_lds_overhead(int nw, int maxrin, int n_qtab, int max_rows);

// Cuts the rings into zones (from the outside in) whose pixel sets fit `budget` floats.  The pixel set of a group of rings is
// found by stamping: sample (dx, dy) of a ring under a sampling centre c + (ox, oy), |ox|, |oy| <= S, c within [0, 1) of the
// crop's integer centre, reads columns floor(dx + ox + frac) and + 1 -- columns floor(dx) - S - 1 .. floor(dx) + S + 3 cover that
// with a pixel to spare on either side for the rounding of the float additions -- and rows alike.
inline bool build_zone_plan(const Geometry &g, int S, int nw, int n_qtab, ZonePlanHost &zp)
{
    zp = ZonePlanHost();
    zp.S = S; zp.nw = nw;
    if (!g.quad_aligned || g.nring < 4 || g.maxrin > 1024 || g.maxrin < 64) return false;
    const int Hmax = g.last_ring + S + 6, W = 2 * Hmax + 1;
    const int nquad = (g.nring + 3) / 4;
    const size_t lds_total = 160 * 1024 - 1024;
    const size_t over = zone_lds_overhead(nw, g.maxrin, n_qtab, W);
    if (over + 16 * 1024 > lds_total) return false;
    const int budget = (int)((lds_total - over) / sizeof(float));
    // per ring quad: row intervals [lo, hi] of needed columns, left and right of the centre column separately is not needed --
    // a full mask per zone is cheap enough (W^2 bytes)
    std::vector<unsigned char> mask((size_t)W * W, 0), trial;
    auto stamp_quad = [&](std::vector<unsigned char> &m, int q) {
        for (int i = 4 * q; i < std::min(4 * q + 4, g.nring); i++) {
            const int kc = g.numr[3 * i + 1] - 1, n = g.numr[3 * i + 2];
            for (int j = 0; j < n; j++) {
                const int fx = (int)std::floor(g.samp_dx[kc + j]), fy = (int)std::floor(g.samp_dy[kc + j]);
                const int x0 = fx - S - 1 + Hmax, x1 = fx + S + 3 + Hmax, y0 = fy - S - 1 + Hmax, y1 = fy + S + 3 + Hmax;
                if (x0 < 0 || y0 < 0 || x1 >= W || y1 >= W) return false;
                for (int y = y0; y <= y1; y++) memset(&m[(size_t)y * W + x0], 1, (size_t)(x1 - x0 + 1));
            }
        }
        return true;
    };
    // stored pixels of a mask: per row the hull of the marked columns, minus the hole around the centre column when it is unmarked
    auto count_pix = [&](const std::vector<unsigned char> &m) {
        int tot = 0;
        for (int y = 0; y < W; y++) {
            const unsigned char *r = &m[(size_t)y * W];
            int lo = 0, hi = W - 1;
            while (lo < W && !r[lo]) lo++;
            if (lo == W) continue;
            while (!r[hi]) hi--;
            tot += hi - lo + 1;
            if (!r[Hmax]) {
                int a = Hmax, b = Hmax;
                while (a - 1 >= lo && !r[a - 1]) a--;
                while (b + 1 <= hi && !r[b + 1]) b++;
                tot -= b - a + 1;
            }
        }
        return tot;
    };
    auto emit_zone = [&](const std::vector<unsigned char> &m, int q0, int q1) {      // quads [q0, q1)
        ZoneDesc z{};
        z.ring0 = 4 * q0; z.nquad = q1 - q0; z.quad0 = q0;
        int ylo = 0, yhi = W - 1;
        auto row_any = [&](int y) { const unsigned char *r = &m[(size_t)y * W]; for (int x = 0; x < W; x++) if (r[x]) return true; return false; };
        while (ylo < W && !row_any(ylo)) ylo++;
        while (yhi >= 0 && !row_any(yhi)) yhi--;
        z.H = std::max(Hmax - ylo, yhi - Hmax);
        z.nrow = 2 * z.H + 1;
        z.pix_off = (int)zp.pixtab.size(); z.row_off = (int)zp.rowtab.size();
        int base = 0;
        for (int yy = -z.H; yy <= z.H; yy++) {
            const unsigned char *r = &m[(size_t)(yy + Hmax) * W];
            int lo = 0, hi = W - 1;
            while (lo < W && !r[lo]) lo++;
            if (lo == W) { zp.rowtab.push_back(make_int2(0, 0)); continue; }      // (a row no sample reads)
            while (!r[hi]) hi--;
            int a = Hmax + 1, b = Hmax;      // hole [a, b] (empty)
            if (!r[Hmax]) {
                a = Hmax; b = Hmax;
                while (a - 1 >= lo && !r[a - 1]) a--;
                while (b + 1 <= hi && !r[b + 1]) b++;
            }
            if (a <= b && a > lo && b < hi) {
                const int nl = a - lo;                       // left segment columns lo .. a - 1, right b + 1 .. hi
                zp.rowtab.push_back(make_int2(base - (lo - Hmax), base + nl - (b + 1 - Hmax)));
                for (int x = lo; x < a; x++) zp.pixtab.push_back((int)(((unsigned)(yy & 0xffff) << 16) | (unsigned)((x - Hmax) & 0xffff)));
                for (int x = b + 1; x <= hi; x++) zp.pixtab.push_back((int)(((unsigned)(yy & 0xffff) << 16) | (unsigned)((x - Hmax) & 0xffff)));
                base += nl + (hi - b);
            } else {
                // no hole (or a hole that reaches an end of the row: keep the hull)
                zp.rowtab.push_back(make_int2(base - (lo - Hmax), base - (lo - Hmax)));
                for (int x = lo; x <= hi; x++) zp.pixtab.push_back((int)(((unsigned)(yy & 0xffff) << 16) | (unsigned)((x - Hmax) & 0xffff)));
                base += hi - lo + 1;
            }
        }
        zp.rowtab.push_back(make_int2(0, 0));       // row H + 1: read by nobody, keeps rt[ry + 1] inside the table
        z.npix = base;
        zp.max_rows = std::max(zp.max_rows, z.nrow);
        zp.max_pix = std::max(zp.max_pix, z.npix);
        zp.zones.push_back(z);
    };
    int q1 = nquad;
    while (q1 > 0) {
        std::fill(mask.begin(), mask.end(), 0);
        int q0 = q1;
        while (q0 > 0) {
            trial = mask;
            if (!stamp_quad(trial, q0 - 1)) return false;
            if (count_pix(trial) > budget) break;
            mask.swap(trial);
            q0--;
        }
        if (q0 == q1) return false;       // one ring quad alone exceeds the LDS
        emit_zone(mask, q0, q1);
        q1 = q0;
    }
    zp.nquad_total = nquad;
    zp.lds_bytes = zone_lds_overhead(nw, g.maxrin, n_qtab, zp.max_rows) + (size_t)((zp.max_pix + 3) & ~3) * sizeof(float);
    zp.ok = zp.lds_bytes <= lds_total;
    return zp.ok;
}

// waves per workgroup.  8: 256 registers per wave, no spills, 3 zones at configs[4] -- measured 9.8 ms per chunk of 330 particles against
// 13.3 ms with 12 waves (170 registers: 27 spilled) and 14.2 ms with 16 (scripts/dev/zone_nw.sh; the global-tap kernel: 13.25 ms)
#ifndef RA_ZONE_NW
#define RA_ZONE_NW 8
#endif
#define RA_ZONE_WB 544        // float2 per wave: the 512 padded slots of the register transforms (slot sl at sl + (sl >> 4))

// LDS of polar_zone_kernel: [nw][RA_ZONE_WB] float2 wave buffers | [maxrin] twiddles | [n_qtab] (sin, cos) tables | [64] W_64^{n1 k0} |
// [max_rows + 2] row table | [2 nbins + 2] ints bin_offp, bin_first | image
inline size_t zone_lds_overhead(int nw, int maxrin, int n_qtab, int max_rows)
{
    return ((size_t)nw * RA_ZONE_WB + maxrin + n_qtab + 64 + (max_rows + 2) + (maxrin / 2 + 2)) * sizeof(float2);
}

// Util::bilinear at 1-based (xo, yo) with the four taps out of a zone's LDS image: bilinear_1b's operations in its order (no
// contraction; v_fract_f32 is x - (float)(int)x for the positive coordinates here, exactly), so the sample equals
// polar_generic_kernel's bit for bit wherever that kernel's taps lie inside the image
// (the row table holds BYTE offsets into the image: tap address = (column left of the centre ? offL : offR) + 4 column)
__device__ __forceinline__ float zone_bilinear(const float *img_s, const int2 *rt_s, int cxi, int cyi_mH, float xo, float yo)
{
#pragma clang fp contract(off)
    const int ix = (int)xo, iy = (int)yo;
    const float ydif = __builtin_amdgcn_fractf(yo), xdif = __builtin_amdgcn_fractf(xo);
    const int rx = ix - cxi;
    const int2 *rp = rt_s + (iy - cyi_mH);
    const int2 t0 = rp[0], t1 = rp[1];
    const bool left = rx < 0;
    const char *ib = reinterpret_cast<const char *>(img_s) + 4 * rx;
    const float *q0 = reinterpret_cast<const float *>(ib + (left ? t0.x : t0.y)), *q1 = reinterpret_cast<const float *>(ib + (left ? t1.x : t1.y));
    const float f00 = q0[0], f10 = q0[1], f01 = q1[0], f11 = q1[1];          // (4-byte aligned pairs: ds_read2_b32)
#ifdef RA_ZONE_LERP
    // the interpolant as two fused lerps, row pair first (bilinear_pad of the particle-resident kernels: 4 instructions instead of 9,
    // within 1 ulp of the taps of Util::bilinear's own operation order)
    const v2f r0 = {f00, f10}, r1 = {f01, f11};
    const v2f gg = __builtin_elementwise_fma((v2f){ydif, ydif}, r1 - r0, r0);
    return __builtin_fmaf(xdif, gg.y - gg.x, gg.x);
#else
    return f00 + ydif * (f01 - f00) + xdif * (f10 - f00 + ydif * (f11 - f10 - f01 + f00));
#endif
}

__device__ __forceinline__ float2 *zone_slot(float2 *wb, int sl) { return wb + sl + (sl >> 4); }

// One ring of NR = 128 R1 samples (1024 | 512) by one wave, everything between the taps and the spectrum in registers:
//   samples      lane l holds z[l + 64 k2] = (x[2 (l + 64 k2)], x[2 (l + 64 k2) + 1]), k2 < R1 -- the four quadrants of a table
//                position share its (sin, cos) read and radius multiply (alrl_ms's mirroring, as ring_pos)
//   transform    H = 64 R1 points as R1 x 8 x 8: DFT-R1 over k2 in registers, twiddle W_H^{n0 l}; the 64-point transforms of
//                the R1 rows as two DFT-8 with a transpose through the wave's LDS buffer each (k = 64 k2 + 8 k1 + k0,
//                n = n0 + R1 n1 + 8 R1 n2); lane 8 n0 + n1 ends with Z[m + 8 R1 n2], n2 < 8, m = n0 + R1 n1 (lanes < 8 R1)
//   split step   X_k from (Z_k, Z_{H-k}): the partner values sit in lane(8 R1 - m) at register 7 - n2 (ds_bpermute); lane 0 (m = 0)
//                pairs inside itself and also holds the Nyquist term X_H
// X[t] = X_{m + 8 R1 t}; nyq = X_H (lane 0).  av / sq += the ring's Normalize_ring sums (weighted once per ring).
template <int R1>
__device__ __forceinline__ void zone_ring_fast(const DevGeom &g, const float *img_s, const int2 *rt_s, const float2 *tw_s, const float2 *qt,
                                               const float2 *twb_s, float2 *wb, int lane, int cxi, int cyi_mH, float cx, float cy, float fr,
                                               float wt, float &av, float &sq, float2 (&X)[8], float &nyq, bool no_taps)
{
    constexpr int H = 64 * R1, NB = R1 == 8 ? 2 : 1;        // table positions per lane and sample parity: 2 l + u (+ 128)
    static_assert(R1 == 8 || R1 == 4, "rings of 1024 or 512 samples");
    float2 v[R1];
    float a = 0.f, qq = 0.f;
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float2 sc = qt[2 * lane + u + 128 * b];
            const float x = __fmul_rn(sc.x, fr), y = __fmul_rn(sc.y, fr);
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const int k2 = R1 == 8 ? 2 * qd + b : qd;
                const float px = qd == 0 ? x : qd == 1 ? y : qd == 2 ? -x : -y;
                const float py = qd == 0 ? y : qd == 1 ? -x : qd == 2 ? -y : x;
                const float sv = no_taps ? px : zone_bilinear(img_s, rt_s, cxi, cyi_mH, __fadd_rn(px, cx), __fadd_rn(py, cy));
                if (u == 0) v[k2].x = sv; else v[k2].y = sv;
                a += sv; qq += sv * sv;
            }
        }
    av += a * wt; sq += qq * wt;
    // ---- stage 1: DFT-R1 over k2, twiddle W_H^{n0 lane}
    Dft<-1, R1>::run(v);
    const int sh = g.lg_maxrin - (R1 == 8 ? 9 : 8);
#pragma unroll
    for (int n0 = 1; n0 < R1; n0++) v[n0] = cmul(v[n0], tw_s[(n0 * lane) << sh]);
#pragma unroll
    for (int n0 = 0; n0 < R1; n0++) *zone_slot(wb, 64 * n0 + lane) = v[n0];
    wave_lds_sync();
    float2 z[8];
    const bool act = lane < 8 * R1;
    const int lc = act ? lane : 0;
    // ---- stage 2: lane = 8 n0 + k0: DFT-8 over k1, twiddle W_64^{n1 k0}, back into the slots it read
    {
        const int n0 = lc >> 3, k0 = lc & 7;
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) z[k1] = *zone_slot(wb, 64 * n0 + 8 * k1 + k0);
        Dft<-1, 8>::run(z);
#pragma unroll
        for (int n1 = 1; n1 < 8; n1++) z[n1] = cmul(z[n1], twb_s[n1 * 8 + k0]);
        if (act) {
#pragma unroll
            for (int n1 = 0; n1 < 8; n1++) *zone_slot(wb, 64 * n0 + 8 * n1 + k0) = z[n1];
        }
    }
    wave_lds_sync();
    // ---- stage 3: lane = 8 n0 + n1: DFT-8 over k0 (8 consecutive slots) -> Z[m + 8 R1 n2]
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) z[k0] = *zone_slot(wb, 8 * lc + k0);
    Dft<-1, 8>::run(z);
    wave_lds_sync();                         // the buffer is free for the next ring (or the re-mapping of this one)
    // ---- split step
    const int m = (lc >> 3) + R1 * (lc & 7);
    const int mp = (8 * R1 - m) & (8 * R1 - 1);                     // partner residue; m = 0 pairs with itself
    const int pl = 8 * (mp & (R1 - 1)) + mp / R1;
    const int lsh = g.lg_maxrin - (R1 == 8 ? 10 : 9);              // twiddle e^{-2 pi i k / NR} = tw[k << lsh]
#pragma unroll
    for (int t = 0; t < 8; t++) {
        // Z_{H-k}, k = m + 8 R1 t: register 7 - t of the partner lane; lane 0: its own register 8 - t (t = 0: Z_H = Z_0)
        float2 zm;
        zm.x = __shfl(z[7 - t].x, pl); zm.y = __shfl(z[7 - t].y, pl);
        if (m == 0) zm = z[(8 - t) & 7];
        const int k = m + 8 * R1 * t;
        v2f xk, xm;
        vsplit_pair(to_v(z[t]), to_v(zm), to_v(tw_s[k << lsh]), xk, xm);
        X[t] = to_f2(xk);
    }
    nyq = z[0].x - z[0].y;
    if (m == 0) X[0] = make_float2(z[0].x + z[0].y, 0.f);
    (void)H;
}

// Polar2Dm (bilinear) + Normalize_ring partial sums + Frngs of the particles of a chunk, image taps from LDS.
//   grid = n * nzone * nchunk workgroups of NW waves: block -> (particle, zone, chunk of search offsets)
//   out: the A blocks of polar_generic_kernel (entry e = p * ent_stride + s lives in block e >> 2, slot e & 3)
//   stats_part [n * ent_stride][nquad_total] {sum w v, sum w v^2} of every (entry, ring quad)
// Work item of a wave = (search offset, ring quad).  Rings of 1024 / 512 samples go through zone_ring_fast; shorter ones are
// sampled into the wave's buffer and transformed there (wave_fft, split_bin).  The lanes of a quad store the bins of ITS longest
// ring's register layout (bins m + 8 R1 t of lane 8 n0 + n1; natural order, lane + 64 t, when the quad's rings have at most 256
// samples); a ring of another length inside the quad (one quad at configs[4]: rings 80 .. 83) passes its spectrum through the
// wave's buffer in natural order.
template <int NW>
__global__ __launch_bounds__(64 * NW) void polar_zone_kernel(DevGeom g, ZonePlanDev zp, const float *__restrict__ images,
                                                             const float *__restrict__ state, int n, float *__restrict__ out,
                                                             float2 *__restrict__ stats_part)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *wb = reinterpret_cast<float2 *>(lds) + (size_t)wave * RA_ZONE_WB;
    float2 *tw_s = reinterpret_cast<float2 *>(lds) + (size_t)NW * RA_ZONE_WB;
    float2 *qt_s = tw_s + g.maxrin;
    float2 *twb_s = qt_s + g.n_qtab;
    int2 *rt_s = reinterpret_cast<int2 *>(twb_s + 64);
    int *bo_s = reinterpret_cast<int *>(rt_s + zp.max_rows + 2);          // bin_offp [nbins + 1]
    int *bf_s = bo_s + g.nbins + 1;                                        // bin_first [nbins]
    float *img_s = reinterpret_cast<float *>(reinterpret_cast<float2 *>(bo_s) + (g.maxrin / 2 + 2));

    const int per_p = zp.nzone * zp.nchunk;
    const int p = blockIdx.x / per_p, rem = blockIdx.x - p * per_p, zi = rem / zp.nchunk, ch = rem - zi * zp.nchunk;
    if (p >= n) return;
    const ZoneDesc zd = zp.zones[zi];
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    const float cxf = (float)g.cnx + w.sxi, cyf = (float)g.cnx + w.syi;         // sampling centre of offset (0, 0), 1-based
    const int cxi = (int)cxf, cyi = (int)cyf;                                   // crop centre (1-based pixel), >= 1: truncation = floor

    for (int i = tid; i < g.maxrin; i += 64 * NW) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += 64 * NW) qt_s[i] = g.qtab[i];
    for (int i = tid; i < 64; i += 64 * NW) twb_s[i] = g.tw[((i >> 3) * (i & 7) * (g.maxrin >> 6)) & (g.maxrin - 1)];      // W_64^{n1 k0}
    for (int i = tid; i <= zd.nrow; i += 64 * NW) { const int2 t = zp.rowtab[zd.row_off + i]; rt_s[i] = make_int2(4 * t.x, 4 * t.y); }
    for (int i = tid; i <= g.nbins; i += 64 * NW) bo_s[i] = g.bin_offp[i];
    for (int i = tid; i < g.nbins; i += 64 * NW) bf_s[i] = g.bin_first[i];
    if (!RA_DBG(g, 4096)) {
        const float *img = images + (size_t)p * g.nx * g.nx;
        const int *pt = zp.pixtab + zd.pix_off;
        // eight pixels per thread in flight; out-of-image sources (only offsets outside the particle's window read them) are clamped
        for (int i0 = tid; i0 < zd.npix; i0 += 8 * 64 * NW) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = min(i0 + u * 64 * NW, zd.npix - 1);
                const int code = pt[i];
                const int dy = code >> 16, dx = (int)(short)(code & 0xffff);
                const int sy = min(max(cyi - 1 + dy, 0), g.nx - 1), sx = min(max(cxi - 1 + dx, 0), g.nx - 1);
                v[u] = img[sy * g.nx + sx];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * 64 * NW < zd.npix) img_s[i0 + u * 64 * NW] = v[u];
        }
    }
    __syncthreads();

    // offsets of this workgroup: a quarter of the offset list -- or, with live-offset lists, of the particle's in-window offsets
    const bool live = g.ent_base != nullptr;
    const int e0 = live ? g.ent_base[p] : 0, nlive = live ? g.ent_base[p + 1] - e0 : g.nshift;
    const int per = (nlive + zp.nchunk - 1) / zp.nchunk;
    const int s_lo = ch * per, s_hi = min(nlive, s_lo + per);
    const int nitem = max(0, s_hi - s_lo) * zd.nquad;
    const int cyi_mH = cyi - zd.H;
    const bool no_taps = RA_DBG(g, 512);
    for (int it = wave; it < nitem; it += NW) {
        const int sl = it / zd.nquad, q = it - sl * zd.nquad;
        const int jl = s_lo + sl, c4 = zd.ring0 + 4 * q;
        const int s = live ? live_shift(g, w, jl) : jl;
        const size_t e = live ? (size_t)e0 + jl : (size_t)p * g.ent_stride + jl;
        float *blk = out + (e >> 2) * g.a_blk;
        const int slot = (int)(e & 3);
        const float cx = cxf + g.shift_x[s], cy = cyf + g.shift_y[s];
        // bins of the lane: the register layout of the quad's longest ring
        const int Lq = g.ringinfo[min(c4 + 3, g.nring - 1)].z, hq = Lq >> 1;
        const int R1q = Lq >> 7;                                             // 8 | 4 for the register layouts
        const bool regmap = Lq >= 512;
        const int mq = (lane >> 3) + R1q * (lane & 7);
        const bool lane_on = !regmap || lane < 8 * R1q;
        auto bin_of = [&](int t) { return regmap ? mq + 8 * R1q * t : lane + 64 * t; };      // t < 8
        float av = 0.f, sq = 0.f;
        float2 xq[4][8];
        float nyq[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = c4 + r;
            nyq[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; t++) xq[r][t] = make_float2(0.f, 0.f);
            if (i >= g.nring) continue;        // uniform
            const int4 ri = g.ringinfo[i];
            const int nlen = ri.z, h = nlen >> 1;
            const float wt = g.ringw[i], fr = (float)ri.y;
            const float2 *qt = qt_s + ri.w;
            if (nlen == 1024 || nlen == 512) {
                float2 X[8];
                float ny;
                if (nlen == 1024) zone_ring_fast<8>(g, img_s, rt_s, tw_s, qt, twb_s, wb, lane, cxi, cyi_mH, cx, cy, fr, wt, av, sq, X, ny, no_taps);
                else zone_ring_fast<4>(g, img_s, rt_s, tw_s, qt, twb_s, wb, lane, cxi, cyi_mH, cx, cy, fr, wt, av, sq, X, ny, no_taps);
                if (nlen == Lq) {
#pragma unroll
                    for (int t = 0; t < 8; t++) xq[r][t] = X[t];
                    nyq[r] = ny;
                } else {
                    // a shorter ring inside the quad: spectrum to the wave's buffer in natural order, read back in the quad's layout
                    const int R1r = nlen >> 7, mr = (lane >> 3) + R1r * (lane & 7);
                    if (lane < 8 * R1r) {
#pragma unroll
                        for (int t = 0; t < 8; t++) wb[mr + 8 * R1r * t] = X[t];
                        if (lane == 0) wb[h] = make_float2(ny, 0.f);
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int k = bin_of(t);
                        xq[r][t] = (lane_on && k <= h) ? wb[min(k, h)] : make_float2(0.f, 0.f);
                    }
                    wave_lds_sync();
                }
            } else {
                // rings of up to 256 samples: samples -> wave buffer, Stockham transform between its two halves, split step by index
                float *xr = reinterpret_cast<float *>(wb);
                float a = 0.f, qq = 0.f;
                for (int j0 = lane; j0 < nlen; j0 += 256) {
                    float sv[4];
                    const int lt = nlen >> 2, lgl = 31 - __clz(lt);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = min(j0 + 64 * u, nlen - 1);
                        const float2 d = ring_pos(qt, lt, lgl, fr, j);
                        sv[u] = no_taps ? d.x : zone_bilinear(img_s, rt_s, cxi, cyi_mH, d.x + cx, d.y + cy);      // bilinear_1b(img, nx, d.x + cx, d.y + cy)
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j0 + 64 * u < nlen) {
                            xr[j0 + 64 * u] = sv[u];
                            a += sv[u]; qq += sv[u] * sv[u];
                        }
                }
                av += a * wt; sq += qq * wt;
                wave_lds_sync();
                const float2 *Z = (h >= 2) ? wave_fft<-1>(wb, wb + 128, h, tw_s, g.maxrin, lane) : wb;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int k = bin_of(t);
                    xq[r][t] = (lane_on && k <= h) ? split_bin(Z, min(k, h), h, tw_s[min(k, h) * (g.maxrin / nlen)]) : make_float2(0.f, 0.f);
                }
                if (regmap) nyq[r] = 0.f;          // (a ring of <= 256 samples has no bin hq >= 256 ... except h == hq, excluded: Lq >= 512 > nlen)
                wave_lds_sync();
            }
        }
        // panel pieces: 16 bytes of four rings per (bin, Re | Im)
        if (!RA_DBG(g, 1024)) {
#pragma unroll
            for (int t = 0; t < 9; t++) {
                // t = 8: the Nyquist bin hq of the register layouts (lane 0); natural layout: bin lane + 64 t of t < 8 covers everything
                const int k = t < 8 ? bin_of(t) : hq;
                const bool on = t < 8 ? (lane_on && k < g.nbins && (regmap ? k < hq : k <= hq)) : (regmap && lane == 0);
                if (!on) continue;
                const int bo = bo_s[k], nsk = (bo_s[k + 1] - bo) >> 2, i0al = bf_s[k] & ~3;
                if (c4 < i0al) continue;
                const int j0 = c4 - i0al;
                const int kk = j0 / nsk, s4 = j0 - kk * nsk;
                float *dst = blk + bo * 8 + 8 * slot + (s4 >> 2) * 128 + kk * 32;
                if (t < 8) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(xq[0][t & 7].x, xq[1][t & 7].x, xq[2][t & 7].x, xq[3][t & 7].x);
                    *reinterpret_cast<float4 *>(dst + 4) = make_float4(xq[0][t & 7].y, xq[1][t & 7].y, xq[2][t & 7].y, xq[3][t & 7].y);
                } else {
                    *reinterpret_cast<float4 *>(dst) = make_float4(nyq[0], nyq[1], nyq[2], nyq[3]);
                    *reinterpret_cast<float4 *>(dst + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        av = wave_sum_dpp(av); sq = wave_sum_dpp(sq);      // fixed order: reproducible
        if (lane == 0) stats_part[e * zp.nquad_total + zd.quad0 + q] = make_float2(av, sq);
    }
}

// Normalize_ring statistics of every entry from the per-quad partial sums of polar_zone_kernel, added in quad order
__global__ void polar_stats_kernel(DevGeom g, const float2 *__restrict__ part, int nent, int nquad, float2 *__restrict__ stats)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (g.ent_total ? *g.ent_total : nent)) return;
    float av = 0.f, sq = 0.f;
    for (int k = 0; k < nquad; k++) {
        const float2 v = part[(size_t)e * nquad + k];
        av += v.x; sq += v.y;
    }
    float avg = 0.f, rsg = 1.f;
    if (g.norm_ring) {
        const float nn = g.nn_weight;
        avg = av / nn;
        rsg = 1.0f / sqrtf((sq - av * av / nn) / nn);
    }
    stats[e] = make_float2(avg, rsg);
}

}  // namespace ralign
