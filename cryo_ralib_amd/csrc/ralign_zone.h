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

// LDS of polar_zone_kernel: [nw][maxrin] float2 transform buffers | [maxrin] twiddles | [n_qtab] (sin, cos) tables |
// [max_rows + 2] row table | image
inline size_t zone_lds_overhead(int nw, int maxrin, int n_qtab, int max_rows)
{
    return ((size_t)nw * maxrin + maxrin + n_qtab + (max_rows + 2)) * sizeof(float2);
}

// Cuts the rings into zones (from the outside in) whose pixel sets fit `budget` floats.  The pixel set of a group of rings is
// found by stamping: sample (dx, dy) of a ring under a sampling centre c + (ox, oy), |ox|, |oy| <= S, c within [0, 1) of the
// crop's integer centre, reads columns floor(dx + ox + frac) and + 1 -- columns floor(dx) - S - 1 .. floor(dx) + S + 3 cover that
// with a pixel to spare on either side for the rounding of the float additions -- and rows alike.
inline bool build_zone_plan(const Geometry &g, int S, int nw, int n_qtab, ZonePlanHost &zp)
{
    zp = ZonePlanHost();
    zp.S = S; zp.nw = nw;
    if (!g.quad_aligned || g.nring < 4 || g.maxrin > 2048) return false;
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

#define RA_ZONE_NW 8

// Polar2Dm (bilinear) + Normalize_ring partial sums + Frngs of the particles of a chunk, image taps from LDS.
//   grid = n * nzone * nchunk workgroups of NW waves: block -> (particle, zone, chunk of search offsets)
//   out: the A blocks of polar_generic_kernel (entry e = p * ent_stride + s lives in block e >> 2, slot e & 3)
//   stats_part [n * ent_stride][nquad_total] {sum w v, sum w v^2} of every (entry, ring quad)
// Util::bilinear at 1-based (xo, yo) with the four taps out of a zone's LDS image: bilinear_1b's operations in its order (no
// contraction), so the sample equals polar_generic_kernel's bit for bit wherever that kernel's taps lie inside the image
__device__ __forceinline__ float zone_bilinear(const float *img_s, const int2 *rt_s, int cxi, int cyi_mH, float xo, float yo)
{
#pragma clang fp contract(off)
    const int ix = (int)xo, iy = (int)yo;
    const float ydif = yo - iy, xdif = xo - ix;
    const int rx = ix - cxi, ry = iy - cyi_mH;
    const int2 t0 = rt_s[ry], t1 = rt_s[ry + 1];
    const int a0 = (rx < 0 ? t0.x : t0.y) + rx, a1 = (rx < 0 ? t1.x : t1.y) + rx;
    const float f00 = img_s[a0], f10 = img_s[a0 + 1], f01 = img_s[a1], f11 = img_s[a1 + 1];
    return f00 + ydif * (f01 - f00) + xdif * (f10 - f00 + ydif * (f11 - f10 - f01 + f00));
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void polar_zone_kernel(DevGeom g, ZonePlanDev zp, const float *__restrict__ images,
                                                             const float *__restrict__ state, int n, float *__restrict__ out,
                                                             float2 *__restrict__ stats_part)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *bx = reinterpret_cast<float2 *>(lds) + (size_t)wave * g.maxrin;     // two buffers of maxrin/2 complex
    float2 *by = bx + g.maxrin / 2;
    float2 *tw_s = reinterpret_cast<float2 *>(lds) + (size_t)NW * g.maxrin;
    float2 *qt_s = tw_s + g.maxrin;
    int2 *rt_s = reinterpret_cast<int2 *>(qt_s + g.n_qtab);
    float *img_s = reinterpret_cast<float *>(rt_s + zp.max_rows + 2);

    const int per_p = zp.nzone * zp.nchunk;
    const int p = blockIdx.x / per_p, rem = blockIdx.x - p * per_p, zi = rem / zp.nchunk, ch = rem - zi * zp.nchunk;
    if (p >= n) return;
    const ZoneDesc zd = zp.zones[zi];
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    const float cxf = (float)g.cnx + w.sxi, cyf = (float)g.cnx + w.syi;         // sampling centre of offset (0, 0), 1-based
    const int cxi = (int)cxf, cyi = (int)cyf;                                   // crop centre (1-based pixel), >= 1: truncation = floor

    for (int i = tid; i < g.maxrin; i += 64 * NW) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += 64 * NW) qt_s[i] = g.qtab[i];
    for (int i = tid; i <= zd.nrow; i += 64 * NW) rt_s[i] = zp.rowtab[zd.row_off + i];
    {
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

    // panel addressing of the lane's bins k = lane + 64 t (polar_generic_kernel, quad_aligned branch)
    constexpr int T = 9;
    int base0[T], i0al[T], nsk[T];
    float rns[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int k = min(lane + 64 * t, g.nbins - 1);
        nsk[t] = (g.bin_offp[k + 1] - g.bin_offp[k]) >> 2;
        rns[t] = 1.0f / (float)nsk[t];
        base0[t] = g.bin_offp[k] * 8;
        i0al[t] = g.bin_first[k] & ~3;
    }

    const int per = (g.nshift + zp.nchunk - 1) / zp.nchunk;
    const int s_lo = ch * per, s_hi = min(g.nshift, s_lo + per);
    const int nitem = max(0, s_hi - s_lo) * zd.nquad;
    float *xr = reinterpret_cast<float *>(bx);
    for (int it = wave; it < nitem; it += NW) {
        const int sl = it / zd.nquad, q = it - sl * zd.nquad;
        const int s = s_lo + sl, c4 = zd.ring0 + 4 * q;
        const size_t e = (size_t)p * g.ent_stride + s;
        float *blk = out + (e >> 2) * g.a_blk;
        const int slot = (int)(e & 3);
        const float cx = cxf + g.shift_x[s], cy = cyf + g.shift_y[s];
        float av = 0.f, sq = 0.f;
        float2 xq[4][T];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = c4 + r;
            if (i < g.nring) {        // uniform
                const int4 ri = g.ringinfo[i];
                const int nlen = ri.z, h = nlen >> 1, lt = nlen >> 2, lgl = 31 - __clz(lt);
                const float wt = g.ringw[i], fr = (float)ri.y;
                const float2 *qt = qt_s + ri.w;
                float a = 0.f, qq = 0.f;
                for (int j0 = lane; j0 < nlen; j0 += 256) {
                    float sv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = min(j0 + 64 * u, nlen - 1);
                        const float2 d = ring_pos(qt, lt, lgl, fr, j);
                        sv[u] = RA_DBG(g, 512) ? d.x : zone_bilinear(img_s, rt_s, cxi, cyi - zd.H, d.x + cx, d.y + cy);      // bilinear_1b(img, nx, d.x + cx, d.y + cy)
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (j0 + 64 * u < nlen) {
                            xr[j0 + 64 * u] = sv[u];
                            a += sv[u] * wt; qq += sv[u] * sv[u] * wt;
                        }
                }
                av += a; sq += qq;
                wave_lds_sync();
                const float2 *Z = (h >= 2 && !RA_DBG(g, 256)) ? wave_fft<-1>(bx, by, h, tw_s, g.maxrin, lane) : bx;
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int k = lane + 64 * t;
                    xq[r][t] = RA_DBG(g, 2048) ? Z[k & 255] : k <= h ? split_bin(Z, k, h, tw_s[k * (g.maxrin / nlen)]) : make_float2(0.f, 0.f);
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
                float *dst = blk + base0[t] + 8 * slot + (s4 >> 2) * 128 + kk * 32;
                if (RA_DBG(g, 1024) && xq[0][t].x != 1.2345f) continue;          // profiling: no panel stores
                *reinterpret_cast<float4 *>(dst) = make_float4(xq[0][t].x, xq[1][t].x, xq[2][t].x, xq[3][t].x);
                *reinterpret_cast<float4 *>(dst + 4) = make_float4(xq[0][t].y, xq[1][t].y, xq[2][t].y, xq[3][t].y);
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
    if (e >= nent) return;
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
