// Host-side geometry of the alignment path: ring layout (Numrinit/ringwe), polar sample
// table (alrl_ms), search offsets, and the bin-major contraction layout the CCF kernel uses.
// Mirrors what the reference takes from SPHIRE (test_mref_gpu_align.py:348-349) and what its
// CUDA library builds in gpu_aln_common.cu:39-84.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

// Environment switches.  The ones a user or a test may set (kernel-path selection, RALIGN_INFO, RALIGN_REFINE; README "Environment")
// are read with getenv in every build.  EXPERIMENT switches -- layout strides, job orders, block shapes that were measured once and
// whose built-in value is the fastest -- exist only in profiling builds (-DRALIGN_PROFILE_SWITCHES): in the shipped library
// RA_EXP_ENV is a null constant, so neither the lookup nor the name is in the binary.
#include <cstdlib>
#ifdef RALIGN_PROFILE_SWITCHES
#define RA_EXP_ENV(name) getenv(name)
#else
#define RA_EXP_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace ralign {

inline int ra_atoi(const char *v) { return v ? atoi(v) : 0; }

// floats between consecutive rings in the padded ring buffer, beyond the ring's n samples: 2 would do for the
// in-place real FFT (slot of X_{n/2}); 16 makes every ring stride (24, 32, 48, 80, 144, 272 floats) an odd multiple of 8
// or 16 dwords, so that the 2 ... 4 rings a wave-instruction touches at once sit in different LDS bank windows
// instead of on top of each other (256 + 2 = 258 = 2 mod 64 put 8 rings within 14 banks of each other)
constexpr int kRingPad = 16;

struct Geometry {
    int nx = 0;
    int first_ring = 1, last_ring = 0, skip = 1;
    int nring = 0, maxrin = 0, lcirc = 0;
    int nbins = 0;                 // maxrin/2 + 1 complex bins of the CCF spectrum
    int LB = 0;                    // sum over rings of (n/2+1): complex entries per particle-shift
    int LBP = 0;                   // same with every bin's ring count padded to a multiple of 4
    int ring_pad = kRingPad;       // floats between consecutive rings of a ring buffer (>= 2: the in-place R2C needs the slot of X_{n/2})
    int lring = 0;                 // lcirc + ring_pad*nring floats: ring buffers padded for in-place R2C
    float nn_weight = 0.f;         // Normalize_ring's float-accumulated sum of weights
    std::vector<int> numr;         // (radius, 1-based offset, length) per ring
    std::vector<float> wr;         // ringwe
    std::vector<int> ring_off;     // offset of ring i in the padded ring buffer (floats)
    std::vector<float> samp_dx, samp_dy;  // [lcirc] sample offsets relative to the centre
    std::vector<int> samp_dst;     // [lcirc] destination float index in the padded ring buffer
    std::vector<float> samp_w;     // [lcirc] Normalize_ring weight r*2pi/n of the sample's ring
    std::vector<int> bin_first;    // [nbins] first ring that has bin k (rings are a suffix)
    std::vector<int> lead;         // [nbins] empty ring slots in front of bin k's rings (align_ring_quads; else 0)
    bool quad_aligned = false;     // ring quads aligned across bins (align_ring_quads)
    std::vector<int> bin_off;      // [nbins+1] prefix of ring counts     (A layout)
    std::vector<int> bin_offp;     // [nbins+1] prefix of padded counts   (B layout)
    std::vector<int> ent_src;      // [LB] float offset (ring_off[i] + 2k) of entry e=(k,i)
    std::vector<float> ent_wgt;    // [LB] Applyws weight of entry (wr, halved on a short ring's Nyquist)
    std::vector<int> a_src;        // [LBP*8]  see build_operand_tables
    std::vector<int> b_src;        // [LBP*16]
    std::vector<int> ent_apos;     // [LB*2] {A-block float offset of entry e at row 0, chunk width}
    // search offsets
    int nkx = 0, nky = 0, nshift = 0, nshift_pad = 0;
    float step = 1.f;
    std::vector<float> shift_x, shift_y;   // [nshift] EMAN2 loop order: y outer, x inner
};

inline int ilog2_floor(int n) { int l = -1; while (n > 0) { n >>= 1; l++; } return l; }

// sp_alignment.Numrinit(first,last,skip,"F") / ringwe(numr,"F")
inline bool build_rings(Geometry &g, int nx, int first_ring, int last_ring, int skip, int ring_pad = kRingPad)
{
    g.ring_pad = ring_pad;
    const int MAXFFT = 32768;
    const double dpi = 2.0 * M_PI;
    g.nx = nx; g.first_ring = first_ring; g.last_ring = last_ring; g.skip = skip;
    g.numr.clear();
    int lcirc = 1;
    if (first_ring < 1 || last_ring < first_ring || skip < 1) return false;
    for (int k = first_ring; k <= last_ring; k += skip) {
        int jp = (int)(dpi * k + 0.5);
        int ip = 1 << (ilog2_floor(jp) + 1);
        if (k + skip <= last_ring && jp > ip + ip / 2) ip = std::min(MAXFFT, 2 * ip);
        if (k + skip > last_ring && jp > ip + ip / 5) ip = std::min(MAXFFT, 2 * ip);
        g.numr.push_back(k); g.numr.push_back(lcirc); g.numr.push_back(ip);
        lcirc += ip;
    }
    g.nring = (int)g.numr.size() / 3;
    g.lcirc = lcirc - 1;
    g.maxrin = g.numr.back();
    g.nbins = g.maxrin / 2 + 1;
    g.wr.resize(g.nring);
    for (int i = 0; i < g.nring; i++) {
        double n = (double)g.numr[3 * i + 2];
        g.wr[i] = (float)((double)g.numr[3 * i] * dpi / n * (double)g.maxrin / n);
    }
    // ring lengths must be non-decreasing so that the rings holding bin k form a suffix
    for (int i = 1; i < g.nring; i++)
        if (g.numr[3 * i + 2] < g.numr[3 * (i - 1) + 2]) return false;

    // padded ring buffer and the alrl_ms sample table (float arithmetic as in EMAN2)
    g.ring_off.resize(g.nring);
    g.samp_dx.assign(g.lcirc, 0.f); g.samp_dy.assign(g.lcirc, 0.f);
    g.samp_dst.assign(g.lcirc, 0); g.samp_w.assign(g.lcirc, 0.f);
    const double qpi = 2 * atan(1.0);
    float nn = 0.f;
    for (int it = 0; it < g.nring; it++) {
        int inr = g.numr[3 * it], kc = g.numr[3 * it + 1] - 1, l = g.numr[3 * it + 2];
        g.ring_off[it] = kc + g.ring_pad * it;
        int lt = l / 4, nsim = lt - 1;
        double dfi = qpi / (nsim + 1);
        float w = (float)(inr * 2 * M_PI / (float)l);
        auto put = [&](int j, float dx, float dy) {
            g.samp_dx[kc + j] = dx; g.samp_dy[kc + j] = dy;
            g.samp_dst[kc + j] = g.ring_off[it] + j; g.samp_w[kc + j] = w;
        };
        put(0, 0.0f, (float)inr); put(lt, (float)inr, 0.0f);
        put(2 * lt, 0.0f, (float)-inr); put(3 * lt, (float)-inr, 0.0f);
        for (int jt = 1; jt <= nsim; jt++) {
            float fi = (float)(dfi * jt);
            float x = sinf(fi) * inr, y = cosf(fi) * inr;
            put(jt, x, y); put(jt + lt, y, -x); put(jt + 2 * lt, -x, -y); put(jt + 3 * lt, -y, x);
        }
        for (int j = 0; j < l; j++) nn += w;   // Normalize_ring: float accumulation in ring order
    }
    g.nn_weight = nn;
    g.lring = g.lcirc + g.ring_pad * g.nring;

    // bin-major contraction layout
    g.bin_first.assign(g.nbins, 0);
    g.lead.assign(g.nbins, 0); g.quad_aligned = false;
    g.bin_off.assign(g.nbins + 1, 0); g.bin_offp.assign(g.nbins + 1, 0);
    for (int k = 0; k < g.nbins; k++) {
        int i0 = 0;
        while (i0 < g.nring && g.numr[3 * i0 + 2] / 2 < k) i0++;
        g.bin_first[k] = i0;
        int cnt = g.nring - i0;
        g.bin_off[k + 1] = g.bin_off[k] + cnt;
        g.bin_offp[k + 1] = g.bin_offp[k] + ((cnt + 3) / 4) * 4;
    }
    g.LB = g.bin_off[g.nbins]; g.LBP = g.bin_offp[g.nbins];
    g.ent_src.resize(g.LB); g.ent_wgt.resize(g.LB);
    for (int k = 0; k < g.nbins; k++)
        for (int i = g.bin_first[k]; i < g.nring; i++) {
            int e = g.bin_off[k] + (i - g.bin_first[k]);
            int n = g.numr[3 * i + 2];
            g.ent_src[e] = g.ring_off[i] + 2 * k;
            // Applyws: every packed element * wr, the Nyquist slot * 0.5 wr unless n == maxrin
            g.ent_wgt[e] = (k == n / 2 && n != g.maxrin) ? 0.5f * g.wr[i] : g.wr[i];
        }
    return true;
}

// Operand panels of the contraction.  For bin k with KP_k = 4*ns padded rings, ring j of the
// bin is MFMA k-slot (kk = j / ns, s = j % ns).  The ns steps of a lane are stored in chunks of
// width w = 4,...,4,[2],[1]; inside a chunk the order is [kk][row][w], so one w-wide load per
// lane is a fully coalesced wave access (lane = kk*16 + row).  An A block holds 8 rows
// (4 particle-offsets x Re/Im), the B tile 16 columns (8 references x Re/Im).
// panel_pos(ns, rows, kk, row, s): float offset inside the bin's panel.
inline int panel_pos(int ns, int rows, int kk, int row, int s)
{
    int off = 0, s0 = 0;
    for (int i = 0; i < (ns >> 2); i++) {
        if (s < s0 + 4) return off + (kk * rows + row) * 4 + (s - s0);
        off += 4 * rows * 4; s0 += 4;
    }
    if (ns & 2) {
        if (s < s0 + 2) return off + (kk * rows + row) * 2 + (s - s0);
        off += 4 * rows * 2; s0 += 2;
    }
    return off + (kk * rows + row);
}

// Size-generic kernels (large boxes): ring slot j of bin k is ring (bin_first[k] & ~3) + j and every bin holds a multiple of
// 16 slots, so that the four rings of a panel float4 are rings 4c .. 4c+3 for EVERY bin.  polar_generic_kernel can then
// keep four rings in registers and store whole 16-byte panel pieces (its 4-byte scattered stores cost 5x the panel size in
// HBM writes); the price is <= 15 zero slots per bin in the contraction.  Call before build_operand_tables.
inline void align_ring_quads(Geometry &g)
{
    g.quad_aligned = true;
    for (int k = 0; k < g.nbins; k++) {
        g.lead[k] = g.bin_first[k] & 3;
        const int cnt = g.nring - g.bin_first[k] + g.lead[k];
        g.bin_offp[k + 1] = g.bin_offp[k] + (cnt + 15) / 16 * 16;
    }
    g.LBP = g.bin_offp[g.nbins];
}

// a_src: per A-block float, the source offset inside the 4 LDS ring buffers (stride sbuf) | offset slot << 24, -1 = 0
// b_src: per B-tile float, (entry e << 4 | col), -1 = 0
inline void build_operand_tables(Geometry &g, int sbuf)
{
    g.a_src.assign((size_t)g.LBP * 8, -1);
    g.b_src.assign((size_t)g.LBP * 16, -1);
    g.ent_apos.assign((size_t)g.LB * 2, 0);
    for (int k = 0; k < g.nbins; k++) {
        const int cnt = g.bin_off[k + 1] - g.bin_off[k], kp = g.bin_offp[k + 1] - g.bin_offp[k];
        const int ns = kp / 4;
        for (int j = 0; j < cnt; j++) {
            const int jj = j + g.lead[k], kk = jj / ns, s = jj % ns, e = g.bin_off[k] + j;
            g.ent_apos[2 * e] = g.bin_offp[k] * 8 + panel_pos(ns, 8, kk, 0, s);
            g.ent_apos[2 * e + 1] = panel_pos(ns, 8, kk, 1, s) - panel_pos(ns, 8, kk, 0, s);
            for (int row = 0; row < 8; row++)
                g.a_src[(size_t)g.bin_offp[k] * 8 + panel_pos(ns, 8, kk, row, s)] =
                    ((row >> 1) * sbuf + g.ent_src[e] + (row & 1)) | ((row >> 1) << 24);
            for (int col = 0; col < 16; col++)
                g.b_src[(size_t)g.bin_offp[k] * 16 + panel_pos(ns, 16, kk, col, s)] = (e << 4) | col;
        }
    }
}

// search offsets in Util::multiref_polar_ali_2d order (y outer, x inner), full window
inline bool build_shifts(Geometry &g, float xrng, float yrng, float step)
{
    if (step <= 0.f || xrng < 0.f || yrng < 0.f) return false;
    g.step = step;
    g.nkx = (int)(xrng / step); g.nky = (int)(yrng / step);
    g.shift_x.clear(); g.shift_y.clear();
    for (int i = -g.nky; i <= g.nky; i++)
        for (int j = -g.nkx; j <= g.nkx; j++) {
            g.shift_x.push_back(j * step); g.shift_y.push_back(i * step);
        }
    g.nshift = (int)g.shift_x.size();
    g.nshift_pad = ((g.nshift + 3) / 4) * 4;
    return true;
}

}  // namespace ralign
