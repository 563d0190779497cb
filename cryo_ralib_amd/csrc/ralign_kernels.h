// Device kernels of the 2-D alignment hot path, written for gfx950 (MI355X, 64-wide waves).
//
//   polar_fft_kernel   Polar2Dm (bilinear) + Normalize_ring + Frngs for 4 search offsets of one
//                      particle at a time; image and ring buffers in LDS; spectra written
//                      bin-major so the contraction streams them in full 128-B lines.
//   ccf_kernel         reference x particle cross-correlation (Crosrng_ms) as a dense f32 MFMA
//                      contraction over rings, per Fourier bin; the CCF spectra of an
//                      8 (particle-offset) x 8 (reference) tile stay in LDS, are inverse-FFTed
//                      there (q and t together as one complex transform) and reduced by a
//                      wavefront argmax; only the peak records leave the CU.
//   finalize_kernel    per-particle reduction over offsets and references with EMAN2's
//                      tie rules, prb1d / ang_n, parameter algebra.
//   transform_kernel   rot_shift2D (quadratic, background) + per-class even/odd accumulation.
//   refprep / pack / update kernels for the reference side.
//
// The arithmetic follows the EMAN2 CPU path; reference call sites are
// cited at each kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ralign_fft.h"
#include "../../include/ralign.h"

// phase-skip switches for in-situ timing experiments exist only in builds made with
// -DRALIGN_PROFILE_SWITCHES (scripts/profile.sh); the shipped library compiles them out, so no
// environment variable can make a production kernel skip work.
#ifdef RALIGN_PROFILE_SWITCHES
#define RA_DBG(g, bits) (((g).dbg & (bits)) != 0)
// wave timeline of the fused search kernel (profiling builds, RALIGN_TIMELINE=<file>): stamp s of pass `grp`
#define RA_STAMP(g, cond, grp, wave, s)                                                                         \
    do { if ((g).timeline && (cond) && (grp) < 64 && threadIdx.x % 64 == 0) (g).timeline[((grp) * 16 + (wave)) * 16 + (s)] = clock64(); } while (0)
#else
#define RA_DBG(g, bits) false
#define RA_STAMP(g, cond, grp, wave, s) do { } while (0)
#endif

namespace ralign {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DevGeom {
    int nx, cnx;                  // cnx = nx/2+1 (1-based SPIDER centre)
    int nring, maxrin, lcirc, lring, nbins, LB, LBP;
    int last_ring;
    int win_ring;                 // the radius of the window rule (particle_window): the caller's last_ring argument for RA_MODE_MREF
                                  // (test_mref_gpu_align.py:740, 761-766), numr[-3] -- the outermost SAMPLED ring, below it when the ring
                                  // step does not divide last_ring - first_ring -- for RA_MODE_REFFREE (ali2d_single_iter: ou = numr[-3])
    int nshift, nshift_pad, nkx, nky;
    int ent_stride;               // entries (particle-offsets) per particle in the A blocks, statistics and candidate records: nshift_pad; the
                                  // size-generic path packs them densely (nshift: 121 instead of 124 at configs[4], no padding offsets in the contraction)
    float step, xrng, yrng;
    float nn_weight;
    float inv_nn_weight;           // 1 / nn_weight, rounded once on the host
    int lg_maxrin;                 // log2(maxrin): twiddle strides are shifts (v_mul_lo_u32 is quarter rate)
    int mode;                     // RA_MODE_*
    int nomirror;                 // ormq(..., nomirror): the mirrored half of Crosrng_ms is not considered
    int norm_ring;                // Normalize_ring between Polar2Dm and Frngs (default: RA_MODE_MREF on, RA_MODE_REFFREE off; ra_set_normalize_ring)
    int interp;                   // RA_INTERP_*: alrl_ms's interpolation (bilinear; quadri: size-generic kernels only, ra_create_ex)
    int quad_aligned;             // generic kernels: ring quads aligned across bins (ralign_geom.h: align_ring_quads)
    int dbg;                      // phase-skip mask of profiling builds (-DRALIGN_PROFILE_SWITCHES); unused otherwise
    unsigned long long *timeline; // profiling builds: [pass][wave][stamp] clock values of workgroup 0's first particle, or null
    int sbuf;                     // LDS stride of one ring buffer (floats)
    int a_blk;                    // floats per A block of 4 particle-offsets: LBP*8 + slack
    int n_itemA, n_itemB, n_itemC;
    int rpt;                      // references per tile of the contraction (<= 8, balanced over the tiles)
    int n_class;                  // bins with the same ring-slot count ns form contiguous classes
    int class_k0[8], class_ns[8]; // class c = bins [class_k0[c], class_k0[c+1]) ; class_k0[n_class] = nbins
    int class_k0_end;
    int class_rot[8];             // wave that takes the first bin of the class (balances the contraction over the waves)
    const float *samp_dx, *samp_dy, *samp_w;
    const int *samp_dst;
    const int *bin_off, *bin_offp;
    const int *ent_src;
    const int4 *a_src4;           // [LBP*2] gather table of the A write-out, 4 floats per entry
    const int *b_src;             // [LBP*16] (entry << 4 | col) of every B-tile float, -1 = zero
    const int2 *ent_apos;         // [LB] {A-block offset of the entry at row 0, row stride}
    const int *bin_first;         // [nbins] first ring holding bin k
    const float *ent_wgt;
    const float *shift_x, *shift_y;
    const float2 *tw;             // e^{-2 pi i k / maxrin}, k < maxrin
    const int4 *itemA, *itemB, *itemC;
    const float *mask;            // model_circle(last_ring) [nx*nx]
    // size-generic class, live-offset lists (round 6): entries of a chunk are the IN-WINDOW offsets of its particles only, in scan
    // order -- particle p owns entries ent_base[p] .. ent_base[p + 1] - 1 (live_scan_kernel), *ent_total of them in the chunk; null:
    // entry of (p, s) = p * ent_stride + s
    const int *ent_base, *ent_total;
    // copies of a reference inside the stack (ref_groups_kernel, ralign_exact.h): ref_dup[2 r] = the first reference whose exact
    // spectrum equals reference r's bit for bit (r itself: none before it), ref_dup[2 r + 1] = the next one behind r (-1: none).
    // Their CCFs are equal to the bit, so the CPU scan alone decides among them; a winner with copies goes to refine_winner_kernel,
    // which replays the scan over all of them.  null: no exact re-evaluation (ra_set_refine(0)) / class-resident references
    const int *ref_dup;
    // wave-job schedule of the polar kernel (4 search offsets per pass)
    int n_job, n_qtab, n_inst;
    int n_job_b;                  // search_duo_kernel: jobs [n_job, n_job + n_job_b) = the ring jobs of a pass's second offset (0: as the first)
    int bd, pst;                  // zero border and row stride of the padded LDS image
    const int4 *jobs;             // {size code, first instance, instance count, 0}
    const int4 *inst;             // {offset slot 0..3 | ring << 8, ring_off, qtab offset, radius}, by ring length
    const float *instw;           // Normalize_ring weight of the instance's ring
    const float2 *qtab;           // first-quadrant (sinf, cosf) of alrl_ms per ring length
    const int4 *ringinfo;         // per ring {ring_off, radius, length, qtab offset}
    const float *ringw;           // per ring Normalize_ring weight r*2pi/n
};

// ------------------------------------------------------------------------------------------
// helpers

// exchange with the neighbouring lane (lane ^ 1) through the DPP quad permute [1,0,3,2]: a VALU move modifier, no
// trip through the LDS crossbar (what __shfl_xor(v, 1) compiles to: ds_bpermute_b32 + s_waitcnt)
__device__ __forceinline__ float swap_lane_pair(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// sum over the LR (4 | 8 | 16) lanes of a ring group with DPP row permutes; every lane ends up with the total.
// Fixed pairing order, hence reproducible run to run.
template <int LR> __device__ __forceinline__ float group_sum_dpp(float v)
{
    if constexpr (LR >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // 15 - i
    if constexpr (LR >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));    // 7 - i
    if constexpr (LR >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));     // i ^ 2
    if constexpr (LR >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // i ^ 1
    return v;
}

// sum over the 64 lanes without LDS traffic: row totals by DPP, the four rows through scalar registers, fixed order
// ((r0 + r1) + (r2 + r3)); __shfl_xor is a ds_bpermute, i.e. six dependent LDS round trips per sum
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v = group_sum_dpp<16>(v);
    const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Util::bilinear on a 0-based image pointer with 1-based coordinates (no FMA contraction so
// that samples match the CPU restatement bit for bit)
__device__ __forceinline__ float bilinear_1b(const float *img, int nx, float xold, float yold)
{
#pragma clang fp contract(off)
    int ix = (int)xold, iy = (int)yold;
    float ydif = yold - iy, xdif = xold - ix;
    // a sample may land exactly on the last column / row (zero-weight tap one past the image),
    // and out-of-window offsets (masked later) may leave it altogether: clamp every tap
    const int x0 = min(max(ix, 1), nx) - 1, x1 = min(x0 + 1, nx - 1);
    const int y0 = min(max(iy, 1), nx) - 1, y1 = min(y0 + 1, nx - 1);
    // (the two taps of a row as one 4-byte-aligned 8-byte load: measured SLOWER, 15.7 -> 18.9 ms per chunk of the generic polar
    // stage, whose time is 57 % sampling)
    // (unsigned offsets: a wave-uniform image pointer then stays in scalar registers and every tap is one 32-bit lane offset)
    const unsigned r0 = (unsigned)(y0 * nx), r1 = (unsigned)(y1 * nx);
    const char *ib = reinterpret_cast<const char *>(img);
    auto tap = [&](unsigned idx) { return *reinterpret_cast<const float *>(ib + 4u * idx); };      // 32-bit byte offset (images < 4 GB)
    float f00 = tap(r0 + (unsigned)x0), f10 = tap(r0 + (unsigned)x1), f01 = tap(r1 + (unsigned)x0), f11 = tap(r1 + (unsigned)x1);
    return f00 + ydif * (f01 - f00) + xdif * (f10 - f00 + ydif * (f11 - f10 - f01 + f00));
}

// Util::quadri (the interpolation of alrl_ms in older EMAN2 releases; RA_INTERP_QUADRI): 1-based coordinates, periodic in both
// directions, the operations and their order as in quadri_background_1b below (no contraction)
__device__ __forceinline__ float quadri_1b(const float *fdata, int nx, int ny, float xx, float yy)
{
#pragma clang fp contract(off)
    float x = xx, y = yy;
    while (x < 1.0f) x += nx;
    while (x >= (float)(nx + 1)) x -= nx;
    while (y < 1.0f) y += ny;
    while (y >= (float)(ny + 1)) y -= ny;
    int i = (int)x, j = (int)y;
    float dx0 = x - i, dy0 = y - j;
    int ip1 = i + 1, im1 = i - 1, jp1 = j + 1, jm1 = j - 1;
    if (ip1 > nx) ip1 -= nx;
    if (im1 < 1) im1 += nx;
    if (jp1 > ny) jp1 -= ny;
    if (jm1 < 1) jm1 += ny;
#define RA_FDQ(i_, j_) fdata[((j_) - 1) * nx + ((i_) - 1)]
    float f0 = RA_FDQ(i, j);
    float c1 = RA_FDQ(ip1, j) - f0;
    float c2 = (c1 - f0 + RA_FDQ(im1, j)) * 0.5f;
    float c3 = RA_FDQ(i, jp1) - f0;
    float c4 = (c3 - f0 + RA_FDQ(i, jm1)) * 0.5f;
    float dxb = dx0 - 1, dyb = dy0 - 1;
    int hxc = (dx0 >= 0) ? 1 : -1, hyc = (dy0 >= 0) ? 1 : -1;
    int ic = i + hxc, jc = j + hyc;
    if (ic > nx) ic -= nx; else if (ic < 1) ic += nx;
    if (jc > ny) jc -= ny; else if (jc < 1) jc += ny;
    float c5 = ((RA_FDQ(ic, jc) - f0 - hxc * c1 - (hxc * (hxc - 1.0f)) * c2 - hyc * c3 - (hyc * (hyc - 1.0f)) * c4) * (hxc * hyc));
#undef RA_FDQ
    return f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
}
// alrl_ms's sample at (x, y): Util::bilinear (EMAN2 2.31; the default) or Util::quadri
template <bool QUADRI> __device__ __forceinline__ float polar_sample_1b(const float *img, int nx, float x, float y)
{
    if constexpr (QUADRI) return quadri_1b(img, nx, nx, x, y);
    else return bilinear_1b(img, nx, x, y);
}

// the same interpolation on an LDS image with a zero border of `bd` pixels (row stride `st`):
// every tap of every search offset, including the zero-weight taps one past the image and the
// taps of offsets outside the particle's window (masked later), stays inside the padded
// array, so no clamping is needed.  `base` = img + (bd-1)*st + (bd-1) absorbs the 1-based origin.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bilinear_pad(const float *base, int st, float xold, float yold)
{
#pragma clang fp contract(off)
    // coordinates are >= 1 inside the padded image: x - floor(x) (v_fract_f32, exact) equals the CPU path's
    // xold - (float)(int)xold bit for bit and saves the int -> float round trip (positions, taps and fractions are
    // the CPU path's; only the evaluation order of the interpolant below differs)
    const int ix = (int)xold, iy = (int)yold;
    const float ydif = __builtin_amdgcn_fractf(yold), xdif = __builtin_amdgcn_fractf(xold);
    int idx;                                            // iy * st + ix in ONE full-rate instruction (hipcc otherwise splits
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(idx) : "v"(iy), "s"(st), "v"(ix));    // it into mul24 + two shifts + add3)
    const float *p = base + idx;
#ifdef RALIGN_BILINEAR_EXACT_ORDER
    // Util::bilinear's own operation order, no contraction: bit-identical samples (11 instructions)
    const float f00 = p[0], f10 = p[1], f01 = p[st], f11 = p[st + 1];
    return f00 + ydif * (f01 - f00) + xdif * (f10 - f00 + ydif * (f11 - f10 - f01 + f00));
#else
    // the same interpolant as two fused lerps, row pair first: both rows in one packed subtract and one packed fma, then
    // the column lerp (4 instructions; differs from Util::bilinear's operation order by rounding only, < 1 ulp of the taps)
    const v2f r0 = {p[0], p[1]}, r1 = {p[st], p[st + 1]};
    const v2f g = __builtin_elementwise_fma((v2f){ydif, ydif}, r1 - r0, r0);
    float d = g.y - g.x;
    asm("" : "+v"(d));      // keeps the column lerp scalar: packed over two samples it costs three register moves
    return __builtin_fmaf(xdif, d, g.x);
#endif
}

// search window of one particle: the reset/clamp rule and search_range
// (test_mref_gpu_align.py:1030-1038; ali2d_single_iter for RA_MODE_REFFREE)
struct Window { float sxi, syi; int lkx, rkx, lky, rky; };
__device__ __forceinline__ Window particle_window(const DevGeom &g, float dx, float dy)
{
    Window w;
    const float mashi = (float)(g.cnx - g.win_ring - 2);
    if (g.mode == RA_MODE_MREF) {
        if (fabsf(dx) > mashi || fabsf(dy) > mashi) { dx = 0.f; dy = 0.f; }
    } else {
        dx = fminf(fmaxf(dx, -mashi), mashi);
        dy = fminf(fmaxf(dy, -mashi), mashi);
    }
    w.sxi = dx; w.syi = dy;
    const int cn = g.nx / 2 + 1;
    float qlx = fmaxf(cn + dx - g.win_ring - 2, 0.f), qex = fmaxf(g.nx - cn - dx - g.win_ring, 0.f);
    float qly = fmaxf(cn + dy - g.win_ring - 2, 0.f), qey = fmaxf(g.nx - cn - dy - g.win_ring, 0.f);
    w.lkx = (int)(fminf(qlx, g.xrng) / g.step); w.rkx = (int)(fminf(qex, g.xrng) / g.step);
    w.lky = (int)(fminf(qly, g.yrng) / g.step); w.rky = (int)(fminf(qey, g.yrng) / g.step);
    return w;
}
// the search offsets inside a window, in the order of the offset list (y outer, x inner): how many, the list index of the j-th, and
// the position of list index s among them (-1: outside the window)
__device__ __forceinline__ int live_count(const Window &w) { return (w.lkx + w.rkx + 1) * (w.lky + w.rky + 1); }
__device__ __forceinline__ int live_shift(const DevGeom &g, const Window &w, int j)
{
    const int wx = w.lkx + w.rkx + 1, jy = j / wx, jx = j - jy * wx;
    return (jy - w.lky + g.nky) * (2 * g.nkx + 1) + (jx - w.lkx + g.nkx);
}
__device__ __forceinline__ int live_index(const DevGeom &g, const Window &w, int s)
{
    const int nx1 = 2 * g.nkx + 1, iy = s / nx1 - g.nky, ix = s - (s / nx1) * nx1 - g.nkx;
    if (ix < -w.lkx || ix > w.rkx || iy < -w.lky || iy > w.rky) return -1;
    return (iy + w.lky) * (w.lkx + w.rkx + 1) + (ix + w.lkx);
}
// per-particle entry ranges of a chunk: ent_base[p] = number of in-window offsets of particles 0 .. p - 1 (one workgroup; n <= 32768)
__global__ __launch_bounds__(1024) void live_scan_kernel(DevGeom g, const float *__restrict__ state, int n, int *__restrict__ ent_base,
                                                        int *__restrict__ ent_total)
{
    __shared__ int sc[1024];
    __shared__ int run;
    const int tid = threadIdx.x;
    if (tid == 0) run = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n; b0 += 1024) {
        const int p = b0 + tid;
        const int c = p < n ? live_count(particle_window(g, state[2 * p], state[2 * p + 1])) : 0;
        sc[tid] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = tid >= o ? sc[tid - o] : 0;
            __syncthreads();
            sc[tid] += v;
            __syncthreads();
        }
        const int base = run;
        if (p < n) ent_base[p] = base + sc[tid] - c;
        __syncthreads();
        if (tid == 1023) run = base + sc[1023];
        __syncthreads();
    }
    if (tid == 0) { ent_base[n] = run; *ent_total = run; }
}

// ------------------------------------------------------------------------------------------
// ring FFT passes on LDS ring buffers.  A ring of n reals is the n/2-point complex sequence
// z_m = x_2m + i x_2m+1 already in place; h = n/2 = R1*R2.
//   pass A: R2 items/ring, DFT-R1 over stride R2, twiddle, in place
//   pass B: R1 items/ring, DFT-R2 over contiguous R2, output stride R1 (lock-step in one wave)
//   pass C: split step X_k <- (Z_k, Z_{h-k}), in place, writes X_0 and X_h (padding slot)

template <int R>
__device__ __forceinline__ void passA_item(float *buf, const int4 it, const float2 *__restrict__ tw,
                                           int maxrin, float avg, float rsg)
{
    const int off = it.x, R2 = it.z, b = it.w;
    float2 v[R];
#pragma unroll
    for (int a = 0; a < R; a++) {
        float2 z = *reinterpret_cast<const float2 *>(buf + off + 2 * (R2 * a + b));
        v[a] = make_float2((z.x - avg) * rsg, (z.y - avg) * rsg);
    }
    Dft<-1, R>::run(v);
    const int tstep = maxrin / (R * R2) * b;   // W_h^{b c} = W_maxrin^{b c maxrin/h}
#pragma unroll
    for (int c = 0; c < R; c++) {
        float2 o = v[c];
        if (R2 > 1 && c > 0) o = cmul(o, tw[(tstep * c) & (maxrin - 1)]);
        *reinterpret_cast<float2 *>(buf + off + 2 * (R2 * c + b)) = o;
    }
}

template <int R>
__device__ __forceinline__ void passB_item(float *buf, const int4 it)
{
    const int off = it.x, R1 = it.y, c = it.w;
    float2 v[R];
#pragma unroll
    for (int b = 0; b < R; b++) v[b] = *reinterpret_cast<const float2 *>(buf + off + 2 * (R * c + b));
    Dft<-1, R>::run(v);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int e = 0; e < R; e++) *reinterpret_cast<float2 *>(buf + off + 2 * (c + R1 * e)) = v[e];
}

// item: x = float offset of Z_k, y = float offset of Z_{h-k}, z = twiddle index k*maxrin/n, w = (k==0)
__device__ __forceinline__ void passC_item(float *buf, const int4 it, const float2 *__restrict__ tw)
{
    float2 zk = *reinterpret_cast<const float2 *>(buf + it.x);
    if (it.w) {   // k = 0: X_0 = Zr + Zi, X_h = Zr - Zi, both real
        *reinterpret_cast<float2 *>(buf + it.x) = make_float2(zk.x + zk.y, 0.f);
        *reinterpret_cast<float2 *>(buf + it.y) = make_float2(zk.x - zk.y, 0.f);
        return;
    }
    float2 zm = *reinterpret_cast<const float2 *>(buf + it.y);
    // E = (Z_k + conj Z_m)/2 ; O = (Z_k - conj Z_m)/(2i) ; X_k = E + W^k O ; X_m = conj(E - W^k O)
    float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
    float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
    float2 w = tw[it.z];
    float tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
    *reinterpret_cast<float2 *>(buf + it.x) = make_float2(er + tr, ei + ti);
    if (it.x != it.y) *reinterpret_cast<float2 *>(buf + it.y) = make_float2(er - tr, -(ei - ti));
}

// run the three passes over `nbuf` ring buffers (stride sbuf) with the whole workgroup
__device__ __forceinline__ void ring_fft_all(const DevGeom &g, float *bufs, int nbuf,
                                             const float *avg, const float *rsg)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int idx = tid; idx < g.n_itemA * nbuf; idx += nt) {
        int s = idx / g.n_itemA, i = idx - s * g.n_itemA;
        int4 it = g.itemA[i];
        float *b = bufs + s * g.sbuf;
        float av = avg[s], rs = rsg[s];
        switch (it.y) {
        case 16: passA_item<16>(b, it, g.tw, g.maxrin, av, rs); break;
        case 8:  passA_item<8>(b, it, g.tw, g.maxrin, av, rs); break;
        case 4:  passA_item<4>(b, it, g.tw, g.maxrin, av, rs); break;
        default: passA_item<2>(b, it, g.tw, g.maxrin, av, rs); break;
        }
    }
    __syncthreads();
    // pass B: items of one ring are consecutive and radix-aligned, so they sit in one wave;
    // the trip count is made wave-uniform so the in-wave load/store ordering holds
    const int nB = g.n_itemB * nbuf;
    for (int base = 0; base < nB; base += nt) {
        int idx = base + tid;
        if (idx < nB) {
            int s = idx / g.n_itemB, i = idx - s * g.n_itemB;
            int4 it = g.itemB[i];
            float *b = bufs + s * g.sbuf;
            switch (it.z) {
            case 16: passB_item<16>(b, it); break;
            case 8:  passB_item<8>(b, it); break;
            case 4:  passB_item<4>(b, it); break;
            case 2:  passB_item<2>(b, it); break;
            default: break;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < g.n_itemC * nbuf; idx += nt) {
        int s = idx / g.n_itemC, i = idx - s * g.n_itemC;
        passC_item(bufs + s * g.sbuf, g.itemC[i], g.tw);
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// K1: polar resampling + Normalize_ring + ring FFT of every search offset of one particle.
// Restates Polar2Dm / Normalize_ring / Frngs as called inside Util.multiref_polar_ali_2d
// (reference call site test_mref_gpu_align.py:1043-1044; ormq for RA_MODE_REFFREE).
//   particles [n][nx*nx], state [n][2] (accumulated centre offset), A blocks out.
// One workgroup (16 waves) per particle, 4 search offsets per pass; LDS holds the image and 4
// ring buffers.  Work is cut into wave-jobs: a wave takes 64/LR rings of one length (LR lanes
// per ring), samples them straight into registers (bilinear taps from the LDS image, sample
// positions rebuilt from the alrl_ms quadrant table), runs the n/2-point complex FFT as
// DFT-R1 (registers) -> LDS transpose -> DFT-LR (registers) and the real-FFT split step, with
// wave-local synchronisation only.  Normalize_ring's affine map is applied after the
// (linear) FFT: X_k' = (X_k - avg n [k==0]) / sigma.
// Output: per block of 4 offsets, the bin-major operand panels of ccf_kernel.
#define RA_POLAR_THREADS 1024

// wave-local hand-off through LDS: all earlier LDS traffic of this wave retired, and the
// compiler may not move memory operations across this point
#define RA_WAVE_SYNC()                                               \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");       \
    } while (0)

// `nlive`: offset slots >= nlive are padding and skipped.
// `part` receives the Normalize_ring partial sums {sum w x, sum w x^2} of every (offset slot, ring).  NYQ1 (fused
// search kernel): the Nyquist coefficient of a full-length ring (n == maxrin) is stored in the imaginary slot of
// bin 0, EMAN2's own packing, so that the contraction sees bins 0 .. maxrin/2 - 1 only.
// `sync` runs once, in uniform control flow, between the sampling (which reads the image and the tables only) and the first
// write to the ring buffers: the fused kernel passes the workgroup barrier that ends the previous pass's inverse FFTs there
// (their spectra occupy the ring buffers), so a wave samples while the others still transform.
struct NoSync { __device__ __forceinline__ void operator()() const {} };

template <int R1, int LR, bool NYQ1 = false, class Sync = NoSync>
__device__ __forceinline__ void ring_job(const DevGeom &g, const float *imgb, float *bufs, const float2 *tw_s,
                                         const float2 *qt_s, const float *ctr, float *part, const int4 *inst_s,
                                         const float *instw_s, int inst0, int count, int zero, int sbuf, int nlive = 4,
                                         Sync sync = Sync(), int slot_lo = 0)
{
    // lane id rebuilt from a per-job runtime zero (jobs[].w): keeps the per-variant lane arithmetic
    // inside the job instead of hoisted out of the pass loop for all six variants (VGPR spills)
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));

    constexpr int H = R1 * LR, NR = 2 * H, LT = NR / 4;
    const int sub = lane / LR, t = lane % LR;
    const int subc = min(sub, count - 1);       // lanes past the last instance stay in the control flow (see `sync`), inactive
    const int4 in = inst_s[inst0 + subc];
    const int slot = in.x & 255, ring = in.x >> 8;
    // inactive: no instance, or a padding offset of the last pass (the instances of a slot are contiguous: whole jobs drop out)
    const bool active = sub < count && slot < nlive && slot >= slot_lo;      // [slot_lo, nlive): the offset slots of this call
    float *buf = bufs + (__mul24(slot, sbuf) + in.y);
    const float rad = (float)in.w, wt = instw_s[inst0 + subc];
    const float cx = ctr[2 * slot], cy = ctr[2 * slot + 1];
    const float2 *qt = qt_s + in.z;
    // Normalize_ring partial sums: plain sums of the lane's samples and their squares, two at a time (packed f32),
    // weighted once per ring at the end
    v2f av2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
    const v2f ctr2 = {cx, cy};
    float2 v[R1];
    // Sample j = 2*(LR*a + t) + u of the ring sits in quadrant j / LT at table entry j % LT.  When 2*LR <= LT (QCT) the
    // quadrant depends on `a` alone: a = qd * NB + b with NB = R1 / 4 table positions per lane, and the four quadrants
    // of one position share the table read and the radius multiply -- the loop runs position-major, the alrl_ms
    // mirroring (x,y), (y,-x), (-x,-y), (-y,x) is an operand modifier of the packed add of the centre.
    // The (sin, cos) pairs of the next step are requested before the current one is interpolated: their LDS latency
    // hides behind the taps instead of heading the dependent chain table -> position -> taps -> interpolation.
    constexpr bool QCT = (2 * LR <= LT) && (R1 >= 4);
    constexpr int NB = QCT ? R1 / 4 : R1;
    auto table_index = [&](int step, int u, int &qd) {
        if constexpr (QCT) { qd = 0; return 2 * LR * step + 2 * t + u; }
        else { const int j = 2 * (LR * step + t) + u; qd = j / LT; return j % LT; }
    };
    if (active) {
    int qdn[2];
    float2 scn[2];
#pragma unroll
    for (int u = 0; u < 2; u++) scn[u] = qt[table_index(0, u, qdn[u])];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int qdc[2] = {qdn[0], qdn[1]};
        v2f xy[2];
        {
#pragma clang fp contract(off)
            // both coordinates in one packed multiply and one packed add: the same IEEE operations as the scalar form
            xy[0] = v2f{scn[0].x, scn[0].y} * rad;
            xy[1] = v2f{scn[1].x, scn[1].y} * rad;
        }
        if (b + 1 < NB) {
#pragma unroll
            for (int u = 0; u < 2; u++) scn[u] = qt[table_index(b + 1, u, qdn[u])];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < (QCT ? 4 : 1); q++) {
            const int a = QCT ? q * NB + b : b;
            v2f val;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                v2f o;
                if constexpr (QCT) {
                    o = q == 0 ? ctr2 + xy[u] : q == 1 ? vadd_rot<-1>(ctr2, xy[u]) : q == 2 ? ctr2 - xy[u] : vadd_rot<1>(ctr2, xy[u]);
                } else {
                    const int qd = qdc[u];
                    v2f m = (qd & 1) ? v2f{xy[u].y, xy[u].x} : xy[u];
                    m.x = (qd & 2) ? -m.x : m.x;
                    m.y = ((qd + 1) & 2) ? -m.y : m.y;
                    o = m + ctr2;
                }
                const float s = bilinear_pad(imgb, g.pst, o.x, o.y);
                if (u == 0) val.x = s; else val.y = s;
            }
            av2 += val;
            sq2 += val * val;
            v[a] = make_float2(val.x, val.y);
            if ((NYQ1 && (QCT ? 4 * b + q : b) >= R1 / 2) || ((QCT ? q : b) & 1)) __builtin_amdgcn_sched_barrier(0);   // at most 4 samples' taps in flight (fused kernel: 2 once half of the samples are held in registers)
        }
    }
    }
    sync();
    if (!active) return;
    float av = (av2.x + av2.y) * wt, sq = (sq2.x + sq2.y) * wt;
    if (RA_DBG(g, 256)) {   // diagnostic: leave the raw samples in natural order, no FFT
#pragma unroll
        for (int a = 0; a < R1; a++) *reinterpret_cast<float2 *>(buf + 2 * (LR * a + t)) = v[a];
        return;
    }
    Dft<-1, R1>::run(v);
    constexpr int LGH = __builtin_ctz(H), LGNR = LGH + 1;
    const int tstep = t << (g.lg_maxrin - LGH);      // t * maxrin / H
    // element (row c, column t) is parked at LR*c + ((t + c) mod LR): conflict-free both ways
#pragma unroll
    for (int c = 0; c < R1; c++) {
        float2 o = v[c];
        if (c > 0) o = cmul(o, tw_s[__mul24(tstep, c)]);      // t c < H: no wrap; 24-bit multiplies are full rate, v_mul_lo_u32 is not
        *reinterpret_cast<float2 *>(buf + 2 * (LR * c + ((t + c) & (LR - 1)))) = o;
    }
    RA_WAVE_SYNC();
    // real-FFT split step of one pair: X_k and X_{H-k} from Z_k, Z_{H-k} (k = 0 gives X_0 and the Nyquist term X_H)
    auto split_pair = [&](float2 zk, float2 zm, int k, float2 &xk, float2 &xm) {
        v2f a, b;
        vsplit_pair(to_v(zk), to_v(zm), to_v(tw_s[k << (g.lg_maxrin - LGNR)]), a, b);
        xk = to_f2(a); xm = to_f2(b);
    };
    if constexpr (R1 == 2 * LR) {
        // Two rows per lane, chosen so that the split step stays inside the lane: lane t transforms rows t and R1 - t
        // (lane 0: rows 0 and R1/2), and Z_{H-k} of every k = row + R1 e of one row is element LR-1-e of the other
        // (rows 0 and R1/2 pair with themselves).  The second-pass output never goes back to LDS: one write and one
        // read of every element and one wave hand-off less than the generic schedule below.
        const int rowA = t, rowB = t ? R1 - t : R1 / 2;
        float2 z[2][LR];
#pragma unroll
        for (int b = 0; b < LR; b++) z[0][b] = *reinterpret_cast<const float2 *>(buf + 2 * (LR * rowA + ((b + rowA) & (LR - 1))));
#pragma unroll
        for (int b = 0; b < LR; b++) z[1][b] = *reinterpret_cast<const float2 *>(buf + 2 * (LR * rowB + ((b + rowB) & (LR - 1))));
        Dft<-1, LR>::run(z[0]);
        Dft<-1, LR>::run(z[1]);
        RA_WAVE_SYNC();                  // every lane of the ring has its rows: the buffer may be overwritten
        const bool l0 = t == 0;
#pragma unroll
        for (int i = 0; i < LR; i++) {
            // lane t > 0: (Z[t + R1 i], Z[R1 - t + R1 (LR-1-i)]); lane 0: row R1/2 for i < LR/2, then row 0 (e = 1 .. LR/2)
            const int e0 = i < LR / 2 ? i : i - LR / 2 + 1;
            const float2 zk0 = i < LR / 2 ? z[1][e0] : z[0][e0];
            const float2 zm0 = i < LR / 2 ? z[1][LR - 1 - e0] : z[0][(LR - e0) & (LR - 1)];
            const int k0 = i < LR / 2 ? LR + R1 * e0 : R1 * e0;
            float2 zk, zm;
            zk.x = l0 ? zk0.x : z[0][i].x; zk.y = l0 ? zk0.y : z[0][i].y;
            zm.x = l0 ? zm0.x : z[1][LR - 1 - i].x; zm.y = l0 ? zm0.y : z[1][LR - 1 - i].y;
            const int k = l0 ? k0 : t + R1 * i;
            float2 xk, xm;
            split_pair(zk, zm, k, xk, xm);
            *reinterpret_cast<float2 *>(buf + 2 * (H - k)) = xm;      // lane 0, last step: k = H/2 = H - k, X_k lands last
            *reinterpret_cast<float2 *>(buf + 2 * k) = xk;
        }
        if (l0) {
            const float2 zk = z[0][0];
            if (NYQ1 && NR == g.maxrin) {
                *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, zk.x - zk.y);
            } else {
                *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, 0.f);
                *reinterpret_cast<float2 *>(buf + 2 * H) = make_float2(zk.x - zk.y, 0.f);
            }
        }
    } else {
        // second pass: the R1 rows of the LR-point transforms are dealt to the LR lanes of the ring
        // (one row per lane when R1 <= LR, R1/LR rows per lane otherwise: no idle half-groups)
        constexpr int NROW = (R1 + LR - 1) / LR;
        float2 z[NROW][LR];
#pragma unroll
        for (int m = 0; m < NROW; m++) {
            const int row = t + LR * m;
            if (row < R1) {
#pragma unroll
                for (int b = 0; b < LR; b++) z[m][b] = *reinterpret_cast<const float2 *>(buf + 2 * (LR * row + ((b + row) & (LR - 1))));
                Dft<-1, LR>::run(z[m]);
            }
        }
        RA_WAVE_SYNC();
#pragma unroll
        for (int m = 0; m < NROW; m++) {
            const int row = t + LR * m;
            if (row < R1) {
#pragma unroll
                for (int e = 0; e < LR; e++) *reinterpret_cast<float2 *>(buf + 2 * (row + R1 * e)) = z[m][e];
            }
        }
        RA_WAVE_SYNC();
        // split step X_k <- (Z_k, Z_{H-k}), k = 0..H/2, in place; X_0 and X_H are real
        for (int k = t; k <= H / 2; k += LR) {
            float2 zk = *reinterpret_cast<const float2 *>(buf + 2 * k);
            if (k == 0) {
                if (NYQ1 && NR == g.maxrin) {
                    *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, zk.x - zk.y);
                } else {
                    *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, 0.f);
                    *reinterpret_cast<float2 *>(buf + 2 * H) = make_float2(zk.x - zk.y, 0.f);
                }
            } else {
                float2 zm = *reinterpret_cast<const float2 *>(buf + 2 * (H - k));
                float2 xk, xm;
                split_pair(zk, zm, k, xk, xm);
                *reinterpret_cast<float2 *>(buf + 2 * k) = xk;
                if (2 * k != H) *reinterpret_cast<float2 *>(buf + 2 * (H - k)) = xm;
            }
        }
    }
    // Normalize_ring partial sums of this ring -> its own slot (summed in ring order later)
    av = group_sum_dpp<LR>(av); sq = group_sum_dpp<LR>(sq);
    if (t == 0) { part[2 * (slot * g.nring + ring)] = av; part[2 * (slot * g.nring + ring) + 1] = sq; }
}

// Rings of 8, 16 and 32 samples in ONE job (code 9): 4 sample pairs per lane and n / 8 lanes per ring (4 | 2 | 1), so all the
// short rings of a pass fit the 64 lanes of one wave and every wave of the workgroup gets exactly one job -- as three jobs
// (one per length) the two shortest cost a second round of ~3 k cycles each on the wave timeline.  The instance table holds
// one entry per LANE: in.x = slot | ring << 8 | log2 n << 16 | lane-in-ring << 20.  Same arithmetic as ring_job<4, LR>.
template <bool NYQ1, class Sync = NoSync>
__device__ __forceinline__ void ring_job_mix(const DevGeom &g, const float *imgb, float *bufs, const float2 *tw_s,
                                             const float2 *qt_s, const float *ctr, float *part, const int4 *inst_s,
                                             const float *instw_s, int inst0, int count, int zero, int sbuf, int nlive = 4,
                                             Sync sync = Sync(), int slot_lo = 0)
{
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
    const int lanec = min(lane, count - 1);     // lanes without an instance stay in the control flow (see ring_job's `sync`)
    const int4 in = inst_s[inst0 + lanec];
    const int slot = in.x & 255, ring = (in.x >> 8) & 255, lg = (in.x >> 16) & 15, t = in.x >> 20;
    const bool active = lane < count && slot < nlive && slot >= slot_lo;
    const int lgLR = lg - 3, LR = 1 << lgLR, H = 4 << lgLR, lgLT = lg - 2, LTm = (1 << lgLT) - 1;
    float *buf = bufs + (__mul24(slot, sbuf) + in.y);
    const float rad = (float)in.w, wt = instw_s[inst0 + lanec];
    const v2f ctr2 = {ctr[2 * slot], ctr[2 * slot + 1]};
    const float2 *qt = qt_s + in.z;
    v2f av2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
    float2 v[4];
    if (active) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
        v2f val;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = 2 * ((a << lgLR) + t) + u, qd = j >> lgLT;
            const float2 sc = qt[j & LTm];
            v2f o;
            {
#pragma clang fp contract(off)
                const v2f xy = v2f{sc.x, sc.y} * rad;
                v2f m = (qd & 1) ? v2f{xy.y, xy.x} : xy;         // alrl_ms quadrant mirroring (x,y), (y,-x), (-x,-y), (-y,x)
                m.x = (qd & 2) ? -m.x : m.x;
                m.y = ((qd + 1) & 2) ? -m.y : m.y;
                o = m + ctr2;
            }
            const float sv = bilinear_pad(imgb, g.pst, o.x, o.y);
            if (u == 0) val.x = sv; else val.y = sv;
        }
        av2 += val;
        sq2 += val * val;
        v[a] = make_float2(val.x, val.y);
    }
    }
    sync();
    if (!active) return;
    Dft<-1, 4>::run(v);
    const int tstep = t << (g.lg_maxrin - (lg - 1));          // t * maxrin / H
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float2 o = v[c];
        if (c > 0) o = cmul(o, tw_s[__mul24(tstep, c)]);      // t c < H: no wrap
        *reinterpret_cast<float2 *>(buf + 2 * ((c << lgLR) + ((t + c) & (LR - 1)))) = o;
    }
    RA_WAVE_SYNC();
    // second pass: every lane owns 4 elements -- one row of 4 (LR = 4), two rows of 2 (LR = 2), four rows of 1 (LR = 1)
    int src[4], dst[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row2 = t + 2 * (i >> 1), b2 = i & 1;
        src[i] = lgLR == 2 ? 4 * t + ((i + t) & 3) : lgLR == 1 ? 2 * row2 + ((b2 + row2) & 1) : i;
        dst[i] = lgLR == 2 ? t + 4 * i : lgLR == 1 ? row2 + 4 * b2 : i;
    }
    float2 z[4];
#pragma unroll
    for (int i = 0; i < 4; i++) z[i] = *reinterpret_cast<const float2 *>(buf + 2 * src[i]);
    if (lgLR == 2) Dft<-1, 4>::run(z);
    if (lgLR == 1) { dft2<-1>(z[0], z[1]); dft2<-1>(z[2], z[3]); }
    RA_WAVE_SYNC();
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<float2 *>(buf + 2 * dst[i]) = z[i];
    RA_WAVE_SYNC();
    // split step X_k <- (Z_k, Z_{H-k}), k = t, t + LR, t + 2 LR while k <= H/2 = 2 LR
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int k = t + (m << lgLR);
        if (k <= 2 * LR) {
            const float2 zk = *reinterpret_cast<const float2 *>(buf + 2 * k);
            if (k == 0) {
                if (NYQ1 && 2 * H == g.maxrin) {
                    *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, zk.x - zk.y);
                } else {
                    *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, 0.f);
                    *reinterpret_cast<float2 *>(buf + 2 * H) = make_float2(zk.x - zk.y, 0.f);
                }
            } else {
                const float2 zm = *reinterpret_cast<const float2 *>(buf + 2 * (H - k));
                v2f xk, xm;
                vsplit_pair(to_v(zk), to_v(zm), to_v(tw_s[k << (g.lg_maxrin - lg)]), xk, xm);
                *reinterpret_cast<float2 *>(buf + 2 * k) = to_f2(xk);
                if (2 * k != H) *reinterpret_cast<float2 *>(buf + 2 * (H - k)) = to_f2(xm);
            }
        }
    }
    // Normalize_ring partial sums over the LR lanes of the ring (fixed pairing order)
    float av = (av2.x + av2.y) * wt, sq = (sq2.x + sq2.y) * wt;
    {
        const float a2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(av), 0x4E, 0xF, 0xF, true));     // i ^ 2
        const float q2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0x4E, 0xF, 0xF, true));
        if (lgLR == 2) { av += a2; sq += q2; }
        const float a1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(av), 0xB1, 0xF, 0xF, true));     // i ^ 1
        const float q1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq), 0xB1, 0xF, 0xF, true));
        if (lgLR >= 1) { av += a1; sq += q1; }
    }
    if (t == 0) { part[2 * (slot * g.nring + ring)] = av; part[2 * (slot * g.nring + ring) + 1] = sq; }
}

// Rings of 512 samples with 32 lanes per ring and EIGHT sample pairs per lane (code 11, search_solo_kernel): the 256-point complex
// transform as 8 x 8 x 4 -- DFT-8 over the stride-32 elements a lane sampled, one transpose through the ring's own buffer, DFT-8,
// and the last DFT-4 across the four lanes of a quad with DPP exchanges -- then the real-FFT split step from the buffer.  Half the
// registers of ring_job<16, 16> (16 values per lane instead of 32 + 32) and twice as many, half as long jobs: a pass of one offset
// (36 - 62 rings) then keeps all 16 waves busy, and the job fits next to a live A slice.
//   m = 32 a + 4 b + c,  k = k1 + 8 k2 + 64 k3,  W = e^{-2 pi i / 256}
//   A1 [k1; b c] = sum_a z[32 a + t] W8^{a k1},   t = 4 b + c;   * W^{t k1}
//   A2 [k1 k2; c] = sum_b A1 W8^{b k2},           lane = (k1, c);   * W^{8 c k2}
//   Z  [k]        = sum_c A2 (-i)^{c k3},         across the quad (k1, 0..3): lane c ends with k3 = (0, 2, 1, 3)[c]
// Same samples, tables and Normalize_ring partial sums as ring_job.
template <bool NYQ1, class Sync = NoSync>
__device__ __forceinline__ void ring_job512(const DevGeom &g, const float *imgb, float *bufs, const float2 *tw_s,
                                            const float2 *qt_s, const float *ctr, float *part, const int4 *inst_s,
                                            const float *instw_s, int inst0, int count, int zero, int sbuf, int nlive = 4,
                                            Sync sync = Sync(), int slot_lo = 0)
{
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
    constexpr int LR = 32, H = 256;
    const int sub = lane >> 5, t = lane & 31;
    const int subc = min(sub, count - 1);
    const int4 in = inst_s[inst0 + subc];
    const int slot = in.x & 255, ring = in.x >> 8;
    const bool active = sub < count && slot < nlive && slot >= slot_lo;
    float *buf = bufs + (__mul24(slot, sbuf) + in.y);
    const float rad = (float)in.w, wt = instw_s[inst0 + subc];
    const v2f ctr2 = {ctr[2 * slot], ctr[2 * slot + 1]};
    const float2 *qt = qt_s + in.z;
    v2f av2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
    float2 v[8];
    // element a of the lane is z[32 a + t] = samples 64 a + 2 t, + 1: quadrant a >> 1, table entries 64 (a & 1) + 2 t + u (of 128):
    // the four quadrants of a table position share the table read and the radius multiply (alrl_ms mirroring (x, y), (y, -x),
    // (-x, -y), (-y, x) as an operand modifier of the packed add of the centre)
    if (active) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            v2f xy[2];
            {
#pragma clang fp contract(off)
                const float2 s0 = qt[64 * b + 2 * t], s1 = qt[64 * b + 2 * t + 1];
                xy[0] = v2f{s0.x, s0.y} * rad;
                xy[1] = v2f{s1.x, s1.y} * rad;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v2f val;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const v2f o = q == 0 ? ctr2 + xy[u] : q == 1 ? vadd_rot<-1>(ctr2, xy[u]) : q == 2 ? ctr2 - xy[u] : vadd_rot<1>(ctr2, xy[u]);
                    const float sv = bilinear_pad(imgb, g.pst, o.x, o.y);
                    if (u == 0) val.x = sv; else val.y = sv;
                }
                av2 += val;
                sq2 += val * val;
                v[2 * q + b] = make_float2(val.x, val.y);
                if (q & 1) __builtin_amdgcn_sched_barrier(0);       // at most 4 samples' taps in flight
            }
        }
    }
    sync();
    if (!active) return;
    float av = (av2.x + av2.y) * wt, sq = (sq2.x + sq2.y) * wt;
    // ---- stage 1: DFT-8 over a, twiddle W^{t k1} = W512^{2 t k1}; row k1 parked rotated by 4 k1 (conflict-free both ways)
    Dft<-1, 8>::run(v);
    const int sh = g.lg_maxrin - 8;                       // W256 in units of the table's W_maxrin
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) {
        float2 o = v[k1];
        if (k1 > 0) o = cmul(o, tw_s[__mul24(t, k1) << sh]);
        *reinterpret_cast<float2 *>(buf + 2 * (32 * k1 + ((t + 4 * k1) & 31))) = o;
    }
    RA_WAVE_SYNC();
    // ---- stage 2: lane = (k1, c): DFT-8 over b, twiddle W^{8 c k2}
    const int k1 = t >> 2, c = t & 3;
#pragma unroll
    for (int b = 0; b < 8; b++) v[b] = *reinterpret_cast<const float2 *>(buf + 2 * (32 * k1 + ((4 * b + c + 4 * k1) & 31)));
    Dft<-1, 8>::run(v);
#pragma unroll
    for (int k2 = 1; k2 < 8; k2++) v[k2] = cmul(v[k2], tw_s[(8 * c * k2) << sh]);
    RA_WAVE_SYNC();                      // every lane of the ring has its row: the buffer may be overwritten
    // ---- stage 3: DFT-4 over c across the quad (DPP), natural order into the buffer: Z[k1 + 8 k2 + 64 k3]
    {
        const float sg2 = (c & 2) ? -1.0f : 1.0f, sg1 = (c & 1) ? -1.0f : 1.0f;
        const bool l3 = c == 3;
        const int k3 = ((c & 1) << 1) | (c >> 1);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            const float2 a = v[k2];
            const float tx = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.x), 0x4E, 0xF, 0xF, true));      // lane ^ 2
            const float ty = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a.y), 0x4E, 0xF, 0xF, true));
            const float bx = __builtin_fmaf(sg2, a.x, tx), by = __builtin_fmaf(sg2, a.y, ty);       // x0 + x2, x1 + x3, x0 - x2, x1 - x3
            const float cx = l3 ? by : bx, cy = l3 ? -bx : by;                                       // lane 3: (x1 - x3) * (-i)
            const float ux = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cx), 0xB1, 0xF, 0xF, true));     // lane ^ 1
            const float uy = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cy), 0xB1, 0xF, 0xF, true));
            *reinterpret_cast<float2 *>(buf + 2 * (k1 + 8 * k2 + 64 * k3)) = make_float2(__builtin_fmaf(sg1, cx, ux), __builtin_fmaf(sg1, cy, uy));
        }
    }
    RA_WAVE_SYNC();
    // ---- split step X_k <- (Z_k, Z_{H-k}), k = 0 .. H/2, in place; X_0 and X_H are real
    for (int k = t; k <= H / 2; k += LR) {
        const float2 zk = *reinterpret_cast<const float2 *>(buf + 2 * k);
        if (k == 0) {
            if (NYQ1 && 2 * H == g.maxrin) {
                *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, zk.x - zk.y);
            } else {
                *reinterpret_cast<float2 *>(buf) = make_float2(zk.x + zk.y, 0.f);
                *reinterpret_cast<float2 *>(buf + 2 * H) = make_float2(zk.x - zk.y, 0.f);
            }
        } else {
            const float2 zm = *reinterpret_cast<const float2 *>(buf + 2 * (H - k));
            v2f xk, xm;
            vsplit_pair(to_v(zk), to_v(zm), to_v(tw_s[k << (g.lg_maxrin - 9)]), xk, xm);
            *reinterpret_cast<float2 *>(buf + 2 * k) = to_f2(xk);
            if (2 * k != H) *reinterpret_cast<float2 *>(buf + 2 * (H - k)) = to_f2(xm);
        }
    }
    // Normalize_ring partial sums over the 32 lanes of the ring: two DPP rows, then the neighbouring row (fixed order)
    av = group_sum_dpp<16>(av); sq = group_sum_dpp<16>(sq);
    av += __shfl_xor(av, 16); sq += __shfl_xor(sq, 16);
    if (t == 0) { part[2 * (slot * g.nring + ring)] = av; part[2 * (slot * g.nring + ring) + 1] = sq; }
}

__global__ __launch_bounds__(RA_POLAR_THREADS) void polar_fft_kernel(DevGeom g, const float *__restrict__ particles,
                                                                     const float *__restrict__ state, int n,
                                                                     float *__restrict__ A)
{
    extern __shared__ __align__(16) float lds[];
    const int npad = g.pst * g.pst;
    float *img = lds;
    float *bufs = lds + ((npad + 3) & ~3);
    float2 *tw_s = reinterpret_cast<float2 *>(bufs + 4 * g.sbuf);     // [maxrin]
    float2 *qt_s = tw_s + g.maxrin;                                    // [n_qtab]
    int4 *inst_s = reinterpret_cast<int4 *>(qt_s + g.n_qtab + (g.n_qtab & 1));   // [n_inst]
    int4 *jobs_s = inst_s + g.n_inst;                                  // [n_job]
    float *instw_s = reinterpret_cast<float *>(jobs_s + g.n_job);      // [n_inst]
    float *red = instw_s + g.n_inst;    // [8] -, [8] avg / rsigma, [8] centres, [4*nring*2] ring partials
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwave = blockDim.x >> 6;
    if (p >= n) return;

    const float *src = particles + (size_t)p * g.nx * g.nx;
    for (int row = wave; row < g.pst; row += nwave) {        // a wave per padded row: no per-pixel division
        const int y = row - g.bd;
        const bool yin = y >= 0 && y < g.nx;
        for (int c = lane; c < g.pst; c += 64) {
            const int x = c - g.bd;
            img[row * g.pst + c] = (yin && x >= 0 && x < g.nx) ? src[y * g.nx + x] : 0.f;
        }
    }
    const float *imgb = img + (g.bd - 1) * g.pst + (g.bd - 1);
    for (int i = tid; i < g.maxrin; i += blockDim.x) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += blockDim.x) qt_s[i] = g.qtab[i];
    for (int i = tid; i < g.n_inst; i += blockDim.x) { inst_s[i] = g.inst[i]; instw_s[i] = g.instw[i]; }
    for (int i = tid; i < g.n_job; i += blockDim.x) jobs_s[i] = g.jobs[i];
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    const float cxf = (float)g.cnx + w.sxi, cyf = (float)g.cnx + w.syi;

    const int ngroup = g.nshift_pad / 4;
    for (int grp = 0; grp < ngroup; grp++) {
        if (tid < 4) {
            int si = min(grp * 4 + tid, g.nshift - 1);
            red[16 + 2 * tid] = cxf + g.shift_x[si];
            red[17 + 2 * tid] = cyf + g.shift_y[si];
        }
        __syncthreads();
        if (!RA_DBG(g, 16)) {
#pragma unroll 1
            for (int job = wave; job < g.n_job; job += nwave) {
                const int4 jd = jobs_s[job];
                switch (__builtin_amdgcn_readfirstlane(jd.x)) {
                case 0: ring_job<8, 16>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 1: ring_job<8, 8>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 2: ring_job<4, 8>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 3: ring_job<4, 4>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 4: ring_job<2, 4>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 6: ring_job<16, 8>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 7: ring_job<8, 4>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                case 9: ring_job_mix<false>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                default: ring_job<1, 4>(g, imgb, bufs, tw_s, qt_s, red + 16, red + 24, inst_s, instw_s, jd.y, jd.z, jd.w, g.sbuf); break;
                }
            }
        }
        __syncthreads();
        // Normalize_ring: avg = av/nn, sigma = sqrt((sq - av^2/nn)/nn); X_0 -= avg*n, then * 1/sigma.
        // Wave s (< 4) reduces the ring partials of offset s with a fixed butterfly (reproducible).
        if (wave < 4) {
            float a = 0.f, q = 0.f;
            for (int i = lane; i < g.nring; i += 64) { a += red[24 + 2 * (wave * g.nring + i)]; q += red[25 + 2 * (wave * g.nring + i)]; }
            a = wave_sum(a); q = wave_sum(q);
            if (lane == 0) {
                float avg = 0.f, rsg = 1.f;
                if (g.norm_ring) {
                    const float nn = g.nn_weight;
                    avg = a / nn;
                    rsg = 1.0f / sqrtf((q - a * a / nn) / nn);
                }
                red[8 + wave] = avg; red[12 + wave] = rsg;
            }
        }
        __syncthreads();
        if (g.norm_ring && tid < 4 * g.nring) {
            const int s = tid / g.nring, i = tid - s * g.nring;
            const int4 ri = g.ringinfo[i];
            bufs[s * g.sbuf + ri.x] -= red[8 + s] * (float)ri.z;
        }
        __syncthreads();
        // write-out through the gather table: 4 consecutive ring slots of one (bin, row) per thread
        float4 *dstA = reinterpret_cast<float4 *>(A + ((size_t)p * ngroup + grp) * g.a_blk);
        if (!RA_DBG(g, 64))
        for (int q = tid; q < g.LBP * 2; q += blockDim.x) {
            const int4 sidx = g.a_src4[q];
            // table element = LDS float index | offset slot << 24 (the slot picks 1/sigma); -1 = padding
            auto fetch = [&](int code) -> float {
                if (code < 0) return 0.f;
                return bufs[code & 0xffffff] * red[12 + (code >> 24)];
            };
            float4 v;
            v.x = fetch(sidx.x); v.y = fetch(sidx.y); v.z = fetch(sidx.z); v.w = fetch(sidx.w);
            dstA[q] = v;
        }
        __syncthreads();
    }
}

// K0a: references -> Polar2Dm(cnx,cny) + Frngs, natural (padded ring buffer) layout in HBM.
// (test_mref_gpu_align.py:1015-1016)
__global__ __launch_bounds__(256) void ref_polar_fft_kernel(DevGeom g, const float *__restrict__ refs,
                                                            int nref, float *__restrict__ out)
{
    extern __shared__ __align__(16) float lds[];
    const int npix = g.nx * g.nx;
    float *img = lds;
    float *bufs = lds + ((npix + 3) & ~3);
    float *red = bufs + g.sbuf;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (r >= nref) return;
    for (int i = tid; i < npix; i += blockDim.x) img[i] = refs[(size_t)r * npix + i];
    for (int i = tid; i < g.sbuf; i += blockDim.x) bufs[i] = 0.f;
    if (tid == 0) { red[0] = 0.f; red[1] = 1.f; }
    __syncthreads();
    const float c = (float)g.cnx;
    for (int i = tid; i < g.lcirc; i += blockDim.x)
        bufs[g.samp_dst[i]] = bilinear_1b(img, g.nx, g.samp_dx[i] + c, g.samp_dy[i] + c);
    __syncthreads();
    ring_fft_all(g, bufs, 1, red, red + 1);
    for (int i = tid; i < g.lring; i += blockDim.x) out[(size_t)r * g.lring + i] = bufs[i];
}

// K0b: Applyws + 1/maxrin, packed as the MFMA B operand (layout: ralign_geom.h panel_pos):
//   B[rtile][LBP*16], column = 2*(ref in tile) + (0: Re, 1: Im); padding zero.
// (Applyws: test_mref_gpu_align.py:1017)
__global__ void pack_refs_kernel(DevGeom g, const float *__restrict__ refspec, int nref, int nrtile,
                                 float *__restrict__ B)
{
    const int per_tile = g.LBP * 16, total = nrtile * per_tile;
    const float inv = 1.0f / (float)g.maxrin;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int rt = idx / per_tile, f = idx - rt * per_tile;
        const int code = g.b_src[f];
        float v = 0.f;
        if (code >= 0) {
            const int e = code >> 4, col = code & 15, ref = rt * g.rpt + (col >> 1);
            if ((col >> 1) < g.rpt && ref < nref) v = refspec[(size_t)ref * g.lring + g.ent_src[e] + (col & 1)] * g.ent_wgt[e] * inv;
        }
        B[idx] = v;
    }
}

// diagnostic: prepared references in EMAN2 packing (after Applyws, without the 1/maxrin)
__global__ void unpack_refs_kernel(DevGeom g, const float *__restrict__ refspec, int nref,
                                   const int *__restrict__ ring_off, const int *__restrict__ numr,
                                   const float *__restrict__ wr, float *__restrict__ out)
{
    int r = blockIdx.x;
    for (int i = 0; i < g.nring; i++) {
        int n = numr[3 * i + 2], o = numr[3 * i + 1] - 1, ro = ring_off[i];
        const float *s = refspec + (size_t)r * g.lring + ro;
        float *d = out + (size_t)r * g.lcirc + o;
        float w = wr[i];
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            float v;
            if (j == 0) v = s[0] * w;
            else if (j == 1) v = s[n] * ((n == g.maxrin) ? w : 0.5f * w);
            else v = s[j] * w;
            d[j] = v;
        }
    }
}

// diagnostic: A panels -> ring spectra in EMAN2 packing [n][nshift][lcirc] (what Frngs leaves in
// `cimage` inside Util.multiref_polar_ali_2d), for bin-for-bin tests of the polar kernel
// stats (generic path only): {avg, 1/sigma} per particle-offset; the generic polar kernel writes raw spectra
__global__ void unpack_spectra_kernel(DevGeom g, const float *__restrict__ A, int n, const int *__restrict__ numr,
                                      float *__restrict__ out, const float2 *__restrict__ stats, const float *__restrict__ state = nullptr)
{
    const int m = blockIdx.x;                  // particle * nshift + shift
    const int p = m / g.nshift, sft = m - p * g.nshift;
    if (p >= n) return;
    size_t ent = (size_t)p * g.ent_stride + sft;          // entry of this particle-offset: block ent >> 2, slot ent & 3
    float *dst = out + (size_t)m * g.lcirc;
    if (g.ent_base) {          // live-offset lists: an offset outside the particle's window has no entry (zeros)
        const int j = live_index(g, particle_window(g, state[2 * p], state[2 * p + 1]), sft);
        if (j < 0) {
            for (int i = threadIdx.x; i < g.lcirc; i += blockDim.x) dst[i] = 0.f;
            return;
        }
        ent = (size_t)g.ent_base[p] + j;
    }
    const float *blk = A + (ent >> 2) * g.a_blk;
    const int row0 = (int)(ent & 3) * 2;
    for (int i = 0; i < g.nring; i++) {
        const int nlen = numr[3 * i + 2], o = numr[3 * i + 1] - 1;
        for (int j = threadIdx.x; j < nlen; j += blockDim.x) {
            // packed slot j: 0 -> Re X0, 1 -> Re X(n/2), 2k -> Re Xk, 2k+1 -> Im Xk
            const int k = (j == 0) ? 0 : (j == 1 ? nlen / 2 : j >> 1), comp = (j < 2) ? 0 : (j & 1);
            const int e = g.bin_off[k] + (i - g.bin_first[k]);
            const int2 ap = g.ent_apos[e];
            float v = blk[ap.x + (row0 + comp) * ap.y];
            if (stats) {
                const float2 st = stats[ent];
                if (j == 0) v -= st.x * (float)nlen;
                v *= st.y;
            }
            dst[o + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: CCF contraction on MFMA + in-LDS inverse FFT + wavefront argmax.
// Restates Util::Crosrng_ms for every (particle-offset, reference) pair of an 8 x 8 tile:
//   per Fourier bin k:  [16 rows = 8 m x (Re,Im) D] x [rings] x [16 cols = 8 refs x (Re,Im) C]
//   a=c1d1 b=c1d2 c=c2d1 d=c2d2 ;  Q_k = (a+d) + i(c-b) ;  T_k = (a-d) - i(b+c)
//   Z_k = Q_k + i T_k,  Z_{N-k} = conj(Q_k) + i conj(T_k)  -> one N-point complex inverse
//   FFT yields q (real part) and t (imaginary part) of Crosrng_ms together.
// 8 waves: phase 1 gives each wave every 8th bin (operands straight from HBM/L2 into registers:
// each MFMA lane owns KP/4 contiguous floats of its row, fetched as 16-B loads one bin ahead);
// phase 2 transforms 4 pairs per wave and round, 16 lanes per transform.
// record of the two-kernel path: the 7-point neighbourhood of the maximum travels with the candidate
// and Util::prb1d runs once per particle in finalize_kernel instead of once per pair
struct CandT { float val; int jtot; int refmir; float t7[7]; };
// The particle-resident kernels keep the best reference per search offset only.  When another reference of the same offset
// comes within RA_TIE_RTOL of it, the record's jtot word carries that runner-up as well, so that finalize_kernel can hand
// both to the exact re-evaluation: bits 0-12 jtot | 13-20 runner-up reference + 1 (0: none) | 21 its mirror flag | 22-31 its jtot
#define RA_TIE_RTOL 3.0e-6f      // f32 peaks closer than this (relative) are re-evaluated in the CPU path's arithmetic
// ... and records of DIFFERENT search offsets under Normalize_ring closer than this: the CPU path's sigma of an offset comes out of a
// float sum over lcirc samples (a rounding walk of ~1e-6 of the peak at 5816 samples, one sigma), so two offsets whose exact maxima
// differ by 3e-6 are still ordered either way by it now and then.  exact_candidate repeats the walk bit for bit.
#define RA_TIE_RTOL_OFFSETS 1.0e-5f
__device__ __forceinline__ int cand_pack_runner(int jtot, const CandT &ru)
{
    return (jtot & 0x1fff) | ((((ru.refmir & 0xffff) + 1) & 0xff) << 13) | (((ru.refmir >> 16) & 1) << 21) | ((ru.jtot & 0x3ff) << 22);
}
__device__ __forceinline__ int cand_jtot(int w) { return w & 0x1fff; }

// Util::prb1d, npoint 7: sub-bin position of the maximum relative to the centre sample
__device__ __forceinline__ float prb1d7(const float *t)
{
    const double t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5], t6 = t[6];
    const double c2 = 49. * t0 + 6. * t1 - 21. * t2 - 32. * t3 - 27. * t4 - 6. * t5 + 31. * t6;
    const double c3 = 5. * t0 - 3. * t2 - 4. * t3 - 3. * t4 + 5. * t6;
    return (c3 != 0.0) ? (float)(c2 / (2.0 * c3) - 4) : 0.f;
}

template <int N> struct IfftPlan;
template <> struct IfftPlan<256> { static constexpr int R1 = 16, R2 = 16; };
template <> struct IfftPlan<128> { static constexpr int R1 = 16, R2 = 8; };
template <> struct IfftPlan<64>  { static constexpr int R1 = 8,  R2 = 8; };
template <> struct IfftPlan<32>  { static constexpr int R1 = 8,  R2 = 4; };

// LDS image of the CCF spectra: one complex slot padded in after every 16, pair stride 2*(N+N/16)+2
// dwords.  This makes (i) the 16 pairs x same-bin ds_write_b64 of phase 1, (ii) the contiguous
// and (iii) the stride-16 accesses of the two FFT passes bank-conflict free, with pairs p and
// p+16 sharing a half-wave in phase 2 (their bases differ by 32 banks mod 64).
template <int N> struct ZLayout {
    static constexpr int kPairStride = 2 * (N + N / 16) + 2;
    static __device__ __forceinline__ int addr(int pair, int slot) { return pair * kPairStride + 2 * (slot + (slot >> 4)); }
};

// phase 1 of ccf_kernel for one class of bins (same number NS of ring slots per MFMA lane, so
// every load and MFMA below has a static shape).  Wave w takes bins k0+w, k0+w+NW, ...; the
// operands of the next bin are requested before the current one is multiplied.
template <int NS> struct Operands { float a[NS], b[NS]; };

template <int NS>
__device__ __forceinline__ void load_operands(Operands<NS> &o, const float *__restrict__ pa, const float *__restrict__ pb,
                                              int la, int lb)
{
    constexpr int N4 = NS >> 2;
#pragma unroll
    for (int q = 0; q < N4; q++) {
        float4 va = *reinterpret_cast<const float4 *>(pa + q * 128 + la * 4);
        float4 vb = *reinterpret_cast<const float4 *>(pb + q * 256 + lb * 4);
        o.a[4 * q] = va.x; o.a[4 * q + 1] = va.y; o.a[4 * q + 2] = va.z; o.a[4 * q + 3] = va.w;
        o.b[4 * q] = vb.x; o.b[4 * q + 1] = vb.y; o.b[4 * q + 2] = vb.z; o.b[4 * q + 3] = vb.w;
    }
    int oa = N4 * 128, ob = N4 * 256;
    if constexpr ((NS & 2) != 0) {
        float2 va = *reinterpret_cast<const float2 *>(pa + oa + la * 2);
        float2 vb = *reinterpret_cast<const float2 *>(pb + ob + lb * 2);
        o.a[4 * N4] = va.x; o.a[4 * N4 + 1] = va.y; o.b[4 * N4] = vb.x; o.b[4 * N4 + 1] = vb.y;
        oa += 64; ob += 128;
    }
    if constexpr ((NS & 1) != 0) { o.a[NS - 1] = pa[oa + la]; o.b[NS - 1] = pb[ob + lb]; }
}

template <int N, int NS>
__device__ __forceinline__ void contract_bin(const Operands<NS> &o, float *Z, int pair, int odd, int k)
{
    typedef ZLayout<N> ZL;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[s], o.b[s], acc, 0, 0, 0);
    // (one accumulation chain: the sum over rings stays the exact f32 fma chain the parity tests pin)
    // 2x2 block exchange between the Re/Im column lanes of one reference
    float s0 = odd ? acc[0] : acc[2], s1 = odd ? acc[1] : acc[3];
    float r0 = swap_lane_pair(s0), r1 = swap_lane_pair(s1);
    float ca = odd ? r0 : acc[0], cb = odd ? r1 : acc[1];
    float cc = odd ? acc[2] : r0, cd = odd ? acc[3] : r1;
    float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
    *reinterpret_cast<float2 *>(Z + ZL::addr(pair, k)) = make_float2(apd + bpc, cmb + amd);
    *reinterpret_cast<float2 *>(Z + ZL::addr(pair, (N - k) & (N - 1))) = make_float2(apd - bpc, amd - cmb);
}

template <int N, int NS, int NW>
__device__ __forceinline__ void contract_class(const float *__restrict__ Ablk, const float *__restrict__ Bt, float *Z,
                                               int kbeg, int kend, int p0_class, int wave, int la, int lb, int pair,
                                               int odd)
{
    // panel of bin k in this class starts at (p0_class + (k - kbeg) * 4*NS) entries
    // (`wave` arrives rotated by the class's start wave)
    int k = kbeg + wave;
    if (k >= kend) return;
    Operands<NS> cur, nxt;
    load_operands<NS>(cur, Ablk + (size_t)(p0_class + (k - kbeg) * 4 * NS) * 8, Bt + (size_t)(p0_class + (k - kbeg) * 4 * NS) * 16, la, lb);
    while (true) {
        const int k1 = k + NW;
        if (k1 < kend)
            load_operands<NS>(nxt, Ablk + (size_t)(p0_class + (k1 - kbeg) * 4 * NS) * 8, Bt + (size_t)(p0_class + (k1 - kbeg) * 4 * NS) * 16, la, lb);
        contract_bin<N, NS>(cur, Z, pair, odd, k);
        if (k1 >= kend) break;
        const int k2 = k1 + NW;
        if (k2 < kend)
            load_operands<NS>(cur, Ablk + (size_t)(p0_class + (k2 - kbeg) * 4 * NS) * 8, Bt + (size_t)(p0_class + (k2 - kbeg) * 4 * NS) * 16, la, lb);
        contract_bin<N, NS>(nxt, Z, pair, odd, k1);
        if (k2 >= kend) break;
        k = k2;
    }
}

// two-kernel path: the same transform, but the 7-point neighbourhood of the maximum is picked out of the REGISTERS
// (lane j holds the outputs with index = j mod R1; every one of the 7 neighbours sits in a different lane, which
// selects it with a compare chain and stores it straight into the candidate record) -- the transformed sequence is
// never written back to LDS.  Records: pc[pair] = {val, jtot, refmir = mirror << 16 | ref0 + (pair & 7), t7[7]}.
template <int N, int NP, int REFMASK = 7>
__device__ __forceinline__ void ifft_argmax(float *Z, CandT *pc, const float2 *twl, int pairA, int pairB, int j, int ref0,
                                            bool nomirror = false)
{
    typedef ZLayout<N> ZL;
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2, TWS = IfftPlan<N>::R2;
    const int pr[2] = {pairA, pairB};
    float2 v[NP][16];
    if (j < R2) {
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int k1 = 0; k1 < R1; k1++) v[q][k1] = *reinterpret_cast<const float2 *>(Z + ZL::addr(pr[q], R2 * k1 + j));
#pragma unroll
        for (int q = 0; q < NP; q++) {
            Dft<1, R1>::run(v[q]);
#pragma unroll
            for (int n0 = 1; n0 < R1; n0++) v[q][n0] = cmul(v[q][n0], twl[n0 * TWS]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (j < R2) {
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int n0 = 0; n0 < R1; n0++) *reinterpret_cast<float2 *>(Z + ZL::addr(pr[q], n0 * R2 + j)) = v[q][n0];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (j < R1) {
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int k0 = 0; k0 < R2; k0++) v[q][k0] = *reinterpret_cast<const float2 *>(Z + ZL::addr(pr[q], j * R2 + k0));
#pragma unroll
        for (int q = 0; q < NP; q++) Dft<1, R2>::run(v[q]);
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        // Maximum first, index second: the running maximum of both sequences is one v_max per value, the 16-lane maximum
        // one DPP v_max per step, and only the sequence that wins (straight on ties: qn >= qm, Util::multiref_polar_ali_2d;
        // nomirror: Crosrng_ns, straight only) is searched for the position of its maximum -- the LAST element equal to it
        // (the CPU scan's ">=" keeps the last maximum; ascending n1 within the lane, the larger index across lanes).
        float bq = -1.0e20f, bt = -1.0e20f;
        if (j < R1) {
#pragma unroll
            for (int n1 = 0; n1 < R2; n1++) { bq = __builtin_fmaxf(bq, v[q][n1].x); bt = __builtin_fmaxf(bt, v[q][n1].y); }
        }
        // DPP row permutes (mirror 15-i, half mirror 7-i, quad [2,3,0,1], quad [1,0,3,2]) reach all 16 lanes of the row
#define RA_DPP_FMAX(X, CTRL) X = __builtin_fmaxf(X, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xF, 0xF, true)))
        RA_DPP_FMAX(bq, 0x140); RA_DPP_FMAX(bt, 0x140); RA_DPP_FMAX(bq, 0x141); RA_DPP_FMAX(bt, 0x141);
        RA_DPP_FMAX(bq, 0x4E); RA_DPP_FMAX(bt, 0x4E); RA_DPP_FMAX(bq, 0xB1); RA_DPP_FMAX(bt, 0xB1);
#undef RA_DPP_FMAX
        const bool mir = !nomirror && !(bq >= bt);
        const float best = mir ? bt : bq;
        float c[R2];
        int jt = -1;
        if (j < R1) {
            int nb = -1;
#pragma unroll
            for (int n1 = 0; n1 < R2; n1++) {
                c[n1] = mir ? v[q][n1].y : v[q][n1].x;
                nb = c[n1] == best ? n1 : nb;
            }
            jt = nb >= 0 ? R1 * nb + j : -1;
        }
#define RA_DPP_IMAX(CTRL) jt = max(jt, __builtin_amdgcn_update_dpp(0, jt, CTRL, 0xF, 0xF, true))
        RA_DPP_IMAX(0x140); RA_DPP_IMAX(0x141); RA_DPP_IMAX(0x4E); RA_DPP_IMAX(0xB1);
#undef RA_DPP_IMAX
        CandT *dst = pc + pr[q];
        // neighbour jt + k (k = -3..3) lives in lane (jt + k) mod R1 at register (jt + k) / R1
        const int d = (j - jt) & (R1 - 1);
        const int k = d <= 3 ? d : d - R1;
        if (j < R1 && k >= -3) {
            const int n1 = ((jt + k + N) & (N - 1)) / R1;
            float val = 0.f;
#pragma unroll
            for (int r = 0; r < R2; r++) val = (r == n1) ? c[r] : val;
            dst->t7[k + 3] = val;
        }
        if (j == 0) {
            dst->val = best;
            dst->jtot = jt + 1;
            dst->refmir = ((mir ? 1 : 0) << 16) | (ref0 + (pr[q] & REFMASK));
        }
    }
}

// One 512-point inverse transform per WAVE (search_solo_kernel): 512 = 8 x 8 x 8, three 8-point DFTs per lane with two transposes
// through the transform's own LDS image in between -- 16 registers of data, so it runs next to a full A slice, and a quarter of the
// dependent chain of the 16-lane form (16 lanes per transform, two 16-point columns and a 32-point row per lane: 13 k cycles per call on the wave timeline, the critical path of a pass).
//   k = 64 k2 + 8 k1 + k0,  n = n0 + 8 n1 + 64 n2,  W = e^{2 pi i / 512}
//   A [n0; k1 k0] = sum_k2 Z[64 k2 + 8 k1 + k0] W8^{k2 n0},   lane = 8 k1 + k0;   then * W^{(8 k1 + k0) n0}      (twa[n0][lane])
//   B [n0 n1; k0] = sum_k1 A W8^{k1 n1},                      lane = 8 n0 + k0;   then * W^{8 k0 n1}             (twb[n1][k0])
//   x [n]         = sum_k0 B W8^{k0 n2},                      lane = 8 n0 + n1:   outputs n = n0 + 8 n1 + 64 n2 in registers
// Argmax with the CPU scan's ">=" semantics (the LAST maximum; straight beats mirrored on equality) and the 7-point neighbourhood
// of the maximum picked out of the registers, as in ifft_argmax.  twa: [8][64], twb: [8][8] (ifft512_twiddles).
__device__ __forceinline__ void ifft512_twiddles(const float2 *__restrict__ tw512, float2 *twa, float2 *twb, int tid, int nthreads)
{
    for (int i = tid; i < 8 * 64; i += nthreads) {
        const float2 w = tw512[((i >> 6) * (i & 63)) & 511];
        twa[i] = make_float2(w.x, -w.y);
    }
    for (int i = tid; i < 64; i += nthreads) {
        const float2 w = tw512[(8 * (i >> 3) * (i & 7)) & 511];
        twb[i] = make_float2(w.x, -w.y);
    }
}
// an opaque copy of a lane index: what is computed from it stays where it is used.  The persistent kernels hold ~120 live registers
// in their passes; lane-dependent addresses of one-wave tasks (records, statistics, centres) and of the later phases of a pass that the
// compiler hoists out of the particle loop end up in scratch, and every reload is a round trip to L2 in front of a barrier
__device__ __forceinline__ int rf_own_lane(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ float wave_max_dpp(float v)
{
#define RA_DPP_FMAX(X, CTRL) X = __builtin_fmaxf(X, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(X), CTRL, 0xF, 0xF, true)))
    RA_DPP_FMAX(v, 0x140); RA_DPP_FMAX(v, 0x141); RA_DPP_FMAX(v, 0x4E); RA_DPP_FMAX(v, 0xB1);
#undef RA_DPP_FMAX
    const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return __builtin_fmaxf(__builtin_fmaxf(a0, a1), __builtin_fmaxf(a2, a3));
}
__device__ __forceinline__ int wave_imax_dpp(int v)
{
#define RA_DPP_IMAX(CTRL) v = max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true))
    RA_DPP_IMAX(0x140); RA_DPP_IMAX(0x141); RA_DPP_IMAX(0x4E); RA_DPP_IMAX(0xB1);
#undef RA_DPP_IMAX
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ void ifft512_wave_argmax(float *Z, CandT *dst, const float2 *twa, const float2 *twb, int pair, int lane, int refid,
                                                    bool nomirror)
{
    typedef ZLayout<512> ZL;
    float2 v[8];
    float *zp = Z + pair * ZL::kPairStride;
    auto slot = [&](int sl) { return reinterpret_cast<float2 *>(zp + 2 * (sl + (sl >> 4))); };
    // ---- stage 1: DFT-8 over k2
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) v[k2] = *slot(64 * k2 + lane);
    Dft<1, 8>::run(v);
#pragma unroll
    for (int n0 = 1; n0 < 8; n0++) v[n0] = cmul(v[n0], twa[n0 * 64 + lane]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int n0 = 0; n0 < 8; n0++) *slot(64 * n0 + lane) = v[n0];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- stage 2: lane = 8 n0 + k0: DFT-8 over k1
    {
        const int n0 = lane >> 3, k0 = lane & 7;
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) v[k1] = *slot(64 * n0 + 8 * k1 + k0);
        Dft<1, 8>::run(v);
#pragma unroll
        for (int n1 = 1; n1 < 8; n1++) v[n1] = cmul(v[n1], twb[n1 * 8 + k0]);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) *slot(64 * n0 + 8 * n1 + k0) = v[n1];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- stage 3: lane = 8 n0 + n1: DFT-8 over k0 (8 consecutive slots); outputs n = m + 64 n2, m = n0 + 8 n1
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) v[k0] = *slot(8 * lane + k0);
    Dft<1, 8>::run(v);
    const int m = (lane >> 3) + 8 * (lane & 7);
    float bq = -1.0e20f, bt = -1.0e20f;
#pragma unroll
    for (int r = 0; r < 8; r++) { bq = __builtin_fmaxf(bq, v[r].x); bt = __builtin_fmaxf(bt, v[r].y); }
    bq = wave_max_dpp(bq); bt = wave_max_dpp(bt);
    const bool mir = !nomirror && !(bq >= bt);
    const float best = mir ? bt : bq;
    float c[8];
    int nb = -1;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        c[r] = mir ? v[r].y : v[r].x;
        nb = c[r] == best ? r : nb;
    }
    const int jt = wave_imax_dpp(nb >= 0 ? 64 * nb + m : -1);
    // neighbour jt + k (k = -3 .. 3) lives in the lane whose m = (jt + k) mod 64, at register (jt + k) / 64
    const int d = (m - jt) & 63;
    const int k = d <= 3 ? d : d - 64;
    if (k >= -3) {
        const int r7 = ((jt + k + 512) & 511) >> 6;
        float val = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) val = (r == r7) ? c[r] : val;
        dst->t7[k + 3] = val;
    }
    if (lane == 0) {
        dst->val = best;
        dst->jtot = jt + 1;
        dst->refmir = ((mir ? 1 : 0) << 16) | refid;
    }
}

#ifndef RA_CCF_THREADS
#define RA_CCF_THREADS 1024
#endif
#define RA_CCF_MAXNS 12     // KP_k / 4 <= 12 rings per MFMA lane (nring <= 48)

template <int N>
__global__ __launch_bounds__(RA_CCF_THREADS, RA_CCF_THREADS >= 1024 ? 4 : 2) void ccf_kernel(DevGeom g, const float *__restrict__ A,
                                                             const float *__restrict__ B, int n_mtile, int nrtile,
                                                             int nref, CandT *__restrict__ cand)
{
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    typedef ZLayout<N> ZL;
    extern __shared__ __align__(16) float Z[];
    __shared__ CandT pc[64];
    const int mtile = blockIdx.x;
    if (mtile >= n_mtile) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = RA_CCF_THREADS / 64;

    // twiddles e^{+2 pi i n0 j / N} of the first FFT pass, [n0][j] in LDS (lane j reads column j)
    __shared__ float2 tws[R1 * R2];
    for (int i = tid; i < R1 * R2; i += RA_CCF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2) * (g.maxrin / N)) & (g.maxrin - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    const float2 *twl = tws + (lane & 15);

    // the reference tiles are swept by the same workgroup: later sweeps find this tile's A panels
    // in L2 / Infinity Cache instead of HBM
    for (int rtile = 0; rtile < nrtile; rtile++) {
    const int ref0 = rtile * g.rpt;                       // references [ref0, ref0 + nvalid) of this tile
    const int nvalid = min(g.rpt, nref - ref0);
    if (RA_DBG(g, ~0) && tid < 64) {   // profiling builds that skip a phase still emit in-range records
        pc[tid].val = 0.f; pc[tid].jtot = 1; pc[tid].refmir = min(ref0 + (tid & 7), nref - 1);
        for (int k = 0; k < 7; k++) pc[tid].t7[k] = 0.f;
    }
    // ---- phase 1: contraction, class by class (static shapes inside a class)
    if (!RA_DBG(g, 2)) {
        const int r16 = lane & 15, kk = lane >> 4;
        const int mt_eff = RA_DBG(g, 8192) ? (mtile & 63) : mtile;      // profiling: operands from an L2-resident subset
        const float *Ablk = A + (size_t)(2 * mt_eff + (r16 >> 3)) * g.a_blk;
        const float *Bt = B + (size_t)rtile * g.LBP * 16;
        const int odd = lane & 1;
        const int pair = (2 * (lane >> 4) + odd) * 8 + ((lane & 15) >> 1);
        const int la = kk * 8 + (r16 & 7), lb = kk * 16 + r16;
        int p0 = 0;
        for (int c = 0; c < g.n_class; c++) {
            const int kb = g.class_k0[c], ke = (c + 1 < g.n_class) ? g.class_k0[c + 1] : g.class_k0_end, ns = g.class_ns[c];
            switch (ns) {
#define RA_CASE(NSV) case NSV: contract_class<N, NSV, NW>(Ablk, Bt, Z, kb, ke, p0, (wave - g.class_rot[c] + NW) % NW, la, lb, pair, odd); break;
                RA_CASE(1) RA_CASE(2) RA_CASE(3) RA_CASE(4) RA_CASE(5) RA_CASE(6)
                RA_CASE(7) RA_CASE(8) RA_CASE(9) RA_CASE(10) RA_CASE(11) RA_CASE(12)
#undef RA_CASE
            default: break;
            }
            p0 += (ke - kb) * 4 * ns;
        }
    }
    __syncthreads();

    // ---- phase 2: N-point inverse FFT of every live pair, 16 lanes per transform, then argmax.
    // pair = 16*sub + b, b = (ref slot, particle-offset parity): liveness depends on b only,
    // so a wave is either wholly busy or wholly idle in a round.  With 8 waves a wave transforms two
    // pairs per lane group at a time (independent register chains hide the DFT's dependent latency).
    if (!RA_DBG(g, 1)) {
        const int j = lane & 15, sub = lane >> 4;
        const int nlive = 2 * nvalid;                 // live b values: (b & 7) < nvalid
        for (int idx = wave; idx < nlive; idx += 2 * NW) {
            const int idx2 = idx + NW;
            const bool two = NW < 16 && idx2 < nlive;     // 16 waves cover all 16 live slots in one round
            const int bA = (idx / nvalid) * 8 + idx % nvalid;
            const int bB = two ? (idx2 / nvalid) * 8 + idx2 % nvalid : bA;
            if constexpr (NW < 16) { if (two) { ifft_argmax<N, 2>(Z, pc, twl, 16 * sub + bA, 16 * sub + bB, j, ref0, g.nomirror != 0); continue; } }
            ifft_argmax<N, 1>(Z, pc, twl, 16 * sub + bA, 16 * sub + bA, j, ref0, g.nomirror != 0);
        }
    }
    __syncthreads();
    // ---- best reference of the tile per particle-offset (ascending ref, ">=": later wins); the 8 winning records are
    // copied dword by dword (no private copy of the struct)
    if (tid < 8 * (int)(sizeof(CandT) / 4)) {
        constexpr int W = sizeof(CandT) / 4;
        const int o = tid / W, wd = tid - o * W;
        float bv = pc[o * 8].val; int br = 0;
        for (int rr = 1; rr < nvalid; rr++) {
            const float v = pc[o * 8 + rr].val;
            if (v >= bv) { bv = v; br = rr; }
        }
        reinterpret_cast<int *>(cand + ((size_t)mtile * 8 + o) * nrtile + rtile)[wd] = reinterpret_cast<const int *>(pc + o * 8 + br)[wd];
    }
    __syncthreads();
    }   // rtile
}

// ------------------------------------------------------------------------------------------
// K3: per-particle reduction over search offsets (y outer, x inner) and reference tiles with
// the ">=" rule of Util::multiref_polar_ali_2d, ang_n, the ormq tail and combine_params2
// (test_mref_gpu_align.py:1043-1049).
// record of a particle whose sub-bin angle is to be re-evaluated with the CPU path's arithmetic (ralign_exact.h)
// alt[0 .. nalt): every OTHER candidate whose f32 peak lies within RA_TIE_RTOL of the winner's -- a runner-up reference of the winning
// offset (cand_pack_runner), the records of other offsets / reference tiles and their runner-ups -- (offset, reference tile,
// reference | mirror << 16).  The exact re-evaluation decides among all of them with the CPU path's ">=" order.  A ridge of the CCF
// in (shift, angle) -- half- or quarter-pixel steps -- puts three and more offsets within 1e-6 of each other: until round 6 the
// record carried ONE runner-up, and a third candidate that the f64 CCF ranks first was lost (scripts/dev/random_legacy_sweep.py,
// 77 / 29 / ts 0.5).  More than RA_TIE_ALTS near-ties: the first RA_TIE_ALTS in scan order.
#define RA_TIE_ALTS 7
struct RefineAlt { int bs, rt, refmir; };
struct RefineRec { int p, ref, mirror, jtot, bs, brt; float sxi, syi; int nalt; RefineAlt alt[RA_TIE_ALTS]; };

// rlist / rcount / rthr: particles whose prb1d is ill-conditioned (|c3| < rthr x max |b|; rthr < 0: every particle) are
// appended to rlist for refine_winner_kernel; rlist = null: none
// everything behind the scan over the records: prb1d / ang_n / ormq tail / combine_params2 of the winner (bs, brt), the result
// record and the new state, and the hand-over to refine_winner_kernel (flat peaks and float ties; second = the second-largest
// record peak; scan(fn) calls fn(offset, reference tile, record) for every in-window record of the particle, in scan order)
template <class Scan>
__device__ __forceinline__ void finalize_tail(const DevGeom &g, int p, const Window &w,
                                              CandT best, int bs, int brt, float second, float *__restrict__ state,
                                              ra_result *__restrict__ res, RefineRec *__restrict__ rlist, int *__restrict__ rcount,
                                              float rthr, int p_base, Scan scan)
{
    // (no contraction: the ormq tail sxs = sx co - sy so, sys = sx so + sy co is float arithmetic of two rounded products in the CPU
    // path; an fma moves sys by one float ulp on ~6 % of the particles -- scripts/dev/alpha_ulp.py)
#pragma clang fp contract(off)
    const float peak = best.val;
    const int mirror = best.refmir >> 16, ref = best.refmir & 0xffff;
    // Util::prb1d on the winner's neighbourhood, then Util::ang_n, mode F
    const int jword = best.jtot;         // jtot and, possibly, a runner-up reference of the same offset (cand_pack_runner)
    best.jtot = cand_jtot(jword);
    const float tot = (float)best.jtot + prb1d7(best.t7);
    const float ang = fmodf(((tot - 1.0f) / g.maxrin + 1.0f) * 360.0f, 360.0f);
    const float ixw = g.shift_x[bs], iyw = g.shift_y[bs];
    const float sx = -ixw, sy = -iyw;
    // the tail of the search: sxs = sx co - sy so, sys = sx so + sy co -- FLOAT arithmetic with float co, so in
    // Util::multiref_polar_ali_2d, Python doubles in sp_alignment.ormq
    const double a = (double)ang * M_PI / 180.0, c = cos(a), s = sin(a);
    double sxs, sys;
    if (g.mode == RA_MODE_MREF) {
        const float co = (float)c, so = (float)(-s);
        sxs = (double)(sx * co - sy * so); sys = (double)(sx * so + sy * co);
    } else {
        const double co = c, so = -s;
        sxs = (double)sx * co - (double)sy * so; sys = (double)sx * so + (double)sy * co;
    }
    // combine_params2(0,-sxi,-syi,0, ang,sxs,sys,mirror) in double: rotate the pre-shift, add
    const double tx = c * (double)(-w.sxi) + s * (double)(-w.syi) + sxs;
    const double ty = -s * (double)(-w.sxi) + c * (double)(-w.syi) + sys;
    double alpha = atan2(s, c) * 180.0 / M_PI;
    alpha = fmod(alpha, 360.0);
    if (alpha < 0) alpha += 360.0;
    if (alpha >= 360.0) alpha -= 360.0;
    ra_result r;
    r.alpha = (float)alpha; r.sx = (float)tx; r.sy = (float)ty;
    r.mirror = mirror; r.ref_id = ref; r.peak = peak; r.angle_bin = best.jtot; r.shift_idx = bs;
    res[p] = r;
    state[2 * p] = w.sxi + ixw;
    state[2 * p + 1] = w.syi + iyw;
    if (rlist) {
        const float *b = best.t7;
        const float c3 = 5.f * b[0] - 3.f * b[2] - 4.f * b[3] - 3.f * b[4] + 5.f * b[6];
        float tmax = 0.f, nb = -1.0e23f;
        for (int k = 0; k < 7; k++) { tmax = fmaxf(tmax, fabsf(b[k])); if (k != 3) nb = fmaxf(nb, b[k]); }
        // float ties: a neighbouring angular bin or another offset / reference tile within RA_TIE_RTOL of the winner (the f32
        // and the f64 CCF may order them differently); refine_winner_kernel decides them in the CPU path's arithmetic
        const float tol = RA_TIE_RTOL * fabsf(b[3]);
        const bool tie_bin = nb >= b[3] - tol;
        // (the walk grows with the sample count: 3.4e-8 sqrt(lcirc / 3) of sq, one sigma -- 1.1e-6 between two offsets at 5816 samples,
        // 3.7e-6 at the 71 k samples of a 256-pixel box with ou = 120: 1e-5 up to 10 k samples, 1e-7 sqrt(lcirc) beyond)
        const float thr = peak - (g.norm_ring ? fmaxf(RA_TIE_RTOL_OFFSETS, 1.0e-7f * sqrtf((float)g.lcirc)) : RA_TIE_RTOL) * fabsf(peak);
        const bool tie_rec = second >= thr;
        const int runner = (jword >> 13) & 0xff;          // another reference of the winning offset within the tolerance
        const bool copies = g.ref_dup && (g.ref_dup[2 * ref] != ref || g.ref_dup[2 * ref + 1] >= 0);      // the winner has copies in the stack
        if (rthr < 0.f || fabsf(c3) < rthr * tmax || tie_bin || tie_rec || runner || copies) {
            RefineRec *rec = rlist + atomicAdd(rcount, 1);
            // (p_base: index of the chunk's first particle when the list spans a whole call)
            rec->p = p_base + p; rec->ref = ref; rec->mirror = mirror; rec->jtot = best.jtot; rec->bs = bs; rec->brt = brt;
            rec->sxi = w.sxi; rec->syi = w.syi;
            int nalt = 0;
            auto add = [&](int as, int art, int aref, int amir) {
                if (nalt < RA_TIE_ALTS) { rec->alt[nalt].bs = as; rec->alt[nalt].rt = art; rec->alt[nalt].refmir = aref | (amir << 16); nalt++; }
            };
            if (runner) add(bs, brt, runner - 1, (jword >> 21) & 1);
            if (tie_rec)
                scan([&](int cs, int crt, const CandT &cr) {
                    if ((cs == bs && crt == brt) || !(cr.val >= thr)) return;
                    add(cs, crt, cr.refmir & 0xffff, cr.refmir >> 16);
                    const int ru = (cr.jtot >> 13) & 0xff;
                    if (ru) add(cs, crt, ru - 1, (cr.jtot >> 21) & 1);
                });
            rec->nalt = nalt;
        }
    }
}

__global__ void finalize_kernel(DevGeom g, const CandT *__restrict__ cand, int nrtile, int n,
                                float *__restrict__ state, ra_result *__restrict__ res,
                                const float *__restrict__ cs, RefineRec *__restrict__ rlist, int *__restrict__ rcount, float rthr, int p_base = 0)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    float peak = -1.0e23f, second = -1.0e23f;
    CandT best; best.val = peak; best.jtot = 1; best.refmir = 0;
    for (int k = 0; k < 7; k++) best.t7[k] = 0.f;
    int bs = 0, brt = 0;
    const int nx1 = 2 * g.nkx + 1;
    if (g.ent_base) {          // live-offset lists: the particle's records are those of its in-window offsets, in scan order
        const int e0 = g.ent_base[p], L = g.ent_base[p + 1] - e0;
        auto scan = [&](auto fn) {
            for (int j = 0; j < L; j++) {
                const int s = live_shift(g, w, j);
                for (int rt = 0; rt < nrtile; rt++) fn(s, rt, cand[((size_t)e0 + j) * nrtile + rt]);
            }
        };
        scan([&](int s, int rt, const CandT &c) {
            const float v = c.val;
            if (v >= peak) { second = peak; peak = v; best = c; bs = s; brt = rt; }
            else if (v >= second) second = v;
        });
        finalize_tail(g, p, w, best, bs, brt, second, state, res, rlist, rcount, rthr, p_base, scan);
        return;
    }
    auto scan = [&](auto fn) {
        for (int s = 0; s < g.nshift; s++) {
            const int iy = s / nx1 - g.nky, ix = s % nx1 - g.nkx;
            if (ix < -w.lkx || ix > w.rkx || iy < -w.lky || iy > w.rky) continue;
            for (int rt = 0; rt < nrtile; rt++) fn(s, rt, cand[((size_t)p * g.ent_stride + s) * nrtile + rt]);
        }
    };
    scan([&](int s, int rt, const CandT &c) {
        const float v = c.val;
        if (v >= peak) { second = peak; peak = v; best = c; bs = s; brt = rt; }
        else if (v >= second) second = v;
    });
    finalize_tail(g, p, w, best, bs, brt, second, state, res, rlist, rcount, rthr, p_base, scan);
}

// the same with one WAVE per particle, for paths that leave many records per particle (the generic kernels: 121 offsets x 13
// reference tiles at 256 x 256 / 100 references, where one thread per particle scans 63 KB on its own): lane l scans records
// l, l + 64, .. in scan order (offsets, then reference tiles), the lanes' (best, second) pairs are merged with the order of the
// sequential ">=" scan -- of equal peaks the later record wins -- and lane 0 finishes the particle
__global__ __launch_bounds__(64) void finalize_wave_kernel(DevGeom g, const CandT *__restrict__ cand, int nrtile, int n,
                                                           float *__restrict__ state, ra_result *__restrict__ res,
                                                           RefineRec *__restrict__ rlist, int *__restrict__ rcount, float rthr, int p_base = 0)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n) return;
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    // live-offset lists: the particle's records are those of its L in-window offsets, record li = j * nrtile + rt at entry e0 + j
    const bool live = g.ent_base != nullptr;
    const int e0 = live ? g.ent_base[p] : p * g.ent_stride, L = live ? g.ent_base[p + 1] - e0 : g.nshift;
    const int nx1 = 2 * g.nkx + 1, nrec = L * nrtile;
    float bv = -1.0e23f, sv = -1.0e23f;
    int bi = -1, si = -1;                              // linear record index s * nrtile + rt of the best / second record
    for (int li = lane; li < nrec; li += 64) {
        const int s = li / nrtile;
        const int iy = s / nx1 - g.nky, ix = s % nx1 - g.nkx;
        if (!live && (ix < -w.lkx || ix > w.rkx || iy < -w.lky || iy > w.rky)) continue;
        const float v = cand[(size_t)e0 * nrtile + li].val;
        if (v >= bv) { sv = bv; si = bi; bv = v; bi = li; }
        else if (v >= sv) { sv = v; si = li; }
    }
    auto better = [](float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia > ib); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float obv = __shfl_xor(bv, o), osv = __shfl_xor(sv, o);
        const int obi = __shfl_xor(bi, o), osi = __shfl_xor(si, o);
        // new best = the better of the two bests; new second = the better of (the other best, the two seconds)
        float lv, cv; int lix, cix;
        if (better(obv, obi, bv, bi)) { lv = bv; lix = bi; bv = obv; bi = obi; } else { lv = obv; lix = obi; }
        if (better(osv, osi, sv, si)) { cv = osv; cix = osi; } else { cv = sv; cix = si; }
        if (better(lv, lix, cv, cix)) { sv = lv; si = lix; } else { sv = cv; si = cix; }
    }
    if (lane != 0) return;
    CandT best; best.val = -1.0e23f; best.jtot = 1; best.refmir = 0;
    for (int k = 0; k < 7; k++) best.t7[k] = 0.f;
    int bs = 0, brt = 0;
    if (bi >= 0) { best = cand[(size_t)e0 * nrtile + bi]; bs = bi / nrtile; brt = bi - bs * nrtile; }
    if (live) bs = bi >= 0 ? live_shift(g, w, bs) : 0;          // position in the live list -> index of the offset list
    auto scan = [&](auto fn) {
        for (int li = 0; li < nrec; li++) {
            int s = li / nrtile;
            const int rt = li - s * nrtile;
            if (live) s = live_shift(g, w, s);
            else {
                const int iy = s / nx1 - g.nky, ix = s % nx1 - g.nkx;
                if (ix < -w.lkx || ix > w.rkx || iy < -w.lky || iy > w.rky) continue;
            }
            fn(s, rt, cand[(size_t)e0 * nrtile + li]);
        }
    };
    finalize_tail(g, p, w, best, bs, brt, sv, state, res, rlist, rcount, rthr, p_base, scan);
}

// ------------------------------------------------------------------------------------------
// K4: rot_shift2D(img, alpha, sx, sy, mirror) ("quadratic", "background") and per-class
// even/odd accumulation (test_mref_gpu_align.py:1055-1057; kernel_sum_oe :48-80).
// The arithmetic follows the reference tree's own restatement of
// rot_scale_trans2D_background / quadri_background (notebook/02 cell 2).
__device__ __forceinline__ float quadri_background_1b(const float *fdata, int nx, int ny, float xx, float yy,
                                                      int xnew, int ynew)
{
#pragma clang fp contract(off)
    float x = xx, y = yy;
    // (written as the negation of "inside": a NaN position -- the parameters of a NaN particle -- takes the background branch too,
    // instead of (int)NaN = 0 indexing one row in front of the image)
    if (!((x >= 1.0f) && (x < (float)(nx + 1)) && (y >= 1.0f) && (y < (float)(ny + 1)))) { x = (float)xnew; y = (float)ynew; }
    int i = (int)x, j = (int)y;
    float dx0 = x - i, dy0 = y - j;
    int ip1 = i + 1, im1 = i - 1, jp1 = j + 1, jm1 = j - 1;
    if (ip1 > nx) ip1 -= nx;
    if (im1 < 1) im1 += nx;
    if (jp1 > ny) jp1 -= ny;
    if (jm1 < 1) jm1 += ny;
#define RA_FD(i_, j_) fdata[((j_) - 1) * nx + ((i_) - 1)]
    float f0 = RA_FD(i, j);
    float c1 = RA_FD(ip1, j) - f0;
    float c2 = (c1 - f0 + RA_FD(im1, j)) * 0.5f;
    float c3 = RA_FD(i, jp1) - f0;
    float c4 = (c3 - f0 + RA_FD(i, jm1)) * 0.5f;
    float dxb = dx0 - 1, dyb = dy0 - 1;
    int hxc = (dx0 >= 0) ? 1 : -1, hyc = (dy0 >= 0) ? 1 : -1;
    int ic = i + hxc, jc = j + hyc;
    if (ic > nx) ic -= nx; else if (ic < 1) ic += nx;
    if (jc > ny) jc -= ny; else if (jc < 1) jc += ny;
    float c5 = ((RA_FD(ic, jc) - f0 - hxc * c1 - (hxc * (hxc - 1.0f)) * c2 - hyc * c3 - (hyc * (hyc - 1.0f)) * c4) * (hxc * hyc));
#undef RA_FD
    return f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
}

// 512 threads per particle: the image fills a quarter of the LDS, so four workgroups share a CU -- with 256 threads that is four
// waves per SIMD for a kernel paced by its chain of address -> six LDS taps -> interpolate (243 -> 200 us per 7143 particles)
#define RA_XF_THREADS 512
__global__ __launch_bounds__(RA_XF_THREADS) void transform_kernel(int nx, const float *__restrict__ particles, int n,
                                                        int index0, const ra_result *__restrict__ res,
                                                        float *__restrict__ aligned, float *__restrict__ sums,
                                                        int *__restrict__ counts)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) float img[];
    const int p = blockIdx.x, tid = threadIdx.x, npix = nx * nx;
    if (p >= n) return;
    __shared__ float trig[2];
    const float *src = particles + (size_t)p * npix;
    for (int i = tid; i < npix; i += blockDim.x) img[i] = src[i];
    const ra_result r = res[p];
    const float ang = r.alpha * (float)M_PI / 180.0f;
    if (tid == 0) { trig[0] = (float)cos((double)ang); trig[1] = (float)sin((double)ang); }     // once per particle, not per thread
    __syncthreads();
    float delx = r.sx, dely = r.sy;
    while (delx >= (float)nx) delx -= nx;
    while (delx <= -(float)nx) delx += nx;
    while (dely >= (float)nx) dely -= nx;
    while (dely <= -(float)nx) dely += nx;
    const int xc = nx / 2, yc = nx / 2;
    const float shiftxc = xc + delx, shiftyc = yc + dely;
    const float cang = trig[0], sang = trig[1];
    float *dsum = sums ? sums + ((size_t)r.ref_id * 2 + ((index0 + p) & 1)) * npix : nullptr;
    float *dal = aligned ? aligned + (size_t)p * npix : nullptr;
    const int mstart = 1 - nx % 2;
    const unsigned nx_rcp = 0xFFFFFFFFu / (unsigned)nx + 1u;      // i / nx = (i * nx_rcp) >> 32, exact for i * nx < 2^32
    for (int i = tid; i < npix; i += blockDim.x) {
        const int iy = (int)__umulhi((unsigned)i, nx_rcp), ix = i - iy * nx;
        float y = (float)iy - shiftyc;
        float ycang = y * cang + yc;
        float ysang = -y * sang + xc;
        float x = (float)ix - shiftxc;
        float xold = x * cang + ysang;
        float yold = x * sang + ycang;
        float v = quadri_background_1b(img, nx, nx, xold + 1.0f, yold + 1.0f, ix + 1, iy + 1);
        // xform.mirror(x): columns [1 - nx%2, nx) reversed
        int ox = ix;
        if (r.mirror && ix >= mstart) ox = mstart + (nx - 1) - ix;
        const int o = iy * nx + ox;
        if (dal) dal[o] = v;
        if (dsum) atomicAdd(dsum + o, v);
    }
    if (counts && tid == 0) atomicAdd(counts + r.ref_id, 1);
}

// K4b: per-class even/odd sums in PARTICLE ORDER (the order Util.add_img is called in on the CPU
// path, test_mref_gpu_align.py:1056-1057), hence bitwise reproducible: one workgroup per
// (class, parity, 256-pixel tile) walks the batch and adds the aligned images that belong to it.
// Same scan-all-particles shape as the reference's cu_average_batch_m (gpu_aln_noref.cu:1232-1274).
// With few classes a (class, parity) holds thousands of images and there are only 2 nref x tiles workgroups: the member
// list is then cut into gridDim.z contiguous runs (`partial` != null): workgroup z adds its run in particle order into
// partial[z][class, parity][pixel], and class_sum_combine_kernel adds the runs to the sums in run order -- a fixed
// association ((sums + run0) + run1) + ..., bitwise reproducible run to run like the single-run case.
// member lists of a chunk, once per chunk instead of once per workgroup of class_sum_kernel: segment = (class, parity),
// one wave per segment compacts the particle indices in particle order (ballot + prefix count keeps the order)
__global__ __launch_bounds__(64) void class_members_kernel(const ra_result *__restrict__ res, int n, int index0,
                                                           int *__restrict__ members, int *__restrict__ mcount)
{
    const int seg = blockIdx.x, cls = seg >> 1, par = seg & 1, lane = threadIdx.x;
    int *mem = members + (size_t)seg * n;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool mine = i < n && res[i].ref_id == cls && ((index0 + i) & 1) == par;
        const unsigned long long m = __ballot(mine);
        if (mine) mem[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        base += __popcll(m);
    }
    if (lane == 0) mcount[seg] = base;
}

// gmembers / gcount: the lists of class_members_kernel ([segments][n], [segments]); null: every workgroup compacts its own
// list into LDS (segment counts too large for the list buffer)
__global__ __launch_bounds__(256) void class_sum_kernel(int npix, const float *__restrict__ aligned,
                                                        const ra_result *__restrict__ res, int n, int index0,
                                                        float *__restrict__ sums, int *__restrict__ counts,
                                                        float *__restrict__ partial, const int *__restrict__ gmembers,
                                                        const int *__restrict__ gcount)
{
    extern __shared__ int lmembers[];         // [n] particle indices of this (class, parity), ascending (gmembers == null only)
    __shared__ int nmem;
    const int seg = blockIdx.x, cls = seg >> 1, par = seg & 1;
    const int pix = blockIdx.y * blockDim.x + threadIdx.x;
    const bool live = pix < npix;
    const int *members = gmembers ? gmembers + (size_t)seg * n : lmembers;
    int cnt;
    if (gmembers) {
        cnt = gcount[seg];
    } else {
        // wave 0 compacts the member list in particle order (ballot + prefix count keeps the order)
        if (threadIdx.x < 64) {
            int base = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + threadIdx.x;
                const bool mine = i < n && res[i].ref_id == cls && ((index0 + i) & 1) == par;
                const unsigned long long m = __ballot(mine);
                if (mine) lmembers[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = i;
                base += __popcll(m);
            }
            if (threadIdx.x == 0) nmem = base;
        }
        __syncthreads();
        cnt = nmem;
    }
    const int nrun = gridDim.z, run = blockIdx.z;
    const int j0 = (int)((long long)cnt * run / nrun), j1 = (int)((long long)cnt * (run + 1) / nrun);
    if (live) {
        float acc = partial ? 0.f : sums[(size_t)seg * npix + pix];
        int j = j0;
        for (; j + 4 <= j1; j += 4) {          // four loads in flight, added in particle order
            const float a0 = aligned[(size_t)members[j] * npix + pix], a1 = aligned[(size_t)members[j + 1] * npix + pix];
            const float a2 = aligned[(size_t)members[j + 2] * npix + pix], a3 = aligned[(size_t)members[j + 3] * npix + pix];
            acc += a0; acc += a1; acc += a2; acc += a3;
        }
        for (; j < j1; j++) acc += aligned[(size_t)members[j] * npix + pix];
        if (partial) partial[((size_t)run * gridDim.x + seg) * npix + pix] = acc;
        else sums[(size_t)seg * npix + pix] = acc;
    }
    if (counts && blockIdx.y == 0 && run == 0 && threadIdx.x == 0 && cnt) atomicAdd(counts + cls, cnt);
}

__global__ __launch_bounds__(256) void class_sum_combine_kernel(int npix, int nseg, int nrun, const float *__restrict__ partial,
                                                                float *__restrict__ sums)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nseg * npix) return;
    float acc = sums[i];
    for (int r = 0; r < nrun; r++) acc += partial[(size_t)r * nseg * npix + i];
    sums[i] = acc;
}

// K4c: rot_shift2D and the class sums in one pass: the aligned image is never written.  Workgroup (segment =
// (class, parity), run) walks over its contiguous share of the segment's member list IN PARTICLE ORDER: image -> LDS, every
// thread interpolates its pixels (the arithmetic of transform_kernel, bit for bit) and adds them to accumulators it keeps in
// registers; the next member's image is on its way from HBM meanwhile (register prefetch); cos / sin come from
// transform_trig_kernel.  partial[run][segment][pixel] receives the run's sum; class_sum_combine_kernel adds the runs in run
// order: a fixed association, bitwise reproducible run to run (the same shape as class_sum_kernel's runs).
// Replaces cu_transform_batch + cu_average_batch_m / kernel_sum_oe (cuda/gpu_aln_noref.cu:1145-1197, 1232-1274;
// test_mref_gpu_align.py:48-80, 449-453), which write and re-read the aligned stack.
#define RA_XS_THREADS 1024
// quadri_background_1b on a copy of the image with a wrap-around border of one pixel (row stride pst, P = pointer to the
// element of 1-based pixel (0, 0)): the periodic neighbours ip1 / im1 / jp1 / jm1 / (ic, jc) are plain offsets.  The float
// operations are those of quadri_background_1b in the same order; two simplifications are exact: hxc = hyc = 1 always (x >= 1
// after the background rule, so dx0 = x - (int)x >= 0), which turns hxc * c1 into c1 and drops the terms multiplied by
// hxc (hxc - 1) = 0.
__device__ __forceinline__ float quadri_background_wrap(const float *P, int pst, int nx, float xx, float yy, int xnew, int ynew)
{
#pragma clang fp contract(off)
    float x = xx, y = yy;
    if (!((x >= 1.0f) && (x < (float)(nx + 1)) && (y >= 1.0f) && (y < (float)(nx + 1)))) { x = (float)xnew; y = (float)ynew; }          // (NaN: background)
    const int i = (int)x, j = (int)y;
    const float dx0 = x - i, dy0 = y - j;
    const float *q = P + j * pst + i;
    const float f0 = q[0];
    const float c1 = q[1] - f0;
    const float c2 = (c1 - f0 + q[-1]) * 0.5f;
    const float c3 = q[pst] - f0;
    const float c4 = (c3 - f0 + q[-pst]) * 0.5f;
    const float dxb = dx0 - 1, dyb = dy0 - 1;
    const float c5 = q[pst + 1] - f0 - c1 - c3;
    return f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
}

// Thread (tx, ty) = (tid % nx, tid / nx) owns column tx of rows ty, ty + RY, ty + 2 RY, .. (RY = 1024 / nx rows per sweep of
// the workgroup, NPT sweeps): the column terms of the rotation -- mirror, x, x cos, x sin -- are formed once per member, pixel
// indices are tid + k RY nx, and nothing per-pixel has to live in registers besides the accumulator and the prefetched value.
// NPF sweeps fill the LDS image (NPF * (1024 / nx) >= nx), NPT sweeps of output rows per workgroup: larger boxes are cut into
// gridDim.z row bands (NPT * gridDim.z >= NPF) so that a thread keeps at most 9 accumulators -- every band loads the whole image
// (any pixel may be a tap), but 16 - 20 accumulators per thread next to their prefetch registers spilled 75 - 126 registers.
template <int NPT, int NPF = NPT>
__global__ __launch_bounds__(RA_XS_THREADS) void transform_sum_kernel(int nx, const float *__restrict__ particles, int n, int index0,
                                                                      const ra_result *__restrict__ res, const float2 *__restrict__ trig,
                                                                      const int *__restrict__ members,
                                                                      const int *__restrict__ mcount, int mstride,
                                                                      float *__restrict__ partial)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) float img[];          // [(nx + 2)][pst]: rows / columns 0 and nx + 1 repeat nx and 1
    const int seg = blockIdx.x, run = blockIdx.y, nrun = gridDim.y, tid = threadIdx.x, npix = nx * nx;
    const int pst = (nx + 2) | 1;
    const int cnt = mcount[seg];
    const int j0 = (int)((long long)cnt * run / nrun), j1 = (int)((long long)cnt * (run + 1) / nrun);
    const int *mem = members + (size_t)seg * mstride;
    const unsigned nx_rcp = 0xFFFFFFFFu / (unsigned)nx + 1u;      // i / nx = (i * nx_rcp) >> 32, exact for i * nx < 2^32
    const int RY = RA_XS_THREADS / nx, S = RY * nx;               // rows, pixels per sweep
    const int ty = (int)__umulhi((unsigned)tid, nx_rcp), tx = tid - ty * nx;
    const bool live = tid < S;
    const int band = blockIdx.z * NPT;                            // first output sweep of this workgroup's row band
    float acc[NPT], pre[NPF];
#pragma unroll
    for (int k = 0; k < NPT; k++) acc[k] = 0.f;
    // wrap-around border of the padded copy: 4 nx + 4 elements, one per thread tid < 4 nx + 4 (padded (row, col) <- pixel)
    int bdst = -1, bsrc = 0;
    if (tid < 4 * nx + 4) {
        int pr, pc;
        if (tid < nx + 2) { pr = 0; pc = tid; }
        else if (tid < 2 * nx + 4) { pr = nx + 1; pc = tid - (nx + 2); }
        else if (tid < 3 * nx + 4) { pr = tid - (2 * nx + 4) + 1; pc = 0; }
        else { pr = tid - (3 * nx + 4) + 1; pc = nx + 1; }
        const int sr = pr == 0 ? nx - 1 : pr == nx + 1 ? 0 : pr - 1, sc = pc == 0 ? nx - 1 : pc == nx + 1 ? 0 : pc - 1;
        bdst = pr * pst + pc; bsrc = sr * nx + sc;
    }
    float preb = 0.f;
    const int xc = nx / 2, yc = nx / 2, mstart = 1 - nx % 2;
    // (tid and ty pass through an empty asm in every trip of the member loop: what derives from them per sweep -- pixel
    // indices, LDS addresses, row coordinates -- is then rebuilt with a few instructions instead of being hoisted out of the
    // loop into ~8 registers per sweep, more than the kernel has)
    auto fetch = [&](int j) {
        int t0 = tid;
        asm volatile("" : "+v"(t0));
        const float *src = particles + (size_t)mem[j] * npix;
#pragma unroll
        for (int k = 0; k < NPF; k++) pre[k] = src[min(t0 + k * S, npix - 1)];
        preb = src[bsrc];
    };
    if (j0 < j1) fetch(j0);
#pragma unroll 1
    for (int j = j0; j < j1; j++) {
        __syncthreads();                                  // the previous member's taps are read
        int t1 = tid, tyv = ty;
        asm volatile("" : "+v"(t1), "+v"(tyv));
        float *fill = img + (tyv + 1) * pst + tx + 1;
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (t1 < S && t1 + k * S < npix) fill[k * RY * pst] = pre[k];
        if (bdst >= 0) img[bdst] = preb;
        const ra_result r = res[mem[j]];
        const float2 cs = trig[mem[j]];
        __syncthreads();
        if (j + 1 < j1) fetch(j + 1);
        float delx = r.sx, dely = r.sy;
        while (delx >= (float)nx) delx -= nx;
        while (delx <= -(float)nx) delx += nx;
        while (dely >= (float)nx) dely -= nx;
        while (dely <= -(float)nx) dely += nx;
        const float shiftxc = xc + delx, shiftyc = yc + dely;
        const float cang = cs.x, sang = cs.y;
        // destination column tx; xform.mirror(x) reverses columns [1 - nx%2, nx): its source column is ix
        const int ix = (r.mirror && tx >= mstart) ? mstart + (nx - 1) - tx : tx;
        const float x = (float)ix - shiftxc;
        const float xcang = x * cang, xsang = x * sang;
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            const int iy = min(tyv + (band + k) * RY, nx - 1);     // rows past the image (last sweep, idle threads) shadow the last row
            const float y = (float)iy - shiftyc;
            const float ycang = y * cang + yc;
            const float ysang = -y * sang + xc;
            const float xold = xcang + ysang;
            const float yold = xsang + ycang;
            acc[k] += quadri_background_wrap(img, pst, nx, xold + 1.0f, yold + 1.0f, ix + 1, iy + 1);
        }
    }
    float *dst = partial + ((size_t)run * gridDim.x + seg) * npix;
#pragma unroll
    for (int k = 0; k < NPT; k++) { const int o = tid + (band + k) * S; if (live && o < npix) dst[o] = acc[k]; }
}

// K4d: the same for boxes whose image does not fit the LDS (transform_sum_kernel stops at ~140 pixels): workgroup = (segment, run,
// OUTPUT TILE of 64 x 64 pixels).  The taps of a tile's pixels lie in the rotated tile's bounding box in the source image, at most
// 63 (|cos| + |sin|) <= 89.1 pixels wide plus the neighbours of quadri and a margin for rounding: a box of 98 x 98 source pixels is brought in per member --
// global -> LDS without a stop in registers, periodic in both directions as quadri's own neighbour rule (ip1 / im1 wrap around), two
// boxes so that the next member's box travels while this one is interpolated -- and every thread adds its 8 pixels (column tx,
// rows ty + 8 k) to accumulators in registers (256 threads: 3.1 ms per 8192 x 256 x 256, 512: 1.9 ms, 1024: 2.6 ms).  A pixel whose source position lies outside the image takes the particle's own
// pixel (quadri_background's rule: the interpolation at the integer position (xnew, ynew) is that pixel, to the bit), read from
// global memory.  Arithmetic and association of the sums are transform_sum_kernel's: partial[run][segment][pixel], runs added in
// run order by class_sum_combine_kernel.
#define RA_XT_TS 64
#define RA_XT_BB 98
#define RA_XT_THREADS 512
__global__ __launch_bounds__(RA_XT_THREADS) void transform_sum_tile_kernel(int nx, const float *__restrict__ particles, int n, int index0,
                                                                           const ra_result *__restrict__ res, const float2 *__restrict__ trig,
                                                                           const int *__restrict__ members,
                                                                           const int *__restrict__ mcount, int mstride,
                                                                           float *__restrict__ partial)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) float box[];          // [2][RA_XT_BB * RA_XT_BB]
    constexpr int BB = RA_XT_BB, NB = BB * BB, NW = RA_XT_THREADS / 64, NK = RA_XT_TS / NW;
    const int seg = blockIdx.x, run = blockIdx.y, nrun = gridDim.y, tid = threadIdx.x, npix = nx * nx;
    const int ntx = (nx + RA_XT_TS - 1) / RA_XT_TS;
    const int tx0 = (blockIdx.z % ntx) * RA_XT_TS, ty0 = (blockIdx.z / ntx) * RA_XT_TS;
    const int cnt = mcount[seg];
    const int j0 = (int)((long long)cnt * run / nrun), j1 = (int)((long long)cnt * (run + 1) / nrun);
    const int *mem = members + (size_t)seg * mstride;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = tx0 + lane, ty = ty0 + wave;            // destination column; first destination row (then + NW k)
    const int xc = nx / 2, yc = nx / 2, mstart = 1 - nx % 2;
    float acc[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) acc[k] = 0.f;
    struct Pose { float shiftxc, shiftyc, cang, sang; int mirror, bx0, by0; };
    // the member's pose and the origin (1-based source coordinates of box element (0, 0)) of its tile's box
    auto pose_of = [&](int j) {
        Pose q;
        const ra_result r = res[mem[j]];
        const float2 cs = trig[mem[j]];
        float delx = r.sx, dely = r.sy;
        while (delx >= (float)nx) delx -= nx;
        while (delx <= -(float)nx) delx += nx;
        while (dely >= (float)nx) dely -= nx;
        while (dely <= -(float)nx) dely += nx;
        q.shiftxc = xc + delx; q.shiftyc = yc + dely; q.cang = cs.x; q.sang = cs.y; q.mirror = r.mirror;
        // source columns of the tile's destination columns (mirror reverses [mstart, nx)), rows ty0 .. ty0 + 63
        // (even nx: column 0 is not mirrored and its source lies a box away from its neighbours': it goes through global memory)
        const int c0 = (r.mirror && tx0 < mstart) ? mstart : tx0, c1 = min(tx0 + RA_XT_TS - 1, nx - 1);
        const int ia = r.mirror ? mstart + (nx - 1) - c1 : c0, ib = r.mirror ? mstart + (nx - 1) - c0 : c1;
        const float xa = (float)ia - q.shiftxc, xb = (float)ib - q.shiftxc;
        const float ya = (float)ty0 - q.shiftyc, yb = (float)min(ty0 + RA_XT_TS - 1, nx - 1) - q.shiftyc;
        float lox = 3.0e38f, loy = 3.0e38f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float x = (c & 1) ? xb : xa, y = (c & 2) ? yb : ya;
            lox = fminf(lox, x * q.cang - y * q.sang); loy = fminf(loy, x * q.sang + y * q.cang);
        }
        q.bx0 = (int)floorf(lox + xc + 1.0f) - 3; q.by0 = (int)floorf(loy + yc + 1.0f) - 3;
        return q;
    };
    // box element e = (row e / BB, column e % BB) <- source pixel ((by0 + row - 1) mod nx, (bx0 + col - 1) mod nx)
    auto stage = [&](int j, const Pose &q, float *dstbox) {
        const float *src = particles + (size_t)mem[j] * npix;
        int cb = (q.bx0 - 1) % nx, rb = (q.by0 - 1) % nx;
        if (cb < 0) cb += nx;
        if (rb < 0) rb += nx;
#pragma unroll 1
        for (int e0 = wave * 64; e0 < NB; e0 += RA_XT_THREADS) {
            const int e = min(e0 + lane, NB - 1), row = e / BB, col = e - row * BB;
            int sr = rb + row, sc = cb + col;
            if (sr >= nx) sr -= nx;
            if (sc >= nx) sc -= nx;
            if (sr >= nx) sr %= nx;              // boxes wider than the image (nx < 96): not the geometry this kernel is for, but exact
            if (sc >= nx) sc %= nx;
            if (e0 + lane < NB)
                __builtin_amdgcn_global_load_lds(src + sr * nx + sc, (__attribute__((address_space(3))) void *)(dstbox + e0), 4, 0, 0);
        }
    };
    Pose cur{}, nxt{};
    if (j0 < j1) { cur = pose_of(j0); stage(j0, cur, box); }
#pragma unroll 1
    for (int j = j0; j < j1; j++) {
        float *bx = box + ((j - j0) & 1) * NB;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // this member's box is complete; the other box is free
        if (j + 1 < j1) { nxt = pose_of(j + 1); stage(j + 1, nxt, box + ((j + 1 - j0) & 1) * NB); }
        if (tx < nx) {
            const int ix = (cur.mirror && tx >= mstart) ? mstart + (nx - 1) - tx : tx;
            const float x = (float)ix - cur.shiftxc;
            const float xcang = x * cur.cang, xsang = x * cur.sang;
            const float *P = bx - (cur.by0 * BB + cur.bx0);
            const float *own = particles + (size_t)mem[j] * npix + ix;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                const int iy = min(ty + NW * k, nx - 1);
                const float y = (float)iy - cur.shiftyc;
                const float ycang = y * cur.cang + yc;
                const float ysang = -y * cur.sang + xc;
                const float xold = xcang + ysang;
                const float yold = xsang + ycang;
                const float X = xold + 1.0f, Y = yold + 1.0f;
                float v;
                if (!((X >= 1.0f) && (X < (float)(nx + 1)) && (Y >= 1.0f) && (Y < (float)(nx + 1)))) {          // (NaN: background)
                    v = own[iy * nx];
                } else if (cur.mirror && tx < mstart) {
                    v = quadri_background_1b(particles + (size_t)mem[j] * npix, nx, nx, X, Y, ix + 1, iy + 1);
                } else {
                    const int i = (int)X, jj = (int)Y;
                    const float dx0 = X - i, dy0 = Y - jj;
                    const float *q = P + jj * BB + i;
                    const float f0 = q[0];
                    const float c1 = q[1] - f0;
                    const float c2 = (c1 - f0 + q[-1]) * 0.5f;
                    const float c3 = q[BB] - f0;
                    const float c4 = (c3 - f0 + q[-BB]) * 0.5f;
                    const float dxb = dx0 - 1, dyb = dy0 - 1;
                    const float c5 = q[BB + 1] - f0 - c1 - c3;
                    v = f0 + dx0 * (c1 + dxb * c2 + dy0 * c5) + dy0 * (c3 + dyb * c4);
                }
                acc[k] += v;
            }
        }
        cur = nxt;
    }
    float *dst = partial + ((size_t)run * gridDim.x + seg) * npix;
    if (tx < nx) {
#pragma unroll
        for (int k = 0; k < NK; k++) { const int iy = ty + NW * k; if (iy < nx) dst[iy * nx + tx] = acc[k]; }
    }
}

// (float)cos / sin of the angle in double, once per particle, exactly as transform_kernel forms them
__global__ void transform_trig_kernel(const ra_result *__restrict__ res, int n, float2 *__restrict__ trig)
{
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float ang = res[p].alpha * (float)M_PI / 180.0f;
    trig[p] = make_float2((float)cos((double)ang), (float)sin((double)ang));
}

// member lists of a whole batch: segment = (class, parity); 16 waves per segment, wave w compacts the particles of its
// contiguous 1/16 of the batch in particle order (ballot + prefix count), after a counting pass that gives every wave its
// offset.  counts: += members per class (even + odd), added once.
__global__ __launch_bounds__(1024) void class_members_wide_kernel(const ra_result *__restrict__ res, int n, int index0,
                                                                  int *__restrict__ members, int *__restrict__ mcount, int mstride,
                                                                  int *__restrict__ counts)
{
    __shared__ int wcnt[16];
    const int seg = blockIdx.x, cls = seg >> 1, par = seg & 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = ((n + 15) / 16 + 63) & ~63;
    const int lo = min(n, wave * per), hi = min(n, lo + per);
    int c = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool mine = i < hi && res[i].ref_id == cls && ((index0 + i) & 1) == par;
        c += __popcll(__ballot(mine));
    }
    if (lane == 0) wcnt[wave] = c;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) base += wcnt[w]; total += wcnt[w]; }
    int *mem = members + (size_t)seg * mstride;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool mine = i < hi && res[i].ref_id == cls && ((index0 + i) & 1) == par;
        const unsigned long long m = __ballot(mine);
        if (mine) mem[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        base += __popcll(m);
    }
    if (threadIdx.x == 0) {
        mcount[seg] = total;
        if (counts && total) atomicAdd(counts + cls, total);
    }
}

// K5: new references from the class sums: (even + odd) * (1/count), then
// normalize.mask(no_sigma=1) (test_mref_gpu_align.py:534-535, 563)
__global__ __launch_bounds__(256) void update_refs_kernel(int nx, const float *__restrict__ sums,
                                                          const int *__restrict__ counts, int min_count,
                                                          const float *__restrict__ mask, float *__restrict__ refs)
{
    __shared__ double sh[2][4];
    __shared__ int shn[4];
    const int r = blockIdx.x, tid = threadIdx.x, npix = nx * nx;
    const int cnt = counts[r];
    if (cnt < min_count) return;
    const float sc = (float)(1.0 / (double)(float)cnt);
    const float *ev = sums + (size_t)r * 2 * npix, *od = ev + npix;
    float *dst = refs + (size_t)r * npix;
    double s = 0, q = 0; int nm = 0;
    for (int i = tid; i < npix; i += blockDim.x) {
        float v = (ev[i] + od[i]) * sc;
        dst[i] = v;
        if (mask[i] > 0.5f) { s += v; q += v * (double)v; nm++; }
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); nm += __shfl_xor(nm, o); }
    if ((tid & 63) == 0) { sh[0][tid >> 6] = s; sh[1][tid >> 6] = q; shn[tid >> 6] = nm; }
    __syncthreads();
    s = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    q = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    nm = shn[0] + shn[1] + shn[2] + shn[3];
    const float mean = (float)s / nm;
    const float sigma = sqrtf((float)((q - s * s / nm) / (nm - 1)));
    for (int i = tid; i < npix; i += blockDim.x) dst[i] = (dst[i] - mean) / sigma;
}

// particle preprocessing: subtract the mean under the mask (normalize.mask no_sigma=0,
// test_mref_gpu_align.py:342)
__global__ __launch_bounds__(256) void normalize_particles_kernel(int nx, const float *__restrict__ mask,
                                                                  float *__restrict__ particles, int n)
{
    __shared__ double sh[4];
    __shared__ int shn[4];
    const int p = blockIdx.x, tid = threadIdx.x, npix = nx * nx;
    if (p >= n) return;
    float *img = particles + (size_t)p * npix;
    double s = 0; int nm = 0;
    for (int i = tid; i < npix; i += blockDim.x)
        if (mask[i] > 0.5f) { s += img[i]; nm++; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); nm += __shfl_xor(nm, o); }
    if ((tid & 63) == 0) { sh[tid >> 6] = s; shn[tid >> 6] = nm; }
    __syncthreads();
    s = sh[0] + sh[1] + sh[2] + sh[3];
    nm = shn[0] + shn[1] + shn[2] + shn[3];
    const float mean = nm ? (float)s / nm : 0.f;
    for (int i = tid; i < npix; i += blockDim.x) img[i] = (img[i] - mean) / 1.0f;
}

}  // namespace ralign
