// Particle-resident search kernel for gfx950 (MI355X): the whole search of one particle -- polar resampling,
// Normalize_ring, ring FFTs, the reference x particle contraction (Crosrng_ms), inverse FFTs and argmax -- runs inside
// one workgroup (16 waves); the particle spectra never leave the CU.  HBM sees the image once (32 KB at 90 x 90) and a
// few candidate records; the prepared references (the MFMA B operand) stream from L2.
//
// Structure ("ring streaming"): a pass handles one row of the search grid (<= 8 x-offsets = the 16 rows of a
// v_mfma_f32_16x16x4_f32 tile: offset x (Re, Im)).  The rings are cut into k-slots of 4 rings (the K = 4 of the MFMA),
// k-slots into steps of up to 16 wave-jobs.  Per step the waves sample the step's rings for every offset of the row and
// FFT them in an LDS stage (ring_job, shared with polar_fft_kernel), then multiply-accumulate them into CCF-spectrum
// accumulators that stay in REGISTERS for the whole pass:
//   * references 0..7: v_mfma_f32_16x16x4_f32, wave w owns the Fourier bins k = w (mod 16): 8 bins x 4 VGPRs;
//   * references 8..11 (pairs): v_mfma_f32_4x4x1_16b_f32, one instruction = 16 bins x (2 offsets x Re/Im) x
//     (2 references x Re/Im) for one ring; wave w owns bins 16 (w >> 1) .. +15 and two offset pairs: 8 VGPRs per pair
//     of references instead of the 32 a second, mostly empty 16-column tile would take.
// So 10 references need 40 accumulator VGPRs per wave, which leaves a 16-wave workgroup (128 VGPRs per wave) enough
// registers for the ring jobs.  Only a stage of 8 ... 20 rings x 8 offsets is ever in LDS.  After the last step the
// accumulators hold all bins of every (offset, reference) pair of the row; they are combined into Z_k = Q_k + i T_k,
// written to LDS (the stage space, idle by then) in rounds of <= 5 references, inverse-transformed there and reduced
// by the wavefront argmax of ralign_kernels.h.  Normalize_ring is linear, so it is applied after the contraction: the
// DC bin is corrected by avg * sum_r n_r C_r(0) and the peak values are scaled by 1/sigma.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq as called from test_mref_gpu_align.py:1043-1044 and
// sp_alignment.ali2d_single_iter (test_reffree_gpu_align.py:844-847); SURVEY.md Appendix A.3-A.9.
#pragma once

#include <algorithm>
#include <vector>

#include "ralign_geom.h"
#include "ralign_kernels.h"

namespace ralign {

constexpr int RF_WAVES = 16;
constexpr int RF_THREADS = RF_WAVES * 64;
constexpr int RF_MAXKS = 16;      // k-slots of 4 rings: nring <= 64
constexpr int RF_MAXSTEP = 8;
constexpr int RF_MAXREF = 12;     // 8 on the 16x16x4 tile + 2 pairs on 4x4x1 blocks

struct FusedGeom {
    int on;                        // plan valid for the current window
    int nxo;                       // live x-offsets per pass (row of the search grid), <= 8
    int npass;                     // rows of the search grid
    int nks, nstep;
    int nrp;                       // reference pairs beyond the first 8 references (0..2)
    int rz, nzr;                   // references per inverse-FFT round, rounds
    int r_floats;                  // LDS region shared by the stage and the CCF spectra
    int n_inst, n_job;
    int b_floats;                  // prepared-reference stream
    int step_ks0[RF_MAXSTEP + 1];  // k-slots [step_ks0[s], step_ks0[s+1]) form step s
    int step_os[RF_MAXSTEP];       // stride between offset slots in the stage (== 2 mod 32)
    int step_job0[RF_MAXSTEP + 1];
    int ks_base[RF_MAXKS];         // float offset of the k-slot inside an offset slot
    int ks_rs[RF_MAXKS];           // stride between the 4 rings of the k-slot (== 16 mod 32)
    int ks_nbin[RF_MAXKS];         // bins 0 .. nbin-1 exist in the k-slot
    int ks_boff[RF_MAXKS];         // float offset of the k-slot's B block for the 16x16x4 tile: [wave][chunk][lane][4]
    int ks_bw[RF_MAXKS];           // floats per wave inside that block (chunks of 256)
    int ks_bxoff[RF_MAXKS];        // float offset of the k-slot's B block for the 4x4x1 pairs: [pair][bin group][lane][4 rings]
    const int4 *jobs;              // {size code, first instance, count, 0}
    const int4 *inst;              // {offset slot | ring << 8, float offset inside the slot, qtab offset, radius}
    const float *instw;
    const int *bsrc;               // [b_floats] (entry << 5 | reference << 1 | imaginary part), -1 = 0
    const float *cdc_w;            // [nring] n_r * Applyws weight of bin 0 / maxrin: DC correction weights
};

// ------------------------------------------------------------------------------------------
// host: plan of the fused kernel for geometry g and the current search window
struct FusedPlanHost {
    FusedGeom f{};
    std::vector<int4> jobs, inst;
    std::vector<float> instw, cdc_w;
    std::vector<int> bsrc;
    size_t lds_bytes = 0;
};

inline int rf_align_up(int v, int a) { return (v + a - 1) / a * a; }

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
    return (e << 5) | (ref << 1) | part;
}

// qoff[log2 n] = offset of the ring length's quadrant table inside qtab; ringw = Normalize_ring weights
inline bool build_fused_plan(const Geometry &g, int nref, int pst, int n_qtab, const std::vector<int> &qoff,
                             const std::vector<float> &ringw, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.jobs.clear(); out.inst.clear(); out.instw.clear(); out.bsrc.clear(); out.cdc_w.clear();
    const int nx1 = 2 * g.nkx + 1;
    if (g.step != 1.0f || nx1 > 8 || nref > RF_MAXREF || g.nring > 4 * RF_MAXKS) return false;
    if (!(g.maxrin == 256 || g.maxrin == 128) || g.numr[2] < 8) return false;
    f.nxo = nx1; f.npass = 2 * g.nky + 1;
    f.nrp = nref > 8 ? (nref - 8 + 1) / 2 : 0;
    f.nks = (g.nring + 3) / 4;
    const int nbw_all = g.maxrin / 32;         // bins per wave at full length (bins 0 .. maxrin/2-1, Nyquist merged into bin 0)
    const int zstride = 2 * (g.maxrin + g.maxrin / 16) + 2;
    // k-slot geometry
    std::vector<int> nmax(f.nks), rs(f.nks), nbin(f.nks);
    for (int q = 0; q < f.nks; q++) {
        const int last = std::min(4 * q + 3, g.nring - 1);
        nmax[q] = g.numr[3 * last + 2];
        rs[q] = rf_align_up(nmax[q] + 2, 32) + 16;
        nbin[q] = (nmax[q] == g.maxrin) ? g.maxrin / 2 : nmax[q] / 2 + 1;
    }
    // ring-job shapes by ring length: code (ring_job variant) and instances per wave-job
    auto code_of = [](int n) { switch (n) { case 256: return 0; case 128: return 1; case 64: return 7; case 32: return 3; case 16: return 4; case 8: return 5; default: return -1; } };
    auto per_job = [](int n) { switch (n) { case 256: return 4; case 128: return 8; default: return 16; } };
    auto jobs_of = [&](int q0, int q1) {
        int cnt[9] = {0};
        for (int q = q0; q < q1; q++)
            for (int r = 4 * q; r < std::min(4 * q + 4, g.nring); r++) cnt[ilog2_floor(g.numr[3 * r + 2])] += f.nxo;
        int j = 0;
        for (int lg = 3; lg <= 8; lg++) j += (cnt[lg] + per_job(1 << lg) - 1) / per_job(1 << lg);
        return j;
    };
    // steps: consecutive k-slots while the ring jobs of the step fit one round of the 16 waves and the stage stays
    // below ~74 KB
    f.nstep = 0;
    int q = 0;
    const int stage_cap = 18944;               // floats
    int stage_floats = 0;
    while (q < f.nks) {
        if (f.nstep == RF_MAXSTEP) return false;
        int q1 = q + 1, os = 4 * rs[q] + 2;
        while (q1 < f.nks && jobs_of(q, q1 + 1) <= RF_WAVES && 8 * (os + 4 * rs[q1]) <= stage_cap) { os += 4 * rs[q1]; q1++; }
        f.step_ks0[f.nstep] = q; f.step_os[f.nstep] = os;
        int base = 0;
        for (int qq = q; qq < q1; qq++) { f.ks_base[qq] = base; f.ks_rs[qq] = rs[qq]; f.ks_nbin[qq] = nbin[qq]; base += 4 * rs[qq]; }
        stage_floats = std::max(stage_floats, 8 * os);
        f.nstep++;
        q = q1;
    }
    f.step_ks0[f.nstep] = f.nks;
    // inverse-FFT rounds: as many references per round as 64 lane groups and ~80 KB of spectra allow, balanced
    int rzmax = std::min(64 / f.nxo, 20480 / (f.nxo * zstride));
    rzmax = std::max(1, std::min(rzmax, 8));
    f.nzr = (nref + rzmax - 1) / rzmax;
    f.rz = (nref + f.nzr - 1) / f.nzr;
    f.r_floats = rf_align_up(std::max(stage_floats, f.nxo * f.rz * zstride), 64);

    // ring jobs per step: instances (offset slot, ring) by ring length (longest first)
    for (int s = 0; s < f.nstep; s++) {
        f.step_job0[s] = (int)out.jobs.size();
        for (int lg = 8; lg >= 3; lg--) {
            const int n = 1 << lg, code = code_of(n);
            std::vector<int4> cls; std::vector<float> clsw;
            for (int o = 0; o < f.nxo; o++)
                for (int qq = f.step_ks0[s]; qq < f.step_ks0[s + 1]; qq++)
                    for (int kk = 0; kk < 4; kk++) {
                        const int r = 4 * qq + kk;
                        if (r >= g.nring || g.numr[3 * r + 2] != n) continue;
                        cls.push_back(make_int4(o | (r << 8), f.ks_base[qq] + kk * f.ks_rs[qq], qoff[lg], g.numr[3 * r]));
                        clsw.push_back(ringw[r]);
                    }
            const int pj = per_job(n);
            for (size_t b = 0; b < cls.size(); b += pj) {
                const int cnt = (int)std::min<size_t>(pj, cls.size() - b);
                out.jobs.push_back(make_int4(code, (int)out.inst.size(), cnt, 0));
                for (int c = 0; c < cnt; c++) { out.inst.push_back(cls[b + c]); out.instw.push_back(clsw[b + c]); }
            }
        }
    }
    f.step_job0[f.nstep] = (int)out.jobs.size();
    f.n_inst = (int)out.inst.size(); f.n_job = (int)out.jobs.size();

    // B stream
    int boff = 0;
    for (int qq = 0; qq < f.nks; qq++) {
        const int nbw_max = std::min(nbw_all, (nbin[qq] + RF_WAVES - 1) / RF_WAVES);
        const int nch = (nbw_max + 3) / 4;
        f.ks_boff[qq] = boff; f.ks_bw[qq] = nch * 256;
        boff += RF_WAVES * nch * 256;
        f.ks_bxoff[qq] = boff;
        boff += f.nrp * ((nbin[qq] + 15) / 16) * 256;
    }
    f.b_floats = boff;
    out.bsrc.assign(boff, -1);
    for (int qq = 0; qq < f.nks; qq++) {
        // 16x16x4 tile: wave w, bin k = 16 i + w, lane = (ring kk, column = reference x Re/Im)
        for (int w = 0; w < RF_WAVES; w++)
            for (int i = 0; RF_WAVES * i + w < nbin[qq]; i++)
                for (int lane = 0; lane < 64; lane++) {
                    const int kk = lane >> 4, col = lane & 15;
                    out.bsrc[f.ks_boff[qq] + w * f.ks_bw[qq] + ((i >> 2) * 64 + lane) * 4 + (i & 3)] =
                        rf_bsrc_code(g, RF_WAVES * i + w, 4 * qq + kk, col >> 1, col & 1);
                }
        // 4x4x1 pairs: bin group m (bins 16 m + b), lane = (block b, column j = reference in pair x Re/Im), 4 rings
        const int nm = (nbin[qq] + 15) / 16;
        for (int rp = 0; rp < f.nrp; rp++)
            for (int m = 0; m < nm; m++)
                for (int lane = 0; lane < 64; lane++)
                    for (int kk = 0; kk < 4; kk++) {
                        const int k = 16 * m + (lane >> 2), j = lane & 3;
                        if (k >= nbin[qq]) continue;
                        out.bsrc[f.ks_bxoff[qq] + ((rp * nm + m) * 64 + lane) * 4 + kk] =
                            rf_bsrc_code(g, k, 4 * qq + kk, 8 + 2 * rp + (j >> 1), j & 1);
                    }
    }
    out.cdc_w.resize(g.nring);
    for (int r = 0; r < g.nring; r++) out.cdc_w[r] = (float)g.numr[3 * r + 2] * g.wr[r] / (float)g.maxrin;
    size_t fl = ((size_t)(pst * pst + 3) & ~(size_t)3) + f.r_floats + 2 * g.maxrin + 2 * n_qtab + 2 + 4 * f.n_inst + 4 * f.n_job + 4 +
                f.n_inst + 32 + 16 * g.nring + 2 * g.maxrin + 8 * RF_MAXREF * (sizeof(CandT) / 4) + 64;
    out.lds_bytes = fl * sizeof(float);
    f.on = out.lds_bytes <= 160 * 1024;
    return f.on != 0;
}

// ------------------------------------------------------------------------------------------
// device

// prepared references -> B stream of the fused kernel (Applyws weights and 1/maxrin folded in) and the per-reference
// DC weights Cdc[ref] = sum_r n_r * B_r(bin 0, Re)
__global__ void pack_refs_fused_kernel(DevGeom g, FusedGeom f, const float *__restrict__ refspec, int nref,
                                       float *__restrict__ Bf, float *__restrict__ cdc)
{
    const float inv = 1.0f / (float)g.maxrin;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < f.b_floats; idx += gridDim.x * blockDim.x) {
        const int code = f.bsrc[idx];
        float v = 0.f;
        if (code >= 0) {
            const int e = code >> 5, ref = (code >> 1) & 15;
            if (ref < nref) v = refspec[(size_t)ref * g.lring + g.ent_src[e] + (code & 1)] * g.ent_wgt[e] * inv;
        }
        Bf[idx] = v;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < nref) {
        const int ref = threadIdx.x;
        float s = 0.f;
        for (int r = 0; r < g.nring; r++) s += refspec[(size_t)ref * g.lring + g.ringinfo[r].x] * f.cdc_w[r];
        cdc[ref] = s;
    }
}

__device__ __forceinline__ float rf_f4(const float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// Z_k = Q_k + i T_k and Z_{N-k} = conj Q_k + i conj T_k from the four products a = c1 d1, b = c1 d2, c = c2 d1,
// d = c2 d2 summed over rings (Util::Crosrng_ms: Q = (a + d) + i (c - b), T = (a - d) - i (b + c)); bin 0 carries the
// DC term in a (minus the Normalize_ring mean) and the Nyquist term of the full-length rings in d
template <int N>
__device__ __forceinline__ void rf_store_z(float *Z, int zslot, int k, float ca, float cb, float cc, float cd, float dcw)
{
    typedef ZLayout<N> ZL;
    if (k == 0) {
        const float a0 = ca - dcw;
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, 0)) = make_float2(a0, a0);
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, N / 2)) = make_float2(cd, cd);
    } else {
        const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, k)) = make_float2(apd + bpc, cmb + amd);
        *reinterpret_cast<float2 *>(Z + ZL::addr(zslot, N - k)) = make_float2(apd - bpc, amd - cmb);
    }
}

template <int N, int NRP>
__global__ __launch_bounds__(RF_THREADS) void search_fused_kernel(DevGeom g, FusedGeom f, const float *__restrict__ particles,
                                                                  const float *__restrict__ state, int n,
                                                                  const float *__restrict__ Bf,
                                                                  const float *__restrict__ cdc, int nref,
                                                                  CandT *__restrict__ cand)
{
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    constexpr int NBW = N / 32;                 // bins per wave of the 16x16x4 tile (k = 16 i + wave)
    constexpr int NCH = (NBW + 3) / 4;          // float4 B registers per k-slot
    constexpr int NRPA = NRP > 0 ? NRP : 1;
    extern __shared__ __align__(16) float lds[];
    const int npad = g.pst * g.pst;
    float *img = lds;
    float *R = lds + ((npad + 3) & ~3);                                   // [r_floats] stage | CCF spectra
    float2 *tw_s = reinterpret_cast<float2 *>(R + f.r_floats);            // [maxrin]
    float2 *qt_s = tw_s + g.maxrin;                                        // [n_qtab]
    int4 *inst_s = reinterpret_cast<int4 *>(qt_s + g.n_qtab + (g.n_qtab & 1));
    int4 *jobs_s = inst_s + f.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + f.n_job);          // [n_inst]
    float *ctr = instw_s + ((f.n_inst + 3) & ~3);                          // [16] sampling centres of the 8 offset slots
    float *nrm = ctr + 16;                                                 // [8] avg, [8] 1/sigma
    float *part = nrm + 16;                                                // [8][nring][2] Normalize_ring partial sums
    float2 *tws = reinterpret_cast<float2 *>(part + 16 * g.nring);         // [R1*R2] inverse-FFT twiddles
    CandT *pc = reinterpret_cast<CandT *>(tws + R1 * R2);                  // [8][nref] records of the pass
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (p >= n) return;

    // ---- particle image into the zero-bordered LDS array, tables
    const float *src = particles + (size_t)p * g.nx * g.nx;
    for (int row = wave; row < g.pst; row += RF_WAVES) {
        const int y = row - g.bd;
        const bool yin = y >= 0 && y < g.nx;
        for (int c = lane; c < g.pst; c += 64) {
            const int x = c - g.bd;
            img[row * g.pst + c] = (yin && x >= 0 && x < g.nx) ? src[y * g.nx + x] : 0.f;
        }
    }
    const float *imgb = img + (g.bd - 1) * g.pst + (g.bd - 1);
    for (int i = tid; i < g.maxrin; i += RF_THREADS) tw_s[i] = g.tw[i];
    for (int i = tid; i < g.n_qtab; i += RF_THREADS) qt_s[i] = g.qtab[i];
    for (int i = tid; i < f.n_inst; i += RF_THREADS) { inst_s[i] = f.inst[i]; instw_s[i] = f.instw[i]; }
    for (int i = tid; i < f.n_job; i += RF_THREADS) jobs_s[i] = f.jobs[i];
    for (int i = tid; i < R1 * R2; i += RF_THREADS) {
        const float2 t = g.tw[((i / R2) * (i % R2) * (g.maxrin / N)) & (g.maxrin - 1)];
        tws[i] = make_float2(t.x, -t.y);
    }
    for (int i = tid; i < f.r_floats; i += RF_THREADS) R[i] = 0.f;        // stage slack must hold finite values
    const Window w = particle_window(g, state[2 * p], state[2 * p + 1]);
    const float cxf = (float)g.cnx + w.sxi, cyf = (float)g.cnx + w.syi;
    const float2 *twl = tws + (lane & 15);

    // lane roles.  16x16x4: A row = lane & 15 = (offset slot, Re/Im), k-slot ring = lane >> 4; C column = lane & 15 =
    // (reference, Re/Im); after the 2x2 exchange the even lane keeps offset 2*(lane>>4), the odd lane the next one.
    // 4x4x1: block = lane >> 2 = bin in the group, A row = lane & 3 = (offset in pair, Re/Im), C column = lane & 3.
    const int kk = lane >> 4, arow = lane & 15, odd = lane & 1;
    const int o_epi = 2 * kk + odd, r8 = (lane & 15) >> 1;
    const int xm = wave >> 1, xb = lane >> 2, xj = lane & 3;

    const int nx1 = 2 * g.nkx + 1;
    for (int pass = 0; pass < f.npass; pass++) {
        __syncthreads();                       // previous pass has left the stage / spectra region and `ctr`
        if (tid < 8) {
            const int si = pass * nx1 + min(tid, f.nxo - 1);
            ctr[2 * tid] = cxf + g.shift_x[si];
            ctr[2 * tid + 1] = cyf + g.shift_y[si];
        }
        f32x4 acc[NBW];
        f32x4 accx[NRPA][2];
#pragma unroll
        for (int i = 0; i < NBW; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rp = 0; rp < NRPA; rp++) { accx[rp][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[rp][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        __syncthreads();
        for (int st = 0; st < f.nstep; st++) {
            const int os = f.step_os[st];
            // ---- ring jobs of the step: sampling + Normalize_ring statistics + ring FFT into the stage
            if (!RA_DBG(g, 16)) {
                const int j1 = f.step_job0[st + 1];
#pragma unroll 1
                for (int job = f.step_job0[st] + wave; job < j1; job += RF_WAVES) {
                    const int4 jd = jobs_s[job];
                    switch (__builtin_amdgcn_readfirstlane(jd.x)) {
                    case 0: ring_job<8, 16, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    case 1: ring_job<8, 8, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    case 7: ring_job<8, 4, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    case 3: ring_job<4, 4, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    case 4: ring_job<2, 4, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    default: ring_job<1, 4, true>(g, imgb, R, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
                    }
                }
            }
            // ---- contraction of the step: accumulate its k-slots into the bins this wave owns.  The B operands of the
            // next k-slot are requested while the current one is multiplied; those of the first k-slot before the barrier.
            float4 breg[2][NCH], bx[2][NRPA];
            auto load_b = [&](int q, float4 (&b1)[NCH], float4 (&b2)[NRPA]) {
                const float *bp = Bf + f.ks_boff[q] + wave * f.ks_bw[q] + lane * 4;
                const int nch = f.ks_bw[q] >> 8, nbq = f.ks_nbin[q];
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (c < nch) b1[c] = *reinterpret_cast<const float4 *>(bp + c * 256);
                if (NRP > 0 && 16 * xm < nbq) {
                    const int nm = (nbq + 15) >> 4;
#pragma unroll
                    for (int rp = 0; rp < NRP; rp++)
                        b2[rp] = *reinterpret_cast<const float4 *>(Bf + f.ks_bxoff[q] + ((rp * nm + xm) * 64 + lane) * 4);
                }
            };
            auto mul_q = [&](int q, const float4 (&b1)[NCH], const float4 (&b2)[NRPA]) {
                const int nb = f.ks_nbin[q], rs = f.ks_rs[q], kb = f.ks_base[q];
                const float *ab = R + (arow >> 1) * os + kb + kk * rs + (arow & 1) + 2 * wave;
#pragma unroll
                for (int i = 0; i < NBW; i++)
                    if (RF_WAVES * i + wave < nb)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ab[2 * RF_WAVES * i], rf_f4(b1[i >> 2], i & 3), acc[i], 0, 0, 0);
                if (NRP > 0 && 16 * xm < nb) {
#pragma unroll
                    for (int opi = 0; opi < 2; opi++) {
                        const int op = 2 * (wave & 1) + opi;
                        const float *ax = R + (2 * op + (xj >> 1)) * os + kb + (xj & 1) + 2 * (16 * xm + xb);
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const float a = ax[k4 * rs];
#pragma unroll
                            for (int rp = 0; rp < NRP; rp++)
                                accx[rp][opi] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, rf_f4(b2[rp], k4), accx[rp][opi], 0, 0, 0);
                        }
                    }
                }
            };
            const int q0 = f.step_ks0[st], q1 = f.step_ks0[st + 1];
            if (!RA_DBG(g, 2)) load_b(q0, breg[0], bx[0]);
            __syncthreads();
            if (!RA_DBG(g, 2)) {
                for (int q = q0; q < q1; q += 2) {
                    if (q + 1 < q1) load_b(q + 1, breg[1], bx[1]);
                    mul_q(q, breg[0], bx[0]);
                    if (q + 1 < q1) {
                        if (q + 2 < q1) load_b(q + 2, breg[0], bx[0]);
                        mul_q(q + 1, breg[1], bx[1]);
                    }
                }
            }
            __syncthreads();
        }
        // ---- Normalize_ring statistics of every offset slot (ring order, fixed butterfly: reproducible)
        if (wave < f.nxo) {
            float a = 0.f, q2 = 0.f;
            for (int i = lane; i < g.nring; i += 64) { a += part[2 * (wave * g.nring + i)]; q2 += part[2 * (wave * g.nring + i) + 1]; }
            a = wave_sum(a); q2 = wave_sum(q2);
            if (lane == 0) {
                float avg = 0.f, rsg = 1.f;
                if (g.mode == RA_MODE_MREF) {
                    const float nn = g.nn_weight;
                    avg = a / nn;
                    rsg = 1.0f / sqrtf((q2 - a * a / nn) / nn);
                }
                nrm[wave] = avg; nrm[8 + wave] = rsg;
            }
        }
        __syncthreads();
        // ---- CCF spectra -> LDS, inverse FFT, argmax: rounds of f.rz references
        if (RA_DBG(g, ~0) && tid < f.nxo * nref) {   // profiling builds that skip a phase still emit in-range records
            pc[tid].val = 0.f; pc[tid].jtot = 1; pc[tid].refmir = tid % nref;
            for (int k7 = 0; k7 < 7; k7++) pc[tid].t7[k7] = 0.f;
        }
        for (int zr = 0; zr < f.nzr && !RA_DBG(g, 4); zr++) {
            const int ref_lo = zr * f.rz, nrz = min(f.rz, nref - ref_lo);
            {   // 16x16x4 accumulators: this lane holds (offset o_epi, reference r8) of the bins 16 i + wave
                const int rr = r8 - ref_lo;
                const bool mine = o_epi < f.nxo && r8 < nref && rr >= 0 && rr < nrz;
                const int zslot = o_epi * f.rz + rr;
                const float dcw = mine ? cdc[r8] * nrm[o_epi] : 0.f;
                if (ref_lo < 8) {
#pragma unroll
                    for (int i = 0; i < NBW; i++) {
                        const f32x4 c4 = acc[i];
                        const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                        const float r0 = swap_lane_pair(s0), r1 = swap_lane_pair(s1);
                        const float ca = odd ? r0 : c4[0], cb = odd ? r1 : c4[1];
                        const float cc = odd ? c4[2] : r0, cd = odd ? c4[3] : r1;
                        if (mine) rf_store_z<N>(R, zslot, RF_WAVES * i + wave, ca, cb, cc, cd, dcw);
                    }
                }
            }
            if (NRP > 0 && ref_lo + nrz > 8 && 16 * xm < N / 2) {
                // 4x4x1 accumulators: (offset 2 op + odd, reference 8 + 2 rp + (column >> 1)) of bin 16 xm + xb
#pragma unroll
                for (int rp = 0; rp < NRP; rp++) {
                    const int ref = 8 + 2 * rp + (xj >> 1), rr = ref - ref_lo;
#pragma unroll
                    for (int opi = 0; opi < 2; opi++) {
                        const int o = 2 * (2 * (wave & 1) + opi) + odd;
                        const f32x4 c4 = accx[rp][opi];
                        const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                        const float r0 = swap_lane_pair(s0), r1 = swap_lane_pair(s1);
                        const float ca = odd ? r0 : c4[0], cb = odd ? r1 : c4[1];
                        const float cc = odd ? c4[2] : r0, cd = odd ? c4[3] : r1;
                        if (o < f.nxo && ref < nref && rr >= 0 && rr < nrz)
                            rf_store_z<N>(R, o * f.rz + rr, 16 * xm + xb, ca, cb, cc, cd, cdc[ref] * nrm[o]);
                    }
                }
            }
            __syncthreads();
            {
                // lane groups (sub, sub ^ 1) of a half-wave take spectrum slots zs and zs + 16: their LDS images are 32 banks apart
                const int j = lane & 15, sub = lane >> 4, u = 2 * wave + (sub >> 1);
                const int zs = (u & 15) + 16 * (sub & 1) + 32 * (u >> 4);
                const int rr = zs % f.rz, o = zs / f.rz;
                if (zs < f.nxo * f.rz && rr < nrz && !RA_DBG(g, 1))      // uniform over the 16-lane group
                    ifft_argmax<N, 1, 0>(R, pc + (o * nref + ref_lo + rr) - zs, twl, zs, zs, j, ref_lo + rr);
            }
            __syncthreads();
        }
        // best reference per offset of the row (ascending reference, ">=": later wins), scaled by 1/sigma
        if (tid < f.nxo * (int)(sizeof(CandT) / 4)) {
            constexpr int W = sizeof(CandT) / 4;
            const int o = tid / W, wd = tid - o * W;
            float bv = pc[o * nref].val; int br = 0;
            for (int q3 = 1; q3 < nref; q3++) {
                const float v = pc[o * nref + q3].val;
                if (v >= bv) { bv = v; br = q3; }
            }
            const int sft = pass * nx1 + o;
            int word = reinterpret_cast<const int *>(pc + o * nref + br)[wd];
            if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * nrm[8 + o]);     // val, t7[]
            reinterpret_cast<int *>(cand + (size_t)p * g.nshift_pad + sft)[wd] = word;
        }
    }
}

}  // namespace ralign
