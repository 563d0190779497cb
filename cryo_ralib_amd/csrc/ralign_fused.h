// Particle-resident search kernel for gfx950 (MI355X): the whole search of one particle -- polar resampling,
// Normalize_ring, ring FFTs, the reference x particle contraction (Crosrng_ms), inverse FFTs and argmax -- runs inside
// one workgroup; the particle spectra never leave the CU.  HBM sees the image once (32 KB at 90 x 90) and a few
// candidate records; the prepared references (the MFMA B operand) stream from L2.
//
// Structure ("ring streaming"): a pass handles one row of the search grid (<= 8 x-offsets = the 16 rows of a
// v_mfma_f32_16x16x4_f32 tile: offset x (Re, Im)).  The rings are cut into k-slots of 4 rings (the K = 4 of the MFMA),
// k-slots into steps.  Per step the waves sample the step's rings for every offset of the row, FFT them in an LDS
// stage buffer (ring_job, shared with polar_fft_kernel) and then multiply-accumulate them into CCF-spectrum
// accumulators that stay in REGISTERS for the whole pass: wave w owns the Fourier bins k = w (mod 8), 16 bins x
// up to 2 reference tiles x 4 VGPRs = 128 VGPRs.  Only a stage of 4 ... 12 rings x 8 offsets is ever in LDS
// (double buffered, so sampling of step s+1 runs beside the MFMAs of step s).  After the last step the accumulators
// hold all bins of every (offset, reference) pair of the row; they are combined into Z_k = Q_k + i T_k, written to LDS
// (the stage space, idle by then) in rounds of <= 4 references, inverse-transformed there and reduced by the
// wavefront argmax of ralign_kernels.h.  Normalize_ring is linear, so it is applied after the contraction: the DC bin
// is corrected by avg * sum_r n_r C_r(0) and the peak values are scaled by 1/sigma.
//
// Reference call sites restated: Util.multiref_polar_ali_2d / ormq as called from test_mref_gpu_align.py:1043-1044 and
// sp_alignment.ali2d_single_iter (test_reffree_gpu_align.py:844-847); SURVEY.md Appendix A.3-A.9.
#pragma once

#include <algorithm>
#include <vector>

#include "ralign_geom.h"
#include "ralign_kernels.h"

namespace ralign {

constexpr int RF_WAVES = 8;
constexpr int RF_THREADS = RF_WAVES * 64;
constexpr int RF_MAXKS = 16;      // k-slots of 4 rings: nring <= 64
constexpr int RF_MAXSTEP = 12;
constexpr int RF_ZREFS = 4;       // references per inverse-FFT round

struct FusedGeom {
    int on;                        // plan valid for the current window
    int nxo;                       // live x-offsets per pass (row of the search grid), <= 8
    int npass;                     // rows of the search grid
    int nks, nstep;
    int nt;                        // reference tiles of 8 (1 | 2)
    int nzr;                       // inverse-FFT rounds = ceil(nref / 4)
    int stage_floats;              // one stage buffer
    int r_floats;                  // LDS region shared by the two stage buffers and the CCF spectra
    int n_inst, n_job;
    int b_floats;                  // prepared-reference stream
    int step_ks0[RF_MAXSTEP + 1];  // k-slots [step_ks0[s], step_ks0[s+1]) form step s
    int step_os[RF_MAXSTEP];       // stride between offset slots in the stage (== 2 mod 32)
    int step_job0[RF_MAXSTEP + 1];
    int ks_base[RF_MAXKS];         // float offset of the k-slot inside an offset slot
    int ks_rs[RF_MAXKS];           // stride between the 4 rings of the k-slot (== 16 mod 32)
    int ks_nbin[RF_MAXKS];         // bins 0 .. nbin-1 exist in the k-slot
    int ks_boff[RF_MAXKS];         // float offset of the k-slot's B block
    int ks_bw[RF_MAXKS];           // floats per wave inside that block (chunks of 256)
    const int4 *jobs;              // {size code, first instance, count, 0}
    const int4 *inst;              // {offset slot | ring << 8, float offset inside the slot, qtab offset, radius}
    const float *instw;
    const int *bsrc;               // [b_floats] (entry << 5 | reference slot << 1 | imaginary part), -1 = 0
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

// qoff[log2 n] = offset of the ring length's quadrant table inside qtab; ringw = Normalize_ring weights
inline bool build_fused_plan(const Geometry &g, int nref, int pst, int n_qtab, const std::vector<int> &qoff,
                             const std::vector<float> &ringw, FusedPlanHost &out)
{
    FusedGeom &f = out.f;
    f = FusedGeom{};
    out.jobs.clear(); out.inst.clear(); out.instw.clear(); out.bsrc.clear(); out.cdc_w.clear();
    const int nx1 = 2 * g.nkx + 1;
    if (g.step != 1.0f || nx1 > 8 || nref > 16 || g.nring > 4 * RF_MAXKS) return false;
    if (!(g.maxrin == 256 || g.maxrin == 128 || g.maxrin == 64 || g.maxrin == 32) || g.numr[2] < 8) return false;
    f.nxo = nx1; f.npass = 2 * g.nky + 1;
    f.nt = (nref + 7) / 8;
    f.nzr = (nref + RF_ZREFS - 1) / RF_ZREFS;
    f.nks = (g.nring + 3) / 4;
    const int nbw_all = g.maxrin / 16;         // bins per wave at full length (bins 0 .. maxrin/2-1, Nyquist merged into bin 0)
    // k-slot geometry
    std::vector<int> nmax(f.nks), rs(f.nks), nbin(f.nks);
    for (int q = 0; q < f.nks; q++) {
        const int last = std::min(4 * q + 3, g.nring - 1);
        nmax[q] = g.numr[3 * last + 2];
        rs[q] = rf_align_up(nmax[q] + 2, 32) + 16;
        nbin[q] = (nmax[q] == g.maxrin) ? g.maxrin / 2 : nmax[q] / 2 + 1;
    }
    // steps: consecutive k-slots while the ring jobs of the step fill at most ~one round of the 8 waves and the
    // stage stays small
    auto jobs_of = [&](int q0, int q1) {
        int cnt[9] = {0};
        for (int q = q0; q < q1; q++)
            for (int r = 4 * q; r < std::min(4 * q + 4, g.nring); r++) cnt[ilog2_floor(g.numr[3 * r + 2])] += f.nxo;
        const int per[9] = {0, 0, 0, 16, 16, 16, 8, 8, 4};      // instances per wave-job by log2(ring length)
        int j = 0;
        for (int lg = 3; lg <= 8; lg++) j += (cnt[lg] + per[lg] - 1) / per[lg];
        return j;
    };
    f.nstep = 0;
    int q = 0;
    const int stage_cap = 10 * 1024;           // floats
    while (q < f.nks) {
        if (f.nstep == RF_MAXSTEP) return false;
        int q1 = q + 1, os = 4 * rs[q] + 2;
        while (q1 < f.nks && jobs_of(q, q1 + 1) <= RF_WAVES && 8 * (os + 4 * rs[q1]) <= stage_cap) { os += 4 * rs[q1]; q1++; }
        f.step_ks0[f.nstep] = q; f.step_os[f.nstep] = os;
        int base = 0;
        for (int qq = q; qq < q1; qq++) { f.ks_base[qq] = base; f.ks_rs[qq] = rs[qq]; f.ks_nbin[qq] = nbin[qq]; base += 4 * rs[qq]; }
        f.nstep++;
        q = q1;
    }
    f.step_ks0[f.nstep] = f.nks;
    f.stage_floats = 0;
    for (int s = 0; s < f.nstep; s++) f.stage_floats = std::max(f.stage_floats, 8 * f.step_os[s]);
    f.stage_floats = rf_align_up(f.stage_floats, 64);
    const int zfloats = f.nxo * RF_ZREFS * (2 * (g.maxrin + g.maxrin / 16) + 2);
    f.r_floats = rf_align_up(std::max(2 * f.stage_floats, zfloats), 64);

    // ring jobs per step: instances (offset slot, ring) by ring length (longest first), 64 / LR instances per wave-job
    auto code_of = [](int n) { switch (n) { case 256: return 0; case 128: return 1; case 64: return 2; case 32: return 3; case 16: return 4; case 8: return 5; default: return -1; } };
    const int lanes_of[6] = {16, 8, 8, 4, 4, 4};
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
            const int per_job = 64 / lanes_of[code];
            for (size_t b = 0; b < cls.size(); b += per_job) {
                const int cnt = (int)std::min<size_t>(per_job, cls.size() - b);
                out.jobs.push_back(make_int4(code, (int)out.inst.size(), cnt, 0));
                for (int c = 0; c < cnt; c++) { out.inst.push_back(cls[b + c]); out.instw.push_back(clsw[b + c]); }
            }
        }
    }
    f.step_job0[f.nstep] = (int)out.jobs.size();
    f.n_inst = (int)out.inst.size(); f.n_job = (int)out.jobs.size();

    // B stream: per k-slot, per wave, chunks of 4 consecutive MFMAs (m = i * nt + t), [chunk][lane][4]
    int boff = 0;
    for (int qq = 0; qq < f.nks; qq++) {
        const int nbw_max = std::min(nbw_all, (nbin[qq] + RF_WAVES - 1) / RF_WAVES);
        const int nch = (nbw_max * f.nt + 3) / 4;
        f.ks_boff[qq] = boff; f.ks_bw[qq] = nch * 256;
        boff += RF_WAVES * nch * 256;
    }
    f.b_floats = boff;
    out.bsrc.assign(boff, -1);
    for (int qq = 0; qq < f.nks; qq++)
        for (int w = 0; w < RF_WAVES; w++)
            for (int i = 0; RF_WAVES * i + w < nbin[qq]; i++)
                for (int t = 0; t < f.nt; t++) {
                    const int m = i * f.nt + t, k = RF_WAVES * i + w;
                    for (int lane = 0; lane < 64; lane++) {
                        const int kk = lane >> 4, col = lane & 15, r = 4 * qq + kk;
                        if (r >= g.nring) continue;
                        const int n = g.numr[3 * r + 2];
                        int ks = k, part = col & 1;
                        if (k == 0 && part == 1) {             // Nyquist of a full-length ring rides in the imaginary slot of bin 0
                            if (n != g.maxrin) continue;
                            ks = n / 2; part = 0;
                        } else if (k > n / 2 || (k == n / 2 && n == g.maxrin)) continue;
                        else if (k == n / 2 && part == 1) continue;   // Nyquist of a shorter ring is real
                        const int e = g.bin_off[ks] + (r - g.bin_first[ks]);
                        const int refslot = t * 8 + (col >> 1);
                        out.bsrc[f.ks_boff[qq] + w * f.ks_bw[qq] + ((m >> 2) * 64 + lane) * 4 + (m & 3)] = (e << 5) | (refslot << 1) | part;
                    }
                }
    out.cdc_w.resize(g.nring);
    for (int r = 0; r < g.nring; r++) out.cdc_w[r] = (float)g.numr[3 * r + 2] * g.wr[r] / (float)g.maxrin;
    const int R1R2 = g.maxrin;
    size_t fl = ((size_t)(pst * pst + 3) & ~(size_t)3) + f.r_floats + 2 * g.maxrin + 2 * n_qtab + 2 + 4 * f.n_inst + 4 * f.n_job + 4 +
                f.n_inst + 32 + 16 * g.nring + 2 * R1R2 + 32 * (sizeof(CandT) / 4) + 64;
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

template <int NT> __device__ __forceinline__ float rf_bsel(const float4 (&b)[4 * NT], int m)
{
    const float4 v = b[m >> 2];
    switch (m & 3) { case 0: return v.x; case 1: return v.y; case 2: return v.z; default: return v.w; }
}

template <int N, int NT>
__global__ __launch_bounds__(RF_THREADS, 2) void search_fused_kernel(DevGeom g, FusedGeom f, const float *__restrict__ particles,
                                                                     const float *__restrict__ state, int n,
                                                                     const float *__restrict__ Bf,
                                                                     const float *__restrict__ cdc, int nref,
                                                                     CandT *__restrict__ cand)
{
    typedef ZLayout<N> ZL;
    constexpr int R1 = IfftPlan<N>::R1, R2 = IfftPlan<N>::R2;
    constexpr int NBW = N / 16;                 // bins per wave (k = 8 i + wave, i < NBW)
    constexpr int NCH = (NBW * NT + 3) / 4;     // float4 B registers per k-slot
    extern __shared__ __align__(16) float lds[];
    const int npad = g.pst * g.pst;
    float *img = lds;
    float *R = lds + ((npad + 3) & ~3);                                   // [r_floats] stage buffers | CCF spectra
    float2 *tw_s = reinterpret_cast<float2 *>(R + f.r_floats);            // [maxrin]
    float2 *qt_s = tw_s + g.maxrin;                                        // [n_qtab]
    int4 *inst_s = reinterpret_cast<int4 *>(qt_s + g.n_qtab + (g.n_qtab & 1));
    int4 *jobs_s = inst_s + f.n_inst;
    float *instw_s = reinterpret_cast<float *>(jobs_s + f.n_job);          // [n_inst]
    float *ctr = instw_s + ((f.n_inst + 3) & ~3);                          // [16] sampling centres of the 8 offset slots
    float *nrm = ctr + 16;                                                 // [8] avg, [8] 1/sigma
    float *part = nrm + 16;                                                // [8][nring][2] Normalize_ring partial sums
    float2 *tws = reinterpret_cast<float2 *>(part + 16 * g.nring);         // [R1*R2] inverse-FFT twiddles
    CandT *pc = reinterpret_cast<CandT *>(tws + R1 * R2);                  // [32]
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

    // MFMA lane roles: A row = lane & 15 = (offset slot, Re/Im), k-slot ring = lane >> 4; C column = lane & 15 =
    // (reference in tile, Re/Im).  After the 2x2 exchange the even lane keeps offset 2*(lane>>4), the odd lane the next.
    const int kk = lane >> 4, arow = lane & 15, odd = lane & 1;
    const int o_epi = 2 * kk + odd, r8 = (lane & 15) >> 1;

    auto run_jobs = [&](int st, float *stg) {
        const int j1 = f.step_job0[st + 1], os = f.step_os[st];
#pragma unroll 1
        for (int job = f.step_job0[st] + wave; job < j1; job += RF_WAVES) {
            const int4 jd = jobs_s[job];
            switch (__builtin_amdgcn_readfirstlane(jd.x)) {
            case 0: ring_job<8, 16, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            case 1: ring_job<8, 8, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            case 2: ring_job<4, 8, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            case 3: ring_job<4, 4, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            case 4: ring_job<2, 4, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            default: ring_job<1, 4, true>(g, imgb, stg, tw_s, qt_s, ctr, part, inst_s, instw_s, jd.y, jd.z, jd.w, os); break;
            }
        }
    };

    const int nx1 = 2 * g.nkx + 1;
    for (int pass = 0; pass < f.npass; pass++) {
        __syncthreads();                       // previous pass has left the stage / spectra region and `ctr`
        if (tid < 8) {
            const int si = pass * nx1 + min(tid, f.nxo - 1);
            ctr[2 * tid] = cxf + g.shift_x[si];
            ctr[2 * tid + 1] = cyf + g.shift_y[si];
        }
        f32x4 acc[NBW][NT];
#pragma unroll
        for (int i = 0; i < NBW; i++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        run_jobs(0, R);
        __syncthreads();
        for (int st = 0; st < f.nstep; st++) {
            float *cur = R + (st & 1) * f.stage_floats, *nxt = R + ((st + 1) & 1) * f.stage_floats;
            if (st + 1 < f.nstep) run_jobs(st + 1, nxt);
            // ---- contraction of step st: accumulate the step's k-slots into the bins this wave owns
            const int os = f.step_os[st];
            for (int q = f.step_ks0[st]; q < f.step_ks0[st + 1]; q++) {
                const float *bp = Bf + f.ks_boff[q] + wave * f.ks_bw[q] + lane * 4;
                const int nch = f.ks_bw[q] >> 8, nb = f.ks_nbin[q];
                float4 breg[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (c < nch) breg[c] = *reinterpret_cast<const float4 *>(bp + c * 256);
                const float *ab = cur + (arow >> 1) * os + f.ks_base[q] + kk * f.ks_rs[q] + (arow & 1) + 2 * wave;
#pragma unroll
                for (int i = 0; i < NBW; i++) {
                    if (RF_WAVES * i + wave < nb) {
                        const float a = ab[2 * RF_WAVES * i];
#pragma unroll
                        for (int t = 0; t < NT; t++) {
                            const int m = i * NT + t;
                            const float4 bv = breg[m >> 2];
                            const float b = (m & 3) == 0 ? bv.x : (m & 3) == 1 ? bv.y : (m & 3) == 2 ? bv.z : bv.w;
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][t], 0, 0, 0);
                        }
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
        // ---- CCF spectra -> LDS, inverse FFT, argmax: rounds of <= 4 references
        const float avg_o = nrm[min(o_epi, 7)];
        for (int zr = 0; zr < f.nzr; zr++) {
            const int ref_lo = zr * RF_ZREFS, nrz = min(RF_ZREFS, nref - ref_lo);
            const int tsel = ref_lo >> 3, r8lo = ref_lo & 7;
            const int rr = r8 - r8lo;
            const bool mine = o_epi < f.nxo && rr >= 0 && rr < nrz;
            const int zslot = o_epi * RF_ZREFS + rr;
            const float dcw = mine ? cdc[ref_lo + rr] * avg_o : 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t != tsel) continue;
#pragma unroll
                for (int i = 0; i < NBW; i++) {
                    const int k = RF_WAVES * i + wave;
                    const f32x4 c4 = acc[i][t];
                    const float s0 = odd ? c4[0] : c4[2], s1 = odd ? c4[1] : c4[3];
                    const float r0 = swap_lane_pair(s0), r1 = swap_lane_pair(s1);
                    const float ca = odd ? r0 : c4[0], cb = odd ? r1 : c4[1];
                    const float cc = odd ? c4[2] : r0, cd = odd ? c4[3] : r1;
                    if (mine) {
                        if (i == 0 && wave == 0) {
                            // bin 0: a = C0 X0 (minus the Normalize_ring mean), d = C_nyq X_nyq of the full-length rings
                            const float a0 = ca - dcw;
                            *reinterpret_cast<float2 *>(R + ZL::addr(zslot, 0)) = make_float2(a0, a0);
                            *reinterpret_cast<float2 *>(R + ZL::addr(zslot, N / 2)) = make_float2(cd, cd);
                        } else {
                            const float apd = ca + cd, amd = ca - cd, bpc = cb + cc, cmb = cc - cb;
                            *reinterpret_cast<float2 *>(R + ZL::addr(zslot, k)) = make_float2(apd + bpc, cmb + amd);
                            *reinterpret_cast<float2 *>(R + ZL::addr(zslot, N - k)) = make_float2(apd - bpc, amd - cmb);
                        }
                    }
                }
            }
            __syncthreads();
            {
                const int j = lane & 15, zs = wave * 4 + (lane >> 4);
                if (zs < f.nxo * RF_ZREFS && (zs & (RF_ZREFS - 1)) < nrz)      // uniform over the 16-lane group
                    ifft_argmax<N, 1, RF_ZREFS - 1>(R, pc, twl, zs, zs, j, ref_lo);
            }
            __syncthreads();
            // best reference of the round per offset (ascending reference, ">=": later wins), scaled by 1/sigma
            if (tid < f.nxo * (int)(sizeof(CandT) / 4)) {
                constexpr int W = sizeof(CandT) / 4;
                const int o = tid / W, wd = tid - o * W;
                float bv = pc[o * RF_ZREFS].val; int br = 0;
                for (int q3 = 1; q3 < nrz; q3++) {
                    const float v = pc[o * RF_ZREFS + q3].val;
                    if (v >= bv) { bv = v; br = q3; }
                }
                const int s = pass * nx1 + o;
                int word = reinterpret_cast<const int *>(pc + o * RF_ZREFS + br)[wd];
                if (wd == 0 || wd >= 3) word = __float_as_int(__int_as_float(word) * nrm[8 + o]);     // val, t7[]
                reinterpret_cast<int *>(cand + ((size_t)p * g.nshift_pad + s) * f.nzr + zr)[wd] = word;
            }
            __syncthreads();
        }
    }
}

}  // namespace ralign
