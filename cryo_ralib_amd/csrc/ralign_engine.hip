// libralign_hip.so -- host side of the MI355X 2-D alignment engine and its C ABI
// (include/ralign.h).  Owns geometry tables, workspaces and the kernel schedule; exposes
// the handle-based ra_* API (device pointers in, asynchronous on one HIP stream) and the
// reference's ctypes surface (cuda/gpu_aln_noref.h:52-113) on top of it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ralign_geom.h"
#include "ralign_kernels.h"
#include "ralign_generic.h"
#include "ralign_zone.h"
#include "ralign_fused.h"
#include "ralign_tiled.h"
#include "ralign_solo.h"
#include "ralign_duo.h"
#include "ralign_pair.h"
#include "ralign_exact.h"
#include "ralign_refine.h"

using namespace ralign;

static thread_local std::string g_last_error;

// The size-generic kernels for a geometry the LDS-resident ones cover: RALIGN_GENERIC=1 (A/B runs, tests), or an engine whose
// options only they implement (ra_create_ex: RA_INTERP_QUADRI).  The option is engine state; the flag below carries it into the
// planning helpers that see a geometry but no engine (fits_specialised_kernels, resident_expected) while that engine is planned.
static thread_local bool g_force_generic = false;
// ... and: plan this engine in the size-generic CLASS (crop / pair / solo kernels allowed) although the LDS-resident kernels would hold its
// image -- the second attempt of ra_create_ex for more than 16 references in boxes whose tiled plan does not fit next to the whole image
static thread_local bool g_generic_class = false;
static bool generic_forced() { return g_force_generic || (getenv("RALIGN_GENERIC") && atoi(getenv("RALIGN_GENERIC")) != 0); }
struct ForceGenericScope {
    bool prev;
    explicit ForceGenericScope(bool on) : prev(g_force_generic) { g_force_generic = g_force_generic || on; }
    ~ForceGenericScope() { g_force_generic = prev; }
};

#define RA_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t err__ = (call);                                                            \
        if (err__ != hipSuccess) {                                                            \
            char buf__[512];                                                                  \
            snprintf(buf__, sizeof(buf__), "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), \
                     __FILE__, __LINE__);                                                     \
            g_last_error = buf__;                                                             \
            return RA_ERR_HIP;                                                                \
        }                                                                                     \
    } while (0)

// device workspace of one engine, shared by ra_create and the size checks of the reference surface
// (pre_align_size_check / ref_free_alignment_2D_size_check), so that the estimate cannot drift from what is allocated
#define RA_XS_LDS_MAX ((size_t)80 * 1024)      // padded image copy of transform_sum_kernel: two workgroups of 1024 threads share a CU

struct WorkspacePlan {
    int chunk;
    size_t a_floats, cand_recs, refspec_floats, b_floats, alscratch_floats, zscr_recs;
    size_t bytes;          // everything ra_create and the first reference update take from the device
};

struct ra_engine {
    ra_config cfg{};
    Geometry geo;
    DevGeom dg{};
    hipStream_t stream = nullptr;
    int chunk = 0;
    int shift_cap = 0, pad_cap = 0;     // search offsets the tables / workspace were sized for
    int nrtile = 1;
    std::vector<void *> owned;          // device allocations freed at destroy
    float *d_A = nullptr;               // [chunk * ngroup + 2][a_blk]
    CandT *d_cand = nullptr;             // [(chunk * nshift_pad + 8)][nrtile]
    float *d_refspec = nullptr;         // [nref][lring]
    float *d_B = nullptr;               // [nrtile][LBP][16]
    float *d_cs = nullptr;              // [2]
    float *d_alscratch = nullptr;       // [chunk][nx*nx] aligned images of one chunk (deterministic class sums)
    // sub-bin angle refinement with the CPU path's arithmetic (ralign_exact.h)
    float refine_thr = 0.02f;           // flag |c3| < thr x max |b|; < 0: every particle; 0: off
    bool refine_ok = false;             // tables built
    bool refine_gm = false;             // ring buffers of the exact kernels in global memory (2 lcirc floats exceed the LDS: large boxes)
    float *d_rscratch = nullptr;        // [refine_grid][2 lcirc] (refine_gm)
    int refine_grid = 0;                // waves of a refine_winner_kernel launch
    size_t lds_refine = 0;
    float *d_twx = nullptr, *d_refx = nullptr, *d_cls_refx = nullptr;
    unsigned long long *d_refhash = nullptr;        // [nref] hashes of the exact reference spectra (ref_hash_kernel)
    int *d_refdup = nullptr;                        // [2 nref] copies of a reference inside the stack (DevGeom::ref_dup)
    int *d_twxoff = nullptr, *d_rcount = nullptr;
    RefineRec *d_rlist = nullptr;
    int *d_members = nullptr, *d_mcount = nullptr;      // [2 nref][chunk] member lists of a chunk, [2 nref] their lengths (class_members_kernel)
    float *d_sumpart = nullptr;         // [16][2 nref][nx*nx] per-run partial class sums (few classes: class_sum_kernel with runs)
    // transform_sum_kernel (rot_shift2D + class sums in one pass): member lists of a whole batch and per-run partial sums
    int *d_xs_members = nullptr, *d_xs_mcount = nullptr;
    float *d_xs_partial = nullptr;
    float2 *d_xs_trig = nullptr;
    size_t xs_cap_members = 0, xs_cap_partial = 0, xs_cap_trig = 0;
    bool atomic_sums = false;           // RALIGN_ATOMIC_SUMS=1: fp32 atomics instead of particle-order sums
    int *d_ring_off = nullptr, *d_numr = nullptr;
    float *d_wr = nullptr;
    size_t lds_polar = 0, lds_ref = 0, lds_ccf = 0, lds_xf = 0;
    bool force_generic = false;         // options that only the size-generic kernels implement (ra_create_ex)
    bool generic = false;               // size-generic kernels (ralign_generic.h): large boxes, maxrin > 256, > 48 rings
    bool no_tcrop = false;              // the crop plan did not fit the real tables once: planned without it (create_engine)
    bool tcrop = false;                 // generic class, but the search runs search_tiled_kernel over a CROP of the image (tcrop_wanted)
    int crop_S = 0;                     // ... whose side was planned for search shifts of up to this many pixels
    int crop_pst = 0;                   // ... and its row stride in LDS (tcrop_wanted)
    bool xf_generic = false;            // image does not fit LDS in transform_kernel
    float2 *d_zscr = nullptr;           // [g_nblk][maxrin][64 TM TR] CCF spectra scratch of ccf_generic_kernel
    // polar stage with the image in LDS, ring zone by ring zone (ralign_zone.h); the global-tap kernel stays underneath
    ZonePlanHost zplan;
    ZonePlanDev zdev{};
    bool zones = false;
    float2 *d_stats_part = nullptr;     // [chunk * nshift_pad + 8][nquad_total] Normalize_ring partial sums of every (entry, ring quad)
    size_t zone_cap_rows = 0, zone_cap_pix = 0, zone_cap_zones = 0, stats_part_cap = 0;
    int2 *d_zone_rows = nullptr; int *d_zone_pix = nullptr; ZoneDesc *d_zone_desc = nullptr;
    int *d_ent_base = nullptr, *d_ent_total = nullptr;      // live-offset lists of a chunk (size-generic path): [chunk + 1] entry ranges, their total
    float2 *d_gstats = nullptr;         // [chunk * nshift_pad + 8] Normalize_ring {avg, 1/sigma} of every particle-offset (generic path)
    unsigned long long *d_timeline = nullptr;      // profiling builds only (RALIGN_TIMELINE)
    float *d_cls_refspec = nullptr, *d_cls_Bf = nullptr;      // class-resident mode: [cls_cap][lring], [cls_cap][b_floats]
    int cls_cap = 0, cls_ready = 0;
    float *d_gcdc = nullptr;            // [nref] DC weights of the references (generic and fused paths)
    int g_nblk = 0, g_P = 0;
    size_t lds_gpolar = 0, lds_gccf = 0;
    // reference-update workspace (ralign_refine.h), allocated on first use
    int rf_cap = 0;                     // images the workspace holds
    double2 *d_rfT = nullptr, *d_rfF = nullptr, *d_rftw = nullptr;
    float *d_rfmean = nullptr, *d_rffsc = nullptr, *d_rfcs = nullptr;
    const int *d_fsc_off = nullptr, *d_fsc_idx = nullptr;      // coefficients of every Fourier shell (fsc_shell_table)
    bool refs_ready = false;
    // particle-resident search kernel (ralign_fused.h)
    FusedPlanHost fplan;
    std::vector<int> qoff;              // quadrant-table offset per log2(ring length)
    std::vector<float> ringw_h;
    bool fused = false;                 // plan valid and not disabled (RALIGN_FUSED=0)
    bool tiled = false;                 // the plan is search_tiled_kernel's (ralign_tiled.h: reference tiles, more than RF_MAXREF references)
    bool solo = false;                  // search_solo_kernel (ralign_solo.h: maxrin 512, one offset resident per pass) on top of the generic tables
    bool duo = false;                   // ... as search_duo_kernel (ralign_duo.h): two offsets per pass through the one ring buffer
    bool pair = false;                  // ... search_pair_kernel (ralign_pair.h): maxrin 256 in boxes too large for four ring buffers, two per pass
    float *d_Bf = nullptr;
    int *d_fbsrc = nullptr;
    size_t f_cap_b = 0;
    CandT *d_fcand = nullptr;           // [(fcand_cap * nshift_pad + 8)] one record per particle-offset of a resident-kernel launch
    int fcand_cap = 0, rlist_cap = 0;   // particles the candidate records / the refine list hold (grown on demand: ensure_resident_ws)
    int n_cu = 256;
    bool unfused_ws = false;            // spectra workspace of the two-kernel path allocated
    WorkspacePlan wp{};
    // kernel timing
    bool timing = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_ccf, ev_polar;
    size_t ev_used_ccf = 0, ev_used_polar = 0;
};

template <typename T> static int upload(ra_engine *e, const std::vector<T> &h, const T **dptr)
{
    void *d = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    RA_HIP(hipMalloc(&d, bytes));
    e->owned.push_back(d);
    if (!h.empty()) RA_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *dptr = (const T *)d;
    return RA_OK;
}

static bool fft_plan(int h, int &R1, int &R2)
{
    switch (h) {
    case 4: R1 = 4; R2 = 1; return true;
    case 8: R1 = 8; R2 = 1; return true;
    case 16: R1 = 16; R2 = 1; return true;
    case 32: R1 = 8; R2 = 4; return true;
    case 64: R1 = 8; R2 = 8; return true;
    case 128: R1 = 16; R2 = 8; return true;
    case 256: R1 = 16; R2 = 16; return true;
    default: return false;
    }
}

// polar stage of the size-generic class: alrl_ms with Util::bilinear (default) or Util::quadri (ra_create_ex)
typedef void (*gpolar_fn)(DevGeom, const float *, const float *, int, float *, float2 *);
static gpolar_fn gpolar_kernel(const ra_engine *e, bool refs)
{
    const bool q = e->dg.interp == RA_INTERP_QUADRI;
    return refs ? (q ? (gpolar_fn)polar_generic_kernel<true, true> : (gpolar_fn)polar_generic_kernel<true, false>)
                : (q ? (gpolar_fn)polar_generic_kernel<false, true> : (gpolar_fn)polar_generic_kernel<false, false>);
}

// the particle-resident kernel is planned for this engine (decided before the LDS layout, which differs slightly)
static bool tiled_wanted(const ra_engine *e)
{
    // search_tiled_kernel: more references than one pass of search_fused_kernel accumulates (RALIGN_TILED=1 forces it for fewer)
    const bool force = getenv("RALIGN_TILED") && atoi(getenv("RALIGN_TILED")) != 0;
    if (getenv("RALIGN_TILED") && !force) return false;
    // (from 15 references on: search_fused_kernel needs two spectra rounds per pass from 12 on and is 4 % slower at 15 and 16)
    return (e->cfg.nref >= RT_MINREF || force) && e->geo.maxrin == 256 && e->geo.nring <= 4 * RT_NQ && e->cfg.nref <= 127;
}

// search_solo_kernel is planned for this engine: a geometry of the size-generic class whose rings end at 512 samples
// (RALIGN_SOLO=0: the generic kernels)
static bool solo_wanted(const ra_engine *e)
{
    if (!e->generic || e->geo.maxrin != 512 || e->geo.nring > 4 * RS_NQ || e->geo.numr[2] < 8 || e->cfg.nref > 127) return false;
    if (g_force_generic) return false;      // engine options that only the size-generic kernels implement (RALIGN_GENERIC=1 leaves this class alone, as before)
    return !(getenv("RALIGN_SOLO") && atoi(getenv("RALIGN_SOLO")) == 0);
}

// search_pair_kernel is planned for this engine: a geometry of the size-generic class whose rings end at 256 samples -- a box
// too large for the four ring buffers of the LDS-resident kernels (RALIGN_PAIR=0: the generic kernels)
static bool pair_wanted(const ra_engine *e)
{
    if (!e->generic || e->geo.maxrin != 256 || e->geo.nring > 4 * RP_NQ || e->geo.numr[2] < 8 || e->cfg.nref > 127) return false;
    if (generic_forced()) return false;      // the switch that forces the generic kernels
    return !(getenv("RALIGN_PAIR") && atoi(getenv("RALIGN_PAIR")) == 0);
}

// two offsets per pass (search_duo_kernel, ralign_duo.h) for the solo class: the default (measured against search_solo_kernel:
// +4.5 % at 128 / 60 / nref 10, +24 % at 130 / 52 / nref 50); RALIGN_DUO=0: one offset per pass
static bool duo_wanted(const ra_engine *e)
{
    if (!solo_wanted(e)) return false;
    return !(getenv("RALIGN_DUO") && atoi(getenv("RALIGN_DUO")) == 0);
}

typedef void (*fused_fn)(DevGeom, FusedGeom, const float *, const float *, int, const float *, int, CandT *, const int *);
static fused_fn select_tiled(int nh, int sbuf);
static fused_fn select_fused(int maxrin, int nref, int nzr, int sbuf, bool pack = false, bool crop = false);
// search_tiled_kernel for an engine of the size-generic class: rings of 256 samples (ou <= 36) in a box so much larger than the rings
// that the whole image does not fit next to four ring buffers, but a crop around the particle's sampling centre does (crop_plan,
// ralign_solo.h).  Four offsets per pass with every wave in a ring job instead of the pair kernel's two: 1.6 x its rate.
// RALIGN_TCROP=0: the pair kernel
static bool tcrop_wanted(ra_engine *e)
{
    const Geometry &g = e->geo;
    // (search_tiled_kernel holds slices of at most 36 rings, search_fused_kernel reads its operand from the ring buffers: up to 64)
    if (!e->generic || g.maxrin != 256 || g.nring > 64 || g.numr[2] < 8 || e->cfg.nref > 127) return false;
    if (generic_forced()) return false;
    if (e->no_tcrop || (getenv("RALIGN_TCROP") && atoi(getenv("RALIGN_TCROP")) == 0)) return false;
    if (getenv("RALIGN_FUSED") && atoi(getenv("RALIGN_FUSED")) == 0) return false;
    FusedGeom t{};
    crop_plan(g, t);
    if (!t.s_crop) return false;
    // the job tables and the image stride are laid out for this kernel before its plan is made (build_device_geometry), so the
    // answer has to be the plan's: the same plan on an upper estimate of the polar part's LDS (tables of 4 offset slots)
    const int sbuf = (g.lring + 31) / 32 * 32 + 16;
    FusedPlanHost tmp;
    // row stride of the crop: 101 words -- the stride of the 90 x 90 headline geometry, which the order of the ring jobs' instances was
    // tuned on -- when the crop is narrower and the plan still fits, else the narrowest conflict-poor one (crop_plan).  Measured
    // (search launch per 8192 / 16 384 particles): ou = 36 / 256 x 256: 5.40 ms at 91, 5.35 at 93, 5.39 - 5.41 at 95 - 99, 5.23 at 101;
    // ou = 30 / 160 x 160: 9.56 ms at 73, 9.28 - 9.54 at 75 - 99, 9.35 at 101
    const int pst_min = t.s_pst;
    for (int pst : {pst_min < 101 && !(RA_EXP_ENV("RALIGN_CROP_PST101") && ra_atoi(RA_EXP_ENV("RALIGN_CROP_PST101")) == 0) ? 101 : pst_min, pst_min}) {
    e->crop_pst = pst;
    const size_t polar = (size_t)pst * pst + 4 * (size_t)std::max(sbuf, sbuf <= RF_SBUF_FIXED ? RF_SBUF_FIXED : sbuf) + 2 * g.maxrin + 3000;
    // the kernels' division of the reference counts is that of the 90 x 90 engines: search_fused_kernel up to RT_MINREF - 1
    // references, search_tiled_kernel from there on (RALIGN_TILED=1: from 7 on, the fewest its tile sizes are instantiated for)
    const bool force = getenv("RALIGN_TILED") && atoi(getenv("RALIGN_TILED")) != 0, no_tiled = getenv("RALIGN_TILED") && !force;
    for (int sb : {sbuf <= RF_SBUF_FIXED ? RF_SBUF_FIXED : sbuf, sbuf}) {
        if (!no_tiled && (e->cfg.nref >= RT_MINREF || force) && build_tiled_plan(g, e->cfg.nref, sb, polar, tmp) && select_tiled(tmp.f.nh, sb)) return true;
        if (e->cfg.nref <= RF_MAXREF && build_fused_plan(g, e->cfg.nref, sb, polar, tmp) && select_fused(g.maxrin, e->cfg.nref, tmp.f.nzr, sb, false, true)) return true;
        if (!no_tiled && build_tiled_plan(g, e->cfg.nref, sb, polar, tmp) && select_tiled(tmp.f.nh, sb)) return true;
    }
    }
    return false;
}

static bool fused_wanted(const ra_engine *e)
{
    if (e->generic) return false;
    if (getenv("RALIGN_FUSED") && atoi(getenv("RALIGN_FUSED")) == 0) return false;
    if (tiled_wanted(e)) return true;
    if (e->cfg.nref > RF_MAXREF) return false;
    return e->geo.maxrin == 256 || e->geo.maxrin == 128;
}

static int build_device_geometry(ra_engine *e)
{
    Geometry &g = e->geo;
    DevGeom &d = e->dg;
    d.nx = g.nx; d.cnx = g.nx / 2 + 1;
    d.nring = g.nring; d.maxrin = g.maxrin; d.lcirc = g.lcirc; d.lring = g.lring; d.nbins = g.nbins;
    d.LB = g.LB; d.LBP = g.LBP; d.last_ring = g.last_ring;
    d.win_ring = e->cfg.mode == RA_MODE_MREF ? g.last_ring : g.numr[3 * (g.nring - 1)];
    d.nshift = g.nshift; d.nshift_pad = g.nshift_pad; d.nkx = g.nkx; d.nky = g.nky; d.ent_stride = e->generic ? g.nshift : g.nshift_pad;      // generic class: dense entries
    d.step = g.step; d.xrng = e->cfg.xrng; d.yrng = e->cfg.yrng;
    d.nn_weight = g.nn_weight; d.inv_nn_weight = g.nn_weight > 0.f ? (float)(1.0 / (double)g.nn_weight) : 0.f; d.lg_maxrin = ilog2_floor(g.maxrin); d.mode = e->cfg.mode; d.nomirror = 0; d.norm_ring = e->cfg.mode == RA_MODE_MREF ? 1 : 0; d.interp = RA_INTERP_BILINEAR; d.quad_aligned = g.quad_aligned ? 1 : 0;
#ifdef RALIGN_PROFILE_SWITCHES
    d.dbg = RA_EXP_ENV("RALIGN_DEBUG") ? ra_atoi(RA_EXP_ENV("RALIGN_DEBUG")) : 0;
    d.timeline = nullptr;
    if (RA_EXP_ENV("RALIGN_TIMELINE")) {          // 64 passes x 16 waves x 16 stamps, written by workgroup 0 for its first particle
        if (!e->d_timeline && hipMalloc(&e->d_timeline, 64 * 16 * 16 * sizeof(unsigned long long)) != hipSuccess) e->d_timeline = nullptr;
        if (e->d_timeline) (void)hipMemset(e->d_timeline, 0, 64 * 16 * 16 * sizeof(unsigned long long));
        d.timeline = e->d_timeline;
    }
#else
    d.dbg = 0;
    d.timeline = nullptr;
#endif
    // ring-buffer stride.  Kernel pair: == 8 (mod 32), the 4 offsets of an entry hit disjoint banks in the write-out gather.
    // Fused kernel: == 16 (mod 32), the two offsets a 4x4x1 MFMA A operand reads (16 bins x Re/Im each) sit in disjoint
    // halves of the 32 banks (RALIGN_SBUF_PAD overrides: experiments)
    int sbuf = (g.lring + 31) / 32 * 32 + (RA_EXP_ENV("RALIGN_SBUF_PAD") ? ra_atoi(RA_EXP_ENV("RALIGN_SBUF_PAD")) : ((fused_wanted(e) || e->tcrop) ? 16 : 8));
    // fused kernel at maxrin 256: pad the stride to the compile-time value of its fixed-stride instantiations when the image
    // and four such buffers (+ 16 KB of tables and records) still fit the LDS (RALIGN_SBUF_FIXED=0: keep the run-time stride)
    if ((fused_wanted(e) || e->tcrop) && g.maxrin == 256 && sbuf <= RF_SBUF_FIXED && !RA_EXP_ENV("RALIGN_SBUF_PAD") &&
        !(RA_EXP_ENV("RALIGN_SBUF_FIXED") && ra_atoi(RA_EXP_ENV("RALIGN_SBUF_FIXED")) == 0)) {
        const int bd0 = (int)std::ceil(std::max(e->cfg.xrng, e->cfg.yrng)) + 2, pst0 = e->tcrop ? e->crop_pst : g.nx + 2 * bd0 + 3;
        if ((size_t)(pst0 * pst0 + 4 * RF_SBUF_FIXED + 3400) * sizeof(float) <= 160 * 1024) sbuf = RF_SBUF_FIXED;
    }
    d.sbuf = sbuf;
    d.a_blk = g.LBP * 8 + 64;
    // classes of bins with equal ring-slot count
    d.n_class = 0;
    for (int k = 0; k < g.nbins && !e->generic; k++) {
        int ns = (g.bin_offp[k + 1] - g.bin_offp[k]) / 4;
        if (d.n_class == 0 || d.class_ns[d.n_class - 1] != ns) {
            if (d.n_class == 8) { g_last_error = "too many bin classes"; return RA_ERR_ARG; }
            d.class_k0[d.n_class] = k; d.class_ns[d.n_class] = ns; d.n_class++;
        }
    }
    d.class_k0_end = g.nbins;
    if (!e->generic) {   // start wave of every class: greedy choice that keeps the per-wave MFMA count level
        const int NWV = RA_CCF_THREADS / 64;
        std::vector<long> load(NWV, 0);
        for (int c = 0; c < d.n_class; c++) {
            const int nb = ((c + 1 < d.n_class) ? d.class_k0[c + 1] : g.nbins) - d.class_k0[c], w = d.class_ns[c] + 3;
            int best_r = 0; long best_max = -1, best_sq = 0;
            for (int r = 0; r < NWV; r++) {
                long mx = 0, sq = 0;
                for (int wv = 0; wv < NWV; wv++) {
                    const int first = (wv - r + NWV) % NWV;                 // index of this wave's first bin in the class
                    const long t = load[wv] + (first < nb ? (long)((nb - first + NWV - 1) / NWV) * w : 0);
                    mx = std::max(mx, t); sq += t * t;
                }
                if (best_max < 0 || mx < best_max || (mx == best_max && sq < best_sq)) { best_max = mx; best_sq = sq; best_r = r; }
            }
            d.class_rot[c] = best_r;
            for (int wv = 0; wv < NWV; wv++) {
                const int first = (wv - best_r + NWV) % NWV;
                if (first < nb) load[wv] += (long)((nb - first + NWV - 1) / NWV) * w;
            }
        }
    }
    build_operand_tables(g, sbuf);

    // FFT work lists
    std::vector<int4> A, B, C;
    struct RingPlan { int i, R1, R2; };
    std::vector<RingPlan> plans;
    for (int i = 0; i < g.nring && !e->generic; i++) {
        int n = g.numr[3 * i + 2], R1, R2;
        if (!fft_plan(n / 2, R1, R2)) { g_last_error = "unsupported ring length"; return RA_ERR_ARG; }
        plans.push_back({i, R1, R2});
        for (int b = 0; b < R2; b++) A.push_back(make_int4(g.ring_off[i], R1, R2, b));
        int h = n / 2;
        for (int k = 0; k <= h / 2; k++)
            C.push_back(make_int4(g.ring_off[i] + 2 * k, g.ring_off[i] + 2 * (h - k), k * (g.maxrin / n), k == 0));
    }
    std::stable_sort(A.begin(), A.end(), [](const int4 &a, const int4 &b) { return a.y > b.y; });
    std::stable_sort(plans.begin(), plans.end(), [](const RingPlan &a, const RingPlan &b) { return a.R1 > b.R1; });
    for (auto &p : plans)
        if (p.R2 > 1)
            for (int c = 0; c < p.R1; c++) B.push_back(make_int4(g.ring_off[p.i], p.R1, p.R2, c));
    while (B.size() % 16) B.push_back(make_int4(0, 0, 0, 0));
    d.n_itemA = (int)A.size(); d.n_itemB = (int)B.size(); d.n_itemC = (int)C.size();

    // wave-job schedule of the polar kernel: ring instances (offset slot, ring) sorted by ring
    // length (longest first), cut into jobs of 64/LR instances of one length
    std::vector<float2> qtab;
    std::vector<int4> ringinfo(g.nring);
    std::vector<float> ringw(g.nring);
    std::vector<int4> inst;
    std::vector<float> instw;
    std::vector<int4> jobs;
    {
        std::vector<int> &qoff = e->qoff;
        qoff.assign(32, -1);
        const double qpi = 2 * atan(1.0);
        for (int i = 0; i < g.nring; i++) {
            const int n = g.numr[3 * i + 2], lg = ilog2_floor(n);
            if (qoff[lg] < 0) {
                qoff[lg] = (int)qtab.size();
                const int lt = n / 4;
                const double dfi = qpi / lt;
                for (int jt = 0; jt < lt; jt++) {
                    float fi = (float)(dfi * jt);
                    qtab.push_back(make_float2(sinf(fi), cosf(fi)));
                }
            }
            ringinfo[i] = make_int4(g.ring_off[i], g.numr[3 * i], n, qoff[lg]);
            ringw[i] = (float)(g.numr[3 * i] * 2 * M_PI / (float)n);
        }
        auto code_of = [](int n) { switch (n) { case 512: return 10; case 256: return 0; case 128: return 1; case 64: return 2; case 32: return 3; case 16: return 4; case 8: return 5; default: return -1; } };
        for (int i = 0; i < g.nring && !e->generic; i++)
            if (code_of(g.numr[3 * i + 2]) < 0 || g.numr[3 * i + 2] > 256) { g_last_error = "ring length not supported by the polar kernel (8..256)"; return RA_ERR_ARG; }
        // job code 6: 256-sample rings with 8 lanes per ring (16 sample pairs per lane, two 8-point rows per lane in
        // the second FFT pass) instead of 16 lanes with half of them idle there; code 7 does the same for 64-sample rings
        // (4 lanes x 8 pairs).  Measured: polar stage 5.64 -> 5.2 ms per 7143 particles.
        // nslot = 4: the four offset slots of a pass of the LDS-resident kernels; nslot = 1: search_solo_kernel (rings up to 512
        // samples, code 10: 16 lanes per ring)
        auto make_jobs = [&](int nslot, std::vector<int4> &J, std::vector<int4> &I, std::vector<float> &W, bool lightjobs) {
            const int lanes_of[12] = {16, 8, 8, 4, 4, 4, 8, 4, 0, 0, 16, 32};
            const bool solo = nslot == 1, pairj = nslot == 2;
            // solo class, lightjobs: the short, register-light jobs -- code 11 (512-sample rings, 32 lanes per ring, 8 x 8 x 4:
            // ring_job512) and code 0 -- instead of codes 10 / 6 (16 sample pairs per lane).  They keep all 16 waves busy but are
            // SLOWER in search_solo_kernel (295 k against 324 k particles/s at 128 / 60 / nref 10: the ring jobs of a pass are bound
            // by the LDS array, ~10 k cycles of tap reads and transposes per offset, not by the number of waves that issue them);
            // search_duo_kernel needs them for its second offset, whose ring jobs run next to a live A slice
            // rings of 8 .. 32 samples share jobs of code 9 (ring_job_mix: n / 8 lanes per ring, one table entry per lane)
            // (RALIGN_MIX_JOBS=0, one job per ring length, is an experiment switch of the kernel pair: the fused kernel carries
            // the job variants of codes 1, 6, 7 and 9 only)
            const bool mixed = solo || pairj || (nslot == 4 && (fused_wanted(e) || e->tcrop || !(RA_EXP_ENV("RALIGN_MIX_JOBS") && ra_atoi(RA_EXP_ENV("RALIGN_MIX_JOBS")) == 0)));
            for (int lg = solo ? 9 : 8; lg >= (mixed ? 6 : 3); lg--) {
                const int n = 1 << lg;
                const int code = lightjobs && n == 512 ? 11 : lightjobs && n == 256 ? 0 :
                                 (n == 256 && (nslot == 4 || solo || pairj)) ? 6 : ((n == 64 && (nslot == 4 || solo || pairj)) ? 7 : code_of(n));
                std::vector<int4> cls;
                std::vector<float> clsw;
                for (int sft = 0; sft < nslot; sft++)
                    for (int i = 0; i < g.nring; i++)
                        if (g.numr[3 * i + 2] == n) {
                            cls.push_back(make_int4(sft | (i << 8), g.ring_off[i], qoff[lg], g.numr[3 * i]));
                            clsw.push_back(ringw[i]);
                        }
                const int per_job = 64 / lanes_of[code];
                for (size_t b = 0; b < cls.size(); b += per_job) {
                    const int cnt = (int)std::min<size_t>(per_job, cls.size() - b);
                    J.push_back(make_int4(code, (int)I.size(), cnt, 0));
                    for (int c = 0; c < cnt; c++) { I.push_back(cls[b + c]); W.push_back(clsw[b + c]); }
                }
            }
            if (mixed) {
                std::vector<int4> lanes;              // lane entries of the current job (aligned groups: 4-lane rings first)
                std::vector<float> lanew;
                auto flush = [&]() {
                    if (lanes.empty()) return;
                    J.push_back(make_int4(9, (int)I.size(), (int)lanes.size(), 0));
                    for (size_t c = 0; c < lanes.size(); c++) { I.push_back(lanes[c]); W.push_back(lanew[c]); }
                    lanes.clear(); lanew.clear();
                };
                for (int lg = 5; lg >= 3; lg--) {
                    const int LR = 1 << (lg - 3);
                    for (int sft = 0; sft < nslot; sft++)
                        for (int i = 0; i < g.nring; i++)
                            if (g.numr[3 * i + 2] == (1 << lg)) {
                                if (lanes.size() + LR > 64) flush();
                                for (int t = 0; t < LR; t++) {
                                    lanes.push_back(make_int4(sft | (i << 8) | (lg << 16) | (t << 20), g.ring_off[i], qoff[lg], g.numr[3 * i]));
                                    lanew.push_back(ringw[i]);
                                }
                            }
                }
                flush();
            }
        };
        const bool light_only = getenv("RALIGN_SOLO_JOBS") && atoi(getenv("RALIGN_SOLO_JOBS")) == 1;
        if (!e->generic) make_jobs(4, jobs, inst, instw, false);
        else if (solo_wanted(e)) {
            // table A (every kernel of the class; the first offset of a duo pass), then -- duo -- table B: the light jobs of the second offset
            make_jobs(1, jobs, inst, instw, light_only);
            d.n_job_b = 0;
            const int na = (int)jobs.size();
            if (duo_wanted(e) && !light_only) { make_jobs(1, jobs, inst, instw, true); d.n_job_b = (int)jobs.size() - na; }
        } else if (e->tcrop) make_jobs(4, jobs, inst, instw, false);
        else if (pair_wanted(e)) make_jobs(2, jobs, inst, instw, false);
        if (const char *po = RA_EXP_ENV("RALIGN_JOB_ORDER")) {       // experiments: wave w of a pass runs job order[w] ("3,2,1,0,...")
            std::vector<int4> perm;
            for (const char *c = po; *c;) {
                const int k = atoi(c);
                if (k >= 0 && k < (int)jobs.size()) perm.push_back(jobs[k]);
                while (*c && *c != ',') c++;
                if (*c == ',') c++;
            }
            if (perm.size() == jobs.size()) jobs = perm;
        }
    }
    if (!(e->generic && solo_wanted(e))) d.n_job_b = 0;
    d.n_job = (int)jobs.size() - d.n_job_b; d.n_qtab = (int)qtab.size(); d.n_inst = (int)inst.size();
    e->ringw_h = ringw;
    d.bd = (int)std::ceil(std::max(e->cfg.xrng, e->cfg.yrng)) + 2;
    d.pst = g.nx + 2 * d.bd;
    if (e->tcrop) {          // borderless crop (search_tiled_kernel: load_image): rows / columns 0 .. side - 1 + one spare
        d.bd = 0; d.pst = e->crop_pst;
    }
    // row stride of the padded LDS image: the lanes of a ring job sit along an arc and across consecutive radii, so
    // bilinear taps step through the image by +-1 column, +-1 row (= pst words) or a diagonal (pst +- 1).  A stride
    // that is a multiple of 32 (96 at nx = 90) puts every vertical neighbour into the same LDS bank; pick the next
    // stride whose residues pst, pst - 1, pst + 1 share at most a factor 4 with the 32 banks.
    if (!(RA_EXP_ENV("RALIGN_PST_RAW") && ra_atoi(RA_EXP_ENV("RALIGN_PST_RAW")) != 0))
        while (!((d.pst & 1) && ((d.pst - 1) & 7) && ((d.pst + 1) & 7))) d.pst++;

    std::vector<float2> tw(g.maxrin);
    for (int k = 0; k < g.maxrin; k++) {
        double a = -2.0 * M_PI * k / g.maxrin;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    std::vector<float> mask((size_t)g.nx * g.nx);
    {
        float radius = (float)g.last_ring;
        for (int j = 0; j < g.nx; j++)
            for (int i = 0; i < g.nx; i++) {
                float x2 = ((float)i - g.nx / 2) * ((float)i - g.nx / 2) / (radius * radius);
                float y2 = ((float)j - g.nx / 2) * ((float)j - g.nx / 2) / (radius * radius);
                mask[(size_t)j * g.nx + i] = (x2 + y2 <= 1.0f) ? 1.0f : 0.0f;
            }
    }
    int rc;
    if ((rc = upload(e, g.samp_dx, &d.samp_dx))) return rc;
    if ((rc = upload(e, g.samp_dy, &d.samp_dy))) return rc;
    if ((rc = upload(e, g.samp_w, &d.samp_w))) return rc;
    if ((rc = upload(e, g.samp_dst, &d.samp_dst))) return rc;
    if ((rc = upload(e, g.bin_off, &d.bin_off))) return rc;
    if ((rc = upload(e, g.bin_offp, &d.bin_offp))) return rc;
    if ((rc = upload(e, g.ent_src, &d.ent_src))) return rc;
    if ((rc = upload(e, g.ent_wgt, &d.ent_wgt))) return rc;
    {
        const int *tmp_a;
        if ((rc = upload(e, g.a_src, &tmp_a))) return rc;
        d.a_src4 = reinterpret_cast<const int4 *>(tmp_a);
    }
    if ((rc = upload(e, g.b_src, &d.b_src))) return rc;
    {
        const int *tmp_p;
        if ((rc = upload(e, g.ent_apos, &tmp_p))) return rc;
        d.ent_apos = reinterpret_cast<const int2 *>(tmp_p);
    }
    if ((rc = upload(e, g.bin_first, &d.bin_first))) return rc;
    // shift tables are sized for the create-time window; ra_reset_shifts rewrites them
    std::vector<float> sx(g.shift_x), sy(g.shift_y);
    if ((rc = upload(e, sx, &d.shift_x))) return rc;
    if ((rc = upload(e, sy, &d.shift_y))) return rc;
    if ((rc = upload(e, tw, &d.tw))) return rc;
    if ((rc = upload(e, A, &d.itemA))) return rc;
    if ((rc = upload(e, B, &d.itemB))) return rc;
    if ((rc = upload(e, C, &d.itemC))) return rc;
    if ((rc = upload(e, mask, &d.mask))) return rc;
    if ((rc = upload(e, jobs, &d.jobs))) return rc;
    if ((rc = upload(e, inst, &d.inst))) return rc;
    if ((rc = upload(e, instw, &d.instw))) return rc;
    if ((rc = upload(e, qtab, &d.qtab))) return rc;
    if ((rc = upload(e, ringinfo, &d.ringinfo))) return rc;
    if ((rc = upload(e, ringw, &d.ringw))) return rc;
    const int *tmp_i; const float *tmp_f;
    if ((rc = upload(e, g.ring_off, &tmp_i))) return rc; e->d_ring_off = (int *)tmp_i;
    if ((rc = upload(e, g.numr, &tmp_i))) return rc; e->d_numr = (int *)tmp_i;
    if ((rc = upload(e, g.wr, &tmp_f))) return rc; e->d_wr = (float *)tmp_f;
    return RA_OK;
}

template <typename T> static int dev_alloc(ra_engine *e, T **p, size_t count, bool zero)
{
    void *d = nullptr;
    RA_HIP(hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(T)));
    e->owned.push_back(d);
    if (zero) RA_HIP(hipMemset(d, 0, std::max<size_t>(count, 1) * sizeof(T)));
    *p = (T *)d;
    return RA_OK;
}

// replace a device buffer by a larger one (the old one is released now, not at destroy)
template <typename T> static int dev_grow(ra_engine *e, T **p, size_t count, bool zero)
{
    if (*p) {
        auto it = std::find(e->owned.begin(), e->owned.end(), (void *)*p);
        if (it != e->owned.end()) e->owned.erase(it);
        (void)hipStreamSynchronize(e->stream);
        (void)hipFree(*p);
        *p = nullptr;
    }
    return dev_alloc(e, p, count, zero);
}

// Will ra_create select a particle-resident kernel (search_fused_kernel / search_tiled_kernel) for this geometry?  The same
// conditions as fused_wanted / tiled_wanted and the LDS estimate of build_device_geometry, evaluated without an engine, so
// that the size checks charge what that path allocates (candidate records, the B stream) instead of the spectra panels of
// the two-kernel path, which are then never allocated (ensure_unfused_ws is lazy).
static bool resident_expected(const Geometry &g, const ra_config &cfg, bool generic, size_t *b_floats)
{
    if (generic) {
        // search_solo_kernel / search_duo_kernel (setup_solo): rings of 512 samples, image and one ring buffer in the LDS;
        // search_pair_kernel: rings of 256 samples, image and two ring buffers
        const bool c512 = g.maxrin == 512 && g.nring <= 4 * RS_NQ && !g_force_generic && !(getenv("RALIGN_SOLO") && atoi(getenv("RALIGN_SOLO")) == 0);
        const bool c256 = g.maxrin == 256 && g.nring <= 4 * RP_NQ && !(getenv("RALIGN_PAIR") && atoi(getenv("RALIGN_PAIR")) == 0) &&
                          !generic_forced();
        if (!(c512 || c256) || g.numr[2] < 8 || cfg.nref > 127) return false;
        // the LDS image is a crop around the particle's centre when the box is larger than the rings need (crop_plan)
        const int S = (int)std::ceil(std::max(g.nkx, g.nky) * g.step - 1e-6), side = 2 * (S + g.last_ring) + 5;
        const int cols = (side < g.nx && !(getenv("RALIGN_CROP") && atoi(getenv("RALIGN_CROP")) == 0)) ? side : g.nx;
        int pst = cols + 1;
        while (!((pst & 1) && ((pst - 1) & 7) && ((pst + 1) & 7))) pst++;
        const int nrp = (cfg.nref + 1) / 2, ntile = (nrp + RS_MAXNH - 1) / RS_MAXNH, nh = (nrp + ntile - 1) / ntile;
        const int sbuf = c256 ? 2 * ((g.lring + 31) / 32 * 32 + 16)
                              : std::max((g.lring + 31) / 32 * 32 + 16, 2 * nh * (2 * (g.maxrin + g.maxrin / 16) + 2));
        if ((size_t)((cols + 1) * pst + sbuf + 4600 + 2 * g.nring + 10 * (g.nring + 16)) * sizeof(float) > 160 * 1024) return false;
        size_t quads = 0;
        for (int m = 0; m < g.maxrin / 32; m++) {
            int r0 = 0;
            while (r0 < g.nring) {
                const int n = g.numr[3 * r0 + 2], nbin = (n == g.maxrin) ? n / 2 : n / 2 + 1;
                if (16 * m < nbin) break;
                r0++;
            }
            quads += (g.nring - r0 + 3) / 4;
        }
        if (b_floats) *b_floats = (size_t)nrp * quads * 256 + 256;
        return true;
    }
    if (getenv("RALIGN_FUSED") && atoi(getenv("RALIGN_FUSED")) == 0) return false;
    const bool tiled = cfg.nref >= RT_MINREF && g.maxrin == 256 && g.nring <= 4 * RT_NQ && cfg.nref <= 127 &&
                       !(getenv("RALIGN_TILED") && atoi(getenv("RALIGN_TILED")) == 0);
    if (!tiled && (cfg.nref > RF_MAXREF || !(g.maxrin == 256 || g.maxrin == 128))) return false;
    if (g.numr[2] < 8 || g.nring > 64) return false;
    const int bd0 = (int)std::ceil(std::max(cfg.xrng, cfg.yrng)) + 2, pst0 = g.nx + 2 * bd0 + 3;
    const int sbuf0 = (g.lring + 31) / 32 * 32 + 16;
    if ((size_t)(pst0 * pst0 + 4 * sbuf0 + 3400 + (tiled ? 1200 : 0)) * sizeof(float) > 160 * 1024) return false;
    size_t quads = 0;
    for (int m = 0; m < g.maxrin / 32; m++) {
        int r0 = 0;
        while (r0 < g.nring) {
            const int n = g.numr[3 * r0 + 2], nbin = (n == g.maxrin) ? n / 2 : n / 2 + 1;
            if (16 * m < nbin) break;
            r0++;
        }
        quads += (g.nring - r0 + 3) / 4;
    }
    if (b_floats) *b_floats = (size_t)((cfg.nref + 1) / 2) * quads * 256 + 256;
    return true;
}

static WorkspacePlan plan_workspace(const Geometry &g, const ra_config &cfg, bool generic)
{
    WorkspacePlan w{};
    const int a_blk = g.LBP * 8 + 64, nrtile = (cfg.nref + 7) / 8;
    int chunk = cfg.chunk > 0 ? cfg.chunk : 8192;
    chunk = std::min((chunk + 1) & ~1, 32768);      // class_sum_kernel keeps a chunk's member list in LDS
    const int ngroup = g.nshift_pad / 4;
    {   // keep the spectra workspace of one chunk within ~12 GB (large boxes: tens of MB per particle)
        const size_t per_particle = (size_t)ngroup * a_blk * sizeof(float);
        const size_t cap = std::max<size_t>(2, ((size_t)12 << 30) / per_particle);
        if ((size_t)chunk > cap) chunk = (int)(cap & ~(size_t)1);
    }
    if (generic && g.maxrin == 1024 && !(getenv("RALIGN_GCCF_SPLIT") && atoi(getenv("RALIGN_GCCF_SPLIT")) == 0) && cfg.chunk <= 0) {
        // split contraction: a chunk is walked in slices of nblk blocks, one block per workgroup, and a slice takes its time
        // whether it is full or not (341 particles = 2580 blocks of 4 x 7 tiles = 10 slices of 256 + one of 20): take the chunk
        // size, up to 64 particles below the cap, with the most particles per slice
        const bool wide = gccf_wide_blocks(nrtile);
        const int tm = wide ? gccf_tm(nrtile, g.maxrin) : 2, tr = wide ? 7 : 2, nblk = (tm >= 4 ? 256 : 512) * gccf_blocks_per_wg();
        int best = chunk;
        double best_pps = 0.0;
        for (int cn = chunk; cn >= std::max(2, chunk - 64); cn -= 2) {
            const long long n_mtile = ((long long)cn * (generic ? g.nshift : g.nshift_pad) + 7) / 8;      // (the generic class packs its entries densely)
            const long long ntask = ((n_mtile + tm - 1) / tm) * ((nrtile + tr - 1) / tr), slices = (ntask + nblk - 1) / nblk;
            const double pps = (double)cn / (double)slices;
            if (pps > best_pps) { best_pps = pps; best = cn; }
        }
        chunk = best;
    }
    w.chunk = chunk;
    w.a_floats = ((size_t)chunk * ngroup + 2) * a_blk;
    w.cand_recs = ((size_t)chunk * g.nshift_pad + 8) * nrtile;
    w.refspec_floats = (size_t)cfg.nref * g.lring;
    w.b_floats = (size_t)nrtile * g.LBP * 16;
    w.alscratch_floats = (size_t)chunk * g.nx * g.nx;
    // CCF-spectra scratch of ccf_generic_kernel: 64 pairs x 7 tiles per workgroup, x 14 for the 2 x 7 blocks (gccf_tm; 4 x 7: half the workgroups)
    const bool wide2 = generic && gccf_tm(nrtile, g.maxrin) >= 2;
    w.zscr_recs = generic ? (size_t)512 * (wide2 ? RA_GCCF_ZPAIRS_MAX : RA_GCCF_ZPAIRS_MAX / 2) * (g.maxrin + 2) : 0;      // + 2: N/2 + 1 bins of two values (split kernels); allocated on first use of the generic search
    if (generic && g.maxrin == 1024) w.zscr_recs *= gccf_blocks_per_wg();
    const size_t nxh = g.nx / 2 + 1, rf_cap = 2 * (size_t)cfg.nref;
    const size_t refine = 2 * rf_cap * g.nx * nxh * sizeof(double2) + rf_cap * (nxh + 3) * sizeof(float) + (size_t)g.nx * sizeof(double2);
    const size_t tables = ((size_t)g.LBP * (8 + 16 + 2) + (size_t)g.lcirc * 4 + (size_t)g.nx * g.nx + (size_t)g.maxrin * 8 + (1 << 16)) * sizeof(float);
    // the particle-resident kernels need one candidate record per particle-offset and their B stream; the spectra panels and
    // per-tile candidates of the two-kernel path are allocated on first use only (never, when the resident kernel runs)
    size_t bf = 0;
    const bool resident = resident_expected(g, cfg, generic, &bf);
    const size_t fcand = (size_t)chunk * g.nshift_pad + 8;
    const size_t search_ws = resident ? bf * (sizeof(float) + sizeof(int)) + fcand * sizeof(CandT)      // B stream + its gather table
                                      : w.a_floats * sizeof(float) + (w.cand_recs + fcand) * sizeof(CandT);
    // rot_shift2D + class sums: member lists, per-run partial sums and cos / sin of a batch (transform_sum_kernel), or -- images
    // that do not fit the LDS -- the aligned images of a chunk, their member lists and 16 runs of partial sums
    const size_t npix = (size_t)g.nx * g.nx, nseg = 2 * (size_t)cfg.nref;
    const bool xs = (size_t)(g.nx + 2) * ((g.nx + 2) | 1) * sizeof(float) <= RA_XS_LDS_MAX && g.nx <= 1024;
    const size_t sums_ws = xs ? nseg * chunk * sizeof(int) + std::min<size_t>(512, (1024 + nseg - 1) / nseg) * nseg * npix * sizeof(float) + (size_t)chunk * sizeof(float2)
                              : w.alscratch_floats * sizeof(float) + nseg * chunk * sizeof(int) + 16 * nseg * npix * sizeof(float);
    // sub-bin refinement (ralign_exact.h): exact reference spectra, the list of flagged particles, global ring buffers of large boxes
    const size_t lds_ref = (size_t)std::max(2 * g.lcirc, g.lcirc + 2 * g.maxrin) * sizeof(float) + RA_EXACT_TABLE_BYTES(g.maxrin);
    const size_t exact_ws = (size_t)cfg.nref * g.lcirc * sizeof(float) + (size_t)chunk * sizeof(RefineRec) +
                            (lds_ref > 160 * 1024 - 1024 ? (size_t)std::max(256, cfg.nref) * 2 * g.lcirc * sizeof(float) : 0);
    w.bytes = (w.refspec_floats + w.b_floats + 2) * sizeof(float) + search_ws + sums_ws + exact_ws +
              (resident ? 0 : w.zscr_recs * sizeof(float2)) + refine + tables;
    // every hipMalloc is rounded up to the allocator's granule; ~40 small tables and buffers
    w.bytes += (size_t)48 * (2 << 20);
    return w;
}

extern "C" const char *ra_last_error(void) { return g_last_error.c_str(); }

typedef void (*ccf_fn)(DevGeom, const float *, const float *, int, int, int, CandT *);
static ccf_fn select_ccf(int maxrin);
static ccf_fn select_ccf(int maxrin)
{
    switch (maxrin) {
    case 256: return ccf_kernel<256>;
    case 128: return ccf_kernel<128>;
    case 64: return ccf_kernel<64>;
    case 32: return ccf_kernel<32>;
    default: return nullptr;
    }
}

static fused_fn select_fused(int maxrin, int nref, int nzr, int sbuf, bool pack, bool crop)
{
    if (nref > RF_MAXREF) return nullptr;
    const int nrp = (nref + 1) / 2;
    const bool one = nzr == 1;      // one store / inverse-FFT round per pass
    if (crop) {                     // the image in LDS is a crop (tcrop_wanted): run-time ring-buffer stride only
        if (maxrin != 256 || (pack && !one)) return nullptr;
        switch ((nrp + 1) / 2) {
        case 1: return pack ? search_fused_kernel<256, 1, true, 0, true, true> : one ? search_fused_kernel<256, 1, true, 0, false, true> : search_fused_kernel<256, 1, false, 0, false, true>;
        case 2: return pack ? search_fused_kernel<256, 2, true, 0, true, true> : one ? search_fused_kernel<256, 2, true, 0, false, true> : search_fused_kernel<256, 2, false, 0, false, true>;
        case 3: return pack ? search_fused_kernel<256, 3, true, 0, true, true> : one ? search_fused_kernel<256, 3, true, 0, false, true> : search_fused_kernel<256, 3, false, 0, false, true>;
        default: return pack ? search_fused_kernel<256, 4, true, 0, true, true> : one ? search_fused_kernel<256, 4, true, 0, false, true> : search_fused_kernel<256, 4, false, 0, false, true>;
        }
    }
    if (pack) {                     // dense offset stream over the workgroup's particles (maxrin 256, one round per pass)
        if (maxrin != 256 || !one) return nullptr;
        if (sbuf == RF_SBUF_FIXED)
            switch ((nrp + 1) / 2) {
            case 1: return search_fused_kernel<256, 1, true, RF_SBUF_FIXED, true>;
            case 2: return search_fused_kernel<256, 2, true, RF_SBUF_FIXED, true>;
            case 3: return search_fused_kernel<256, 3, true, RF_SBUF_FIXED, true>;
            default: return search_fused_kernel<256, 4, true, RF_SBUF_FIXED, true>;
            }
        switch ((nrp + 1) / 2) {
        case 1: return search_fused_kernel<256, 1, true, 0, true>;
        case 2: return search_fused_kernel<256, 2, true, 0, true>;
        case 3: return search_fused_kernel<256, 3, true, 0, true>;
        default: return search_fused_kernel<256, 4, true, 0, true>;
        }
    }
    if (maxrin == 256) {
        if (one && sbuf == RF_SBUF_FIXED)      // compile-time ring-buffer stride
            switch ((nrp + 1) / 2) {
            case 1: return search_fused_kernel<256, 1, true, RF_SBUF_FIXED>;
            case 2: return search_fused_kernel<256, 2, true, RF_SBUF_FIXED>;
            case 3: return search_fused_kernel<256, 3, true, RF_SBUF_FIXED>;
            default: return search_fused_kernel<256, 4, true, RF_SBUF_FIXED>;
            }
        switch ((nrp + 1) / 2) {       // 2 waves per 16-bin group
        case 1: return one ? search_fused_kernel<256, 1, true, 0> : search_fused_kernel<256, 1, false, 0>;
        case 2: return one ? search_fused_kernel<256, 2, true, 0> : search_fused_kernel<256, 2, false, 0>;
        case 3: return one ? search_fused_kernel<256, 3, true, 0> : search_fused_kernel<256, 3, false, 0>;
        default: return one ? search_fused_kernel<256, 4, true, 0> : search_fused_kernel<256, 4, false, 0>;
        }
    }
    if (maxrin == 128) {               // 4 waves per group
        if ((nrp + 3) / 4 == 1) return one ? search_fused_kernel<128, 1, true, 0> : search_fused_kernel<128, 1, false, 0>;
        return one ? search_fused_kernel<128, 2, true, 0> : search_fused_kernel<128, 2, false, 0>;
    }
    return nullptr;
}

static fused_fn select_tiled(int nh, int sbuf)
{
    if (sbuf == RF_SBUF_FIXED)
        switch (nh) {
        case 4: return search_tiled_kernel<256, 4, RF_SBUF_FIXED>;
        case 5: return search_tiled_kernel<256, 5, RF_SBUF_FIXED>;
        default: return nullptr;
        }
    switch (nh) {
    case 4: return search_tiled_kernel<256, 4, 0>;
    case 5: return search_tiled_kernel<256, 5, 0>;
    default: return nullptr;
    }
}

template <typename T> static int grow_upload(ra_engine *e, T **dptr, size_t *cap, const std::vector<T> &h)
{
    if (h.size() > *cap || !*dptr) {
        void *d = nullptr;
        const size_t n = std::max<size_t>(h.size(), 1);
        RA_HIP(hipMalloc(&d, n * sizeof(T)));
        e->owned.push_back(d);
        *dptr = (T *)d; *cap = n;
    }
    if (!h.empty()) RA_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return RA_OK;
}

// dense offset stream (search_fused_kernel's PACK): pays when the last pass of a particle would carry padding offsets
// (RALIGN_PACK=0: off)
static bool pack_ok(const ra_engine *e)
{
    if (!e->fused || e->tiled || (getenv("RALIGN_PACK") && atoi(getenv("RALIGN_PACK")) == 0)) return false;
    if (e->geo.nshift % 4 == 0 || e->geo.nshift < 4) return false;      // (a pass holds the offsets of at most two particles)
    return select_fused(e->geo.maxrin, e->cfg.nref, e->fplan.f.nzr, e->dg.sbuf, true, e->tcrop) != nullptr;
}

// plan of the particle-resident search kernel (ralign_fused.h) and its tables.  It covers every search window of a
// geometry the LDS-resident kernels cover, up to RF_MAXREF references; everything else keeps the two-kernel path.
static int setup_fused(ra_engine *e)
{
    e->fused = false; e->tiled = false;
    e->fplan.f.on = 0;
    if (e->generic && !e->tcrop) return RA_OK;
    if (getenv("RALIGN_FUSED") && atoi(getenv("RALIGN_FUSED")) == 0) return RA_OK;
    const Geometry &g = e->geo;
    FusedPlanHost &fp = e->fplan;
    const bool force_tiled = getenv("RALIGN_TILED") && atoi(getenv("RALIGN_TILED")) != 0, no_tiled = getenv("RALIGN_TILED") && !force_tiled;
    const bool want_tiled = e->tcrop ? (!no_tiled && (e->cfg.nref >= RT_MINREF || force_tiled)) : tiled_wanted(e);
    auto plan_tiled = [&]() { return build_tiled_plan(g, e->cfg.nref, e->dg.sbuf, e->lds_polar / sizeof(float), fp) && select_tiled(fp.f.nh, e->dg.sbuf); };
    auto plan_fused = [&]() { return select_fused(g.maxrin, e->cfg.nref, 1, 0, false, e->tcrop) && build_fused_plan(g, e->cfg.nref, e->dg.sbuf, e->lds_polar / sizeof(float), fp); };
    if (want_tiled && plan_tiled()) e->tiled = true;
    else if (plan_fused()) e->tiled = false;
    else if (e->tcrop && !no_tiled && plan_tiled()) e->tiled = true;
    else if (e->tcrop) {
        g_last_error = "particle-resident search over a crop: the plan tcrop_wanted promised does not fit";
        return RA_ERR_STATE;
    } else return RA_OK;
    if (e->tcrop) {
        FusedGeom tcr{};
        crop_plan(g, tcr);
        fp.f.s_crop = tcr.s_crop; fp.f.s_cropm = tcr.s_cropm;
    }
    int rc;
    if ((rc = grow_upload(e, &e->d_fbsrc, &e->f_cap_b, fp.bsrc))) return rc;
    if (!e->d_Bf && (rc = dev_alloc(e, &e->d_Bf, (size_t)fp.f.b_floats + 256, true))) return rc;
    if (!e->d_gcdc && (rc = dev_alloc(e, &e->d_gcdc, (size_t)e->cfg.nref, true))) return rc;
    fp.f.bsrc = e->d_fbsrc; fp.f.cdc_w = e->d_gcdc;
    for (int pk = 0; pk < (e->tiled ? 1 : 2); pk++) {
        const fused_fn fk = e->tiled ? select_tiled(fp.f.nh, e->dg.sbuf) : select_fused(g.maxrin, e->cfg.nref, fp.f.nzr, e->dg.sbuf, pk != 0, e->tcrop);
        if (!fk) continue;
        hipError_t he = hipFuncSetAttribute((const void *)fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
        if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(fused): ") + hipGetErrorString(he); return RA_ERR_HIP; }
    }
    if (getenv("RALIGN_INFO")) fprintf(stderr, "libralign_hip: %s plan: %zu bytes of LDS (polar part %zu), sbuf %d, pst %d, nzr %d, rz %d\n", e->tiled ? "tiled" : "fused", fp.lds_bytes, e->lds_polar, e->dg.sbuf, e->dg.pst, fp.f.nzr, fp.f.rz);
    e->fused = true;
    return RA_OK;
}

typedef void (*solo_fn)(DevGeom, FusedGeom, const float *, const float *, int, const float *, int, CandT *, float *);
static solo_fn select_solo(int maxrin, int nh, int ntile)
{
    if (maxrin != 512) return nullptr;
    if (ntile > 1)       // 3 .. 5 pairs per tile (build_solo_plan)
        return nh == 3 ? search_solo_kernel<512, 3, false> : nh == 4 ? search_solo_kernel<512, 4, false> : nh == 5 ? search_solo_kernel<512, 5, false> : nullptr;
    switch (nh) {
    case 1: return search_solo_kernel<512, 1, true>;
    case 2: return search_solo_kernel<512, 2, true>;
    case 3: return search_solo_kernel<512, 3, true>;
    case 4: return search_solo_kernel<512, 4, true>;
    case 5: return search_solo_kernel<512, 5, true>;
    default: return nullptr;
    }
}

// NQT: the A slice of a wave holds 14 ring quads when no group of rings has more (ou <= 56), 16 otherwise: the slice lives through
// the second offset's ring jobs and every register counts (nb00: 130 k -> 148 k particles/s with the narrower slice)
template <int NQT>
static solo_fn select_duo_t(int nh)
{
    switch (nh) {
    case 1: return search_duo_kernel<512, 1, NQT>;
    case 2: return search_duo_kernel<512, 2, NQT>;
    case 3: return search_duo_kernel<512, 3, NQT>;
    case 4: return search_duo_kernel<512, 4, NQT>;
    default: return nullptr;
    }
}

static int plan_nqmax(const FusedGeom &f)
{
    int m = 0;
    for (int i = 0; i < 16; i++) m = std::max(m, f.grp_nq[i]);
    return m;
}

static solo_fn select_duo(int maxrin, int nh, int nqmax)
{
    if (maxrin != 512) return nullptr;
    const bool q14 = nqmax <= 14 && !(RA_EXP_ENV("RALIGN_DUO_NQT") && ra_atoi(RA_EXP_ENV("RALIGN_DUO_NQT")) == 16);
    return q14 ? select_duo_t<14>(nh) : select_duo_t<16>(nh);
}

static solo_fn select_pair(int maxrin, int nhw)
{
    if (maxrin != 256) return nullptr;
    switch (nhw) {
    case 1: return search_pair_kernel<256, 1>;
    case 2: return search_pair_kernel<256, 2>;
    case 3: return search_pair_kernel<256, 3>;
    default: return nullptr;
    }
}

// plan of search_solo_kernel (ralign_solo.h) for an engine of the size-generic class whose rings end at 512 samples; the generic
// kernels stay available underneath (reference preparation, RALIGN_SOLO=0, geometries whose image and one ring buffer exceed
// the LDS)
static int setup_solo(ra_engine *e)
{
    e->solo = false; e->duo = false; e->pair = false;
    const Geometry &g = e->geo;
    FusedPlanHost &fp = e->fplan;
    if (e->fused) return RA_OK;          // tcrop: search_tiled_kernel over a crop of the image took this engine of the generic class
    if (pair_wanted(e)) {
        // maxrin 256 in a box too large for four ring buffers: two offsets per pass in two (ralign_pair.h)
        if (!build_pair_plan(g, e->cfg.nref, e->dg.n_qtab, e->dg.n_inst, e->dg.n_job, fp)) { fp.f.on = 0; return RA_OK; }
        const solo_fn fk = select_pair(g.maxrin, fp.f.nrpw);
        if (!fk) { fp.f.on = 0; return RA_OK; }
        int rc;
        if ((rc = grow_upload(e, &e->d_fbsrc, &e->f_cap_b, fp.bsrc))) return rc;
        if (!e->d_Bf && (rc = dev_alloc(e, &e->d_Bf, (size_t)fp.f.b_floats + 256, true))) return rc;
        if (!e->d_gcdc && (rc = dev_alloc(e, &e->d_gcdc, (size_t)e->cfg.nref, true))) return rc;
        fp.f.bsrc = e->d_fbsrc; fp.f.cdc_w = e->d_gcdc;
        hipError_t he = hipFuncSetAttribute((const void *)fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)search_pair_kernel<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
        if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(pair): ") + hipGetErrorString(he); return RA_ERR_HIP; }
        if (getenv("RALIGN_INFO")) fprintf(stderr, "libralign_hip: pair plan: %zu bytes of LDS, image %d x %d, 2 ring buffers of %d floats, %d jobs, %d tiles of 2 x %d reference pairs\n",
                                           fp.lds_bytes, fp.f.s_rows, fp.f.s_pst, fp.f.s_sbuf, e->dg.n_job, fp.f.ntile, fp.f.nrpw);
        e->solo = true; e->pair = true;
        return RA_OK;
    }
    if (!solo_wanted(e)) return RA_OK;
    e->duo = duo_wanted(e) && build_duo_plan(g, e->cfg.nref, e->dg.n_qtab, e->dg.n_inst, e->dg.n_job + e->dg.n_job_b, fp) && select_duo(g.maxrin, fp.f.nh, plan_nqmax(fp.f));
    if (!e->duo && !build_solo_plan(g, e->cfg.nref, e->dg.n_qtab, e->dg.n_inst, e->dg.n_job + e->dg.n_job_b, fp)) { fp.f.on = 0; return RA_OK; }
    const solo_fn fk = e->duo ? select_duo(g.maxrin, fp.f.nh, plan_nqmax(fp.f)) : select_solo(g.maxrin, fp.f.nh, fp.f.ntile);
    if (!fk) { fp.f.on = 0; return RA_OK; }
    int rc;
    if ((rc = grow_upload(e, &e->d_fbsrc, &e->f_cap_b, fp.bsrc))) return rc;
    if (!e->d_Bf && (rc = dev_alloc(e, &e->d_Bf, (size_t)fp.f.b_floats + 256, true))) return rc;
    if (!e->d_gcdc && (rc = dev_alloc(e, &e->d_gcdc, (size_t)e->cfg.nref, true))) return rc;
    fp.f.bsrc = e->d_fbsrc; fp.f.cdc_w = e->d_gcdc;
    hipError_t he = hipFuncSetAttribute((const void *)fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
    if (he == hipSuccess)          // ra_debug_spectra: the polar stage through the solo kernel's debug path
        he = hipFuncSetAttribute((const void *)search_solo_kernel<512, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.lds_bytes);
    if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(solo): ") + hipGetErrorString(he); return RA_ERR_HIP; }
    if (getenv("RALIGN_INFO")) fprintf(stderr, "libralign_hip: %s plan:", e->duo ? "duo" : "solo");
    if (getenv("RALIGN_INFO")) fprintf(stderr, " %zu bytes of LDS, image %d x %d, ring buffer %d floats, %d jobs, %d tiles of %d reference pairs\n",
                                       fp.lds_bytes, fp.f.s_rows, fp.f.s_pst, fp.f.s_sbuf, e->dg.n_job, fp.f.ntile, fp.f.nh);
    e->solo = true;
    return RA_OK;
}

// plan and tables of polar_zone_kernel (ralign_zone.h) for an engine that runs the size-generic polar stage: ring zones whose
// annulus (for every search offset of the window) fits the LDS.  RALIGN_ZONES=0: the global-tap kernel.  Quadri sampling (six
// taps, periodic) stays with the global-tap kernel.
static int setup_zones(ra_engine *e)
{
    e->zones = false;
    if (!e->generic || e->fused || e->solo || e->dg.interp != RA_INTERP_BILINEAR) return RA_OK;
    if (getenv("RALIGN_ZONES") && atoi(getenv("RALIGN_ZONES")) == 0) return RA_OK;
    const Geometry &g = e->geo;
    const int S = (int)std::ceil(std::max(g.nkx, g.nky) * g.step - 1e-6);
    if (!build_zone_plan(g, S, RA_ZONE_NW, e->dg.n_qtab, e->zplan)) return RA_OK;
    ZonePlanHost &zp = e->zplan;
    if (const char *ev = RA_EXP_ENV("RALIGN_ZONE_CHUNKS")) { const int v = ra_atoi(ev); if (v >= 1 && v <= 64) zp.nchunk = v; }
    zp.nchunk = std::min(zp.nchunk, std::max(1, g.nshift));
    int rc;
    if ((rc = grow_upload(e, &e->d_zone_rows, &e->zone_cap_rows, zp.rowtab)) || (rc = grow_upload(e, &e->d_zone_pix, &e->zone_cap_pix, zp.pixtab)) ||
        (rc = grow_upload(e, &e->d_zone_desc, &e->zone_cap_zones, zp.zones))) return rc;
    const size_t need = ((size_t)e->chunk * e->pad_cap + 8) * zp.nquad_total;
    if (need > e->stats_part_cap) {
        if ((rc = dev_grow(e, &e->d_stats_part, need, true))) return rc;
        e->stats_part_cap = need;
    }
    hipError_t he = hipFuncSetAttribute((const void *)polar_zone_kernel<RA_ZONE_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)zp.lds_bytes);
    if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(zones): ") + hipGetErrorString(he); return RA_ERR_HIP; }
    e->zdev.zones = e->d_zone_desc; e->zdev.rowtab = e->d_zone_rows; e->zdev.pixtab = e->d_zone_pix;
    e->zdev.nzone = (int)zp.zones.size(); e->zdev.nchunk = zp.nchunk; e->zdev.nquad_total = zp.nquad_total; e->zdev.max_rows = zp.max_rows;
    if (getenv("RALIGN_INFO")) {
        fprintf(stderr, "libralign_hip: zone plan: %d zones, %zu bytes of LDS, %d offset chunks, S = %d:", e->zdev.nzone, zp.lds_bytes, zp.nchunk, S);
        for (const ZoneDesc &z : zp.zones) fprintf(stderr, " [rings %d..%d: %d px, %d rows]", z.ring0, std::min(z.ring0 + 4 * z.nquad, g.nring) - 1, z.npix, z.nrow);
        fprintf(stderr, "\n");
    }
    e->zones = true;
    return RA_OK;
}

// Live-offset lists (round 6): the size-generic kernels work on the IN-WINDOW search offsets of every particle only -- what
// Util.multiref_polar_ali_2d / ormq loop over (search_range) -- instead of computing all of them and masking in finalize_kernel.  A tight
// box (configs[4]: ou + xr = 125 of 127) leaves a particle that sits 5 pixels off centre 8 of its 11 offsets per axis: 19 % less work
// in all three stages once the states have moved.  The entries of a chunk then depend on its states: live_scan_kernel lays the ranges
// out on the device, the kernels read the total there, the host launches for the full lists (empty slices return at once).
// RALIGN_LIVE_OFFSETS=0: every offset of the list, masked later (as the particle-resident 90 x 90 kernels do).
static int update_live_mode(ra_engine *e)
{
    e->dg.ent_base = nullptr; e->dg.ent_total = nullptr;
    if (!e->generic || e->fused || e->solo) return RA_OK;
    if (getenv("RALIGN_LIVE_OFFSETS") && atoi(getenv("RALIGN_LIVE_OFFSETS")) == 0) return RA_OK;
    int rc;
    if (!e->d_ent_base && ((rc = dev_alloc(e, &e->d_ent_base, (size_t)e->chunk + 8, true)) || (rc = dev_alloc(e, &e->d_ent_total, 4, true)))) return rc;
    e->dg.ent_base = e->d_ent_base; e->dg.ent_total = e->d_ent_total;
    return RA_OK;
}

// Polar2Dm + Normalize_ring statistics + Frngs of `cn` particles into the A blocks / d_gstats of the size-generic path
static int launch_generic_polar(ra_engine *e, const float *part, const float *st, int cn, float *Abuf)
{
    const Geometry &g = e->geo;
    if (e->dg.ent_base) {
        hipLaunchKernelGGL(live_scan_kernel, dim3(1), dim3(1024), 0, e->stream, e->dg, st, cn, e->d_ent_base, e->d_ent_total);
        RA_HIP(hipGetLastError());
    }
    if (e->zones) {
        hipLaunchKernelGGL(polar_zone_kernel<RA_ZONE_NW>, dim3((unsigned)cn * e->zdev.nzone * e->zdev.nchunk), dim3(64 * RA_ZONE_NW), e->zplan.lds_bytes, e->stream,
                           e->dg, e->zdev, part, st, cn, Abuf, e->d_stats_part);
        RA_HIP(hipGetLastError());
        const int nent = cn * e->dg.ent_stride;
        hipLaunchKernelGGL(polar_stats_kernel, dim3((nent + 255) / 256), dim3(256), 0, e->stream, e->dg, (const float2 *)e->d_stats_part, nent,
                           e->zdev.nquad_total, e->d_gstats);
    } else
        hipLaunchKernelGGL(gpolar_kernel(e, false), dim3((unsigned)(((long long)cn * e->dg.ent_stride + 3) / 4)), dim3(RA_GEN_THREADS), e->lds_gpolar, e->stream, e->dg,
                           part, st, cn, Abuf, e->d_gstats);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

// tables and buffers of the sub-bin angle refinement (ralign_exact.h): twiddles (float) of the double-precision cos / sin for
// every power-of-two length, as fftr_q's tables; exact reference spectra; the list of flagged particles of a chunk
static int setup_refine(ra_engine *e)
{
    const Geometry &g = e->geo;
    e->refine_ok = false;
    if (getenv("RALIGN_REFINE")) e->refine_thr = (float)atof(getenv("RALIGN_REFINE"));
    // ring buffers (2 lcirc floats) + the twiddles and samples of the f64 CCF (RA_EXACT_TABLE_BYTES)
    e->lds_refine = (size_t)((std::max(2 * g.lcirc, g.lcirc + 2 * g.maxrin) + 3) & ~3) * sizeof(float) + RA_EXACT_TABLE_BYTES(g.maxrin);
    if (g.lcirc & 1) return RA_OK;
    // large boxes: one offset's rings exceed the LDS (271 KB at 256 x 256 / ou = 120) -- the same kernels with the ring buffers
    // in global scratch
    e->refine_gm = e->lds_refine > 160 * 1024 - 1024;          // (the kernel's static LDS: reduction scratch and the tie candidates, ~0.7 KB)
    e->refine_grid = e->refine_gm ? 256 : 2048;
    if (e->refine_gm) e->lds_refine = RA_EXACT_TABLE_BYTES(g.maxrin);
    std::vector<float> tw;
    std::vector<int> off(32, 0);
    for (int l = 1; (1 << l) <= g.maxrin; l++) {
        const int n = 1 << l, h = n / 2;
        off[l] = (int)tw.size();
        for (int k = 0; k < h; k++) {
            const double a = -2.0 * M_PI * k / n;
            tw.push_back((float)cos(a)); tw.push_back((float)sin(a));
        }
    }
    const float *dtw = nullptr; const int *doff = nullptr;
    int rc;
    if ((rc = upload(e, tw, &dtw)) || (rc = upload(e, off, &doff))) return rc;
    e->d_twx = (float *)dtw; e->d_twxoff = (int *)doff;
    if ((rc = dev_alloc(e, &e->d_refx, (size_t)e->cfg.nref * g.lcirc, true)) ||
        (rc = dev_alloc(e, &e->d_rlist, (size_t)e->chunk, false)) || (e->rlist_cap = e->chunk, 0) ||
        (rc = dev_alloc(e, &e->d_rcount, 1, true)) ||
        (rc = dev_alloc(e, &e->d_refhash, (size_t)e->cfg.nref, true)) || (rc = dev_alloc(e, &e->d_refdup, (size_t)2 * e->cfg.nref, true))) return rc;
    if (e->refine_gm) {
        if ((rc = dev_alloc(e, &e->d_rscratch, (size_t)std::max(e->refine_grid, e->cfg.nref) * 2 * g.lcirc, false))) return rc;
        // the f64 twiddles and samples stay in LDS (24 maxrin bytes: 96 KB at maxrin 4096, beyond the 64 KB a kernel gets unasked)
        hipError_t he = hipFuncSetAttribute((const void *)refine_winner_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_refine);
        if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(refine, global ring buffers): ") + hipGetErrorString(he); return RA_ERR_HIP; }
    } else {
        hipError_t he = hipFuncSetAttribute((const void *)refine_winner_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_refine);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)refspec_exact_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_refine);
        if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute(refine): ") + hipGetErrorString(he); return RA_ERR_HIP; }
    }
    e->refine_ok = true;
    return RA_OK;
}

// finalize + (flagged particles) refine of one chunk: `cand` records with `nrtile` per particle-offset.
// Chunked paths (kernel pair, size-generic kernels) call it with deferred = 1 for every chunk -- the flagged particles of the whole
// call collect in ONE list, indexed from the call's first particle (p_base = the chunk's offset; st / res / part are the chunk's) --
// and once more with deferred = 2 behind the last chunk (st / res / part the call's, cn its particle count): refine_winner_kernel
// handles one image per workgroup and a chunk flags ~10 of its 330 particles at configs[4], so a launch per chunk left the GPU to ten
// workgroups for 0.94 ms, 2.6 % of the iteration; one launch per call fills it.
static int finalize_and_refine(ra_engine *e, const CandT *cand, int nrtile, int cn, float *st, ra_result *res, const float *part,
                               const float *refx, const int *cls, int deferred = 0, int p_base = 0)
{
    const bool refine = e->refine_ok && e->refine_thr != 0.f && refx;
    if (refine && deferred == 0) RA_HIP(hipMemsetAsync(e->d_rcount, 0, sizeof(int), e->stream));
    if (deferred != 2) {
    if ((size_t)e->geo.nshift * nrtile >= 256)          // many records per particle: one wave each
        hipLaunchKernelGGL(finalize_wave_kernel, dim3(cn), dim3(64), 0, e->stream, e->dg, cand, nrtile, cn, st, res,
                           refine ? e->d_rlist : (RefineRec *)nullptr, e->d_rcount, e->refine_thr, p_base);
    else
        hipLaunchKernelGGL(finalize_kernel, dim3((cn + 127) / 128), dim3(128), 0, e->stream, e->dg, cand, nrtile, cn, st, res, (const float *)e->d_cs,
                           refine ? e->d_rlist : (RefineRec *)nullptr, e->d_rcount, e->refine_thr, p_base);
    RA_HIP(hipGetLastError());
    }
    if (refine && deferred != 1) {
        const int grid = std::min(cn, e->refine_grid);
        if (e->refine_gm)
            hipLaunchKernelGGL(refine_winner_kernel<true>, dim3(grid), dim3(RA_EXACT_THREADS), e->lds_refine, e->stream, e->dg, (const int *)e->d_numr,
                               (const float *)e->d_twx, (const int *)e->d_twxoff, part, refx, (const RefineRec *)e->d_rlist, (const int *)e->d_rcount,
                               res, cls, st, e->d_rscratch);
        else
            hipLaunchKernelGGL(refine_winner_kernel<false>, dim3(grid), dim3(RA_EXACT_THREADS), e->lds_refine, e->stream, e->dg, (const int *)e->d_numr,
                               (const float *)e->d_twx, (const int *)e->d_twxoff, part, refx, (const RefineRec *)e->d_rlist, (const int *)e->d_rcount,
                               res, cls, st, (float *)nullptr);
        RA_HIP(hipGetLastError());
    }
    return RA_OK;
}

// One launch of a particle-resident kernel takes up to RA_RESIDENT_BATCH particles (cfg.chunk > 0: that many): every launch
// ends with a tail in which the CUs run dry one by one, ~1 % of a 7 000-particle launch (measured: 7 launches 30.9 ms, 5 launches
// 30.6 ms per 50 000 particles); its candidate records (2 KB per particle at 49 offsets) and the refine list grow on demand.
#define RA_RESIDENT_BATCH 131072
static int resident_batch(const ra_engine *e, int n)
{
    return std::max(1, std::min(n, e->cfg.chunk > 0 ? e->chunk : RA_RESIDENT_BATCH));
}
static int ensure_resident_ws(ra_engine *e, int cn)
{
    int rc;
    if (cn > e->fcand_cap) {
        if ((rc = dev_grow(e, &e->d_fcand, (size_t)cn * e->pad_cap + 8, true))) return rc;
        e->fcand_cap = cn;
    }
    if (e->refine_ok && cn > e->rlist_cap) {
        if ((rc = dev_grow(e, &e->d_rlist, (size_t)cn, false))) return rc;
        e->rlist_cap = cn;
    }
    return RA_OK;
}

// spectra workspace of the two-kernel path, allocated on first use (the fused kernel does not need it)
static int ensure_unfused_ws(ra_engine *e)
{
    if (e->unfused_ws) return RA_OK;
    int rc;
    if ((rc = dev_alloc(e, &e->d_A, e->wp.a_floats, true)) || (rc = dev_alloc(e, &e->d_cand, e->wp.cand_recs, true))) return rc;
    if (e->generic && (rc = dev_alloc(e, &e->d_zscr, e->wp.zscr_recs, false))) return rc;
    e->unfused_ws = true;
    return RA_OK;
}

// the LDS-resident kernels cover <= 48 rings of 8..256 samples and images whose padded copy plus four ring buffers
// fit one CU's LDS; everything else runs the size-generic kernels
static ccf_fn select_ccf(int maxrin);
static bool fits_specialised_kernels(const Geometry &g0, const ra_config &cfg)
{
    bool fast = g0.nring <= 4 * RA_CCF_MAXNS && select_ccf(g0.maxrin) != nullptr && g0.numr[2] >= 8;
    const int bd0 = (int)std::ceil(std::max(cfg.xrng, cfg.yrng)) + 2, pst0 = g0.nx + 2 * bd0;
    const int sbuf0 = (g0.lring + 31) / 32 * 32 + 8;
    if ((size_t)(pst0 * pst0 + 4 * sbuf0 + 2 * g0.maxrin + 4096) * sizeof(float) > 160 * 1024) fast = false;
    int ncls = 0, prev = -1;
    for (int k = 0; k < g0.nbins; k++) {
        const int ns = (g0.bin_offp[k + 1] - g0.bin_offp[k]) / 4;
        if (ns != prev) { ncls++; prev = ns; }
    }
    if (ncls > 8) fast = false;
    if (generic_forced()) fast = false;
    return fast;
}

static int create_engine(ra_engine **out, const ra_config *cfg, const ra_options *opt, bool allow_tcrop);
extern "C" int ra_create(ra_engine **out, const ra_config *cfg) { return ra_create_ex(out, cfg, nullptr); }
extern "C" int ra_create_ex(ra_engine **out, const ra_config *cfg, const ra_options *opt)
{
    if (opt && (opt->interp != RA_INTERP_BILINEAR && opt->interp != RA_INTERP_QUADRI)) { g_last_error = "bad interpolation"; return RA_ERR_ARG; }
    if (opt && (opt->normalize_ring < -1 || opt->normalize_ring > 1)) { g_last_error = "normalize_ring is -1 (by mode), 0 or 1"; return RA_ERR_ARG; }
    ForceGenericScope scope(opt && opt->interp == RA_INTERP_QUADRI);
    int rc = create_engine(out, cfg, opt, true);
    // the plan of the four-offset kernels over a crop of the image is promised by tcrop_wanted on an estimate of the tables; should
    // the real tables miss it, the engine is planned again without that path (pair or generic kernels) instead of failing
    if (rc == RA_ERR_STATE) rc = create_engine(out, cfg, opt, false);
    // More than 16 references in a box whose image fits the LDS but leaves no room for the tiled kernel's plan beside it (100 - 128
    // pixels at ou = 21 .. 32) used to fall to the round-1 kernel pair (3.7 MB of HBM per particle): planned in the size-generic
    // class instead, the search runs search_tiled_kernel over a crop of the image.  Kept only when that plan exists.
    if (rc == RA_OK && *out && !(*out)->generic && !(*out)->fused && cfg->nref > RF_MAXREF && (*out)->geo.maxrin == 256 && !generic_forced() &&
        !(getenv("RALIGN_FUSED") && atoi(getenv("RALIGN_FUSED")) == 0)) {          // (RALIGN_FUSED=0 asks for the kernel pair)
        ra_engine *alt = nullptr;
        g_generic_class = true;
        const int rc2 = create_engine(&alt, cfg, opt, true);
        g_generic_class = false;
        if (rc2 == RA_OK && alt && (alt->fused || alt->solo)) { ra_destroy(*out); *out = alt; }
        else if (alt) ra_destroy(alt);
    }
    return rc;
}
static int create_engine(ra_engine **out, const ra_config *cfg, const ra_options *opt, bool allow_tcrop)
{
    if (!out || !cfg) { g_last_error = "null argument"; return RA_ERR_ARG; }
    *out = nullptr;
    if (cfg->nx < 8 || cfg->nref < 1 || cfg->nref > 65535) { g_last_error = "bad nx/nref"; return RA_ERR_ARG; }
    // "Shift or radius is too large - particle crosses image boundary" (test_mref_gpu_align.py:314)
    if (cfg->last_ring + std::max(cfg->xrng, cfg->yrng) > (float)((cfg->nx - 1) / 2)) {
        g_last_error = "shift or radius too large: particle crosses image boundary";
        return RA_ERR_ARG;
    }
    if (cfg->mode != RA_MODE_MREF && cfg->mode != RA_MODE_REFFREE) { g_last_error = "bad mode"; return RA_ERR_ARG; }
    int ndev = 0;
    RA_HIP(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) { g_last_error = "no such HIP device"; return RA_ERR_HIP; }
    RA_HIP(hipSetDevice(cfg->device));

    ra_engine *e = new ra_engine();
    e->cfg = *cfg;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cu = prop.multiProcessorCount;
        if (getenv("RALIGN_GRID") && atoi(getenv("RALIGN_GRID")) > 0) e->n_cu = atoi(getenv("RALIGN_GRID"));      // experiments: fewer persistent workgroups
    }
    if (!build_rings(e->geo, cfg->nx, cfg->first_ring, cfg->last_ring, cfg->ring_skip > 0 ? cfg->ring_skip : 1) ||
        !build_shifts(e->geo, cfg->xrng, cfg->yrng, cfg->step)) {
        g_last_error = "bad ring / shift geometry";
        delete e;
        return RA_ERR_ARG;
    }
    e->force_generic = g_force_generic;
    e->no_tcrop = !allow_tcrop;
    e->generic = !fits_specialised_kernels(e->geo, *cfg) || g_generic_class;
    e->tcrop = tcrop_wanted(e);
    if (!e->tcrop && allow_tcrop && e->generic && e->geo.maxrin == 256 && !(getenv("RALIGN_TIGHT_RINGS") && atoi(getenv("RALIGN_TIGHT_RINGS")) == 0)) {
        // ou = 37 ... 40: crop + four ring buffers miss the LDS by 4 - 8 KB, of which the 16 padding floats per ring are 10 KB.  With
        // rings 4 floats apart (the in-place real transform needs 2; 4 keeps every ring 16-byte aligned) the four-offset kernels take
        // the class from the pair kernel (RALIGN_TIGHT_RINGS=0: the pair kernel)
        // (RALIGN_TIGHT_RINGS is on / off like every other switch; the distance itself is an experiment of profiling builds,
        // RALIGN_RING_PAD: multiples of 4 only -- 2 would leave the rings 8-byte aligned, a layout no test covers)
        int pad = 4;
        if (const char *rp = RA_EXP_ENV("RALIGN_RING_PAD")) { const int v = atoi(rp); if (v >= 4 && v % 4 == 0 && v < kRingPad) pad = v; }
        Geometry g16 = e->geo, g4;
        if (build_rings(g4, cfg->nx, cfg->first_ring, cfg->last_ring, cfg->ring_skip > 0 ? cfg->ring_skip : 1, pad) &&
            build_shifts(g4, cfg->xrng, cfg->yrng, cfg->step)) {
            e->geo = g4;
            e->tcrop = tcrop_wanted(e);
            if (!e->tcrop) e->geo = g16;
        }
    }
    e->crop_S = (int)std::ceil(std::max(e->geo.nkx, e->geo.nky) * e->geo.step - 1e-6);
    if (e->generic && e->geo.maxrin <= 1024 && !(RA_EXP_ENV("RALIGN_QUAD_ALIGN") && ra_atoi(RA_EXP_ENV("RALIGN_QUAD_ALIGN")) == 0)) align_ring_quads(e->geo);
    if (e->geo.maxrin > 4096) {
        g_last_error = "rings longer than 4096 samples are not supported";
        delete e;
        return RA_ERR_ARG;
    }
    e->nrtile = (cfg->nref + 7) / 8;
    e->dg.rpt = (cfg->nref + e->nrtile - 1) / e->nrtile;      // balanced reference tiles (10 -> 5 + 5)
    int rc = build_device_geometry(e);
    if (rc) { ra_destroy(e); return rc; }
    if (opt) {
        e->dg.interp = opt->interp;
        if (opt->normalize_ring >= 0) e->dg.norm_ring = opt->normalize_ring;
    }

    e->shift_cap = e->geo.nshift; e->pad_cap = e->geo.nshift_pad;
    const Geometry &g = e->geo;
    const int npix_pad = (g.nx * g.nx + 3) & ~3;
    e->lds_polar = (size_t)(((e->dg.pst * e->dg.pst + 3) & ~3) + 4 * e->dg.sbuf + 2 * g.maxrin + 2 * e->dg.n_qtab + 2 + 5 * e->dg.n_inst +
                            4 * e->dg.n_job + 32 + 8 * g.nring) * sizeof(float);
    e->lds_ref = (size_t)(npix_pad + e->dg.sbuf + 8) * sizeof(float);
    e->lds_ccf = (size_t)64 * (2 * (g.maxrin + g.maxrin / 16) + 2) * sizeof(float);
    e->lds_xf = (size_t)npix_pad * sizeof(float);
    const size_t lds_max = 160 * 1024;
    if (!e->generic && (e->lds_polar > lds_max || e->lds_ccf > lds_max)) {
        g_last_error = "image / ring geometry does not fit the 160 KB LDS of one CU";
        ra_destroy(e);
        return RA_ERR_ARG;
    }
    e->xf_generic = e->lds_xf > lds_max;
    // generic contraction: P pairs per inverse-FFT batch, two N-point buffers per pair
    e->g_P = 64;
    while (e->g_P > 1 && (size_t)e->g_P * (g.maxrin + 1) * sizeof(float2) > 66 * 1024) e->g_P >>= 1;      // two workgroups per CU
    e->lds_gccf = ((size_t)e->g_P * (g.maxrin + 1) + g.maxrin) * sizeof(float2);      // pair buffers (in-place transforms) + twiddle table
    e->lds_gpolar = ((size_t)(RA_GEN_THREADS / 64 + 1) * g.maxrin + e->dg.n_qtab) * sizeof(float2);      // per-wave ring buffers + twiddle table + (sinf, cosf) tables
    hipError_t he = hipSuccess;
    if (!e->generic) {
        he = hipFuncSetAttribute((const void *)polar_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_polar);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)ref_polar_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_ref);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)select_ccf(g.maxrin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_ccf);
    } else {
        he = hipFuncSetAttribute((const void *)gpolar_kernel(e, false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_gpolar);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)gpolar_kernel(e, true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_gpolar);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)ccf_generic_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_gccf);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)ccf_generic_kernel<1, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_gccf);
        if (he == hipSuccess) he = hipFuncSetAttribute((const void *)ccf_generic_kernel<2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_gccf);
        if (g.maxrin == 1024) {
            const int lds2 = (int)(((size_t)8 * RA_IFFT3_PSTRIDE + 16 * 16 + 16 * 64) * sizeof(float2));
            if (he == hipSuccess) he = hipFuncSetAttribute((const void *)gccf_ifft_kernel<1, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            if (he == hipSuccess) he = hipFuncSetAttribute((const void *)gccf_ifft_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            if (he == hipSuccess) he = hipFuncSetAttribute((const void *)gccf_ifft_kernel<2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            if (he == hipSuccess) he = hipFuncSetAttribute((const void *)gccf_ifft_kernel<4, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        }
    }
    if (he == hipSuccess && !e->xf_generic) he = hipFuncSetAttribute((const void *)transform_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_xf);
    if (he == hipSuccess) he = hipFuncSetAttribute((const void *)class_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (he != hipSuccess) { g_last_error = std::string("hipFuncSetAttribute: ") + hipGetErrorString(he); ra_destroy(e); return RA_ERR_HIP; }

    // workspace.  The spectra panels and candidate records of the two-kernel path are allocated on first use only
    // (ensure_unfused_ws); the fused kernel needs candidate records alone.
    const WorkspacePlan wp = plan_workspace(g, *cfg, e->generic);
    e->wp = wp;
    e->chunk = wp.chunk;
    if ((rc = dev_alloc(e, &e->d_refspec, wp.refspec_floats, true)) ||
        (rc = dev_alloc(e, &e->d_B, wp.b_floats, true)) ||
        (rc = dev_alloc(e, &e->d_cs, 2, true)) ||
        (rc = dev_alloc(e, &e->d_fcand, (size_t)wp.chunk * g.nshift_pad + 8, true))) {
        ra_destroy(e);
        return rc;
    }
    e->fcand_cap = wp.chunk;
    if (e->generic) {
        e->g_nblk = 512;       // persistent workgroups of ccf_generic_kernel (two per CU)
        if ((rc = dev_alloc(e, &e->d_gstats, (size_t)wp.chunk * g.nshift_pad + 8, true)) ||
            (rc = dev_alloc(e, &e->d_gcdc, (size_t)cfg->nref, true))) { ra_destroy(e); return rc; }
    }
    if ((rc = setup_fused(e))) { ra_destroy(e); return rc; }
    if ((rc = setup_solo(e))) { ra_destroy(e); return rc; }
    if ((rc = setup_zones(e))) { ra_destroy(e); return rc; }
    if ((rc = update_live_mode(e))) { ra_destroy(e); return rc; }
    if ((rc = setup_refine(e))) { ra_destroy(e); return rc; }
    e->atomic_sums = getenv("RALIGN_ATOMIC_SUMS") && atoi(getenv("RALIGN_ATOMIC_SUMS")) != 0;
    *out = e;
    return RA_OK;
}

extern "C" void ra_destroy(ra_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
#ifdef RALIGN_PROFILE_SWITCHES
    if (e->d_timeline && RA_EXP_ENV("RALIGN_TIMELINE")) {      // profiling builds: dump the wave timeline of the last launch
        std::vector<unsigned long long> h(64 * 16 * 16);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(h.data(), e->d_timeline, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *fp = fopen(RA_EXP_ENV("RALIGN_TIMELINE"), "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fp); fclose(fp); }
        }
        (void)hipFree(e->d_timeline);
    }
#endif
    if (e->d_cls_refspec) (void)hipFree(e->d_cls_refspec);
    if (e->d_cls_Bf) (void)hipFree(e->d_cls_Bf);
    if (e->d_cls_refx) (void)hipFree(e->d_cls_refx);
    for (void *p : e->owned) (void)hipFree(p);
    for (auto &pr : e->ev_ccf) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : e->ev_polar) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    delete e;
}

extern "C" int ra_set_stream(ra_engine *e, void *hip_stream)
{
    if (!e) return RA_ERR_ARG;
    e->stream = (hipStream_t)hip_stream;
    return RA_OK;
}

extern "C" int ra_set_mask(ra_engine *e, const float *d_mask)
{
    if (!e || !d_mask) { g_last_error = "null argument"; return RA_ERR_ARG; }
    RA_HIP(hipMemcpyAsync((void *)e->dg.mask, d_mask, (size_t)e->geo.nx * e->geo.nx * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return RA_OK;
}

extern "C" int ra_set_normalize_ring(ra_engine *e, int flag)
{
    if (!e) return RA_ERR_ARG;
    e->dg.norm_ring = flag < 0 ? (e->cfg.mode == RA_MODE_MREF ? 1 : 0) : flag ? 1 : 0;
    return RA_OK;
}
extern "C" int ra_get_options(const ra_engine *e, ra_options *opt)
{
    if (!e || !opt) return RA_ERR_ARG;
    opt->interp = e->dg.interp; opt->normalize_ring = e->dg.norm_ring;
    return RA_OK;
}

extern "C" int ra_set_nomirror(ra_engine *e, int flag)
{
    if (!e) return RA_ERR_ARG;
    e->dg.nomirror = flag ? 1 : 0;
    return RA_OK;
}
extern "C" int ra_set_refine(ra_engine *e, float threshold)
{
    if (!e) return RA_ERR_ARG;
    e->refine_thr = threshold;
    e->refs_ready = false;             // the exact reference spectra are prepared by ra_set_references when the refinement is on
    e->cls_ready = 0;                  // ... and those of the class-resident mode by ra_set_class_references
    return e->refine_ok || threshold == 0.f ? RA_OK : RA_ERR_STATE;
}
// particles the last search launch handed to refine_winner_kernel (flat peaks and float ties); synchronises the stream
extern "C" int ra_last_refine_count(ra_engine *e)
{
    if (!e) return RA_ERR_ARG;
    if (!e->refine_ok || !e->d_rcount) return 0;
    int h = 0;
    if (hipStreamSynchronize(e->stream) != hipSuccess || hipMemcpy(&h, e->d_rcount, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return RA_ERR_HIP;
    return h;
}
extern "C" int ra_search_tiled(const ra_engine *e) { return !e ? RA_ERR_ARG : (e->fused && e->tiled) ? 1 : 0; }
extern "C" int ra_search_path(const ra_engine *e) { return !e ? RA_ERR_ARG : e->solo ? 3 : e->fused ? 1 : e->generic ? 2 : 0; }
extern "C" int ra_search_offsets_per_pass(const ra_engine *e) { return !e ? RA_ERR_ARG : !e->solo ? 0 : (e->duo || e->pair) ? 2 : 1; }
extern "C" int ra_search_skips_offsets(const ra_engine *e) { return !e ? RA_ERR_ARG : (e->solo || e->dg.ent_base) ? 1 : 0; }
extern "C" int ra_num_shifts(const ra_engine *e) { return e ? e->geo.nshift : RA_ERR_ARG; }
extern "C" int ra_maxrin(const ra_engine *e) { return e ? e->geo.maxrin : RA_ERR_ARG; }
extern "C" int ra_lcirc(const ra_engine *e) { return e ? e->geo.lcirc : RA_ERR_ARG; }

extern "C" int ra_reset_shifts(ra_engine *e, float xrng, float yrng, float step)
{
    if (!e) return RA_ERR_ARG;
    ForceGenericScope scope(e->force_generic);
    Geometry g2 = e->geo;
    if (!build_shifts(g2, xrng, yrng, step)) { g_last_error = "bad shift window"; return RA_ERR_ARG; }
    // the reference asserts the offset count does not change (gpu_aln_noref.cu:135); we only
    // require that it does not exceed what ra_create sized
    if (g2.nshift_pad > e->pad_cap || g2.nshift > e->shift_cap) {
        g_last_error = "reset_shifts: more search offsets than the engine was created with";
        return RA_ERR_ARG;
    }
    if ((float)e->cfg.last_ring + std::max(xrng, yrng) > (float)((e->cfg.nx - 1) / 2)) {
        g_last_error = "shift or radius too large: particle crosses image boundary";
        return RA_ERR_ARG;
    }
    // the padded LDS image of the polar stage was sized at ra_create for a border of ceil(max range) + 2 pixels;
    // a wider window (possible at a constant offset count, e.g. xr=1,ts=0.5 -> xr=4,ts=2) would let taps leave it
    if (!e->generic && (int)std::ceil(std::max(xrng, yrng)) + 2 > e->dg.bd) {
        g_last_error = "reset_shifts: search range exceeds the image border the engine was created with";
        return RA_ERR_ARG;
    }
    if (e->tcrop && (int)std::ceil(std::max(g2.nkx, g2.nky) * g2.step - 1e-6) > e->crop_S) {
        g_last_error = "reset_shifts: search range exceeds the image crop the engine was created with";
        return RA_ERR_ARG;
    }
    RA_HIP(hipStreamSynchronize(e->stream));
    RA_HIP(hipMemcpy((void *)e->dg.shift_x, g2.shift_x.data(), g2.nshift * sizeof(float), hipMemcpyHostToDevice));
    RA_HIP(hipMemcpy((void *)e->dg.shift_y, g2.shift_y.data(), g2.nshift * sizeof(float), hipMemcpyHostToDevice));
    e->geo.nkx = g2.nkx; e->geo.nky = g2.nky; e->geo.nshift = g2.nshift; e->geo.step = g2.step;
    e->geo.shift_x = g2.shift_x; e->geo.shift_y = g2.shift_y;
    e->geo.nshift_pad = g2.nshift_pad;
    e->cfg.xrng = xrng; e->cfg.yrng = yrng; e->cfg.step = step;
    e->dg.nkx = g2.nkx; e->dg.nky = g2.nky; e->dg.nshift = g2.nshift; e->dg.nshift_pad = g2.nshift_pad; e->dg.ent_stride = e->generic ? g2.nshift : g2.nshift_pad;
    e->dg.step = step; e->dg.xrng = xrng; e->dg.yrng = yrng;
    int rc = setup_fused(e);
    // the solo / duo / pair kernels keep a crop of the image whose side follows the search range: plan again (a wider range at a
    // constant offset count, e.g. xr = 1, ts = 0.5 -> xr = 4, ts = 2, would otherwise let taps leave the crop)
    // (also when the previous window made the plan fall back to the generic kernels: the new one may fit again)
    if (!rc && e->generic && !e->fused && (e->solo || solo_wanted(e) || pair_wanted(e))) rc = setup_solo(e);
    if (!rc) rc = setup_zones(e);          // the zones' annuli follow the search range
    if (!rc) rc = update_live_mode(e);     // (the plan may have moved between the particle-resident and the size-generic kernels)
    return rc;
}

extern "C" int ra_set_references(ra_engine *e, const float *d_refs)
{
    if (!e || !d_refs) { g_last_error = "null argument"; return RA_ERR_ARG; }
    const Geometry &g = e->geo;
    if (e->generic)
        hipLaunchKernelGGL(gpolar_kernel(e, true), dim3(e->cfg.nref), dim3(RA_GEN_THREADS), e->lds_gpolar, e->stream, e->dg,
                           d_refs, (const float *)nullptr, e->cfg.nref, e->d_refspec, (float2 *)nullptr);
    else
        hipLaunchKernelGGL(ref_polar_fft_kernel, dim3(e->cfg.nref), dim3(256), e->lds_ref, e->stream, e->dg, d_refs,
                           e->cfg.nref, e->d_refspec);
    RA_HIP(hipGetLastError());
    int total = e->nrtile * g.LBP * 16;
    hipLaunchKernelGGL(pack_refs_kernel, dim3((total + 255) / 256), dim3(256), 0, e->stream, e->dg, e->d_refspec,
                       e->cfg.nref, e->nrtile, e->d_B);
    RA_HIP(hipGetLastError());
    if (e->d_Bf) {      // the fused kernel's stream does not depend on the search window: keep it current
        FusedGeom f = e->fplan.f;
        if (f.b_floats > 0 && f.bsrc) {
            hipLaunchKernelGGL(pack_refs_fused_kernel, dim3(std::min(2048, (f.b_floats + 255) / 256)), dim3(256), 0, e->stream, e->dg, f,
                               e->d_refspec, e->cfg.nref, e->d_Bf);
            RA_HIP(hipGetLastError());
        }
    }
    if (e->refine_ok && e->refine_thr != 0.f) {        // the same references with the CPU path's arithmetic, for the sub-bin angle refinement
        if (e->refine_gm)
            hipLaunchKernelGGL(refspec_exact_kernel<true>, dim3(e->cfg.nref), dim3(RA_EXACT_THREADS), 0, e->stream, e->dg, (const int *)e->d_numr,
                               (const float *)e->d_wr, (const float *)e->d_twx, (const int *)e->d_twxoff, d_refs, e->cfg.nref, e->d_refx, e->d_rscratch);
        else
            hipLaunchKernelGGL(refspec_exact_kernel<false>, dim3(e->cfg.nref), dim3(RA_EXACT_THREADS), e->lds_refine, e->stream, e->dg, (const int *)e->d_numr,
                               (const float *)e->d_wr, (const float *)e->d_twx, (const int *)e->d_twxoff, d_refs, e->cfg.nref, e->d_refx, (float *)nullptr);
        RA_HIP(hipGetLastError());
        // copies of a reference inside the stack: their CCFs are equal to the bit and the CPU scan alone decides among them
        hipLaunchKernelGGL(ref_hash_kernel, dim3(e->cfg.nref), dim3(256), 0, e->stream, (const float *)e->d_refx, g.lcirc, e->d_refhash);
        hipLaunchKernelGGL(ref_groups_kernel, dim3(1), dim3(256), 0, e->stream, (const float *)e->d_refx, g.lcirc, e->cfg.nref,
                           (const unsigned long long *)e->d_refhash, e->d_refdup);
        RA_HIP(hipGetLastError());
        e->dg.ref_dup = e->d_refdup;
    } else e->dg.ref_dup = nullptr;
    if (e->generic || e->fused) {
        hipLaunchKernelGGL(ref_dc_weights_kernel, dim3((e->cfg.nref + 63) / 64), dim3(64), 0, e->stream, e->dg, e->d_refspec, e->cfg.nref, e->d_gcdc);
        RA_HIP(hipGetLastError());
    }
    e->refs_ready = true;
    return RA_OK;
}

extern "C" int ra_get_prepared_references(ra_engine *e, float *h_crefim)
{
    if (!e || !h_crefim) return RA_ERR_ARG;
    if (!e->refs_ready) { g_last_error = "ra_set_references has not been called"; return RA_ERR_STATE; }
    float *d_out = nullptr;
    size_t cnt = (size_t)e->cfg.nref * e->geo.lcirc;
    RA_HIP(hipMalloc((void **)&d_out, cnt * sizeof(float)));
    hipLaunchKernelGGL(unpack_refs_kernel, dim3(e->cfg.nref), dim3(256), 0, e->stream, e->dg, e->d_refspec,
                       e->cfg.nref, e->d_ring_off, e->d_numr, e->d_wr, d_out);
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(h_crefim, d_out, cnt * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    RA_HIP(he);
    return RA_OK;
}

static std::pair<hipEvent_t, hipEvent_t> *next_events(std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used)
{
    if (used == pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
        pool.push_back({a, b});
    }
    return &pool[used++];
}

// average-centre correction of ali2d_single_iter: d += R(-alpha) M cs using the previous
// parameters (combine_params2(alpha,sx,sy,mirror, 0,-cs0,-cs1,0) then inverse_transform2)
__global__ void apply_cs_kernel(int n, const float *__restrict__ cs, const ra_result *__restrict__ res,
                                float *__restrict__ state)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double a = (double)res[p].alpha * M_PI / 180.0, c = cos(a), s = sin(a);
    double u = res[p].mirror ? -(double)cs[0] : (double)cs[0], v = (double)cs[1];
    state[2 * p] = (float)((double)state[2 * p] + c * u - s * v);
    state[2 * p + 1] = (float)((double)state[2 * p + 1] + s * u + c * v);
}

// The reference's state round trip (test_mref_gpu_align.py:1024-1026; ali2d_single_iter behind test_reffree_gpu_align.py:844-847):
// the shift a search starts from is rebuilt from the float32 header values (alpha, sx, sy[, mirror]) of the previous
// iteration -- inverse_transform2(alpha, sx, sy) for mref_ali2d, combine_params2(alpha, sx, sy, mirror, 0, -cs0, -cs1, 0)
// then inverse_transform2 for ali2d -- in double, EMAN2 Transform algebra v' = M (R(alpha) v + t), and rounded to float32
// once.  Algebraically d_old + (ix, iy) (what ra_align leaves in d_state); numerically e.g. -6.9999995, which decides
// edge-limited windows (particle_window) the way the reference's loop does.
__device__ __forceinline__ void tf_build_d(double a_deg, double tx, double ty, int m, double T[3][3])
{
    const double a = a_deg * M_PI / 180.0, c = cos(a), s = sin(a), sgn = m ? -1.0 : 1.0;
    T[0][0] = sgn * c; T[0][1] = sgn * s; T[0][2] = sgn * tx;
    T[1][0] = -s; T[1][1] = c; T[1][2] = ty;
    T[2][0] = 0; T[2][1] = 0; T[2][2] = 1;
}
__device__ __forceinline__ void tf_params_d(const double T[3][3], double out[4])
{
#pragma clang fp contract(off)
    const double det = T[0][0] * T[1][1] - T[0][1] * T[1][0];
    const int m = det < 0;
    const double sgn = m ? -1.0 : 1.0, c = sgn * T[0][0], s = sgn * T[0][1];
    double alpha = atan2(s, c) * 180.0 / M_PI;
    alpha = fmod(alpha, 360.0);
    if (alpha < 0) alpha += 360.0;
    if (alpha >= 360.0) alpha -= 360.0;
    out[0] = alpha; out[1] = sgn * T[0][2]; out[2] = T[1][2]; out[3] = m;
}
__global__ void state_from_params_kernel(int n, int mode, const float *__restrict__ cs, const ra_result *__restrict__ res,
                                         float *__restrict__ state)
{
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double A[3][3], B[3][3], C[3][3], I[3][3], prm[4];
    double alpha = (double)res[p].alpha, tx = (double)res[p].sx, ty = (double)res[p].sy;
    if (mode == RA_MODE_REFFREE) {
        tf_build_d(alpha, tx, ty, res[p].mirror, A);
        tf_build_d(0.0, -(double)cs[0], -(double)cs[1], 0, B);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                C[i][j] = 0;
                for (int k = 0; k < 3; k++) C[i][j] += B[i][k] * A[k][j];
            }
        tf_params_d(C, prm);
        alpha = prm[0]; tx = prm[1]; ty = prm[2];
    }
    tf_build_d(alpha, tx, ty, 0, A);
    const double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    I[0][0] = A[1][1] / det; I[0][1] = -A[0][1] / det;
    I[1][0] = -A[1][0] / det; I[1][1] = A[0][0] / det;
    I[0][2] = -(I[0][0] * A[0][2] + I[0][1] * A[1][2]);
    I[1][2] = -(I[1][0] * A[0][2] + I[1][1] * A[1][2]);
    I[2][0] = 0; I[2][1] = 0; I[2][2] = 1;
    tf_params_d(I, prm);
    state[2 * p] = (float)prm[1];
    state[2 * p + 1] = (float)prm[2];
}

extern "C" int ra_state_from_params(ra_engine *e, const ra_result *d_result, int n, const float *cs, float *d_state)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_result || !d_state) { g_last_error = "null argument"; return RA_ERR_ARG; }
    const float zero[2] = {0.f, 0.f};
    RA_HIP(hipMemcpyAsync(e->d_cs, cs ? cs : zero, 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(state_from_params_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, e->cfg.mode, (const float *)e->d_cs, d_result, d_state);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

// the same with the centre correction in device memory (d_cs [2], e.g. the reduced shift sums of the average-centre rule
// divided by the particle count on the device): no host value, no copy
extern "C" int ra_state_from_params_dev(ra_engine *e, const ra_result *d_result, int n, const float *d_cs, float *d_state)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_result || !d_state || !d_cs) { g_last_error = "null argument"; return RA_ERR_ARG; }
    hipLaunchKernelGGL(state_from_params_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, e->cfg.mode, d_cs, d_result, d_state);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

// Class-resident alignment (gpu_aln_noref.cu:559-782, the ISAC mode): every particle against the average of its own class, all
// classes in ONE launch of the fused search kernel.  ra_set_class_references prepares ncls references (Polar2Dm, Frngs,
// Applyws) and one B stream per class; ra_align_classes aligns particle i to reference d_cls[i].  Needs the fused kernel
// with a single reference (RA_MODE_REFFREE, nref = 1); RA_ERR_STATE otherwise (callers then loop over the classes).
extern "C" int ra_set_class_references(ra_engine *e, const float *d_refs, int ncls)
{
    if (!e || !d_refs || ncls <= 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (!e->fused || e->tiled || e->generic || e->cfg.nref != 1 || e->cfg.mode != RA_MODE_REFFREE) { g_last_error = "class-resident launch needs the fused kernel with one reference (and a box its reference preparation holds in LDS)"; return RA_ERR_STATE; }
    const FusedGeom f = e->fplan.f;
    int rc;
    if (ncls > e->cls_cap) {
        if (e->d_cls_refspec) (void)hipFree(e->d_cls_refspec);
        if (e->d_cls_Bf) (void)hipFree(e->d_cls_Bf);
        e->d_cls_refspec = nullptr; e->d_cls_Bf = nullptr; e->cls_cap = 0;
        if (e->d_cls_refx) (void)hipFree(e->d_cls_refx);
        e->d_cls_refx = nullptr;
        if (hipMalloc((void **)&e->d_cls_refspec, (size_t)ncls * e->geo.lring * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&e->d_cls_Bf, ((size_t)ncls * f.b_floats + 256) * sizeof(float)) != hipSuccess ||
            (e->refine_ok && hipMalloc((void **)&e->d_cls_refx, (size_t)ncls * e->geo.lcirc * sizeof(float)) != hipSuccess)) {
            g_last_error = "out of device memory (class references)";
            return RA_ERR_NOMEM;
        }
        e->cls_cap = ncls;
    }
    (void)rc;
    hipLaunchKernelGGL(ref_polar_fft_kernel, dim3(ncls), dim3(256), e->lds_ref, e->stream, e->dg, d_refs, ncls, e->d_cls_refspec);
    RA_HIP(hipGetLastError());
    hipLaunchKernelGGL(pack_refs_fused_kernel, dim3(std::min(64, (f.b_floats + 255) / 256), ncls), dim3(256), 0, e->stream, e->dg, f,
                       (const float *)e->d_cls_refspec, 1, e->d_cls_Bf);
    RA_HIP(hipGetLastError());
    if (e->refine_ok && e->refine_thr != 0.f && e->d_cls_refx) {
        hipLaunchKernelGGL(refspec_exact_kernel<false>, dim3(ncls), dim3(RA_EXACT_THREADS), e->lds_refine, e->stream, e->dg, (const int *)e->d_numr,
                           (const float *)e->d_wr, (const float *)e->d_twx, (const int *)e->d_twxoff, d_refs, ncls, e->d_cls_refx, (float *)nullptr);
        RA_HIP(hipGetLastError());
    }
    e->cls_ready = ncls;
    return RA_OK;
}

extern "C" int ra_align_classes(ra_engine *e, const float *d_particles, int n, float *d_state, ra_result *d_result,
                                const int *d_cls)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_particles || !d_state || !d_result || !d_cls) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (!e->fused || e->tiled || e->generic || e->cfg.nref != 1 || e->cls_ready <= 0) { g_last_error = "ra_set_class_references has not been called"; return RA_ERR_STATE; }
    const Geometry &g = e->geo;
    const int npix = g.nx * g.nx;
    const FusedGeom f = e->fplan.f;
    fused_fn fk = select_fused(g.maxrin, 1, f.nzr, e->dg.sbuf);
    const int rch = resident_batch(e, n);
    {
        int rcw = ensure_resident_ws(e, rch);
        if (rcw) return rcw;
    }
    for (int start = 0; start < n; start += rch) {
        const int cn = std::min(rch, n - start);
        float *st = d_state + (size_t)start * 2;
        hipLaunchKernelGGL(fk, dim3(std::min(cn, e->n_cu)), dim3(RF_THREADS), e->fplan.lds_bytes, e->stream, e->dg, f,
                           d_particles + (size_t)start * npix, (const float *)st, cn, (const float *)e->d_cls_Bf, 1, e->d_fcand, d_cls + start);
        RA_HIP(hipGetLastError());
        int rcf = finalize_and_refine(e, e->d_fcand, 1, cn, st, d_result + start, d_particles + (size_t)start * npix, e->d_cls_refx, d_cls + start);
        if (rcf) return rcf;
    }
    return RA_OK;
}

extern "C" int ra_align(ra_engine *e, const float *d_particles, int n, float *d_state, ra_result *d_result,
                        const float *cs)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_particles || !d_state || !d_result) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (!e->refs_ready) { g_last_error = "ra_set_references has not been called"; return RA_ERR_STATE; }
    const Geometry &g = e->geo;
    const int npix = g.nx * g.nx;
    const int ngroup = g.nshift_pad / 4;
    ccf_fn ccf = e->generic ? nullptr : select_ccf(g.maxrin);
    if (cs && (cs[0] != 0.f || cs[1] != 0.f)) {
        RA_HIP(hipMemcpyAsync(e->d_cs, cs, 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(apply_cs_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, e->d_cs, d_result, d_state);
        RA_HIP(hipGetLastError());
    }
    hipStream_t sp = e->stream;
    if (e->solo) {
        // particle-resident search, one offset resident per pass (ralign_solo.h): one persistent workgroup per CU
        const FusedGeom f = e->fplan.f;
        const solo_fn fk = e->pair ? select_pair(g.maxrin, f.nrpw) : e->duo ? select_duo(g.maxrin, f.nh, plan_nqmax(f)) : select_solo(g.maxrin, f.nh, f.ntile);
        const int rch = resident_batch(e, n);
        {
            int rcw = ensure_resident_ws(e, rch);
            if (rcw) return rcw;
        }
        for (int start = 0; start < n; start += rch) {
            const int cn = std::min(rch, n - start);
            float *st = d_state + (size_t)start * 2;
            std::pair<hipEvent_t, hipEvent_t> *evc = e->timing ? next_events(e->ev_ccf, e->ev_used_ccf) : nullptr;
            if (evc) RA_HIP(hipEventRecord(evc->first, sp));
            hipLaunchKernelGGL(fk, dim3(std::min(cn, e->n_cu)), dim3(RF_THREADS), e->fplan.lds_bytes, sp, e->dg, f, d_particles + (size_t)start * npix,
                               (const float *)st, cn, (const float *)e->d_Bf, e->cfg.nref, e->d_fcand, (float *)nullptr);
            RA_HIP(hipGetLastError());
            if (evc) RA_HIP(hipEventRecord(evc->second, sp));
            int rcf = finalize_and_refine(e, e->d_fcand, 1, cn, st, d_result + start, d_particles + (size_t)start * npix, e->d_refx, nullptr);
            if (rcf) return rcf;
        }
        return RA_OK;
    }
    if (e->fused) {
        // particle-resident search: one workgroup per particle, spectra stay on the CU; launched per chunk so that a
        // launch stays a bounded unit of work (timing, candidate workspace)
        FusedGeom f = e->fplan.f;
        // dense offset stream: a template parameter of search_fused_kernel, a run-time flag of search_tiled_kernel (RALIGN_PACK=0: off)
        f.pack = e->tiled && g.nshift % 4 != 0 && g.nshift >= 4 && !(getenv("RALIGN_PACK") && atoi(getenv("RALIGN_PACK")) == 0);
        fused_fn fk = e->tiled ? select_tiled(f.nh, e->dg.sbuf) : select_fused(g.maxrin, e->cfg.nref, e->fplan.f.nzr, e->dg.sbuf, pack_ok(e), e->tcrop);
        const int rch = resident_batch(e, n);
        {
            int rcw = ensure_resident_ws(e, rch);
            if (rcw) return rcw;
        }
        for (int start = 0; start < n; start += rch) {
            const int cn = std::min(rch, n - start);
            float *st = d_state + (size_t)start * 2;
            std::pair<hipEvent_t, hipEvent_t> *evc = e->timing ? next_events(e->ev_ccf, e->ev_used_ccf) : nullptr;
            if (evc) RA_HIP(hipEventRecord(evc->first, sp));
            // one workgroup per CU (its LDS plan fills the CU); each walks over its share of the chunk
            hipLaunchKernelGGL(fk, dim3(std::min(cn, e->n_cu)), dim3(RF_THREADS), e->fplan.lds_bytes, sp, e->dg, f, d_particles + (size_t)start * npix,
                               (const float *)st, cn, (const float *)e->d_Bf, e->cfg.nref, e->d_fcand, (const int *)nullptr);
            RA_HIP(hipGetLastError());
            if (evc) RA_HIP(hipEventRecord(evc->second, sp));
            int rcf = finalize_and_refine(e, e->d_fcand, 1, cn, st, d_result + start, d_particles + (size_t)start * npix, e->d_refx, nullptr);
            if (rcf) return rcf;
        }
        return RA_OK;
    }
    {
        int rcw = ensure_unfused_ws(e);
        if (rcw) return rcw;
    }
    // the flagged particles of every chunk collect in one list; one refine launch behind the last chunk (finalize_and_refine)
    const bool refine_once = e->refine_ok && e->refine_thr != 0.f && e->d_refx && n > e->chunk;
    if (refine_once) {
        if (n > e->rlist_cap) {
            int rcg = dev_grow(e, &e->d_rlist, (size_t)n, false);
            if (rcg) return rcg;
            e->rlist_cap = n;
        }
        RA_HIP(hipMemsetAsync(e->d_rcount, 0, sizeof(int), e->stream));
    }
    for (int start = 0; start < n; start += e->chunk) {
        const int cn = std::min(e->chunk, n - start);
        float *Abuf = e->d_A;
        CandT *Cbuf = e->d_cand;
        const float *part = d_particles + (size_t)start * npix;
        float *st = d_state + (size_t)start * 2;
        std::pair<hipEvent_t, hipEvent_t> *evp = nullptr, *evc = nullptr;
        if (e->timing) {
            evp = next_events(e->ev_polar, e->ev_used_polar);
            evc = next_events(e->ev_ccf, e->ev_used_ccf);
        }
        if (evp) RA_HIP(hipEventRecord(evp->first, sp));
        if (e->generic) {
            int rcp = launch_generic_polar(e, part, st, cn, Abuf);
            if (rcp) return rcp;
        } else
            hipLaunchKernelGGL(polar_fft_kernel, dim3(cn), dim3(RA_POLAR_THREADS), e->lds_polar, sp, e->dg, part, st, cn, Abuf);
        RA_HIP(hipGetLastError());
        if (evp) RA_HIP(hipEventRecord(evp->second, sp));
        const int n_mtile = (cn * e->dg.ent_stride + 7) / 8;
        if (evc) RA_HIP(hipEventRecord(evc->first, sp));
        // blocks of TM x 7 tiles when the reference tiles come in sevens (gccf_tm): the B stream is read once per 8 TM particle-offsets
        const bool split = e->generic && g.maxrin == 1024 && !(getenv("RALIGN_GCCF_SPLIT") && atoi(getenv("RALIGN_GCCF_SPLIT")) == 0);
        const int tmv = e->generic ? gccf_tm(e->nrtile, g.maxrin) : 1;
        const bool tm2 = tmv >= 2 && gccf_wide_blocks(e->nrtile);
        if (tm2 && !split)
            hipLaunchKernelGGL((ccf_generic_kernel<2, 7>), dim3(std::min((n_mtile + 1) / 2, e->g_nblk)), dim3(RA_GCCF_THREADS), e->lds_gccf, sp, e->dg,
                               Abuf, e->d_B, n_mtile, e->nrtile, e->cfg.nref, Cbuf, e->d_zscr, e->g_P, (const float2 *)e->d_gstats,
                               (const float *)e->d_gcdc);
        else if (split) {
            // maxrin 1024: contraction and inverse transforms as two kernels per slice of g_nblk blocks (the scratch holds one slice)
            const bool wide = gccf_wide_blocks(e->nrtile);
            const int TMv = wide ? tmv : 2, TRv = wide ? 7 : 2;
            const int n_mt2 = (n_mtile + TMv - 1) / TMv, n_rt2 = (e->nrtile + TRv - 1) / TRv, ntask = n_mt2 * n_rt2;
            const size_t lds2 = ((size_t)8 * RA_IFFT3_PSTRIDE + 16 * 16 + 16 * 64) * sizeof(float2);
            const int nwg = TMv >= 4 ? e->g_nblk / 2 : e->g_nblk;           // 4 x 7 blocks: one workgroup per CU (256 registers)
            const int nblk = nwg * gccf_blocks_per_wg();                    // blocks per slice (the scratch holds them)
            auto launch = [&](auto ccfk, auto ifftk, int task0, int nt, int grid2) {
                hipLaunchKernelGGL(ccfk, dim3(std::min(nt, nwg)), dim3(RA_EXP_ENV("RALIGN_GCCF_WAVES") ? 64 * std::max(1, std::min(8, ra_atoi(RA_EXP_ENV("RALIGN_GCCF_WAVES")))) : RA_GCCF_THREADS), 0, sp, e->dg, Abuf, e->d_B, n_mtile, e->nrtile, e->cfg.nref, Cbuf,
                                   e->d_zscr, e->g_P, (const float2 *)e->d_gstats, (const float *)e->d_gcdc, task0, nt);
                if (!(e->dg.dbg & 1))          // (profiling builds, RALIGN_DEBUG=1: the contraction alone)
                hipLaunchKernelGGL(ifftk, dim3(grid2), dim3(RA_GCCF_THREADS), lds2, sp, e->dg, n_mtile, e->nrtile, e->cfg.nref, Cbuf,
                                   (const float2 *)e->d_zscr, (const float2 *)e->d_gstats, task0, nt);
            };
            for (int task0 = 0; task0 < ntask; task0 += nblk) {
                const int nt = std::min(nblk, ntask - task0);
                const int grid2 = std::min(nt * TMv * TRv * 8, (RA_EXP_ENV("RALIGN_IFFT_WGS") ? ra_atoi(RA_EXP_ENV("RALIGN_IFFT_WGS")) : 2) * e->n_cu);
                if (wide && TMv == 4) launch(ccf_generic_kernel<4, 7, true>, gccf_ifft_kernel<4, 7>, task0, nt, grid2);
                else if (wide && TMv == 2) launch(ccf_generic_kernel<2, 7, true>, gccf_ifft_kernel<2, 7>, task0, nt, grid2);
                else if (wide) launch(ccf_generic_kernel<1, 7, true>, gccf_ifft_kernel<1, 7>, task0, nt, grid2);
                else launch(ccf_generic_kernel<2, 2, true>, gccf_ifft_kernel<2, 2>, task0, nt, grid2);
                RA_HIP(hipGetLastError());
            }
        }
        else if (e->generic && gccf_wide_blocks(e->nrtile))
            hipLaunchKernelGGL((ccf_generic_kernel<1, 7>), dim3(std::min(n_mtile, e->g_nblk)), dim3(RA_GCCF_THREADS), e->lds_gccf, sp, e->dg,
                               Abuf, e->d_B, n_mtile, e->nrtile, e->cfg.nref, Cbuf, e->d_zscr, e->g_P, (const float2 *)e->d_gstats,
                               (const float *)e->d_gcdc);
        else if (e->generic)
            hipLaunchKernelGGL((ccf_generic_kernel<2, 2>), dim3(std::min((n_mtile + 1) / 2, e->g_nblk)), dim3(RA_GCCF_THREADS), e->lds_gccf, sp, e->dg,
                               Abuf, e->d_B, n_mtile, e->nrtile, e->cfg.nref, Cbuf, e->d_zscr, e->g_P, (const float2 *)e->d_gstats,
                               (const float *)e->d_gcdc);
        else
            hipLaunchKernelGGL(ccf, dim3(n_mtile), dim3(RA_CCF_THREADS), e->lds_ccf, sp, e->dg, Abuf, e->d_B, n_mtile,
                               e->nrtile, e->cfg.nref, Cbuf);
        RA_HIP(hipGetLastError());
        if (evc) RA_HIP(hipEventRecord(evc->second, sp));
        int rcf = finalize_and_refine(e, Cbuf, e->nrtile, cn, st, d_result + start, part, e->d_refx, nullptr, refine_once ? 1 : 0, start);
        if (rcf) return rcf;
    }
    if (refine_once) {
        int rcf = finalize_and_refine(e, nullptr, e->nrtile, n, d_state, d_result, d_particles, e->d_refx, nullptr, 2, 0);
        if (rcf) return rcf;
    }
    (void)ngroup;
    return RA_OK;
}

extern "C" int ra_debug_spectra(ra_engine *e, const float *d_particles, int n, const float *d_state, float *h_out)
{
    if (!e || !d_particles || !d_state || !h_out || n < 1 || n > e->chunk) { g_last_error = "bad argument"; return RA_ERR_ARG; }
    const Geometry &g = e->geo;
    if (e->solo) {
        // the polar stage of search_solo_kernel: ring buffer and statistics of every in-window (particle, offset)
        const size_t rawcnt = (size_t)n * g.nshift * (g.lring + 2), cnt = (size_t)n * g.nshift * g.lcirc;
        float *d_raw = nullptr, *d_out = nullptr;
        RA_HIP(hipMalloc((void **)&d_raw, rawcnt * sizeof(float)));
        hipError_t he = hipMalloc((void **)&d_out, cnt * sizeof(float));
        if (he == hipSuccess) he = hipMemsetAsync(d_raw, 0, rawcnt * sizeof(float), e->stream);
        if (he == hipSuccess) {
            const FusedGeom f = e->fplan.f;
            const solo_fn dk = e->pair ? (solo_fn)search_pair_kernel<256, 1> : (solo_fn)search_solo_kernel<512, 1, true>;
            hipLaunchKernelGGL(dk, dim3(std::min(n, e->n_cu)), dim3(RF_THREADS), e->fplan.lds_bytes, e->stream, e->dg, f, d_particles,
                               d_state, n, (const float *)e->d_Bf, e->cfg.nref, e->d_fcand, d_raw);
            hipLaunchKernelGGL(unpack_solo_spectra_kernel, dim3(n * g.nshift), dim3(256), 0, e->stream, e->dg, (const float *)d_raw, n,
                               (const int *)e->d_numr, (const int *)e->d_ring_off, d_out);
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (he == hipSuccess) he = hipMemcpy(h_out, d_out, cnt * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(d_raw);
        if (d_out) (void)hipFree(d_out);
        RA_HIP(he);
        return RA_OK;
    }
    {
        int rcw = ensure_unfused_ws(e);
        if (rcw) return rcw;
    }
    if (e->generic) {
        int rcp = launch_generic_polar(e, d_particles, d_state, n, e->d_A);
        if (rcp) return rcp;
    } else
        hipLaunchKernelGGL(polar_fft_kernel, dim3(n), dim3(RA_POLAR_THREADS), e->lds_polar, e->stream, e->dg, d_particles, d_state, n, e->d_A);
    RA_HIP(hipGetLastError());
    float *d_out = nullptr;
    const size_t cnt = (size_t)n * g.nshift * g.lcirc;
    RA_HIP(hipMalloc((void **)&d_out, cnt * sizeof(float)));
    hipLaunchKernelGGL(unpack_spectra_kernel, dim3(n * g.nshift), dim3(256), 0, e->stream, e->dg, e->d_A, n, e->d_numr, d_out,
                       e->generic ? (const float2 *)e->d_gstats : (const float2 *)nullptr, d_state);
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(h_out, d_out, cnt * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    RA_HIP(he);
    return RA_OK;
}

// rot_shift2D + class sums without the aligned stack (transform_sum_kernel): member lists of up to RA_XS_BATCH particles at
// once, every (class, parity) list cut into runs so that ~1000 workgroups share the work
#define RA_XS_BATCH 65536
typedef void (*xs_fn)(int, const float *, int, int, const ra_result *, const float2 *, const int *, const int *, int, float *);
static xs_fn select_xs(int nx, int *nband = nullptr)
{
    if (nband) *nband = 1;
    if (nx < 8 || nx > RA_XS_THREADS) return nullptr;
    const int ry = RA_XS_THREADS / nx, npt = (nx + ry - 1) / ry;      // sweeps of the workgroup over the image
    if (npt <= 1) return transform_sum_kernel<1>;
    if (npt <= 2) return transform_sum_kernel<2>;
    if (npt <= 4) return transform_sum_kernel<4>;
    if (npt <= 6) return transform_sum_kernel<6>;
    if (npt <= 9) return transform_sum_kernel<9>;
    // larger boxes: row bands of at most 8 sweeps per workgroup (9 accumulators and their prefetch registers are what 128 hold)
    if (npt <= 12) { if (nband) *nband = 2; return transform_sum_kernel<6, 12>; }
    if (npt <= 16) { if (nband) *nband = 2; return transform_sum_kernel<8, 16>; }
    if (npt <= 20) { if (nband) *nband = 3; return transform_sum_kernel<7, 20>; }
    if (npt <= 24) { if (nband) *nband = 3; return transform_sum_kernel<8, 24>; }
    return nullptr;
}
// the whole image in LDS (transform_sum_kernel), or -- larger boxes -- output tiles with the source box of each in LDS
// (transform_sum_tile_kernel; RALIGN_XTILE=0: the aligned stack + class_sum_kernel as before)
static bool xs_whole_image(const ra_engine *e)
{
    if (e->xf_generic || !select_xs(e->geo.nx)) return false;
    return (size_t)(e->geo.nx + 2) * ((e->geo.nx + 2) | 1) * sizeof(float) <= RA_XS_LDS_MAX;
}
static bool xs_usable(const ra_engine *e)
{
    if (e->atomic_sums) return false;
    if (getenv("RALIGN_XSUM") && atoi(getenv("RALIGN_XSUM")) == 0) return false;
    if (xs_whole_image(e)) return true;
    return e->geo.nx >= RA_XT_BB && !(getenv("RALIGN_XTILE") && atoi(getenv("RALIGN_XTILE")) == 0);
}
static int transform_sum(ra_engine *e, const float *d_particles, int n, int index0, const ra_result *d_result, float *d_sums, int *d_counts)
{
    const int nx = e->geo.nx, npix = nx * nx, nseg = 2 * e->cfg.nref;
    int nband = 1;
    const bool tiles = !xs_whole_image(e);
    const xs_fn fn = tiles ? transform_sum_tile_kernel : select_xs(nx, &nband);
    const int ntile = ((nx + RA_XT_TS - 1) / RA_XT_TS) * ((nx + RA_XT_TS - 1) / RA_XT_TS);
    const size_t xs_lds = tiles ? (size_t)2 * RA_XT_BB * RA_XT_BB * sizeof(float) : (size_t)(nx + 2) * ((nx + 2) | 1) * sizeof(float);
    if (xs_lds > 64 * 1024) RA_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_lds));
    for (int start = 0; start < n; start += RA_XS_BATCH) {
        const int cn = std::min(RA_XS_BATCH, n - start);
        // runs per segment: ~1024 workgroups, at least ~8 members per run on average
        int nrun = std::max(1, std::min(512, (1024 + nseg - 1) / nseg));
        if (tiles) nrun = std::max(1, std::min(64, 2048 / (nseg * ntile)));      // ntile workgroups per (segment, run) already
        nrun = std::max(1, std::min(nrun, cn / (8 * nseg)));
        const size_t need_m = (size_t)nseg * cn, need_p = (size_t)nrun * nseg * npix;
        if (need_m > e->xs_cap_members) {
            int rc = dev_grow(e, &e->d_xs_members, need_m, false);
            if (rc) return rc;
            e->xs_cap_members = need_m;
        }
        if (!e->d_xs_mcount) { int rc = dev_alloc(e, &e->d_xs_mcount, (size_t)nseg, true); if (rc) return rc; }
        if ((size_t)cn > e->xs_cap_trig) {
            int rc = dev_grow(e, &e->d_xs_trig, (size_t)cn, false);
            if (rc) return rc;
            e->xs_cap_trig = cn;
        }
        hipLaunchKernelGGL(transform_trig_kernel, dim3((cn + 255) / 256), dim3(256), 0, e->stream, d_result + start, cn, e->d_xs_trig);
        RA_HIP(hipGetLastError());
        if (need_p > e->xs_cap_partial) {
            int rc = dev_grow(e, &e->d_xs_partial, need_p, false);
            if (rc) return rc;
            e->xs_cap_partial = need_p;
        }
        hipLaunchKernelGGL(class_members_wide_kernel, dim3(nseg), dim3(1024), 0, e->stream, d_result + start, cn, index0 + start,
                           e->d_xs_members, e->d_xs_mcount, cn, d_counts);
        RA_HIP(hipGetLastError());
        hipLaunchKernelGGL(fn, dim3(nseg, nrun, tiles ? ntile : nband), dim3(tiles ? RA_XT_THREADS : RA_XS_THREADS), xs_lds, e->stream, nx,
                           d_particles + (size_t)start * npix, cn, index0 + start, d_result + start, (const float2 *)e->d_xs_trig, (const int *)e->d_xs_members,
                           (const int *)e->d_xs_mcount, cn, e->d_xs_partial);
        RA_HIP(hipGetLastError());
        hipLaunchKernelGGL(class_sum_combine_kernel, dim3((unsigned)(((size_t)nseg * npix + 255) / 256)), dim3(256), 0, e->stream, npix, nseg, nrun,
                           (const float *)e->d_xs_partial, d_sums);
        RA_HIP(hipGetLastError());
    }
    return RA_OK;
}

extern "C" int ra_transform_accumulate(ra_engine *e, const float *d_particles, int n, int index0,
                                       const ra_result *d_result, float *d_aligned, float *d_sums, int *d_counts)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_particles || !d_result) { g_last_error = "null argument"; return RA_ERR_ARG; }
    const int nx = e->geo.nx, npix = nx * nx;
    if (!d_sums || e->atomic_sums) {
        if (e->xf_generic)
            // (a handful of images -- the references of an update step -- are cut into more row blocks: 10 images x 8 blocks left the GPU idle)
            hipLaunchKernelGGL(transform_generic_kernel, dim3(n, std::max(8, std::min((npix + 255) / 256, (2048 + n - 1) / n))), dim3(256), 0, e->stream, nx, d_particles, n, index0,
                               d_result, d_aligned, d_sums, d_counts);
        else
            hipLaunchKernelGGL(transform_kernel, dim3(n), dim3(RA_XF_THREADS), e->lds_xf, e->stream, nx, d_particles, n, index0,
                               d_result, d_aligned, d_sums, d_counts);
        RA_HIP(hipGetLastError());
        return RA_OK;
    }
    if (!d_aligned && xs_usable(e)) return transform_sum(e, d_particles, n, index0, d_result, d_sums, d_counts);
    // aligned images wanted as well (or an image too large for the LDS): aligned images of a chunk, then particle-order sums per
    // (class, parity)
    if (!d_aligned && !e->d_alscratch) {
        int rca = dev_alloc(e, &e->d_alscratch, e->wp.alscratch_floats, false);
        if (rca) return rca;
    }
    for (int start = 0; start < n; start += e->chunk) {
        const int cn = std::min(e->chunk, n - start);
        float *al = d_aligned ? d_aligned + (size_t)start * npix : e->d_alscratch;
        if (e->xf_generic)
            hipLaunchKernelGGL(transform_generic_kernel, dim3(cn, 8), dim3(256), 0, e->stream, nx, d_particles + (size_t)start * npix,
                               cn, index0 + start, d_result + start, al, (float *)nullptr, (int *)nullptr);
        else
            hipLaunchKernelGGL(transform_kernel, dim3(cn), dim3(RA_XF_THREADS), e->lds_xf, e->stream, nx, d_particles + (size_t)start * npix,
                               cn, index0 + start, d_result + start, al, (float *)nullptr, (int *)nullptr);
        RA_HIP(hipGetLastError());
        // member lists once per chunk (class_members_kernel); every (class, parity) list is then cut into runs so that ~4000
        // workgroups share the additions
        const int nseg = 2 * e->cfg.nref, ntile = (npix + 255) / 256;
        const bool glists = (size_t)nseg * e->chunk * sizeof(int) <= ((size_t)256 << 20);
        if (glists && !e->d_members) {
            int rcm = dev_alloc(e, &e->d_members, (size_t)nseg * e->chunk, false);
            if (!rcm) rcm = dev_alloc(e, &e->d_mcount, (size_t)nseg, true);
            if (rcm) return rcm;
        }
        const int nrun = std::min(16, std::max(1, (glists ? 4096 : 1024) / (nseg * ntile)));
        if (nrun > 1 && !e->d_sumpart) {
            int rcp = dev_alloc(e, &e->d_sumpart, (size_t)16 * nseg * npix, false);
            if (rcp) return rcp;
        }
        if (glists) {
            hipLaunchKernelGGL(class_members_kernel, dim3(nseg), dim3(64), 0, e->stream, d_result + start, cn, index0 + start, e->d_members, e->d_mcount);
            RA_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(class_sum_kernel, dim3(nseg, ntile, nrun), dim3(256), glists ? 0 : (size_t)cn * sizeof(int), e->stream, npix, al,
                           d_result + start, cn, index0 + start, d_sums, d_counts, nrun > 1 ? e->d_sumpart : (float *)nullptr,
                           glists ? (const int *)e->d_members : (const int *)nullptr, (const int *)e->d_mcount);
        RA_HIP(hipGetLastError());
        if (nrun > 1) {
            hipLaunchKernelGGL(class_sum_combine_kernel, dim3((unsigned)(((size_t)nseg * npix + 255) / 256)), dim3(256), 0, e->stream, npix, nseg, nrun,
                               (const float *)e->d_sumpart, d_sums);
            RA_HIP(hipGetLastError());
        }
    }
    return RA_OK;
}

extern "C" int ra_update_references(ra_engine *e, const float *d_sums, const int *d_counts, int min_count,
                                    float *d_refs)
{
    if (!e || !d_sums || !d_counts || !d_refs) { g_last_error = "null argument"; return RA_ERR_ARG; }
    hipLaunchKernelGGL(update_refs_kernel, dim3(e->cfg.nref), dim3(256), 0, e->stream, e->geo.nx, d_sums, d_counts,
                       min_count, e->dg.mask, d_refs);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

extern "C" int ra_normalize_particles(ra_engine *e, float *d_particles, int n)
{
    if (!e || n < 0) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (n == 0) return RA_OK;
    if (!d_particles) { g_last_error = "null argument"; return RA_ERR_ARG; }
    hipLaunchKernelGGL(normalize_particles_kernel, dim3(n), dim3(256), 0, e->stream, e->geo.nx, e->dg.mask,
                       d_particles, n);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

extern "C" int ra_sync(ra_engine *e)
{
    if (!e) return RA_ERR_ARG;
    RA_HIP(hipStreamSynchronize(e->stream));
    return RA_OK;
}

extern "C" int ra_kernel_time(ra_engine *e, int enable, double *ms_ccf, int *launches_ccf, double *ms_polar,
                              int *launches_polar)
{
    if (!e) return RA_ERR_ARG;
    RA_HIP(hipStreamSynchronize(e->stream));
    double a = 0, b = 0;
    for (size_t i = 0; i < e->ev_used_ccf; i++) {
        float ms = 0;
        RA_HIP(hipEventElapsedTime(&ms, e->ev_ccf[i].first, e->ev_ccf[i].second));
        a += ms;
    }
    for (size_t i = 0; i < e->ev_used_polar; i++) {
        float ms = 0;
        RA_HIP(hipEventElapsedTime(&ms, e->ev_polar[i].first, e->ev_polar[i].second));
        b += ms;
    }
    if (ms_ccf) *ms_ccf = a;
    if (launches_ccf) *launches_ccf = (int)e->ev_used_ccf;
    if (ms_polar) *ms_polar = b;
    if (launches_polar) *launches_polar = (int)e->ev_used_polar;
    e->ev_used_ccf = e->ev_used_polar = 0;
    e->timing = enable != 0;
    return RA_OK;
}

// ---------------------------------------------------------------------------------------------
// reference update on the device (kernels in ralign_refine.h)

static int ensure_refine_ws(ra_engine *e, int nimg)
{
    if (nimg <= e->rf_cap) return RA_OK;
    const int nx = e->geo.nx, nxh = nx / 2 + 1;
    const int cap = std::max(nimg, 2 * e->cfg.nref);
    int rc;
    if ((rc = dev_alloc(e, &e->d_rfT, (size_t)cap * nx * nxh, false)) || (rc = dev_alloc(e, &e->d_rfF, (size_t)cap * nx * nxh, false)) ||
        (rc = dev_alloc(e, &e->d_rfmean, (size_t)cap, true)) || (rc = dev_alloc(e, &e->d_rffsc, (size_t)cap * (nx / 2 + 1), true)) ||
        (rc = dev_alloc(e, &e->d_rfcs, (size_t)cap * 2, true)))
        return rc;
    if (!e->d_rftw) {
        std::vector<double2> tw(nx);
        for (int t = 0; t < nx; t++) { const double a = 2.0 * M_PI * t / nx; tw[t] = make_double2(cos(a), sin(a)); }
        const double2 *p = nullptr;
        if ((rc = upload(e, tw, &p))) return rc;
        e->d_rftw = (double2 *)p;
    }
    e->rf_cap = cap;
    return RA_OK;
}

static int forward_dft(ra_engine *e, const float *d_imgs, int nimg, const float *d_mask, const float *d_mean)
{
    const int nx = e->geo.nx;
    hipLaunchKernelGGL(dft_rows_kernel, dim3(nimg, nx), dim3(64), 0, e->stream, nx, d_imgs, d_mask, d_mean, e->d_rftw, e->d_rfT);
    RA_HIP(hipGetLastError());
    hipLaunchKernelGGL(dft_cols_kernel<-1>, dim3(nimg, nx), dim3(64), 0, e->stream, nx, e->d_rfT, e->d_rftw, e->d_rfF);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

extern "C" int ra_fsc_len(const ra_engine *e) { return e ? e->geo.nx / 2 + 1 : RA_ERR_ARG; }

// fsc / fsc_mask of the even and odd sum of every class: e->d_rffsc [nref][2][nx/2+1] = {fsc, points per shell}
static int class_fsc_kernels(ra_engine *e, const float *d_sums, int masked)
{
    const int R = e->cfg.nref, nx = e->geo.nx;
    int rc = ensure_refine_ws(e, 2 * R);
    if (rc) return rc;
    if (masked) {
        hipLaunchKernelGGL(masked_mean_kernel, dim3(2 * R), dim3(256), 0, e->stream, nx * nx, d_sums, e->dg.mask, e->d_rfmean);
        RA_HIP(hipGetLastError());
    }
    if ((rc = forward_dft(e, d_sums, 2 * R, masked ? e->dg.mask : nullptr, masked ? e->d_rfmean : nullptr))) return rc;
    const int len = nx / 2 + 1;
    if (!e->d_fsc_off) {          // the coefficients of every shell, in scan order
        std::vector<int> off, idx;
        fsc_shell_table(nx, off, idx);
        const int *p0 = nullptr, *p1 = nullptr;
        if ((rc = upload(e, off, &p0)) || (rc = upload(e, idx, &p1))) return rc;
        e->d_fsc_off = p0; e->d_fsc_idx = p1;
    }
    const int threads = std::min(1024, (RA_FSC_PARTS * len + 63) / 64 * 64);
    const size_t lds = (size_t)RA_FSC_PARTS * len * 4 * sizeof(double);
    if (lds > 64 * 1024) RA_HIP(hipFuncSetAttribute((const void *)fsc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fsc_kernel, dim3(R), dim3(threads), lds, e->stream, nx, e->d_rfF, e->d_fsc_off, e->d_fsc_idx, e->d_rffsc);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

// Average FSC of the live classes and the tangent-filter fit entirely on the device (fsc_fit_kernel): nothing comes back to
// the host, ra_filter_references_dev takes (fl, aa) from d_fit.  d_fit [5] = {fl, aa clamped, fl, aa fitted, status},
// d_curve [3][nx/2+1] = {frequency, fsc as fit_tanh leaves it, points}; the caller copies them to the host when it wants them
// (bookkeeping).  Asynchronous on the engine's stream.
extern "C" int ra_class_fsc_fit(ra_engine *e, const float *d_sums, const int *d_counts, int min_count, int masked, float fl_lo,
                                float fl_hi, float aa_hi, float *d_fit, float *d_curve)
{
    if (!e || !d_sums || !d_counts || !d_fit || !d_curve) { g_last_error = "null argument"; return RA_ERR_ARG; }
    int rc = class_fsc_kernels(e, d_sums, masked);
    if (rc) return rc;
    const int len = e->geo.nx / 2 + 1;
    hipLaunchKernelGGL(fsc_fit_kernel, dim3(1), dim3(64), (size_t)len * (sizeof(double) + 2 * sizeof(float)), e->stream, e->geo.nx, e->cfg.nref,
                       (const float *)e->d_rffsc, d_counts, min_count, fl_lo, fl_hi, aa_hi, d_fit, d_curve);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

extern "C" int ra_class_fsc(ra_engine *e, const float *d_sums, const int *d_counts, int min_count, int masked, float *h_fsc)
{
    if (!e || !d_sums || !d_counts || !h_fsc) { g_last_error = "null argument"; return RA_ERR_ARG; }
    const int R = e->cfg.nref, nx = e->geo.nx, len = nx / 2 + 1;
    int rc = class_fsc_kernels(e, d_sums, masked);
    if (rc) return rc;
    std::vector<float> all((size_t)R * 2 * len);
    std::vector<int> counts(R);
    RA_HIP(hipMemcpyAsync(all.data(), e->d_rffsc, all.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    RA_HIP(hipMemcpyAsync(counts.data(), d_counts, R * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    RA_HIP(hipStreamSynchronize(e->stream));
    // average over the live classes as the reference does (ave_fsc / c_fsc; kept only if its sum is not 0)
    std::vector<double> ave(len, 0.0);
    int live = 0, last = -1;
    for (int j = 0; j < R; j++) {
        if (counts[j] < min_count) continue;
        for (int i = 0; i < len; i++) ave[i] += all[((size_t)j * 2) * len + i];
        live++; last = j;
    }
    if (live == 0) { g_last_error = "ra_class_fsc: every class is below min_count"; return RA_ERR_STATE; }
    double tot = 0;
    for (int i = 0; i < len; i++) tot += ave[i];
    for (int i = 0; i < len; i++) {
        h_fsc[i] = (float)((double)i / (2.0 * (nx / 2)));
        h_fsc[len + i] = (tot != 0.0) ? (float)(ave[i] / live) : all[((size_t)last * 2) * len + i];
        h_fsc[2 * len + i] = all[((size_t)last * 2 + 1) * len + i];
    }
    return RA_OK;
}

// the per-class curves behind the last ra_class_fsc: h_all [nref][2][nx/2+1] = {fsc, points per shell} of every class (what
// the reference writes to drm%03d%04d.txt, test_mref_gpu_align.py:533)
extern "C" int ra_last_class_fsc(ra_engine *e, float *h_all)
{
    if (!e || !h_all) { g_last_error = "null argument"; return RA_ERR_ARG; }
    if (!e->d_rffsc) { g_last_error = "ra_class_fsc has not been called"; return RA_ERR_STATE; }
    const size_t cnt = (size_t)e->cfg.nref * 2 * (e->geo.nx / 2 + 1);
    RA_HIP(hipMemcpyAsync(h_all, e->d_rffsc, cnt * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    RA_HIP(hipStreamSynchronize(e->stream));
    return RA_OK;
}

// sp_filter.fit_tanh(dres, low = 0.1) with sp_utilities.amoeba (simplex maximisation); host arithmetic in
// double like the Python original.  fsc is edited in place (zeroed after its first drop below `low`).
namespace {
struct TanhFit {
    const float *freq; float *fsc; int n;
    double operator()(const double *a) const
    {
        double v = 0.0;
        if (fsc[0] < 0.0f) fsc[0] *= -1.0f;
        for (int i = 0; i < n; i++) {
            const double r = fsc[i], f = 2 * r / (1.0 + r);
            double qt = 0;
            if (a[0] != 0 && a[1] != 0)
                qt = f - 0.5 * (tanh(M_PI * (freq[i] + a[0]) / 2.0 / a[1] / a[0]) - tanh(M_PI * (freq[i] - a[0]) / 2.0 / a[1] / a[0]));
            v -= qt * qt;
        }
        return v;
    }
};
}  // namespace

extern "C" int ra_fit_tanh(const float *freq, float *fsc, int n, float *fl, float *aa)
{
    if (!freq || !fsc || !fl || !aa || n < 3) { g_last_error = "bad argument"; return RA_ERR_ARG; }
    const double low = 0.1;
    bool setzero = false;
    for (int i = 1; i < n; i++) {
        if (!setzero && 2 * (double)fsc[i] / (1.0 + fsc[i]) < low) setzero = true;
        if (setzero) fsc[i] = 0.0f;
    }
    double f0 = -1.0;
    for (int i = 1; i < n - 1; i++)
        if (2 * (double)fsc[i] / (1.0 + fsc[i]) < 0.5) { f0 = freq[i - 1]; break; }
    if (f0 < 0.0) {
        if (fsc[n - 1] < 0.5f) { *fl = 0.5f; *aa = 0.2f; } else { *fl = 0.49f; *aa = 0.1f; }
        return RA_OK;
    }
    TanhFit func{freq, fsc, n};
    const double scale[2] = {0.05, 0.05}, ftol = 1.e-4, xtol = 1.e-4;
    double sx[3][2] = {{f0, 0.1}, {f0 + scale[0], 0.1}, {f0, 0.1 + scale[1]}}, fv[3];
    for (int i = 0; i < 3; i++) fv[i] = func(sx[i]);
    int iteration = 0, best = 0;
    while (true) {
        int worst = 0; best = 0;
        for (int i = 0; i < 3; i++) { if (fv[i] > fv[best]) best = i; if (fv[i] < fv[worst]) worst = i; }
        double pavg[2] = {0, 0};
        for (int i = 0; i < 3; i++) if (i != worst) { pavg[0] += sx[i][0]; pavg[1] += sx[i][1]; }
        pavg[0] /= 2; pavg[1] /= 2;
        const double simscale = (fabs(pavg[0] - sx[worst][0]) / scale[0] + fabs(pavg[1] - sx[worst][1]) / scale[1]) / 2;
        const double fscale = (fabs(fv[best]) + fabs(fv[worst])) / 2.0;
        const double frange = fscale != 0.0 ? fabs(fv[best] - fv[worst]) / fscale : 0.0;
        if ((frange < ftol && simscale < xtol) || iteration >= 500) break;
        double pnew[2] = {2.0 * pavg[0] - sx[worst][0], 2.0 * pavg[1] - sx[worst][1]};
        double fnew = func(pnew);
        if (fnew <= fv[worst]) {           // worse than the worst: shrink towards the best
            for (int i = 0; i < 3; i++)
                if (i != best && i != worst) {
                    sx[i][0] = 0.5 * sx[best][0] + 0.5 * sx[i][0]; sx[i][1] = 0.5 * sx[best][1] + 0.5 * sx[i][1];
                    fv[i] = func(sx[i]);
                }
            pnew[0] = 0.5 * sx[best][0] + 0.5 * sx[worst][0]; pnew[1] = 0.5 * sx[best][1] + 0.5 * sx[worst][1];
            fnew = func(pnew);
        } else if (fnew >= fv[best]) {     // better than the best: try to expand
            double p2[2] = {3.0 * pavg[0] - 2.0 * sx[worst][0], 3.0 * pavg[1] - 2.0 * sx[worst][1]};
            const double f2 = func(p2);
            if (f2 > fnew) { pnew[0] = p2[0]; pnew[1] = p2[1]; fnew = f2; }
        }
        sx[worst][0] = pnew[0]; sx[worst][1] = pnew[1]; fv[worst] = fnew;
        iteration++;
    }
    *fl = (float)sx[best][0]; *aa = (float)sx[best][1];
    return RA_OK;
}

extern "C" int ra_class_averages(ra_engine *e, const float *d_sums, const int *d_counts, int min_count, float *d_refs)
{
    if (!e || !d_sums || !d_counts || !d_refs) { g_last_error = "null argument"; return RA_ERR_ARG; }
    hipLaunchKernelGGL(class_average_kernel, dim3(e->cfg.nref), dim3(256), 0, e->stream, e->geo.nx * e->geo.nx, d_sums, d_counts,
                       min_count, d_refs);
    RA_HIP(hipGetLastError());
    return RA_OK;
}

// filt_tanl + center_2D + (normalize.mask) of nimg images in place; fl / aa by value or (d_flaa != null) from device memory;
// centres: center = -1 reads d_cs_in [nimg][2] (device), the applied centres land in e->d_rfcs
static int filter_references_core(ra_engine *e, float *d_imgs, int nimg, float fl, float aa, const float *d_flaa, int center,
                                  const float *d_cs_in, int normalize)
{
    const int nx = e->geo.nx;
    int rc;
    if ((rc = forward_dft(e, d_imgs, nimg, nullptr, nullptr))) return rc;
    hipLaunchKernelGGL(filter_center_kernel, dim3(nimg), dim3(256), 0, e->stream, nx, e->d_rfF, fl, aa, center,
                       center == -1 ? d_cs_in : (const float *)nullptr, e->d_rfcs, d_flaa);
    RA_HIP(hipGetLastError());
    hipLaunchKernelGGL(dft_cols_kernel<1>, dim3(nimg, nx), dim3(64), 0, e->stream, nx, e->d_rfF, e->d_rftw, e->d_rfT);
    RA_HIP(hipGetLastError());
    hipLaunchKernelGGL(idft_rows_kernel, dim3(nimg, nx), dim3(128), 0, e->stream, nx, e->d_rfT, e->d_rftw, d_imgs);
    RA_HIP(hipGetLastError());
    if (normalize) {
        hipLaunchKernelGGL(normalize_mask_kernel, dim3(nimg), dim3(256), 0, e->stream, nx * nx, e->dg.mask, d_imgs);
        RA_HIP(hipGetLastError());
    }
    return RA_OK;
}

extern "C" int ra_filter_references(ra_engine *e, float *d_imgs, int nimg, float fl, float aa, int center,
                                    const float *h_cs_in, int normalize, float *h_cs_out)
{
    if (!e || !d_imgs || nimg < 1) { g_last_error = "bad argument"; return RA_ERR_ARG; }
    if (center == -1 && !h_cs_in) { g_last_error = "center = -1 needs the shift list"; return RA_ERR_ARG; }
    int rc = ensure_refine_ws(e, nimg);
    if (rc) return rc;
    if (center == -1) RA_HIP(hipMemcpyAsync(e->d_rfcs, h_cs_in, (size_t)nimg * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if ((rc = filter_references_core(e, d_imgs, nimg, fl, aa, nullptr, center, e->d_rfcs, normalize))) return rc;
    if (h_cs_out) {
        RA_HIP(hipMemcpyAsync(h_cs_out, e->d_rfcs, (size_t)nimg * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        RA_HIP(hipStreamSynchronize(e->stream));
    }
    return RA_OK;
}

// the same without a host value in the path: (fl, aa) = d_flaa[0..1] (ra_class_fsc_fit's d_fit; null: no filter), the centres
// of center = -1 from d_cs_in [nimg][2], the applied centres to d_cs_out [nimg][2] (device, may be null).  Asynchronous.
extern "C" int ra_filter_references_dev(ra_engine *e, float *d_imgs, int nimg, const float *d_flaa, int center,
                                        const float *d_cs_in, int normalize, float *d_cs_out)
{
    if (!e || !d_imgs || nimg < 1) { g_last_error = "bad argument"; return RA_ERR_ARG; }
    if (center == -1 && !d_cs_in) { g_last_error = "center = -1 needs the shift list"; return RA_ERR_ARG; }
    int rc = ensure_refine_ws(e, nimg);
    if (rc) return rc;
    if ((rc = filter_references_core(e, d_imgs, nimg, 0.f, 0.f, d_flaa, center, d_cs_in, normalize))) return rc;
    if (d_cs_out) RA_HIP(hipMemcpyAsync(d_cs_out, e->d_rfcs, (size_t)nimg * 2 * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return RA_OK;
}

// ============================================================================================
// reference-compatible surface (cuda/gpu_aln_noref.h:52-113): one process-global engine,
// synchronous calls, print + abort on failure like the reference (gpu_aln_common.cu:89-103).

namespace {
struct Legacy {
    ra_engine *eng = nullptr;
    AlignConfig cfg{};
    unsigned num_particles = 0;
    int device = -1;
    AlignParam *h_param = nullptr;       // pinned, caller reads / writes in place
    float *d_sbj = nullptr, *d_ref = nullptr, *d_aligned = nullptr, *d_state = nullptr, *d_sums = nullptr;
    int *d_counts = nullptr;
    ra_result *d_res = nullptr, *h_res = nullptr;
    float *h_stage = nullptr, *h_state = nullptr, *h_sums = nullptr;
    int *h_counts = nullptr;
    size_t stage_imgs = 0;
    unsigned sbj_loaded = 0;
    // class-resident (ISAC) mode: particles sorted by class, one reference per class
    bool isac = false;
    std::vector<unsigned> cid_idx;       // [ref_num + 1] first particle of every class
    unsigned *d_cid_idx = nullptr;
    int *d_cls = nullptr;                // [sbj_num] class of every particle (class-resident single launch)
} L;

void die(const char *what)
{
    fprintf(stderr, "libralign_hip: %s: %s\n", what, ra_last_error());
    exit(EXIT_FAILURE);
}
void hip_or_die(hipError_t e, const char *what)
{
    if (e != hipSuccess) { fprintf(stderr, "libralign_hip: %s: %s\n", what, hipGetErrorString(e)); exit(EXIT_FAILURE); }
}

ra_config legacy_config(const AlignConfig *c, unsigned device, int mode)
{
    ra_config rc{};
    rc.nx = (int)c->img_dim; rc.first_ring = 1; rc.last_ring = (int)c->ring_num; rc.ring_skip = 1;
    rc.xrng = c->shift_rng_x; rc.yrng = c->shift_rng_y; rc.step = c->shift_step;
    rc.nref = (int)c->ref_num; rc.mode = mode; rc.device = (int)device; rc.chunk = 0;
    return rc;
}

size_t legacy_bytes(unsigned num_particles, const AlignConfig *c)
{
    // everything pre_align_init takes from the device: the engine's workspace (same plan as ra_create) plus the
    // resident batch (particles, aligned images, state, results), the references and the class sums
    Geometry g;
    if (c->img_dim < 8 || c->ref_num < 1 || !build_rings(g, (int)c->img_dim, 1, (int)c->ring_num, 1) ||
        !build_shifts(g, c->shift_rng_x, c->shift_rng_y, c->shift_step))
        return (size_t)-1;
    ra_config rc = legacy_config(c, 0, RA_MODE_MREF);
    rc.chunk = (int)std::min<unsigned>(8192, std::max(2u, c->sbj_num));
    const bool generic = !fits_specialised_kernels(g, rc);
    if (generic && g.maxrin <= 1024) align_ring_quads(g);      // as ra_create: the panel size follows the layout
    const WorkspacePlan wp = plan_workspace(g, rc, generic);
    const size_t npix = (size_t)c->img_dim * c->img_dim, B = c->sbj_num, R = c->ref_num;
    const size_t batch = B * npix * 4 * 2 + B * (2 * sizeof(float) + sizeof(ra_result)) + R * npix * 4 * 3 + R * 4;
    (void)num_particles;     // the AlignParam array lives in pinned host memory
    return wp.bytes + batch + (size_t)8 * (2 << 20);
}

void run_search(int start, int stop, int mode)
{
    if (!L.eng) { fprintf(stderr, "libralign_hip: *_run before pre_align_init\n"); exit(EXIT_FAILURE); }
    const int n = stop - start;
    if (n <= 0 || (unsigned)n > L.cfg.sbj_num || (unsigned)stop > L.num_particles) {
        fprintf(stderr, "libralign_hip: bad index range [%d,%d)\n", start, stop);
        exit(EXIT_FAILURE);
    }
    L.eng->cfg.mode = mode; L.eng->dg.mode = mode; L.eng->dg.norm_ring = mode == RA_MODE_MREF ? 1 : 0;
    L.eng->dg.win_ring = mode == RA_MODE_MREF ? L.eng->geo.last_ring : L.eng->geo.numr[3 * (L.eng->geo.nring - 1)];      // (Normalize_ring follows the entry point: multiref_polar_ali_2d | ormq)
    for (int i = 0; i < n; i++) { L.h_state[2 * i] = L.h_param[start + i].shift_x; L.h_state[2 * i + 1] = L.h_param[start + i].shift_y; }
    hip_or_die(hipMemcpy(L.d_state, L.h_state, sizeof(float) * 2 * n, hipMemcpyHostToDevice), "state upload");
    if (ra_align(L.eng, L.d_sbj, n, L.d_state, L.d_res, nullptr)) die("ra_align");
}

void fetch_results(int start, int stop)
{
    const int n = stop - start;
    if (ra_sync(L.eng)) die("sync");
    hip_or_die(hipMemcpy(L.h_res, L.d_res, sizeof(ra_result) * n, hipMemcpyDeviceToHost), "result download");
    hip_or_die(hipMemcpy(L.h_state, L.d_state, sizeof(float) * 2 * n, hipMemcpyDeviceToHost), "state download");
    for (int i = 0; i < n; i++) {
        AlignParam &a = L.h_param[start + i];
        a.ref_id = L.h_res[i].ref_id;
        a.shift_x = L.h_state[2 * i]; a.shift_y = L.h_state[2 * i + 1];
        a.angle = L.h_res[i].alpha;
        a.mirror = L.h_res[i].mirror != 0;
    }
}
}  // namespace

extern "C" void print_gpu_info(const unsigned int device_idx)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, (int)device_idx) != hipSuccess) { printf("GPU[%u]: not available\n", device_idx); return; }
    size_t fr = 0, tot = 0;
    (void)hipSetDevice((int)device_idx);
    (void)hipMemGetInfo(&fr, &tot);
    printf("GPU[%u]: %s (%s), %d CUs, %.1f GiB total, %.1f GiB free, LDS/CU %zu KB, wave %d\n", device_idx, p.name,
           p.gcnArchName, p.multiProcessorCount, tot / 1073741824.0, fr / 1073741824.0,
           (size_t)p.maxSharedMemoryPerMultiProcessor / 1024, p.warpSize);
}

extern "C" void gpu_clear(void)
{
    if (L.eng) { ra_destroy(L.eng); L.eng = nullptr; }
    if (L.h_param) (void)hipHostFree(L.h_param);
    if (L.h_stage) (void)hipHostFree(L.h_stage);
    if (L.h_state) (void)hipHostFree(L.h_state);
    if (L.h_sums) (void)hipHostFree(L.h_sums);
    if (L.h_counts) (void)hipHostFree(L.h_counts);
    if (L.h_res) (void)hipHostFree(L.h_res);
    for (void *p : {(void *)L.d_sbj, (void *)L.d_ref, (void *)L.d_aligned, (void *)L.d_state, (void *)L.d_sums,
                    (void *)L.d_counts, (void *)L.d_res, (void *)L.d_cid_idx, (void *)L.d_cls})
        if (p) (void)hipFree(p);
    int dev = L.device;
    L = Legacy();
    L.device = dev;   // the reference pins the process to one device id (gpu_aln_noref.cu:105)
}

extern "C" AlignParam *pre_align_init(const unsigned int num_particles, const AlignConfig *aln_cfg,
                                      const unsigned int device_id)
{
    if (!aln_cfg) { fprintf(stderr, "libralign_hip: pre_align_init: null config\n"); exit(EXIT_FAILURE); }
    if (L.device != -1 && L.device != (int)device_id) {
        fprintf(stderr, "libralign_hip: device id may not change within a process\n");
        exit(EXIT_FAILURE);
    }
    if (L.eng) gpu_clear();
    L.device = (int)device_id;
    L.cfg = *aln_cfg;
    L.num_particles = num_particles;
    ra_config rc = legacy_config(aln_cfg, device_id, RA_MODE_MREF);
    rc.chunk = (int)std::min<unsigned>(8192, std::max(2u, aln_cfg->sbj_num));
    if (ra_create(&L.eng, &rc)) die("pre_align_init");
    const size_t npix = (size_t)aln_cfg->img_dim * aln_cfg->img_dim;
    const size_t B = aln_cfg->sbj_num, R = aln_cfg->ref_num;
    hip_or_die(hipHostMalloc((void **)&L.h_param, sizeof(AlignParam) * std::max(1u, num_particles)), "param alloc");
    for (unsigned i = 0; i < num_particles; i++) {
        L.h_param[i].sbj_id = -1; L.h_param[i].ref_id = 0; L.h_param[i].shift_x = 0; L.h_param[i].shift_y = 0;
        L.h_param[i].angle = 0; L.h_param[i].mirror = false;
    }
    L.stage_imgs = std::max(B, R);
    hip_or_die(hipHostMalloc((void **)&L.h_stage, L.stage_imgs * npix * sizeof(float)), "stage alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_state, B * 2 * sizeof(float)), "state alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_res, B * sizeof(ra_result)), "res alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_sums, R * 2 * npix * sizeof(float)), "sums alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_counts, R * sizeof(int)), "counts alloc");
    hip_or_die(hipMalloc((void **)&L.d_sbj, B * npix * sizeof(float)), "sbj alloc");
    hip_or_die(hipMalloc((void **)&L.d_aligned, B * npix * sizeof(float)), "aligned alloc");
    hip_or_die(hipMalloc((void **)&L.d_ref, R * npix * sizeof(float)), "ref alloc");
    hip_or_die(hipMalloc((void **)&L.d_state, B * 2 * sizeof(float)), "state alloc");
    hip_or_die(hipMalloc((void **)&L.d_res, B * sizeof(ra_result)), "res alloc");
    hip_or_die(hipMalloc((void **)&L.d_sums, R * 2 * npix * sizeof(float)), "sums alloc");
    hip_or_die(hipMalloc((void **)&L.d_counts, R * sizeof(int)), "counts alloc");
    hip_or_die(hipMemset(L.d_res, 0, B * sizeof(ra_result)), "res clear");
    return L.h_param;
}

extern "C" bool pre_align_size_check(const unsigned int num_particles, const AlignConfig *cfg,
                                     const unsigned int device_id, const float request, const bool verbose)
{
    if (!cfg) return false;
    if (hipSetDevice((int)device_id) != hipSuccess) return false;
    size_t need = legacy_bytes(num_particles, cfg);
    size_t fr = 0, tot = 0;
    if (need == (size_t)-1 || hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    if (verbose)
        printf("GPU[%u] SIZE CHECK: need %zu MB of %zu MB free (request %.2f)\n", device_id, need >> 20, fr >> 20, request);
    return (double)need <= (double)fr * request;
}

extern "C" void pre_align_fetch(const float **img_data, const unsigned int img_num, const char *batch_type)
{
    if (!L.eng) { fprintf(stderr, "libralign_hip: pre_align_fetch before pre_align_init\n"); exit(EXIT_FAILURE); }
    const size_t npix = (size_t)L.cfg.img_dim * L.cfg.img_dim;
    const bool is_sbj = batch_type && strcmp(batch_type, "sbj_batch") == 0;
    const bool is_ref = batch_type && strcmp(batch_type, "ref_batch") == 0;
    if (!is_sbj && !is_ref) {
        // same message and behaviour as gpu_aln_noref.cu:373-376
        printf("ERROR! fetch_data() :: Unknown batch type '%s' specified.\n", batch_type ? batch_type : "(null)");
        return;
    }
    const unsigned cap = is_sbj ? L.cfg.sbj_num : L.cfg.ref_num;
    if (img_num > cap || !img_data) { fprintf(stderr, "libralign_hip: pre_align_fetch: %u images exceed the batch (%u)\n", img_num, cap); exit(EXIT_FAILURE); }
    // gather into one pinned block and ship with a single copy
    for (unsigned i = 0; i < img_num; i++) memcpy(L.h_stage + (size_t)i * npix, img_data[i], npix * sizeof(float));
    float *dst = is_sbj ? L.d_sbj : L.d_ref;
    hip_or_die(hipMemcpy(dst, L.h_stage, (size_t)img_num * npix * sizeof(float), hipMemcpyHostToDevice), "image upload");
    if (is_sbj) L.sbj_loaded = img_num;
    else if (ra_set_references(L.eng, L.d_ref)) die("ra_set_references");
}

extern "C" void pre_align_run(const int start_idx, const int stop_idx)
{
    run_search(start_idx, stop_idx, RA_MODE_REFFREE);
    fetch_results(start_idx, stop_idx);
}

extern "C" void *pre_align_run_m(const int start_idx, const int stop_idx)
{
    run_search(start_idx, stop_idx, RA_MODE_REFFREE);
    if (ra_transform_accumulate(L.eng, L.d_sbj, stop_idx - start_idx, start_idx, L.d_res, L.d_aligned, nullptr, nullptr)) die("transform");
    fetch_results(start_idx, stop_idx);
    return L.d_aligned;
}

extern "C" void *mref_align_run(const int start_idx, const int stop_idx)
{
    run_search(start_idx, stop_idx, RA_MODE_MREF);
    if (ra_transform_accumulate(L.eng, L.d_sbj, stop_idx - start_idx, start_idx, L.d_res, L.d_aligned, nullptr, nullptr)) die("transform");
    fetch_results(start_idx, stop_idx);
    return L.d_aligned;
}

extern "C" float *mref_align_run_m(const int start_idx, const int stop_idx)
{
    const size_t npix = (size_t)L.cfg.img_dim * L.cfg.img_dim, R = L.cfg.ref_num;
    run_search(start_idx, stop_idx, RA_MODE_MREF);
    hip_or_die(hipMemsetAsync(L.d_sums, 0, R * 2 * npix * sizeof(float), L.eng->stream), "sums clear");
    hip_or_die(hipMemsetAsync(L.d_counts, 0, R * sizeof(int), L.eng->stream), "counts clear");
    if (ra_transform_accumulate(L.eng, L.d_sbj, stop_idx - start_idx, start_idx, L.d_res, L.d_aligned, L.d_sums, L.d_counts)) die("transform");
    fetch_results(start_idx, stop_idx);
    // reference layout: all even averages, then all odd ones (test_mref_cheng_yu_bdb_cuda.py:550-551)
    std::vector<float> tmp(R * 2 * npix);
    hip_or_die(hipMemcpy(tmp.data(), L.d_sums, tmp.size() * sizeof(float), hipMemcpyDeviceToHost), "sums download");
    for (size_t r = 0; r < R; r++) {
        memcpy(L.h_sums + r * npix, tmp.data() + (r * 2) * npix, npix * sizeof(float));
        memcpy(L.h_sums + (R + r) * npix, tmp.data() + (r * 2 + 1) * npix, npix * sizeof(float));
    }
    hip_or_die(hipMemcpy(L.h_counts, L.d_counts, R * sizeof(int), hipMemcpyDeviceToHost), "counts download");
    return L.h_sums;
}

extern "C" int *get_num_ref(void) { return L.h_counts; }

extern "C" void reset_shifts(const float shift_range, const float shift_step)
{
    if (!L.eng) { fprintf(stderr, "libralign_hip: reset_shifts before pre_align_init\n"); exit(EXIT_FAILURE); }
    if (ra_reset_shifts(L.eng, shift_range, shift_range, shift_step)) die("reset_shifts");
}

// ---------------------------------------------------------------------------------------------
// class-resident reference-free alignment (cuda/gpu_aln_noref.h:94-109, gpu_aln_noref.cu:559-782; SURVEY.md
// section 8 row f-3): particles arrive sorted by class, every particle is aligned to the average of its own class
// (single-reference search with sp_alignment.ormq semantics), transformed, and the class averages are rebuilt
// on the device from the aligned images; ref_free_alignment_2D_filter_references applies the tangent low-pass.

// mean of the aligned images of the contiguous class range [cid_idx[r], cid_idx[r+1]) in particle order
// (cu_average_batch, gpu_aln_noref.cu:1199-1229); an empty class keeps its previous reference
__global__ __launch_bounds__(256) void class_mean_kernel(int npix, const float *__restrict__ aligned,
                                                         const unsigned *__restrict__ cid_idx, float *__restrict__ refs)
{
    const int r = blockIdx.x;
    const unsigned b = cid_idx[r], e = cid_idx[r + 1];
    if (e <= b) return;
    for (int pix = blockIdx.y * blockDim.x + threadIdx.x; pix < npix; pix += gridDim.y * blockDim.x) {
        float avg = 0.f;
        for (unsigned i = b; i < e; i++) avg += aligned[(size_t)i * npix + pix];
        refs[(size_t)r * npix + pix] = avg / (float)(e - b);
    }
}

static size_t isac_bytes(const AlignConfig *c)
{
    AlignConfig one = *c;
    one.ref_num = 1;
    const size_t npix = (size_t)c->img_dim * c->img_dim;
    size_t need = legacy_bytes(c->sbj_num, &one);
    if (need == (size_t)-1) return need;
    return need + (size_t)c->ref_num * npix * 4 + ((size_t)c->ref_num + 1) * 4;
}

extern "C" AlignParam *ref_free_alignment_2D_init(const AlignConfig *aln_cfg, const float **sbj_data_list,
                                                  const float **ref_data_list, const int *sbj_cid_list,
                                                  const unsigned int device_id)
{
    if (!aln_cfg || !sbj_data_list || !ref_data_list || !sbj_cid_list) {
        fprintf(stderr, "libralign_hip: ref_free_alignment_2D_init: null argument\n");
        exit(EXIT_FAILURE);
    }
    if (L.device != -1 && L.device != (int)device_id) {
        fprintf(stderr, "libralign_hip: device id may not change within a process\n");
        exit(EXIT_FAILURE);
    }
    if (L.eng) gpu_clear();
    L.device = (int)device_id;
    L.cfg = *aln_cfg;
    L.num_particles = aln_cfg->sbj_num;
    L.isac = true;
    const size_t npix = (size_t)aln_cfg->img_dim * aln_cfg->img_dim;
    const size_t B = aln_cfg->sbj_num, R = aln_cfg->ref_num;
    // class index list as the reference builds it (gpu_aln_noref.cu:611-620): a new class starts where the id changes
    L.cid_idx.assign(R + 1, (unsigned)B);
    {
        int cid = -1; size_t idx = 0;
        for (size_t i = 0; i < B; i++)
            if (sbj_cid_list[i] != cid) {
                if (idx >= R) { fprintf(stderr, "libralign_hip: ref_free_alignment_2D_init: more class runs than references\n"); exit(EXIT_FAILURE); }
                L.cid_idx[idx++] = (unsigned)i; cid = sbj_cid_list[i];
            }
    }
    ra_config rc = legacy_config(aln_cfg, device_id, RA_MODE_REFFREE);
    rc.nref = 1;
    rc.chunk = (int)std::min<unsigned>(8192, std::max(2u, aln_cfg->sbj_num));
    if (ra_create(&L.eng, &rc)) die("ref_free_alignment_2D_init");
    hip_or_die(hipHostMalloc((void **)&L.h_param, sizeof(AlignParam) * std::max<size_t>(1, B)), "param alloc");
    for (size_t i = 0; i < B; i++) {
        L.h_param[i].sbj_id = -1; L.h_param[i].ref_id = sbj_cid_list[i]; L.h_param[i].shift_x = 0; L.h_param[i].shift_y = 0;
        L.h_param[i].angle = 0; L.h_param[i].mirror = false;
    }
    L.stage_imgs = std::max(B, R);
    hip_or_die(hipHostMalloc((void **)&L.h_stage, L.stage_imgs * npix * sizeof(float)), "stage alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_state, B * 2 * sizeof(float)), "state alloc");
    hip_or_die(hipHostMalloc((void **)&L.h_res, B * sizeof(ra_result)), "res alloc");
    hip_or_die(hipMalloc((void **)&L.d_sbj, B * npix * sizeof(float)), "sbj alloc");
    hip_or_die(hipMalloc((void **)&L.d_aligned, B * npix * sizeof(float)), "aligned alloc");
    hip_or_die(hipMalloc((void **)&L.d_ref, R * npix * sizeof(float)), "ref alloc");
    hip_or_die(hipMalloc((void **)&L.d_state, B * 2 * sizeof(float)), "state alloc");
    hip_or_die(hipMalloc((void **)&L.d_res, B * sizeof(ra_result)), "res alloc");
    hip_or_die(hipMalloc((void **)&L.d_cid_idx, (R + 1) * sizeof(unsigned)), "cid alloc");
    hip_or_die(hipMemset(L.d_res, 0, B * sizeof(ra_result)), "res clear");
    hip_or_die(hipMemcpy(L.d_cid_idx, L.cid_idx.data(), (R + 1) * sizeof(unsigned), hipMemcpyHostToDevice), "cid upload");
    {
        std::vector<int> cls(B);
        for (unsigned r = 0; r < R; r++)
            for (unsigned i = L.cid_idx[r]; i < L.cid_idx[r + 1]; i++) cls[i] = (int)r;
        hip_or_die(hipMalloc((void **)&L.d_cls, B * sizeof(int)), "class index alloc");
        hip_or_die(hipMemcpy(L.d_cls, cls.data(), B * sizeof(int), hipMemcpyHostToDevice), "class index upload");
    }
    for (size_t i = 0; i < B; i++) memcpy(L.h_stage + i * npix, sbj_data_list[i], npix * sizeof(float));
    hip_or_die(hipMemcpy(L.d_sbj, L.h_stage, B * npix * sizeof(float), hipMemcpyHostToDevice), "image upload");
    for (size_t i = 0; i < R; i++) memcpy(L.h_stage + i * npix, ref_data_list[i], npix * sizeof(float));
    hip_or_die(hipMemcpy(L.d_ref, L.h_stage, R * npix * sizeof(float), hipMemcpyHostToDevice), "reference upload");
    L.sbj_loaded = (unsigned)B;
    return L.h_param;
}

extern "C" bool ref_free_alignment_2D_size_check(const AlignConfig *cfg, const unsigned int device_id, const float request,
                                                 const bool verbose)
{
    if (!cfg) return false;
    if (hipSetDevice((int)device_id) != hipSuccess) return false;
    const size_t need = isac_bytes(cfg);
    size_t fr = 0, tot = 0;
    if (need == (size_t)-1 || hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    if (verbose)
        printf("GPU[%u] SIZE CHECK: need %zu MB of %zu MB free (request %.2f)\n", device_id, need >> 20, fr >> 20, request);
    return (double)need <= (double)fr * request;
}

extern "C" void ref_free_alignment_2D(void)
{
    if (!L.eng || !L.isac) { fprintf(stderr, "libralign_hip: ref_free_alignment_2D before ref_free_alignment_2D_init\n"); exit(EXIT_FAILURE); }
    const size_t npix = (size_t)L.cfg.img_dim * L.cfg.img_dim;
    const unsigned B = L.cfg.sbj_num, R = L.cfg.ref_num;
    for (unsigned i = 0; i < B; i++) { L.h_state[2 * i] = L.h_param[i].shift_x; L.h_state[2 * i + 1] = L.h_param[i].shift_y; }
    hip_or_die(hipMemcpy(L.d_state, L.h_state, sizeof(float) * 2 * B, hipMemcpyHostToDevice), "state upload");
    // all classes in one launch where the fused search kernel covers the geometry, class by class otherwise
    if (ra_set_class_references(L.eng, L.d_ref, (int)R) == RA_OK) {
        if (ra_align_classes(L.eng, L.d_sbj, (int)B, L.d_state, L.d_res, L.d_cls)) die("ra_align_classes");
    } else {
        for (unsigned r = 0; r < R; r++) {
            const unsigned b = L.cid_idx[r], e = L.cid_idx[r + 1];
            if (e <= b) continue;
            if (ra_set_references(L.eng, L.d_ref + (size_t)r * npix)) die("ra_set_references");
            if (ra_align(L.eng, L.d_sbj + (size_t)b * npix, (int)(e - b), L.d_state + 2 * (size_t)b, L.d_res + b, nullptr)) die("ra_align");
        }
    }
    if (ra_transform_accumulate(L.eng, L.d_sbj, (int)B, 0, L.d_res, L.d_aligned, nullptr, nullptr)) die("transform");
    hipLaunchKernelGGL(class_mean_kernel, dim3(R, 8), dim3(256), 0, L.eng->stream, (int)npix, L.d_aligned, L.d_cid_idx, L.d_ref);
    hip_or_die(hipGetLastError(), "class_mean_kernel");
    if (ra_sync(L.eng)) die("sync");
    hip_or_die(hipMemcpy(L.h_res, L.d_res, sizeof(ra_result) * B, hipMemcpyDeviceToHost), "result download");
    hip_or_die(hipMemcpy(L.h_state, L.d_state, sizeof(float) * 2 * B, hipMemcpyDeviceToHost), "state download");
    for (unsigned i = 0; i < B; i++) {      // ref_id keeps the class id given at init (gpu_aln_noref.cu:607-608)
        AlignParam &a = L.h_param[i];
        a.shift_x = L.h_state[2 * i]; a.shift_y = L.h_state[2 * i + 1];
        a.angle = L.h_res[i].alpha;
        a.mirror = L.h_res[i].mirror != 0;
    }
}

extern "C" void ref_free_alignment_2D_filter_references(const float cutoff_freq, const float falloff)
{
    if (!L.eng || !L.isac) { fprintf(stderr, "libralign_hip: filter_references before ref_free_alignment_2D_init\n"); exit(EXIT_FAILURE); }
    if (ra_filter_references(L.eng, L.d_ref, (int)L.cfg.ref_num, cutoff_freq, falloff, 0, nullptr, 0, nullptr)) die("ra_filter_references");
    if (ra_sync(L.eng)) die("sync");
}

// extension (not in the reference header): copy the current class averages [ref_num][nx][nx] to host memory
extern "C" int ra_isac_get_references(float *h_out)
{
    if (!L.eng || !L.isac || !h_out) { g_last_error = "class-resident mode is not initialised"; return RA_ERR_STATE; }
    const size_t npix = (size_t)L.cfg.img_dim * L.cfg.img_dim;
    RA_HIP(hipMemcpy(h_out, L.d_ref, (size_t)L.cfg.ref_num * npix * sizeof(float), hipMemcpyDeviceToHost));
    return RA_OK;
}

// diagnostic: the device-memory estimate behind pre_align_size_check, in bytes ((size_t)-1 = bad geometry)
extern "C" size_t ra_legacy_bytes(const unsigned int num_particles, const AlignConfig *cfg)
{
    return cfg ? legacy_bytes(num_particles, cfg) : (size_t)-1;
}
