/*
 * ralign.h -- C ABI of the MI355X-native 2-D alignment engine (libralign_hip.so).
 *
 * Two layers, both plain C (pointers and sizes only):
 *
 *  (1) the reference's own ctypes surface, symbol for symbol
 *      (/root/reference/cuda/gpu_aln_noref.h:52-113, cuda/gpu_aln_common.h:62-83,
 *       ctypes mirrors at test_mref_gpu_align.py:91-149), so that the reference drivers can
 *      load this library in place of cuda/gpu_aln_pack.so;
 *  (2) a handle-based `ra_*` API with int error codes that takes DEVICE pointers
 *      (inputs already resident in HBM) and an optional HIP stream; this is what the
 *      Python host side (cryo_ralib_amd/) and bench.py call.
 *
 * Results follow the EMAN2 CPU path (Util.multiref_polar_ali_2d / ormq semantics),
 * the API shape follows the reference's CUDA library.  See DESIGN.md.
 */
#ifndef RALIGN_H
#define RALIGN_H

#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ PODs */

/* reference: cuda/gpu_aln_common.h:62-74 ; ctypes test_mref_gpu_align.py:112-123 (32 bytes) */
typedef struct AlignConfig {
    unsigned int sbj_num;    /* particles per device batch                       */
    unsigned int ref_num;    /* references                                       */
    unsigned int img_dim;    /* nx (square images)                               */
    unsigned int ring_num;   /* numr[-3]: outer radius `ou` (rings 1..ring_num)  */
    unsigned int ring_len;   /* reference passes 256; informational here: ring lengths follow Numrinit */
    float shift_step;
    float shift_rng_x;
    float shift_rng_y;
} AlignConfig;

/* reference: cuda/gpu_aln_common.h:76-83 ; ctypes test_mref_gpu_align.py:125-134 (24 bytes).
 * shift_x/shift_y hold the accumulated centre offset (== inverse_transform2 of xform.align2d);
 * it is both input and output of every *_run call.  The caller converts to EMAN2 parameters
 * exactly as test_mref_gpu_align.py:578-588 does. */
typedef struct AlignParam {
    int   sbj_id;
    int   ref_id;
    float shift_x;
    float shift_y;
    float angle;
    bool  mirror;
} AlignParam;

/* ------------------------------------------- (1) reference-compatible names */

/* cuda/gpu_aln_noref.h:52 */
void print_gpu_info(const unsigned int device_idx);
/* cuda/gpu_aln_noref.h:58 */
void gpu_clear(void);
/* cuda/gpu_aln_noref.h:62-65.  Returns host-visible (pinned) memory of num_particles entries. */
AlignParam *pre_align_init(const unsigned int num_particles, const AlignConfig *aln_cfg,
                           const unsigned int device_id);
/* cuda/gpu_aln_noref.h:67-72 */
bool pre_align_size_check(const unsigned int num_particles, const AlignConfig *cfg,
                          const unsigned int device_id, const float request, const bool verbose);
/* cuda/gpu_aln_noref.h:74-77 ; batch_type "sbj_batch" | "ref_batch" */
void pre_align_fetch(const float **img_data, const unsigned int img_num, const char *batch_type);
/* cuda/gpu_aln_noref.h:81 : single-reference search, parameters only */
void pre_align_run(const int start_idx, const int stop_idx);
/* cuda/gpu_aln_noref.h:82 : single-reference search + transform; returns DEVICE pointer to the
 * aligned images [stop-start][nx][nx] (borrowed until the next run/fetch) */
void *pre_align_run_m(const int start_idx, const int stop_idx);
/* cuda/gpu_aln_noref.h:83 : multi-reference search + transform; same return */
void *mref_align_run(const int start_idx, const int stop_idx);
/* cuda/gpu_aln_noref.h:84 : multi-reference search + transform + per-class even/odd sums;
 * returns HOST-readable float[2][R][nx][nx] (even block, then odd block;
 * test_mref_cheng_yu_bdb_cuda.py:550-551) */
float *mref_align_run_m(const int start_idx, const int stop_idx);
/* cuda/gpu_aln_noref.h:113 : per-class member counts of the last mref_align_run_m */
int *get_num_ref(void);
/* cuda/gpu_aln_noref.cu:119 (exported, not in the header) */
void reset_shifts(const float shift_range, const float shift_step);

/* class-resident reference-free alignment (GPU-ISAC), cuda/gpu_aln_noref.h:94-109, gpu_aln_noref.cu:559-782:
 * particles sorted by class (sbj_cid_list non-decreasing runs), one reference per class; every call aligns each
 * particle to the average of its own class, transforms it and rebuilds the class averages on the device.
 * cuda/gpu_aln_noref.h:94-99.  Returns host-visible AlignParam[sbj_num]; ref_id holds the class id. */
AlignParam *ref_free_alignment_2D_init(const AlignConfig *aln_cfg, const float **sbj_data_list,
                                       const float **ref_data_list, const int *sbj_cid_list,
                                       const unsigned int device_id);
/* cuda/gpu_aln_noref.h:101-105 */
bool ref_free_alignment_2D_size_check(const AlignConfig *cfg, const unsigned int device_id, const float request,
                                      const bool verbose);
/* cuda/gpu_aln_noref.h:107 */
void ref_free_alignment_2D(void);
/* cuda/gpu_aln_noref.h:109 ; tangent low-pass of every class average (gpu_aln_noref.cu:786-816) */
void ref_free_alignment_2D_filter_references(const float cutoff_freq, const float falloff);

/* ------------------------------------------------------ (2) handle-based API */

#define RA_OK            0
#define RA_ERR_ARG      -1   /* invalid argument / geometry                   */
#define RA_ERR_HIP      -2   /* HIP runtime failure (message via ra_last_error) */
#define RA_ERR_STATE    -3   /* call out of protocol order                    */
#define RA_ERR_NOMEM    -4

#define RA_MODE_MREF     0   /* Util.multiref_polar_ali_2d: Normalize_ring, out-of-range shifts reset */
#define RA_MODE_REFFREE  1   /* sp_alignment.ormq: no Normalize_ring, shifts clamped                 */

typedef struct ra_config {
    int   nx;             /* image size (square)                                   */
    int   first_ring;     /* --ir                                                  */
    int   last_ring;      /* --ou                                                  */
    int   ring_skip;      /* --rs                                                  */
    float xrng, yrng;     /* --xr --yr                                             */
    float step;           /* --ts                                                  */
    int   nref;           /* references (1 in RA_MODE_REFFREE)                     */
    int   mode;           /* RA_MODE_*                                             */
    int   device;         /* HIP ordinal                                           */
    int   chunk;          /* particles per internal pass (0 = auto)                */
} ra_config;

typedef struct ra_engine ra_engine;

/* per-particle result record written by ra_align (device or host memory, 32 bytes) */
typedef struct ra_result {
    float alpha, sx, sy;  /* EMAN2 xform.align2d parameters (combine_params2 output) */
    int   mirror;
    int   ref_id;
    float peak;           /* CCF peak (qn or qm)                                   */
    int   angle_bin;      /* jtot: 1-based integer angular bin of the peak         */
    int   shift_idx;      /* index of the winning search offset (y outer, x inner) */
} ra_result;

/* Hedges for the two choices of the EMAN2 CPU path that the reference tree does not pin (its arithmetic lives in EMAN2 2.31, which
 * is not part of the reference: SURVEY.md Appendix A.3 / A.4; call sites test_mref_gpu_align.py:1015, 1043-1044):
 *   interp          Util::alrl_ms's interpolation.  RA_INTERP_BILINEAR (Util::bilinear, EMAN2 2.31; default) or RA_INTERP_QUADRI
 *                   (Util::quadri, older releases).  Quadri runs through the size-generic kernels (ra_search_path == 2) whatever
 *                   the geometry -- the particle-resident kernels sample bilinearly only --, and the exact re-evaluation of
 *                   ra_set_refine follows the option.
 *   normalize_ring  Util::Normalize_ring between Polar2Dm and Frngs: -1 = by mode (RA_MODE_MREF on, as inside
 *                   Util.multiref_polar_ali_2d; RA_MODE_REFFREE off, as sp_alignment.ormq), 0 = off, 1 = on.  Every kernel family
 *                   honours it; everything else stays the mode's: the search window rule (mref_ali2d resets a shift beyond
 *                   cnx - last_ring - 2 and cuts the windows with its last_ring argument; ali2d_single_iter clamps, with
 *                   ou = numr[-3]) and the scan over offsets and references (Util::multiref_polar_ali_2d compares every candidate
 *                   with its running peak ROUNDED TO FLOAT; ormq keeps a double) -- both reproduced literally, down to float ties. */
#define RA_INTERP_BILINEAR 0
#define RA_INTERP_QUADRI   1
typedef struct ra_options {
    int interp;           /* RA_INTERP_*                      */
    int normalize_ring;   /* -1 by mode (default), 0 off, 1 on */
} ra_options;

const char *ra_last_error(void);
int  ra_create(ra_engine **out, const ra_config *cfg);
/* ra_create with options (NULL: the defaults = ra_create) */
int  ra_create_ex(ra_engine **out, const ra_config *cfg, const ra_options *opt);
/* switch Normalize_ring for subsequent ra_align calls (flag < 0: back to the mode's default); any time, any kernel path */
int  ra_set_normalize_ring(ra_engine *e, int flag);
/* the options in force (normalize_ring resolved to 0 / 1) */
int  ra_get_options(const ra_engine *e, ra_options *opt);
void ra_destroy(ra_engine *e);
/* use `hip_stream` (a hipStream_t) for all subsequent work; NULL = default stream */
int  ra_set_stream(ra_engine *e, void *hip_stream);
/* geometry queries */
int  ra_num_shifts(const ra_engine *e);
int  ra_maxrin(const ra_engine *e);
int  ra_lcirc(const ra_engine *e);
/* which kernels ra_align runs for the current geometry and window: 1 = particle-resident fused / tiled search kernel (four
 * offsets per pass: rings of up to 256 samples, ou <= 36 -- <= 39 with at most 16 references --, in any box: beyond ~93 pixels over a crop of the
 * image that follows the particle's centre), 0 = polar + contraction kernel pair, 2 = size-generic kernels (rings beyond 512
 * samples, more than 64 rings), 3 = particle-resident search with one or two ring buffers next to the image (search_pair_kernel:
 * ou = 40, and 37 .. 39 with more than 16 references; search_solo_kernel / search_duo_kernel: rings of 512 samples, ou = 41 .. 60) */
int  ra_search_path(const ra_engine *e);
/* with ra_search_path == 3: search offsets per pass, 2 (search_duo_kernel, the default) or 1 (search_solo_kernel); 0 otherwise */
int  ra_search_offsets_per_pass(const ra_engine *e);
/* 1 when the search kernels evaluate the IN-WINDOW offsets of a particle only (sp_alignment.search_range, as the CPU path does): the
 * solo / duo / pair kernels and, since round 6, the size-generic kernels (live-offset lists); 0 when every offset of the list is
 * computed and the out-of-window ones are masked afterwards (fused / tiled kernels, kernel pair) */
int  ra_search_skips_offsets(const ra_engine *e);
/* 1 when the particle-resident path is search_tiled_kernel (reference tiles: 15 and more references), 0 otherwise */
int  ra_search_tiled(const ra_engine *e);
/* change the search window without re-allocating (reset_shifts analogue); the number of
 * offsets may not grow beyond what ra_create sized */
int  ra_reset_shifts(ra_engine *e, float xrng, float yrng, float step);

/* a user mask [nx][nx] (device) instead of model_circle(last_ring) for normalize.mask and the masked statistics
 * (the drivers' optional maskfile argument, test_mref_gpu_align.py:317-321): pixels > 0.5 are inside */
int  ra_set_mask(ra_engine *e, const float *d_mask);

/* --nomirror (test_reffree_gpu_align.py:921, passed to ali2d_single_iter -> ormq): only the straight half of
 * Crosrng_ms takes part in the search (Util.Crosrng_ns); flag != 0 switches it on for subsequent ra_align calls */
int  ra_set_nomirror(ra_engine *e, int flag);

/* references: d_refs [nref][nx][nx] device, ALREADY normalised under the mask
 * (test_mref_gpu_align.py:336).  Polar transform, ring FFT and ring weights
 * (Polar2Dm/Frngs/Applyws, :1015-1017) happen on the device. */
int  ra_set_references(ra_engine *e, const float *d_refs);
/* diagnostic: copy the prepared references in EMAN2 packing [nref][lcirc] to host */
int  ra_get_prepared_references(ra_engine *e, float *h_crefim);

/* search: d_particles [n][nx][nx] device; d_state [n][2] device, accumulated centre offset
 * (in/out); d_result [n] device.  cs = average-centre correction (RA_MODE_REFFREE; NULL = 0).
 * Asynchronous on the engine's stream. */
int  ra_align(ra_engine *e, const float *d_particles, int n, float *d_state,
              ra_result *d_result, const float *cs);
/* Sub-bin angle refinement.  The search finds the winning (reference, offset, mirror, angular bin) in f32; Util::prb1d's
 * second difference c3 amplifies the f32-vs-f64 difference of the 7 CCF samples around a FLAT peak into degrees.  Particles
 * whose |c3| < threshold x max |b| get their 7 samples re-evaluated with the CPU path's own arithmetic (bilinear samples,
 * fftr_q's radix-2 ring FFTs in f32, f64 accumulation of the ring products, f64 inverse) and alpha / sx / sy rewritten.
 * With the refinement on (threshold != 0), float ties are decided the same way: a neighbouring angular bin, another search
 * offset or another reference whose f32 peak lies within 3e-6 (relative) of the winner's is re-evaluated too, and the CPU
 * path's ">=" order picks the integer assignment (angle_bin, shift_idx, ref_id, mirror, peak and d_state follow).
 * threshold < 0: every particle (+25 % of a headline iteration); 0: off; default 0.02 -- no measurable cost, no particle
 * beyond 2e-3 degrees of the CPU path on any tested workload (environment RALIGN_REFINE overrides the default).  Call before
 * ra_set_references.  Geometries whose rings exceed the LDS (256 x 256 / ou = 120: 271 KB per offset) run the same kernels on
 * global scratch.  RA_ERR_STATE only when the ring layout has an odd length (never with Numrinit's powers of two). */
int  ra_set_refine(ra_engine *e, float threshold);
/* particles the last search launch re-evaluated (flat peaks and float ties); synchronises the stream (diagnostics) */
int  ra_last_refine_count(ra_engine *e);
/* the reference's state round trip: rebuild the shift the next search starts from (d_state [n][2]) from the float32
 * parameters of the previous iteration in d_result -- inverse_transform2(alpha, sx, sy) in RA_MODE_MREF
 * (test_mref_gpu_align.py:1024-1026), combine_params2(alpha, sx, sy, mirror, 0, -cs[0], -cs[1], 0) then
 * inverse_transform2 in RA_MODE_REFFREE (ali2d_single_iter; cs = host float[2] or NULL).  Call it before ra_align
 * (then without cs) to follow the reference's loop to the rounding of the header values; without it ra_align
 * continues from the exact d_state it left. */
int  ra_state_from_params(ra_engine *e, const ra_result *d_result, int n, const float *cs, float *d_state);
/* the same with the centre correction in device memory (d_cs [2]): no host value in the path, nothing to wait for */
int  ra_state_from_params_dev(ra_engine *e, const ra_result *d_result, int n, const float *d_cs, float *d_state);
/* class-resident alignment (the ISAC mode behind ref_free_alignment_2D, cuda/gpu_aln_noref.cu:559-782): every particle
 * against the average of its own class, all classes in one launch.  ra_set_class_references prepares ncls references
 * (d_refs [ncls][nx][nx] device); ra_align_classes aligns particle i to reference d_cls[i] (device, [n]); results and
 * state as ra_align.  Needs RA_MODE_REFFREE with nref = 1 on a geometry the fused search kernel covers with the whole
 * image in LDS (ra_search_path == 1 and a box of up to ~93 pixels at ou = 36); RA_ERR_STATE otherwise -- loop over the
 * classes with ra_set_references / ra_align then. */
int  ra_set_class_references(ra_engine *e, const float *d_refs, int ncls);
int  ra_align_classes(ra_engine *e, const float *d_particles, int n, float *d_state,
                      ra_result *d_result, const int *d_cls);
/* apply rot_shift2D with the parameters in d_result and write the aligned images
 * (d_aligned [n][nx][nx], may be NULL) and/or add them into the class sums
 * (d_sums [nref][2][nx][nx] +=, d_counts [nref] +=, may be NULL);
 * even/odd = (index0 + i) % 2 (test_mref_gpu_align.py:1056). */
int  ra_transform_accumulate(ra_engine *e, const float *d_particles, int n, int index0,
                             const ra_result *d_result, float *d_aligned,
                             float *d_sums, int *d_counts);
/* new references from the (all-reduced) class sums: (even+odd)/count then
 * normalize.mask(no_sigma=1) under model_circle(last_ring)
 * (test_mref_gpu_align.py:534-535, 563).  Classes with count < min_count are left
 * untouched in d_refs (the caller re-seeds them, :523-528). */
int  ra_update_references(ra_engine *e, const float *d_sums, const int *d_counts,
                          int min_count, float *d_refs);
/* particle preprocessing on device: subtract the mean under model_circle(last_ring)
 * (normalize.mask no_sigma=0, test_mref_gpu_align.py:342), in place */
int  ra_normalize_particles(ra_engine *e, float *d_particles, int n);
/* diagnostic: run only the polar / ring-FFT stage on n <= chunk particles and return the ring
 * spectra of every search offset in EMAN2 packing, h_out [n][num_shifts][lcirc] (what
 * Polar2Dm -> Normalize_ring -> Frngs leave in `cimage` inside Util.multiref_polar_ali_2d) */
int  ra_debug_spectra(ra_engine *e, const float *d_particles, int n, const float *d_state, float *h_out);
/* ---- reference update of one iteration on the device (what the reference's main node does on the
 * CPU, test_mref_gpu_align.py:517-564 / test_reffree_gpu_align.py:374-429, default user function) */
/* length of an FSC curve: nx/2 + 1 */
int  ra_fsc_len(const ra_engine *e);
/* sp_statistics.fsc (masked = 0, :531) or fsc_mask (masked = 1, test_reffree_gpu_align.py:384) between
 * the even and odd sums of every class with count >= min_count, averaged over those classes (:537-548).
 * h_fsc [3][ra_fsc_len]: frequencies, fsc, points per shell (host). */
int  ra_class_fsc(ra_engine *e, const float *d_sums, const int *d_counts, int min_count, int masked,
                  float *h_fsc);
/* per-class curves of the last ra_class_fsc: h_all [nref][2][nx/2+1] = {fsc, points per shell} (the reference writes one
 * drm%03d%04d.txt per class and iteration, test_mref_gpu_align.py:533) */
int  ra_last_class_fsc(ra_engine *e, float *h_all);
/* sp_filter.fit_tanh(dres, low=0.1): host arithmetic; fsc is edited in place like the original. */
int  ra_fit_tanh(const float *freq, float *fsc, int n, float *fl, float *aa);
/* ra_class_fsc + the average over the live classes + fit_tanh + the clamps of ref_ali2d (fl_lo <= fl <= fl_hi, aa <= aa_hi) on
 * the DEVICE (test_mref_gpu_align.py:531-548, sp_user_functions.ref_ali2d): d_fit [5] = {fl, aa clamped; fl, aa as fitted;
 * status (1: every class below min_count)}, d_curve [3][nx/2+1] = {frequency, fsc as fit_tanh leaves it, points per shell}.
 * Asynchronous, no host round trip: ra_filter_references_dev reads (fl, aa) from d_fit. */
int  ra_class_fsc_fit(ra_engine *e, const float *d_sums, const int *d_counts, int min_count, int masked, float fl_lo,
                      float fl_hi, float aa_hi, float *d_fit, float *d_curve);
/* (even + odd) / count without normalisation (:534-535); classes below min_count untouched */
int  ra_class_averages(ra_engine *e, const float *d_sums, const int *d_counts, int min_count,
                       float *d_refs);
/* sp_user_functions.ref_ali2d body on nimg device images, in place: filt_tanl(fl, aa) (fl <= 0: no
 * filter), center_2D: center = 1 phase_cog + fshift, center = -1 fshift by -h_cs_in[i] (average-centre
 * rule, test_reffree_gpu_align.py:403-410), center = 0 none; then normalize.mask(no_sigma=1) under
 * model_circle(last_ring) if normalize != 0 (:563).  h_cs_out [nimg][2] (may be NULL) = applied centres. */
int  ra_filter_references(ra_engine *e, float *d_imgs, int nimg, float fl, float aa, int center,
                          const float *h_cs_in, int normalize, float *h_cs_out);
/* the same with every input in device memory: (fl, aa) = d_flaa[0..1] (NULL: no filter), centres of center = -1 from
 * d_cs_in [nimg][2], applied centres to d_cs_out [nimg][2] (device, may be NULL).  Asynchronous. */
int  ra_filter_references_dev(ra_engine *e, float *d_imgs, int nimg, const float *d_flaa, int center,
                              const float *d_cs_in, int normalize, float *d_cs_out);

/* class-resident mode only (extension, no counterpart in the reference header, where the averages stay in
 * device textures): copy the current class averages [ref_num][nx][nx] to host memory */
int  ra_isac_get_references(float *h_out);

/* diagnostic: the device-memory estimate (bytes) behind pre_align_size_check; (size_t)-1 for a bad geometry */
size_t ra_legacy_bytes(const unsigned int num_particles, const AlignConfig *cfg);

/* block until the engine's stream is idle */
int  ra_sync(ra_engine *e);

/* timing of the dominant kernel with HIP events on the engine's stream:
 * accumulated milliseconds and launch count since the last reset */
int  ra_kernel_time(ra_engine *e, int enable, double *ms_ccf, int *launches_ccf,
                    double *ms_polar, int *launches_polar);

#ifdef __cplusplus
}
#endif
#endif
