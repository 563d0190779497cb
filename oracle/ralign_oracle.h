/*
 * ralign_oracle.h -- CPU restatement of the EMAN2 2.31 / SPHIRE 2-D alignment
 * path that phonchi/Cryo-RAlib's CPU drivers call (path "(C)" of SURVEY.md).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic lives in EMAN2 2.31 (libEM + sphire/libpy),
 * which is NOT vendored in the reference tree and not installed here.  The
 * restatement follows the reference's call sites
 *   test_mref_gpu_align.py:1007-1060 (mref_ali2d), :738-786 (mref_ali2d_MPI)
 *   test_reffree_gpu_align.py:756-847 (ali2d_base -> ali2d_single_iter -> ormq)
 * and EMAN2's published algorithm.  It is pinned only by
 * the fragments the reference tree itself holds (tests/golden/known_answers.json):
 *   combine_params2 / inverse_transform2 known answers (cuda/EMAN2_test.ipynb 23-25),
 *   prb1d coefficients (cuda/gpu_aln_noref.cu:1434-1442),
 *   quadri_background / rot_scale_trans2D_background / mirror rule
 *   (notebook/02_CuPy_Image_Processing_rot_shift2d.ipynb cell 2),
 * and by planted-truth self-consistency tests.
 */
#ifndef RALIGN_ORACLE_H
#define RALIGN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXRING 512

/* switches for the three parity-critical uncertainties (SURVEY.md §7 "hard parts") */
#define ORC_INTERP_BILINEAR 0   /* Util::bilinear in alrl_ms (EMAN2 2.31 default) */
#define ORC_INTERP_QUADRI   1   /* Util::quadri (older releases)                  */

typedef struct {
    int   nring;
    int   maxrin;               /* numr[-1]                                   */
    int   lcirc;                /* total packed length                        */
    int   first_ring, last_ring, skip;
    int   numr[3 * ORC_MAXRING];/* (radius, 1-based offset, length) triplets  */
    float wr[ORC_MAXRING];      /* ringwe                                     */
} orc_rings;

/* sp_alignment.Numrinit + ringwe, mode "F"  (test_mref_gpu_align.py:984-985) */
int  orc_rings_init(orc_rings *rg, int first_ring, int last_ring, int skip);

/* Util::Polar2Dm -> alrl_ms (test_mref_gpu_align.py:1015); cnx,cny 1-based SPIDER centre */
void orc_polar2dm(const float *img, int nx, int ny, float cnx, float cny,
                  const orc_rings *rg, float *circ, int interp);
/* Util::Normalize_ring (called inside Util::multiref_polar_ali_2d) */
void orc_normalize_ring(float *circ, const orc_rings *rg);
/* Util::Frngs : in-place forward real FFT of every ring, EMAN2 packing */
void orc_frngs(float *circ, const orc_rings *rg);
/* Util::Applyws (test_mref_gpu_align.py:1017) */
void orc_applyws(float *circ, const orc_rings *rg);
/* Util::Crosrng_ms.  jtot_* are the 1-based integer peak bins (extra outputs). */
void orc_crosrng_ms(const float *circ1, const float *circ2, const orc_rings *rg,
                    double *qn, float *tot, double *qm, float *tmt,
                    int *jtot_n, int *jtot_m);
/* Util::prb1d (npoint = 7), Util::ang_n mode "F" */
float orc_prb1d7(const double *b);
float orc_ang_n(float tot, int maxrin);

/* extra diagnostics returned by the search routines */
typedef struct {
    float ix, iy;      /* winning search offset (sx = -ix, sy = -iy)  */
    int   jtot;        /* winning 1-based angular bin                 */
    float tot;         /* jtot + parabolic offset                     */
} orc_search_info;

/* Util::multiref_polar_ali_2d (test_mref_gpu_align.py:1043-1044).
 * crefim: [nref][lcirc] prepared references.  xrng = {left,right}, yrng likewise.
 * out = {ang, sxs, sys, mirror, nref, peak}.  normalize_ring!=0 reproduces the call
 * to Util::Normalize_ring inside the EMAN2 routine. */
void orc_multiref_polar_ali_2d(const float *img, int nx, int ny,
                               const float *crefim, int nref,
                               const float xrng[2], const float yrng[2], float step,
                               const orc_rings *rg, float cnx, float cny,
                               int interp, int normalize_ring,
                               float out[6], orc_search_info *info);
/* sp_alignment.ormq (single reference, no Normalize_ring; test_reffree_gpu_align.py:844-847).
 * out = {ang, sxs, sys, mirror, peak} */
void orc_set_nomirror(int flag);
void orc_set_ormq_normalize(int flag);   /* ormq on normalised rings (the engine option normalize_ring = 1 in reference-free mode) */
void orc_ormq(const float *img, int nx, int ny, const float *crefim,
              const float xrng[2], const float yrng[2], float step,
              const orc_rings *rg, float cnx, float cny, int interp,
              double out[5], orc_search_info *info);

/* sp_utilities.model_circle (edge_le!=0: r^2 <= radius^2 is inside) */
void orc_model_circle(float radius, int nx, int ny, float *mask, int edge_le);
/* processor "normalize.mask" {mask,no_sigma} (test_mref_gpu_align.py:336,342) */
void orc_normalize_mask(float *img, const float *mask, int npix, int no_sigma);
/* sp_fundamentals.rot_shift2D, "quadratic"/"background" (test_mref_gpu_align.py:1055) */
void orc_rot_shift2d(const float *in, float *out, int nx, int ny,
                     float ang_deg, float sx, float sy, int mirror);
/* sp_utilities.combine_params2 / inverse_transform2 (test_mref_gpu_align.py:1026,1048) */
void orc_combine_params2(double a1, double sx1, double sy1, int m1,
                         double a2, double sx2, double sy2, int m2,
                         double out[4]);
void orc_inverse_transform2(double alpha, double tx, double ty, int mirror, double out[4]);
void orc_state_from_params(const float *params, int n, int mode, const float cs[2], float *d);
/* sp_alignment.search_range; returns the pair already swapped as the caller does
 * (test_mref_gpu_align.py:1035-1038): out = {left, right} */
void orc_search_range(int n, float radius, float shift, float range, float out[2]);

/* references: normalize.mask(no_sigma=1) -> Polar2Dm -> Frngs -> Applyws
 * (test_mref_gpu_align.py:1013-1018).  refs are normalised IN PLACE. */
void orc_prepare_refs(float *refs, int nref, int nx, const float *mask,
                      const orc_rings *rg, int interp, float *crefim);

/* One iteration of the per-particle loop of mref_ali2d (test_mref_gpu_align.py:1023-1060,
 * 754-786).  State per particle is the accumulated centre offset d = (sxi, syi)
 * (== inverse_transform2 of the stored xform.align2d).
 *   particles [n][nx*nx] (already mean-subtracted under the mask)
 *   d         [n][2]   in/out
 *   params    [n][6]   out: alphan, sxn, syn, mirror, iref, peak
 *   sums      [nref][2][nx*nx] accumulated (+=) ; counts[nref] accumulated
 *   index0    global index of particles[0] (even/odd = (index0+i)%2)
 *   nthreads  OpenMP threads (1 = the scalar port; sums are then in particle order)
 */
void orc_mref_iteration(const float *particles, int n, int nx,
                        const float *crefim, int nref, const orc_rings *rg,
                        float xrng, float yrng, float step, int interp,
                        int normalize_ring,
                        float *d, float *params, orc_search_info *infos,
                        float *sums, int *counts, int index0, int nthreads);

/* Reference-free single-reference iteration (ali2d_single_iter,
 * test_reffree_gpu_align.py:844-847): same state / outputs with nref = 1;
 * cs is the average-centre shift applied before the search. */
void orc_reffree_iteration(const float *particles, int n, int nx,
                           const float *crefim, const orc_rings *rg,
                           float xrng, float yrng, float step, int interp,
                           const float cs[2],
                           float *d, float *params, orc_search_info *infos,
                           float *sums, int index0, int nthreads,
                           double sxsy_sum[2]);

#ifdef __cplusplus
}
#endif
#endif
