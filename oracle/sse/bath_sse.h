/* bath_sse.h -- SSE2 (128-bit) striped restatement of the reference's impl_sse filter kernels.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/: this is the "impl_sse-equivalent" CPU
 * baseline that bench.py times next to the GPU (cpu_baseline.kind = "impl_sse-equivalent restatement, SSE2 128-bit").
 * The reference's own impl_sse cannot be built here (it needs the un-vendored easel library), so the same
 * algorithms and data layouts are written from scratch:
 *
 *   bs_ssvfilter      <- p7_SSVFilter     impl_sse/ssvfilter.c:832-925   16 x int8 stripes, bands of diagonals in registers
 *   bs_msvfilter      <- p7_MSVFilter     impl_sse/msvfilter.c:74-208    16 x uint8 stripes, J state
 *   bs_vitfilter      <- p7_ViterbiFilter impl_sse/vitfilter.c:83-248    8 x int16 stripes, lazy-F D->D
 *   bs_vitfilter_bath <- p7_ViterbiFilter_BATH vitfilter.c:286-465       the same + hit windows
 *   bs_forward_parser <- p7_ForwardParser impl_sse/fwdback.c:132,256-463 4 x fp32 stripes, odds ratios, sparse rescaling
 *   bs_fs3_forward_parser <- p7_ForwardParser_Frameshift_3Codons impl_sse/fwdback_fs.c:97-533 (sse_fs.c)
 *
 * Validated against the scalar oracle (tests/test_sse_cpu.py): integer filters bit-exact (score and status), Forward
 * within 1e-5 relative (fp32 sums in striped order).  Striping: node k (1..M) lives in vector q = (k-1) % Q,
 * element z = (k-1) / Q (impl_sse.h:57-71).
 */
#ifndef BATH_SSE_H
#define BATH_SSE_H

#include <emmintrin.h>
#include "../bath_oracle.h"

typedef struct {
  int M, Qb, Qw, Qf;
  const bo_oprofile *om;     /* scalars (tjb_b, xw, xf change with the target length) are read from here at call time */
  __m128i *sbv;              /* [Kp][2*Qb] signed SSV costs min(rb - bias, 127) (sf_conversion, p7_oprofile.c:751), each row stored TWICE back to
                                back so that a band of diagonals that wraps around the model reads contiguous vectors */
  __m128i *rbv;              /* [Kp][Qb]   biased byte costs (MSV) */
  __m128i *rwv;              /* [Kp][Qw]   word scores */
  __m128i *twv;              /* [Qw][7] {BM,MM,IM,DM,MD,MI,II} then [Qw] DD (impl_sse.h:73 order) */
  __m128  *rfv;              /* [Kp][Qf]   odds ratios */
  __m128  *tfv;              /* as twv */
  __m128i *dpb;              /* row buffers, grown on demand */
  __m128i *dpw;
  __m128  *dpf;
  void    *mem[8];
} bs_oprofile;

bs_oprofile *bs_oprofile_create(const bo_oprofile *om);     /* p7_oprofile_Convert's striping, p7_oprofile.c:667-921 */
void         bs_oprofile_free(bs_oprofile *so);

int bs_ssvfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc);
int bs_msvfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc);            /* SSV first, full MSV on eslENORESULT */
int bs_msv_full(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc);            /* the byte recurrence with the J state, msvfilter.c:106-207 */
int bs_vitfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc);
int bs_vitfilter_bath(const uint8_t *dsq, int L, bs_oprofile *so, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc);
int bs_forward_parser(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc);

/* the oracle's cascade (oracle/pipeline.c) with these kernels in place of the scalar ones: 0 scalar, 1 SSE2 striped */
void bo_pipeline_use_sse(int on);

/* ---- the frameshift stage (sse_fs.c): p7_ForwardParser_Frameshift_3Codons, impl_sse/fwdback_fs.c:97-533, striped in probability space */
typedef struct {
  int M, Q, ncodons;
  const bo_fs_profile *gm;   /* the special-state transitions are read from here at call time (they follow the length configuration) */
  __m128 *rfv;               /* [338][Q]  emission odds ratios of every codon / quasi-codon row */
  __m128 *tfv;               /* [Q][7] {BM, MM, IM, DM of node k-1; MD, MI, II of node k} then [Q] DD */
  __m128 *rows;              /* four rows of {M, D, I} and three rows of IVX */
} bs_fsprofile;
bs_fsprofile *bs_fsprofile_create(const bo_fs_profile *gm3);
void          bs_fsprofile_free(bs_fsprofile *so);
int bs_fs3_forward_parser(const uint8_t *dsq, int L, bs_fsprofile *so, float *xmx_log /* (L+1) x 5, log space, or NULL */, float *ret_sc);
int bs_fs3_backward_parser(const uint8_t *dsq, int L, bs_fsprofile *so, float *xmx_log /* (L+1) x 5, log space, or NULL */, float *ret_sc);   /* fwdback_fs.c:565 */
/* the oracle's --fs pipeline (fs_pipeline.c, fs_domaindef.c) with the striped Forward parser in place of the scalar log-space one */
void bo_fs_use_sse(int on);

#endif
