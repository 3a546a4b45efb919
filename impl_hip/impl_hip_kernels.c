/* impl_hip_kernels.c -- the single-target kernel prototypes of impl_sse.h on the MI355X backend: every function is one call of
 * the batched C ABI (include/bath_hip.h) on a block of ONE target (see impl_hip.h for why this layer exists).
 *
 *   filters      p7_MSVFilter, p7_SSVFilter, p7_SSVFilter_BATH, p7_ViterbiFilter, p7_ViterbiFilter_BATH
 *   parsers      p7_ForwardParser, p7_BackwardParser, p7_ForwardParser_Frameshift_3Codons, p7_BackwardParser_Frameshift_3Codons,
 *                p7_DomainDecoding, p7_DomainDecoding_Frameshift
 *   envelopes    p7_Forward / p7_Backward / p7_Decoding / p7_OptimalAccuracy / p7_OATrace / p7_Null2_ByExpectation and their
 *                _Frameshift twins, p7_Null2_ByTrace, p7_StochasticTrace, p7_StochasticTrace_Frameshift
 *
 * Device passes.  The reference calls Forward, Backward, Decoding, OptimalAccuracy, OATrace, Null2 one after the other on the
 * same envelope, handing the matrices from call to call (p7_domaindef.c:1016-1083, 1209-1262).  On the GPU these stages are one
 * fused pass over matrices that stay in HBM (bath_hip_fs5_envelopes_x, bath_hip_std_envelopes).  The first call of the sequence
 * (Forward) runs that pass; the P7_OMX objects the reference threads through the sequence carry a reference to its results
 * (struct impl_hip_pass), and each later function returns its own part -- the Backward score, the decoding status, the
 * optimal-accuracy score, the traceback (a serial walk over the posterior and optimal-accuracy matrices, done here on the
 * host copies of those two matrices exactly as the reference walks its own), the null2 vector.  A function handed a matrix
 * that carries no pass for its input throws eslEINVAL, as the reference functions do for an unprepared matrix.
 *
 * Length configuration.  The device tables hold every per-length quantity for every length and the kernels index them with
 * the target's own length, which is what p7_Pipeline_BATH configures before each call (p7_pipeline.c:1643-1645, 1446-1450;
 * p7_domaindef.c:1019, 1206).  The two places where the reference runs a kernel with ANOTHER length -- the multihit Forward of a
 * multi-domain region with the ORF's / model's saved length (p7_domaindef.c:411, 557) -- pass om->L as the configuration.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "easel.h"
#include "esl_alphabet.h"      /* esl_abc_FAvgScVec */
#include "esl_random.h"        /* esl_random, esl_rnd_FChoose */
#include "esl_vectorops.h"     /* esl_vec_FNorm, esl_vec_FLogNorm */

#include "hmmer.h"

#define IH_KP 29

/* ------------------------------------------------------------------ one-target blocks */

static int ih_block(const ESL_DSQ *dsq, int L, bath_hip_seqs **ret)
{
  int64_t off[2] = { 0, L };
  static const uint8_t none = 0;
  return bath_hip_seqs_create(impl_hip_context(), L > 0 ? dsq + 1 : &none, off, 1, ret) == BATH_OK ? eslOK : eslFAIL;
}
static int ih_status(int st) { return st; }           /* BATH_OK / BATH_ERANGE / BATH_ENORESULT are easel's codes */

/* ------------------------------------------------------------------ filters */

static int ih_filter(int (*fn)(bath_hip_ctx *, const bath_hip_oprofile *, const bath_hip_seqs *, float *, int32_t *),
                     const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, float *ret_sc)
{
  bath_hip_seqs *sq = NULL;
  int32_t status = 0;
  if (!om->dev) ESL_EXCEPTION(eslEINVAL, "profile not converted");
  if (ih_block(dsq, L, &sq) != eslOK) return eslFAIL;
  const int st = fn(impl_hip_context(), om->dev, sq, ret_sc, &status);
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context()));
  return ih_status(status);
}

int p7_SSVFilter(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, float *ret_sc)                       /* ssvfilter.c:876 */
{
  return ih_filter(bath_hip_ssvfilter, dsq, L, om, ret_sc);
}
int p7_MSVFilter(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, float *ret_sc)           /* msvfilter.c:74 */
{
  if (ox) { ox->M = om->M; ox->L = L; }
  return ih_filter(bath_hip_msvfilter, dsq, L, om, ret_sc);
}
int p7_ViterbiFilter(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, float *ret_sc)       /* vitfilter.c:83 */
{
  if (ox) { ox->M = om->M; ox->L = L; }
  return ih_filter(bath_hip_vitfilter, dsq, L, om, ret_sc);
}

static void ih_windows(P7_HMM_WINDOWLIST *wl, const bath_hmm_window *w, int64_t n, int L, int with_score)
{
  for (int64_t i = 0; i < n; i++)
    p7_hmmwindow_new(wl, 0, (uint32_t) w[i].n, (uint32_t) w[i].k, (uint32_t) w[i].length, with_score ? w[i].score : 0.0f, p7_NOCOMPLEMENT, (uint32_t) L);
}

int p7_ViterbiFilter_BATH(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, const P7_SCOREDATA *ssvdata, float filtersc,
                          double P, P7_HMM_WINDOWLIST *windowlist, float *ret_sc)                       /* vitfilter.c:286 */
{
  bath_hip_seqs *sq = NULL;
  const bath_hmm_window *w = NULL;
  int64_t nw = 0;
  int32_t status = 0;
  (void) ssvdata;                                   /* the SSV emission bytes it carries are the profile's own (p7_scoredata.c:57) */
  if (ox) { ox->M = om->M; ox->L = L; }
  if (ih_block(dsq, L, &sq) != eslOK) return eslFAIL;
  const int st = bath_hip_vitfilter_bath(impl_hip_context(), om->dev, sq, &filtersc, P, ret_sc, &status, &w, &nw);
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context()));
  ih_windows(windowlist, w, nw, L, FALSE);
  return ih_status(status);
}

int p7_SSVFilter_BATH(const ESL_DSQ *dsq, int L, P7_OPROFILE *om, P7_OMX *ox, const P7_SCOREDATA *msvdata, P7_BG *bg, double P,
                      P7_HMM_WINDOWLIST *windowlist)                                                    /* msvfilter.c:250 */
{
  bath_hip_seqs *sq = NULL;
  const bath_hmm_window *w = NULL;
  int64_t nw = 0;
  (void) msvdata;
  if (ox) { ox->M = om->M; ox->L = L; }
  p7_bg_SetLength(bg, L);                           /* the side effects the reference leaves behind (msvfilter.c:308-309) */
  p7_oprofile_ReconfigMSVLength(om, L);
  if (ih_block(dsq, L, &sq) != eslOK) return eslFAIL;
  const int st = bath_hip_ssvfilter_bath(impl_hip_context(), om->dev, sq, P, &w, &nw);
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context()));
  ih_windows(windowlist, w, nw, L, TRUE);
  return eslOK;
}

/* ------------------------------------------------------------------ device passes */

enum { IH_STD_PARSER = 1, IH_STD_ENV, IH_STD_REGION, IH_FS_ENV, IH_FS_REGION };
enum { IH_ROLE_FWD = 1, IH_ROLE_BCK, IH_ROLE_PP, IH_ROLE_OA };

struct impl_hip_pass {
  int      refs, kind;
  int      L, M;
  uint64_t key;                    /* digest of the target the pass was computed for */
  float    fwdsc, bcksc, oasc;
  int      fwd_status, bck_status, ok;
  float    null2[p7_MAXCODE];
  float   *pp, *oa, *ppx, *oax;    /* envelope kinds: posterior and optimal-accuracy matrices and their special-state rows */
  float   *fwd, *fx;               /* region kinds: the Forward matrix and its special-state rows                         */
  float   *bx;                     /* parser kind: the Backward parser's special-state rows                                */
};

static uint64_t ih_key(const ESL_DSQ *dsq, int L)
{
  uint64_t h = 1469598103934665603ull ^ (uint64_t) L;
  for (int i = 1; i <= L; i++) { h ^= dsq[i]; h *= 1099511628211ull; }
  return h;
}
static struct impl_hip_pass *ih_pass_new(int kind, const ESL_DSQ *dsq, int L, int M)
{
  struct impl_hip_pass *p = calloc(1, sizeof *p);
  if (p) { p->refs = 1; p->kind = kind; p->L = L; p->M = M; p->key = ih_key(dsq, L); }
  return p;
}
void impl_hip_pass_release(struct impl_hip_pass *p)
{
  if (!p || --p->refs > 0) return;
  free(p->pp); free(p->oa); free(p->ppx); free(p->oax); free(p->fwd); free(p->fx); free(p->bx);
  free(p);
}
static void ih_attach(P7_OMX *ox, struct impl_hip_pass *p, int role)
{
  if (ox->pass == p) { ox->role = role; return; }
  if (ox->pass) impl_hip_pass_release(ox->pass);
  ox->pass = p; ox->role = role; p->refs++;
  ox->M = p->M; ox->L = p->L;
}
/* main cell (i,k,s) of the matrix <ox> stands for, when the pass keeps it on the host (p7_omx_FDeconvert) */
int impl_hip_pass_cells(const P7_OMX *ox, int i, int k, int s, float *ret)
{
  const struct impl_hip_pass *p = ox->pass;
  if (!p) return eslEINVAL;
  const size_t W = (size_t)(p->M + 1);
  if ((p->kind == IH_STD_ENV) && ox->role == IH_ROLE_PP && p->pp) { *ret = p->pp[((size_t) i * W + k) * 3 + s]; return eslOK; }
  if ((p->kind == IH_STD_ENV) && ox->role == IH_ROLE_OA && p->oa) { *ret = p->oa[((size_t) i * W + k) * 3 + s]; return eslOK; }
  if ((p->kind == IH_STD_REGION) && p->fwd) { *ret = p->fwd[((size_t) i * W + k) * 3 + s]; return eslOK; }
  return eslEINVAL;
}

/* ------------------------------------------------------------------ standard parsers and domain decoding */

/* p7_ForwardParser (fwdback.c:132).  Both parsers are one device call; the Backward rows wait in the pass for p7_BackwardParser. */
int p7_ForwardParser(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *fwd, float *opt_sc)
{
  bath_hip_seqs *sq = NULL;
  const int64_t xoff[2] = { 0, ((int64_t) L + 1) * 6 };
  float fsc = 0.f, bsc = 0.f;
  int32_t fst = 0, bst = 0;
  if (!om->dev) ESL_EXCEPTION(eslEINVAL, "profile not converted");
  if (fwd->allocXR < L + 1) ESL_EXCEPTION(eslEINVAL, "matrix too small");
  struct impl_hip_pass *p = ih_pass_new(IH_STD_PARSER, dsq, L, om->M);
  if (!p) return eslEMEM;
  if (!(p->bx = malloc(sizeof(float) * (size_t)(L + 1) * 6))) { impl_hip_pass_release(p); return eslEMEM; }
  if (ih_block(dsq, L, &sq) != eslOK) { impl_hip_pass_release(p); return eslFAIL; }
  const int st = bath_hip_fwdback_parser(impl_hip_context(), om->dev, sq, xoff, &fsc, &bsc, &fst, &bst, fwd->xmx, p->bx);
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) { impl_hip_pass_release(p); ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context())); }
  p->fwdsc = fsc; p->bcksc = bsc; p->fwd_status = fst; p->bck_status = bst;
  ih_attach(fwd, p, IH_ROLE_FWD);
  impl_hip_pass_release(p);
  fwd->totscale = 0.0f; fwd->has_own_scales = TRUE;
  if (opt_sc) *opt_sc = fsc;
  return ih_status(fst);
}
int p7_BackwardParser(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *fwd, P7_OMX *bck, float *opt_sc)     /* fwdback.c:236 */
{
  const struct impl_hip_pass *p = fwd->pass;
  (void) om;
  if (!p || p->kind != IH_STD_PARSER || p->L != L || p->key != ih_key(dsq, L)) ESL_EXCEPTION(eslEINVAL, "Backward needs the Forward parser's matrix of the same target");
  if (bck->allocXR < L + 1) ESL_EXCEPTION(eslEINVAL, "matrix too small");
  memcpy(bck->xmx, p->bx, sizeof(float) * (size_t)(L + 1) * 6);
  ih_attach(bck, (struct impl_hip_pass *) p, IH_ROLE_BCK);
  bck->has_own_scales = FALSE;
  if (opt_sc) *opt_sc = p->bcksc;
  return ih_status(p->bck_status);
}

/* p7_DomainDecoding (decoding.c:143-189): O(L) arithmetic on the two parsers' special-state rows */
int p7_DomainDecoding(const P7_OPROFILE *om, const P7_OMX *oxf, const P7_OMX *oxb, P7_DOMAINDEF *ddef)
{
  const int L = oxf->L;
  float scaleproduct = 1.0f / oxb->xmx[p7X_N];
  float njcp;
  ddef->btot[0] = 0.0f; ddef->etot[0] = 0.0f; ddef->mocc[0] = 0.0f;
  for (int i = 1; i <= L; i++) {
    ddef->btot[i] = ddef->btot[i - 1] + oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_B] * oxb->xmx[(i - 1) * p7X_NXCELLS + p7X_B] * oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_SCALE] * scaleproduct;
    if (oxb->has_own_scales) scaleproduct *= oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_SCALE] / oxb->xmx[(i - 1) * p7X_NXCELLS + p7X_SCALE];
    ddef->etot[i] = ddef->etot[i - 1] + oxf->xmx[i * p7X_NXCELLS + p7X_E] * oxb->xmx[i * p7X_NXCELLS + p7X_E] * oxf->xmx[i * p7X_NXCELLS + p7X_SCALE] * scaleproduct;
    njcp  = oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_N] * oxb->xmx[i * p7X_NXCELLS + p7X_N] * om->xf[p7O_N][p7O_LOOP] * scaleproduct;
    njcp += oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_J] * oxb->xmx[i * p7X_NXCELLS + p7X_J] * om->xf[p7O_J][p7O_LOOP] * scaleproduct;
    njcp += oxf->xmx[(i - 1) * p7X_NXCELLS + p7X_C] * oxb->xmx[i * p7X_NXCELLS + p7X_C] * om->xf[p7O_C][p7O_LOOP] * scaleproduct;
    ddef->mocc[i] = 1.0f - njcp;
  }
  ddef->L = L;
  if (isinf(scaleproduct)) return eslERANGE;
  return eslOK;
}

/* ------------------------------------------------------------------ frameshift parsers and domain decoding */

static int ih_fs3(int backward, const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, P7_OMX *ox, float *opt_sc)
{
  bath_hip_seqs *sq = NULL;
  const int64_t xoff[2] = { 0, ((int64_t) L + 1) * 5 };
  float sc = 0.f;
  if (!om_fs->dev || om_fs->codon_lengths != 3) ESL_EXCEPTION(eslEINVAL, "needs the converted 3-codon profile");
  if (ox->allocXR < L + 1) ESL_EXCEPTION(eslEINVAL, "matrix too small");
  float *x5 = malloc(sizeof(float) * (size_t)(L + 1) * 5);
  if (!x5) return eslEMEM;
  if (ih_block(dsq, L, &sq) != eslOK) { free(x5); return eslFAIL; }
  const int st = (backward ? bath_hip_fs3_backward_parser : bath_hip_fs3_forward_parser)(impl_hip_context(), om_fs->dev, sq, BATH_LOGSUM_CONTEXT, &sc, x5, xoff);
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) { free(x5); ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context())); }
  for (int i = 0; i <= L; i++) {                    /* {E,N,J,B,C} -> the six-cell rows of P7_OMX */
    for (int s = 0; s < 5; s++) ox->xmx[i * p7X_NXCELLS + s] = x5[i * 5 + s];
    ox->xmx[i * p7X_NXCELLS + p7X_SCALE] = 1.0f;
  }
  free(x5);
  ox->M = om_fs->M; ox->L = L;
  if (opt_sc) *opt_sc = sc;
  return (sc == -eslINFINITY) ? eslERANGE : eslOK;
}
int p7_ForwardParser_Frameshift_3Codons(const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, P7_OMX *ox, P7_OIVX *ov, float *opt_sc)      /* fwdback_fs.c:97 */
{
  (void) ov;
  return ih_fs3(FALSE, dsq, L, om_fs, ox, opt_sc);
}
int p7_BackwardParser_Frameshift_3Codons(const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *fwd, P7_OMX *bck, P7_OIVX *ov, float *opt_sc)   /* :565 */
{
  (void) ov; (void) fwd;
  return ih_fs3(TRUE, dsq, L, om_fs, bck, opt_sc);
}

/* p7_DomainDecoding_Frameshift (decoding_fs.c; generic form generic_decoding_frameshift.c:204-290): log-space rows, steps of
 * three nucleotides */
int p7_DomainDecoding_Frameshift(const P7_FS_OPROFILE *om_fs, const P7_OMX *oxf, const P7_OMX *oxb, P7_DOMAINDEF *ddef)
{
  const int L = oxf->L;
  const float *F = oxf->xmx, *B = oxb->xmx;
  const float loop = om_fs->xf[p7O_N][p7O_LOOP];
  const float Z = p7_FLogsum(B[0 * p7X_NXCELLS + p7X_N], p7_FLogsum(B[1 * p7X_NXCELLS + p7X_N], B[2 * p7X_NXCELLS + p7X_N]));
#define IH_EM(s, a, b) expf(F[(a) * p7X_NXCELLS + (s)] + B[(b) * p7X_NXCELLS + (s)] + loop - Z)
  for (int i = 0; i < 3 && i <= L; i++) ddef->btot[i] = ddef->etot[i] = ddef->mocc[i] = 0.0f;
  for (int i = 3; i <= L; i++) {
    ddef->btot[i] = ddef->btot[i - 3] + expf(F[(i - 3) * p7X_NXCELLS + p7X_B] + B[(i - 3) * p7X_NXCELLS + p7X_B] - Z);
    ddef->etot[i] = ddef->etot[i - 3] + expf(F[i * p7X_NXCELLS + p7X_E] + B[i * p7X_NXCELLS + p7X_E] - Z);
    float njcp = 0.0f;
    if (i < L - 1) {
      njcp += IH_EM(p7X_N, i - 3, i); njcp += IH_EM(p7X_N, i - 2, i + 1); njcp += IH_EM(p7X_N, i - 1, i + 2);
      njcp += IH_EM(p7X_C, i - 3, i); njcp += IH_EM(p7X_C, i - 2, i + 1); njcp += IH_EM(p7X_C, i - 1, i + 2);
      njcp += IH_EM(p7X_J, i - 3, i); njcp += IH_EM(p7X_J, i - 2, i + 1); njcp += IH_EM(p7X_J, i - 1, i + 2);
    } else if (i == L - 1) {
      njcp += IH_EM(p7X_N, L - 4, L - 1); njcp += IH_EM(p7X_N, L - 3, L); njcp += IH_EM(p7X_C, L - 4, L - 1);
      njcp += IH_EM(p7X_C, L - 3, L);     njcp += IH_EM(p7X_J, L - 4, L - 1); njcp += IH_EM(p7X_J, L - 3, L);
    } else {
      njcp += IH_EM(p7X_N, L - 3, L); njcp += IH_EM(p7X_C, L - 3, L); njcp += IH_EM(p7X_J, L - 3, L);
    }
    ddef->mocc[i] = 1.0f - njcp;
  }
#undef IH_EM
  ddef->L = L;
  return eslOK;
}

/* ------------------------------------------------------------------ standard envelopes and regions */

int p7_Forward(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *fwd, float *opt_sc)           /* fwdback.c:94 */
{
  bath_hip_seqs *sq = NULL;
  const size_t ncell = (size_t)(L + 1) * (om->M + 1) * 3;
  int st;
  if (!om->dev) ESL_EXCEPTION(eslEINVAL, "profile not converted");
  if (fwd->allocXR < L + 1) ESL_EXCEPTION(eslEINVAL, "matrix too small");
  struct impl_hip_pass *p = ih_pass_new(om->nj == 0.0f ? IH_STD_ENV : IH_STD_REGION, dsq, L, om->M);
  if (!p) return eslEMEM;
  if (ih_block(dsq, L, &sq) != eslOK) { impl_hip_pass_release(p); return eslFAIL; }
  if (p->kind == IH_STD_ENV) {                      /* unihit envelope (p7_domaindef.c:1206-1209): the whole fused pass */
    bath_std_result r;
    p->pp = malloc(sizeof(float) * ncell); p->oa = malloc(sizeof(float) * ncell);
    p->ppx = malloc(sizeof(float) * (size_t)(L + 1) * 5); p->oax = malloc(sizeof(float) * (size_t)(L + 1) * 5);
    if (!p->pp || !p->oa || !p->ppx || !p->oax) { bath_hip_seqs_destroy(sq); impl_hip_pass_release(p); return eslEMEM; }
    st = bath_hip_std_envelopes(impl_hip_context(), om->dev, sq, &r, p->pp, p->oa, p->ppx, p->oax);
    if (st == BATH_OK) {
      p->fwdsc = r.fwdsc; p->bcksc = r.bcksc; p->oasc = r.oasc; p->fwd_status = r.fwd_status; p->bck_status = r.bck_status; p->ok = r.ok;
      memcpy(p->null2, r.null2, sizeof(float) * IH_KP);
    }
  } else {                                          /* multihit region with the configured length om->L (p7_domaindef.c:557-560) */
    const int32_t cfg = om->L;
    p->fwd = malloc(sizeof(float) * ncell); p->fx = malloc(sizeof(float) * (size_t)(L + 1) * 6);
    if (!p->fwd || !p->fx) { bath_hip_seqs_destroy(sq); impl_hip_pass_release(p); return eslEMEM; }
    st = bath_hip_forward_full(impl_hip_context(), om->dev, sq, &cfg, 0, &p->fwdsc, &p->fwd_status, p->fwd, p->fx);
    if (st == BATH_OK) memcpy(fwd->xmx, p->fx, sizeof(float) * (size_t)(L + 1) * 6);
  }
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) { impl_hip_pass_release(p); ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context())); }
  ih_attach(fwd, p, IH_ROLE_FWD);
  impl_hip_pass_release(p);
  if (opt_sc) *opt_sc = p->fwdsc;
  return ih_status(p->fwd_status);
}

static const struct impl_hip_pass *ih_need(const P7_OMX *ox, int kind, const char *what)
{
  (void) what;
  return (ox && ox->pass && ox->pass->kind == kind) ? ox->pass : NULL;
}

int p7_Backward(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *fwd, P7_OMX *bck, float *opt_sc)          /* fwdback.c:201 */
{
  const struct impl_hip_pass *p = ih_need(fwd, IH_STD_ENV, "Backward");
  (void) om;
  if (!p || p->L != L || p->key != ih_key(dsq, L)) ESL_EXCEPTION(eslEINVAL, "Backward needs the Forward matrix of the same envelope");
  ih_attach(bck, (struct impl_hip_pass *) p, IH_ROLE_BCK);
  if (opt_sc) *opt_sc = p->bcksc;
  return ih_status(p->bck_status);
}
int p7_Decoding(const P7_OPROFILE *om, const P7_OMX *oxf, P7_OMX *oxb, P7_OMX *pp)                      /* decoding.c:61 */
{
  const struct impl_hip_pass *p = ih_need(oxf, IH_STD_ENV, "Decoding");
  (void) om;
  if (!p || oxb->pass != p) ESL_EXCEPTION(eslEINVAL, "Decoding needs the Forward and Backward matrices of one envelope");
  ih_attach(pp, (struct impl_hip_pass *) p, IH_ROLE_PP);
  for (int i = 0; i <= p->L && i < pp->allocXR; i++) {
    for (int s = 0; s < 5; s++) pp->xmx[i * p7X_NXCELLS + s] = p->ppx[i * 5 + s];
    pp->xmx[i * p7X_NXCELLS + p7X_SCALE] = 1.0f;
  }
  return p->ok ? eslOK : eslERANGE;
}
int p7_OptimalAccuracy(const P7_OPROFILE *om, const P7_OMX *pp, P7_OMX *ox, float *ret_e)               /* optacc.c:58 */
{
  const struct impl_hip_pass *p = ih_need(pp, IH_STD_ENV, "OptimalAccuracy");
  (void) om;
  if (!p) ESL_EXCEPTION(eslEINVAL, "OptimalAccuracy needs a posterior matrix");
  ih_attach(ox, (struct impl_hip_pass *) p, IH_ROLE_OA);
  for (int i = 0; i <= p->L && i < ox->allocXR; i++)
    for (int s = 0; s < 5; s++) ox->xmx[i * p7X_NXCELLS + s] = p->oax[i * 5 + s];
  *ret_e = p->oasc;
  return eslOK;
}
int p7_Null2_ByExpectation(const P7_OPROFILE *om, const P7_OMX *pp, float *null2)                       /* null2.c:50 */
{
  const struct impl_hip_pass *p = ih_need(pp, IH_STD_ENV, "Null2");
  (void) om;
  if (!p) ESL_EXCEPTION(eslEINVAL, "Null2_ByExpectation needs a posterior matrix");
  memcpy(null2, p->null2, sizeof(float) * IH_KP);
  return eslOK;
}

/* p7_OATrace (optacc.c:225-430): the walk back through the optimal-accuracy matrix, on the host copies of the two matrices the
 * device pass left; select_e visits the cells in the striped order of the reference so that ties resolve alike. */
int p7_OATrace(const P7_OPROFILE *om, const P7_OMX *ppm, const P7_OMX *ox, P7_TRACE *tr)
{
  const struct impl_hip_pass *p = ih_need(ox, IH_STD_ENV, "OATrace");
  if (!p || ppm->pass != p) ESL_EXCEPTION(eslEINVAL, "OATrace needs the posterior and optimal-accuracy matrices of one envelope");
  if (tr->N != 0) ESL_EXCEPTION(eslEINVAL, "trace not empty; needs to be Reuse()'d?");
  const int M = p->M, L = p->L, Q = p7O_NQF(M);
  const size_t W = (size_t)(M + 1) * 3;
  const float *O = p->oa, *P = p->pp, *OX = p->oax, *PX = p->ppx, *tf = om->tf_host;
  enum { gMM = 0, gIM, gDM, gBM, gMD, gDD, gMI, gII };
#define IH_PATH(t, v) ((t) == 0.0f ? -eslINFINITY : (v))
  int i = L, k = 0, s0 = p7T_C, status;
  if ((status = p7_trace_AppendWithPP(tr, p7T_T, k, i, 0.0)) != eslOK) return status;
  if ((status = p7_trace_AppendWithPP(tr, p7T_C, k, i, 0.0)) != eslOK) return status;
  while (s0 != p7T_S) {
    int s1 = -1;
    switch (s0) {
    case p7T_M: {
      const float *t = tf + (size_t) k * 8, *pr = O + (size_t)(i - 1) * W;
      const float pm = IH_PATH(t[gMM], pr[(size_t)(k - 1) * 3 + p7X_M]), pi = IH_PATH(t[gIM], pr[(size_t)(k - 1) * 3 + p7X_I]);
      const float pd = IH_PATH(t[gDM], pr[(size_t)(k - 1) * 3 + p7X_D]), pb = IH_PATH(t[gBM], OX[(size_t)(i - 1) * 5 + p7X_B]);
      float b = pm; s1 = p7T_M;
      if (pi > b) { b = pi; s1 = p7T_I; }
      if (pd > b) { b = pd; s1 = p7T_D; }
      if (pb > b) { b = pb; s1 = p7T_B; }
      k--; i--; break; }
    case p7T_D: {
      const float *t = tf + (size_t)(k - 1) * 8, *c = O + (size_t) i * W;
      const float pm = (k - 1 >= 1) ? IH_PATH(t[gMD], c[(size_t)(k - 1) * 3 + p7X_M]) : -eslINFINITY;
      const float pd = (k - 1 >= 1) ? IH_PATH(t[gDD], c[(size_t)(k - 1) * 3 + p7X_D]) : -eslINFINITY;
      s1 = pm >= pd ? p7T_M : p7T_D; k--; break; }
    case p7T_I: {
      const float *t = tf + (size_t) k * 8, *pr = O + (size_t)(i - 1) * W;
      s1 = IH_PATH(t[gMI], pr[(size_t) k * 3 + p7X_M]) >= IH_PATH(t[gII], pr[(size_t) k * 3 + p7X_I]) ? p7T_M : p7T_I; i--; break; }
    case p7T_N: s1 = (i == 0) ? p7T_S : p7T_N; break;
    case p7T_C: s1 = (OX[(size_t)(i - 1) * 5 + p7X_C] + PX[(size_t) i * 5 + p7X_C] > OX[(size_t) i * 5 + p7X_E]) ? p7T_C : p7T_E; break;
    case p7T_J: s1 = p7T_J; break;                  /* unihit: E->J is impossible (optacc.c:384) */
    case p7T_E: {
      const float *c = O + (size_t) i * W;
      float mx = -eslINFINITY; int smax = -1, kmax = -1;
      for (int q = 0; q < Q; q++) {
        for (int r = 0; r < 4; r++) { const int kk = r * Q + q + 1; if (kk <= M && c[(size_t) kk * 3 + p7X_M] >= mx) { mx = c[(size_t) kk * 3 + p7X_M]; smax = p7T_M; kmax = kk; } }
        for (int r = 0; r < 4; r++) { const int kk = r * Q + q + 1; if (kk <= M && c[(size_t) kk * 3 + p7X_D] >  mx) { mx = c[(size_t) kk * 3 + p7X_D]; smax = p7T_D; kmax = kk; } }
      }
      k = kmax; s1 = smax; break; }
    case p7T_B: s1 = (OX[(size_t) i * 5 + p7X_N] > OX[(size_t) i * 5 + p7X_J]) ? p7T_N : p7T_J; break;
    default: ESL_EXCEPTION(eslEINVAL, "bogus state in traceback");
    }
    if (s1 == -1 || i < 0 || k < 0) ESL_EXCEPTION(eslEINVAL, "OA traceback choice failed");
    float postprob = 0.0f;                          /* get_postprob, optacc.c:264-280 */
    if      (s1 == p7T_M) postprob = P[(size_t) i * W + (size_t) k * 3 + p7X_M];
    else if (s1 == p7T_I) postprob = P[(size_t) i * W + (size_t) k * 3 + p7X_I];
    else if (s1 == s0 && s1 == p7T_N) postprob = PX[(size_t) i * 5 + p7X_N];
    else if (s1 == s0 && s1 == p7T_C) postprob = PX[(size_t) i * 5 + p7X_C];
    else if (s1 == s0 && s1 == p7T_J) postprob = PX[(size_t) i * 5 + p7X_J];
    if ((status = p7_trace_AppendWithPP(tr, s1, k, i, postprob)) != eslOK) return status;
    if ((s1 == p7T_N || s1 == p7T_J || s1 == p7T_C) && s1 == s0) i--;
    s0 = s1;
  }
#undef IH_PATH
  tr->M = M; tr->L = L;
  return p7_trace_Reverse(tr);
}

/* p7_Null2_ByTrace (null2.c:131-215): O(N + M Kp) bookkeeping over a trace and the profile's emission odds ratios */
int p7_Null2_ByTrace(const P7_OPROFILE *om, const P7_TRACE *tr, int zstart, int zend, P7_OMX *wrk, float *null2)
{
  const int M = om->M;
  int Ld = 0;
  (void) wrk;
  float *cnt = calloc((size_t) M + 1, sizeof(float));
  if (!cnt) return eslEMEM;
  for (int z = zstart; z <= zend; z++) {
    if (tr->st[z] == p7T_M || tr->st[z] == p7T_I) { Ld++; cnt[tr->k[z]] += 1.0f; }      /* inserts land in the match slot (null2.c:160-166) */
  }
  const float norm = (float)(1.0 / (float) Ld);
  for (int k = 1; k <= M; k++) cnt[k] *= norm;
  for (int x = 0; x < om->abc->K; x++) {
    const float *e = om->rf_host + (size_t) x * (M + 1);
    float sv = 0.0f;
    for (int k = 1; k <= M; k++) sv += cnt[k] * e[k];
    null2[x] = sv;
  }
  free(cnt);
  esl_abc_FAvgScVec(om->abc, null2);                /* degenerate residues: mean over their members */
  null2[om->abc->K]      = 1.0;                     /* gap */
  null2[om->abc->Kp - 2] = 1.0;                     /* nonresidue */
  null2[om->abc->Kp - 1] = 1.0;                     /* missing data */
  return eslOK;
}

/* p7_StochasticTrace (stotrace.c:71-300) on the region's Forward matrix the device pass left on the host: a walk of dependent
 * random choices drawn from the CALLER's generator, one call to esl_random per choice as in the reference */
int p7_StochasticTrace(ESL_RANDOMNESS *rng, const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *ox, P7_TRACE *tr)
{
  const struct impl_hip_pass *p = ox ? ox->pass : NULL;
  if (!p || (p->kind != IH_STD_REGION) || p->L != L || p->key != ih_key(dsq, L)) ESL_EXCEPTION(eslEINVAL, "StochasticTrace needs the (multihit) Forward matrix of this target");
  if (tr->N != 0) ESL_EXCEPTION(eslEINVAL, "trace not empty");
  const int M = p->M, Q = p7O_NQF(M);
  const size_t W = (size_t)(M + 1) * 3;
  const float *fwd = p->fwd, *fx = p->fx, *tf = om->tf_host;
  enum { gMM = 0, gIM, gDM, gBM, gMD, gDD, gMI, gII };
  int i = L, k = 0, s0 = p7T_C, status;
  if ((status = p7_trace_Append(tr, p7T_T, k, i)) != eslOK) return status;
  if ((status = p7_trace_Append(tr, p7T_C, k, i)) != eslOK) return status;
  while (s0 != p7T_S) {
    int s1 = -1;
    float path[4];
    switch (s0) {
    case p7T_M: {
      static const int state[4] = { p7T_B, p7T_M, p7T_I, p7T_D };
      const float *tk = tf + (size_t) k * 8, *pr = fwd + (size_t)(i - 1) * W;
      path[0] = fx[(size_t)(i - 1) * 6 + p7X_B] * tk[gBM]; path[1] = pr[(size_t)(k - 1) * 3 + p7X_M] * tk[gMM];
      path[2] = pr[(size_t)(k - 1) * 3 + p7X_I] * tk[gIM]; path[3] = pr[(size_t)(k - 1) * 3 + p7X_D] * tk[gDM];
      esl_vec_FNorm(path, 4); s1 = state[esl_rnd_FChoose(rng, path, 4)]; k--; i--; break; }
    case p7T_D: {
      const float *c = fwd + (size_t) i * W;
      path[0] = k - 1 >= 1 ? c[(size_t)(k - 1) * 3 + p7X_M] * tf[(size_t)(k - 1) * 8 + gMD] : 0.0f;
      path[1] = k - 1 >= 1 ? c[(size_t)(k - 1) * 3 + p7X_D] * tf[(size_t)(k - 1) * 8 + gDD] : 0.0f;
      esl_vec_FNorm(path, 2); s1 = esl_rnd_FChoose(rng, path, 2) == 0 ? p7T_M : p7T_D; k--; break; }
    case p7T_I: {
      const float *pr = fwd + (size_t)(i - 1) * W;
      path[0] = pr[(size_t) k * 3 + p7X_M] * tf[(size_t) k * 8 + gMI]; path[1] = pr[(size_t) k * 3 + p7X_I] * tf[(size_t) k * 8 + gII];
      esl_vec_FNorm(path, 2); s1 = esl_rnd_FChoose(rng, path, 2) == 0 ? p7T_M : p7T_I; i--; break; }
    case p7T_N: s1 = (i == 0) ? p7T_S : p7T_N; break;
    case p7T_C:
      path[0] = fx[(size_t)(i - 1) * 6 + p7X_C] * om->xf[p7O_C][p7O_LOOP];
      path[1] = fx[(size_t) i * 6 + p7X_E] * om->xf[p7O_E][p7O_MOVE] * fx[(size_t) i * 6 + p7X_SCALE];
      esl_vec_FNorm(path, 2); s1 = esl_rnd_FChoose(rng, path, 2) == 0 ? p7T_C : p7T_E; break;
    case p7T_J:
      path[0] = fx[(size_t)(i - 1) * 6 + p7X_J] * om->xf[p7O_J][p7O_LOOP];
      path[1] = fx[(size_t) i * 6 + p7X_E] * om->xf[p7O_E][p7O_LOOP] * fx[(size_t) i * 6 + p7X_SCALE];
      esl_vec_FNorm(path, 2); s1 = esl_rnd_FChoose(rng, path, 2) == 0 ? p7T_J : p7T_E; break;
    case p7T_E: {                                   /* select_e: cumulative sum in double over the cells in striped order (stotrace.c:262-300) */
      const float *c = fwd + (size_t) i * W;
      const double roll = esl_random(rng);
      const float norm = (float)(1.0 / fx[(size_t) i * 6 + p7X_E]);
      double sum = 0.0;
      for (int pass = 0; pass < 4 && s1 < 0; pass++)
        for (int q = 0; q < Q && s1 < 0; q++) {
          for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += kk <= M ? c[(size_t) kk * 3 + p7X_M] * norm : 0.0f; if (roll < sum) { k = kk; s1 = p7T_M; } }
          for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += kk <= M ? c[(size_t) kk * 3 + p7X_D] * norm : 0.0f; if (roll < sum) { k = kk; s1 = p7T_D; } }
        }
      break; }
    case p7T_B:
      path[0] = fx[(size_t) i * 6 + p7X_N] * om->xf[p7O_N][p7O_MOVE]; path[1] = fx[(size_t) i * 6 + p7X_J] * om->xf[p7O_J][p7O_MOVE];
      esl_vec_FNorm(path, 2); s1 = esl_rnd_FChoose(rng, path, 2) == 0 ? p7T_N : p7T_J; break;
    default: ESL_EXCEPTION(eslEINVAL, "bogus state in traceback");
    }
    if (s1 == -1 || i < 0 || k < 0) ESL_EXCEPTION(eslEINVAL, "Stochastic traceback choice failed");
    if ((status = p7_trace_Append(tr, s1, k, i)) != eslOK) return status;
    if ((s1 == p7T_N || s1 == p7T_J || s1 == p7T_C) && s1 == s0) i--;
    s0 = s1;
  }
  tr->M = M; tr->L = L;
  return p7_trace_Reverse(tr);
}

/* ------------------------------------------------------------------ frameshift envelopes and regions */

int p7_Forward_Frameshift(const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, P7_OMX *ox, P7_OIVX *ov, float *opt_sc)    /* fwdback_fs.c:2054 */
{
  bath_hip_seqs *sq = NULL;
  const int M = om_fs->M;
  const size_t rows = (size_t) L + 1, ncell = rows * (M + 1);
  int st;
  (void) ov;
  if (!om_fs->dev || om_fs->codon_lengths != 5) ESL_EXCEPTION(eslEINVAL, "needs the converted 5-codon profile");
  struct impl_hip_pass *p = ih_pass_new(om_fs->nj == 0.0f ? IH_FS_ENV : IH_FS_REGION, dsq, L, M);
  if (!p) return eslEMEM;
  if (ih_block(dsq, L, &sq) != eslOK) { impl_hip_pass_release(p); return eslFAIL; }
  if (p->kind == IH_FS_ENV) {                       /* unihit envelope, L/3 (p7_domaindef.c:1019-1021): the fused envelope pass */
    bath_fs5_result r;
    p->pp = malloc(sizeof(float) * ncell * 8); p->oa = malloc(sizeof(float) * ncell * 3);
    p->ppx = malloc(sizeof(float) * rows * 5); p->oax = malloc(sizeof(float) * rows * 5);
    if (!p->pp || !p->oa || !p->ppx || !p->oax) { bath_hip_seqs_destroy(sq); impl_hip_pass_release(p); return eslEMEM; }
    st = bath_hip_fs5_envelopes_x(impl_hip_context(), om_fs->dev, sq, BATH_LOGSUM_CONTEXT, 0, &r, p->pp, p->oa, p->ppx, p->oax);
    if (st == BATH_OK) { p->fwdsc = r.fwdsc; p->bcksc = r.bcksc; p->oasc = r.oasc; p->ok = 1; memcpy(p->null2, r.null2, sizeof(float) * IH_KP); }
  } else {                                          /* multihit region in the configuration of amino length om_fs->L (:411-414) */
    p->fwd = malloc(sizeof(float) * ncell * 8); p->fx = malloc(sizeof(float) * rows * 5);
    if (!p->fwd || !p->fx) { bath_hip_seqs_destroy(sq); impl_hip_pass_release(p); return eslEMEM; }
    st = bath_hip_fs5_forward_full(impl_hip_context(), om_fs->dev, sq, om_fs->L, &p->fwdsc, p->fwd, p->fx);
  }
  bath_hip_seqs_destroy(sq);
  if (st != BATH_OK) { impl_hip_pass_release(p); ESL_EXCEPTION(eslFAIL, "impl_hip: %s", bath_hip_last_error(impl_hip_context())); }
  ih_attach(ox, p, IH_ROLE_FWD);
  impl_hip_pass_release(p);
  if (opt_sc) *opt_sc = p->fwdsc;
  return (p->fwdsc == -eslINFINITY) ? eslERANGE : eslOK;
}
int p7_Backward_Frameshift(const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *fwd, P7_OMX *bck, P7_OIVX *ov, float *opt_sc)     /* :2634 */
{
  const struct impl_hip_pass *p = ih_need(fwd, IH_FS_ENV, "Backward");
  (void) om_fs; (void) ov;
  if (!p || p->L != L || p->key != ih_key(dsq, L)) ESL_EXCEPTION(eslEINVAL, "Backward needs the Forward matrix of the same envelope");
  ih_attach(bck, (struct impl_hip_pass *) p, IH_ROLE_BCK);
  if (opt_sc) *opt_sc = p->bcksc;
  return (p->bcksc == -eslINFINITY) ? eslERANGE : eslOK;
}
int p7_Decoding_Frameshift(const P7_FS_OPROFILE *om_fs, P7_OMX *fwd, const P7_OMX *bck)                 /* decoding_fs.c */
{
  const struct impl_hip_pass *p = ih_need(fwd, IH_FS_ENV, "Decoding");
  (void) om_fs;
  if (!p || bck->pass != p) ESL_EXCEPTION(eslEINVAL, "Decoding needs the Forward and Backward matrices of one envelope");
  fwd->role = IH_ROLE_PP;                           /* the posteriors overwrite Forward, as in the reference */
  for (int i = 0; i <= p->L && i < fwd->allocXR; i++) {
    for (int s = 0; s < 5; s++) fwd->xmx[i * p7X_NXCELLS + s] = p->ppx[i * 5 + s];
    fwd->xmx[i * p7X_NXCELLS + p7X_SCALE] = 1.0f;
  }
  return eslOK;
}
int p7_OptimalAccuracy_Frameshift(const P7_FS_OPROFILE *om_fs, const P7_OMX *pp, P7_OMX *ox, float *ret_e)    /* optacc_fs.c */
{
  const struct impl_hip_pass *p = ih_need(pp, IH_FS_ENV, "OptimalAccuracy");
  (void) om_fs;
  if (!p) ESL_EXCEPTION(eslEINVAL, "OptimalAccuracy needs a posterior matrix");
  ih_attach(ox, (struct impl_hip_pass *) p, IH_ROLE_OA);
  for (int i = 0; i <= p->L && i < ox->allocXR; i++)
    for (int s = 0; s < 5; s++) ox->xmx[i * p7X_NXCELLS + s] = p->oax[i * 5 + s];
  *ret_e = p->oasc;
  return eslOK;
}
int p7_Null2_fs_ByExpectation(const P7_FS_OPROFILE *om_fs, P7_OMX *pp, float *null2)                    /* null2_fs.c */
{
  const struct impl_hip_pass *p = ih_need(pp, IH_FS_ENV, "Null2");
  (void) om_fs;
  if (!p) ESL_EXCEPTION(eslEINVAL, "Null2_fs_ByExpectation needs a posterior matrix");
  memcpy(null2, p->null2, sizeof(float) * IH_KP);
  return eslOK;
}

/* p7_OATrace_Frameshift (optacc_fs.c:546-593; generic form generic_optacc_frameshift.c:373-588).  TSCDELTA: 1 for a possible
 * transition, FLT_MIN for an impossible one. */
int p7_OATrace_Frameshift(const P7_FS_OPROFILE *om_fs, const P7_OMX *ppm, const P7_OMX *ox, P7_TRACE *tr)
{
  const struct impl_hip_pass *p = ih_need(ox, IH_FS_ENV, "OATrace");
  if (!p || ppm->pass != p) ESL_EXCEPTION(eslEINVAL, "OATrace needs the posterior and optimal-accuracy matrices of one envelope");
  if (tr->N != 0) ESL_EXCEPTION(eslEINVAL, "trace not empty; needs to be Reuse()'d?");
  const int M = p->M, L = p->L;
  const float *P = p->pp, *PX = p->ppx, *O = p->oa, *OX = p->oax, *tsc = om_fs->gm_copy->tsc;
  const float tiny = 1.17549435e-38f;
#define IH_DL(kk, s) (((kk) >= 0 && (kk) < M && tsc[(size_t)(kk) * p7P_NTRANS + (s)] != -eslINFINITY) ? 1.0f : tiny)
#define IH_OM(i, k) O[((size_t)(i) * (M + 1) + (k)) * 3 + 2]
#define IH_OI(i, k) O[((size_t)(i) * (M + 1) + (k)) * 3 + 1]
#define IH_OD(i, k) O[((size_t)(i) * (M + 1) + (k)) * 3 + 0]
  int i = L, k = 0, c = 0, sprv = p7T_C, status;
  if ((status = p7_trace_fs_AppendWithPP(tr, p7T_T, k, i, c, 0.0f)) != eslOK) return status;
  if ((status = p7_trace_fs_AppendWithPP(tr, p7T_C, k, i, c, 0.0f)) != eslOK) return status;
  while (sprv != p7T_S) {
    int scur = -1;
    switch (sprv) {
    case p7T_M: {                                   /* transitions into node k are the generic tsc of node k-1 */
      const float p0 = IH_DL(k - 1, p7P_MM) * IH_OM(i, k - 1), p1 = IH_DL(k - 1, p7P_IM) * IH_OI(i, k - 1);
      const float p2 = IH_DL(k - 1, p7P_DM) * IH_OD(i, k - 1), p3 = IH_DL(k - 1, p7P_BM) * OX[(size_t) i * 5 + p7X_B];
      float b = p0; scur = p7T_M;
      if (p1 > b) { b = p1; scur = p7T_I; }
      if (p2 > b) { b = p2; scur = p7T_D; }
      if (p3 > b) { b = p3; scur = p7T_B; }
      k--; break; }
    case p7T_D: {
      const float p0 = IH_DL(k - 1, p7P_MD) * IH_OM(i, k - 1), p1 = IH_DL(k - 1, p7P_DD) * IH_OD(i, k - 1);
      scur = p0 >= p1 ? p7T_M : p7T_D; k--; break; }
    case p7T_I: {
      const float p0 = IH_DL(k, p7P_MI) * IH_OM(i - 3, k), p1 = IH_DL(k, p7P_II) * IH_OI(i - 3, k);
      scur = p0 >= p1 ? p7T_M : p7T_I; i -= 3; break; }
    case p7T_N: scur = (i == 0) ? p7T_S : p7T_N; break;
    case p7T_C: {
      if (i < 4) { scur = p7T_E; break; }
      const float p0 = OX[(size_t)(i - 3) * 5 + p7X_C] + PX[(size_t) i * 5 + p7X_C];
      const float p1 = (i < L)     ? OX[(size_t)(i - 2) * 5 + p7X_C] + PX[(size_t)(i + 1) * 5 + p7X_C] : tiny;
      const float p2 = (i < L - 1) ? OX[(size_t)(i - 1) * 5 + p7X_C] + PX[(size_t)(i + 2) * 5 + p7X_C] : tiny;
      const float p3 = OX[(size_t) i * 5 + p7X_E];
      float b = p0; scur = p7T_C;
      if (p1 > b) b = p1;
      if (p2 > b) b = p2;
      if (p3 > b) { b = p3; scur = p7T_E; }
      break; }
    case p7T_J: {
      if (i <= 5) { scur = p7T_E; break; }
      const float p0 = OX[(size_t) i * 5 + p7X_J] + PX[(size_t) i * 5 + p7X_J], p1 = tiny * OX[(size_t) i * 5 + p7X_E];      /* unihit */
      scur = (p1 > p0) ? p7T_E : p7T_J; break; }
    case p7T_E: {
      float mx = -eslINFINITY; int smax = -1, kmax = -1;
      for (int q = 1; q <= M; q++) {
        const float m = IH_OM(i, q), d = IH_OD(i, q);
        if (m > mx) { mx = m; smax = p7T_M; kmax = q; }
        if (d > mx) { mx = d; smax = p7T_D; kmax = q; }
      }
      k = kmax; scur = smax; break; }
    case p7T_B: scur = (OX[(size_t) i * 5 + p7X_N] > OX[(size_t) i * 5 + p7X_J]) ? p7T_N : p7T_J; break;
    default: ESL_EXCEPTION(eslEINVAL, "bogus state in OA FS traceback");
    }
    if (scur == -1 || k < 0 || i < 0) ESL_EXCEPTION(eslEINVAL, "OA FS traceback choice failed");
    const float *cell = P + ((size_t) i * (M + 1) + k) * 8;
    float postprob = 0.0f;                          /* get_postprob_fs, optacc_fs.c:300-319 */
    if      (scur == p7T_M) postprob = cell[2];
    else if (scur == p7T_I) postprob = cell[1];
    else if (scur == sprv && scur == p7T_N) postprob = PX[(size_t) i * 5 + p7X_N];
    else if (scur == sprv && scur == p7T_C) postprob = PX[(size_t) i * 5 + p7X_C];
    else if (scur == sprv && scur == p7T_J) postprob = PX[(size_t) i * 5 + p7X_J];
    if (scur == p7T_M) {                            /* select_codon_fs: the codon length with the largest posterior */
      float b = cell[3]; c = 1;
      for (int q = 1; q < 5; q++) if (cell[3 + q] > b) { b = cell[3 + q]; c = q + 1; }
    } else c = 0;
    if ((status = p7_trace_fs_AppendWithPP(tr, scur, k, i, c, postprob)) != eslOK) return status;
    if ((scur == p7T_N || scur == p7T_C || scur == p7T_J) && scur == sprv) i--;
    sprv = scur;
    i -= c;
  }
#undef IH_DL
#undef IH_OM
#undef IH_OI
#undef IH_OD
  tr->M = M; tr->L = L;
  return p7_trace_fs_Reverse(tr);
}

/* p7_StochasticTrace_Frameshift (stotrace_fs.c:72; generic form generic_stotrace_frameshift.c:40-215) on the region's Forward
 * matrix (log space): esl_vec_FLogNorm + esl_rnd_FChoose per choice, from the caller's generator */
int p7_StochasticTrace_Frameshift(ESL_RANDOMNESS *rng, const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *ox, P7_TRACE *tr)
{
  const struct impl_hip_pass *p = ox ? ox->pass : NULL;
  if (!p || p->kind != IH_FS_REGION || p->L != L || p->key != ih_key(dsq, L)) ESL_EXCEPTION(eslEINVAL, "StochasticTrace needs the (multihit) Forward matrix of this target");
  if (tr->N != 0) ESL_EXCEPTION(eslEINVAL, "trace not empty");
  const int M = p->M;
  const size_t W = (size_t)(M + 1) * 8;
  const float *fwd = p->fwd, *fx = p->fx, *tsc = om_fs->gm_copy->tsc;
  const float xNL = om_fs->xf[p7O_N][p7O_LOOP], xNM = om_fs->xf[p7O_N][p7O_MOVE], xE = om_fs->xf[p7O_E][p7O_MOVE];
  float *sc = malloc(sizeof(float) * ((size_t) 2 * M + 8));
  if (!sc) return eslEMEM;
#define IH_DP(i, k, s) fwd[(size_t)(i) * W + (size_t)(k) * 8 + (s)]
#define IH_X(i, s)     fx[(size_t)(i) * 5 + (s)]
#define IH_TS(s, k)    tsc[(size_t)(k) * p7P_NTRANS + (s)]
#define IH_CHOOSE(n)   (esl_vec_FLogNorm(sc, (n)), esl_rnd_FChoose(rng, sc, (n)))
#define IH_FAIL        do { free(sc); ESL_EXCEPTION(eslEINVAL, "impossible state reached in stochastic traceback"); } while (0)
  int i = L, k = 0, c = 0, sprv = p7T_C, status;
  if ((status = p7_trace_fs_Append(tr, p7T_T, k, i, c)) != eslOK) { free(sc); return status; }
  if ((status = p7_trace_fs_Append(tr, p7T_C, k, i, c)) != eslOK) { free(sc); return status; }
  while (sprv != p7T_S) {
    int scur = -1;
    switch (sprv) {
    case p7T_C:
      if (IH_X(i, p7X_C) == -eslINFINITY) IH_FAIL;
      if (i < 4) { scur = p7T_E; break; }
      sc[0] = IH_X(i - 3, p7X_C) + xNL; sc[1] = IH_X(i - 2, p7X_C) + xNL; sc[2] = IH_X(i - 1, p7X_C) + xNL; sc[3] = IH_X(i, p7X_E) + xE;
      scur = IH_CHOOSE(4) < 3 ? p7T_C : p7T_E; break;
    case p7T_E:
      if (IH_X(i, p7X_E) == -eslINFINITY) IH_FAIL;
      sc[0] = sc[(size_t) M + 1] = -eslINFINITY;
      for (int q = 1; q <= M; q++) sc[q] = IH_DP(i, q, 2);
      for (int q = 2; q <= M; q++) sc[q + M] = IH_DP(i, q, 0);
      k = IH_CHOOSE(2 * M + 1);
      if (k <= M) scur = p7T_M; else { k -= M; scur = p7T_D; }
      break;
    case p7T_M: {
      static const int state[4] = { p7T_B, p7T_M, p7T_I, p7T_D };
      sc[0] = IH_X(i, p7X_B) + IH_TS(p7P_BM, k - 1); sc[1] = IH_DP(i, k - 1, 2) + IH_TS(p7P_MM, k - 1);
      sc[2] = IH_DP(i, k - 1, 1) + IH_TS(p7P_IM, k - 1); sc[3] = IH_DP(i, k - 1, 0) + IH_TS(p7P_DM, k - 1);
      scur = state[IH_CHOOSE(4)]; k--; break; }
    case p7T_D:
      if (IH_DP(i, k, 0) == -eslINFINITY) IH_FAIL;
      sc[0] = IH_DP(i, k - 1, 2) + IH_TS(p7P_MD, k - 1); sc[1] = IH_DP(i, k - 1, 0) + IH_TS(p7P_DD, k - 1);
      scur = IH_CHOOSE(2) == 0 ? p7T_M : p7T_D; k--; break;
    case p7T_I:
      if (IH_DP(i, k, 1) == -eslINFINITY || i < 3) IH_FAIL;
      sc[0] = IH_DP(i - 3, k, 2) + IH_TS(p7P_MI, k); sc[1] = IH_DP(i - 3, k, 1) + IH_TS(p7P_II, k);
      scur = IH_CHOOSE(2) == 0 ? p7T_M : p7T_I; i -= 3; break;
    case p7T_N:
      if (IH_X(i, p7X_N) == -eslINFINITY) IH_FAIL;
      scur = (i == 0) ? p7T_S : p7T_N; break;
    case p7T_B:
      if (IH_X(i, p7X_B) == -eslINFINITY) IH_FAIL;
      sc[0] = IH_X(i, p7X_N) + xNM; sc[1] = IH_X(i, p7X_J) + xNM;
      scur = IH_CHOOSE(2) == 0 ? p7T_N : p7T_J; break;
    case p7T_J:
      if (IH_X(i, p7X_J) == -eslINFINITY) IH_FAIL;
      if (i < 4) { scur = p7T_E; break; }
      sc[0] = IH_X(i - 3, p7X_J) + xNL; sc[1] = IH_X(i - 2, p7X_J) + xNL; sc[2] = IH_X(i - 1, p7X_J) + xNL; sc[3] = IH_X(i, p7X_E) + om_fs->xf[p7O_E][p7O_LOOP];
      scur = IH_CHOOSE(4) < 3 ? p7T_J : p7T_E; break;
    default: IH_FAIL;
    }
    if (scur == p7T_M) {                            /* the codon length, from the C1..C5 cells */
      for (int q = 0; q < 5; q++) sc[q] = IH_DP(i, k, 3 + q);
      c = IH_CHOOSE(5) + 1;
      if (i - c < 0) scur = p7T_B;
    } else c = 0;
    if (scur < 0 || k < 0 || i < 0) IH_FAIL;
    if ((status = p7_trace_fs_Append(tr, scur, k, i, c)) != eslOK) { free(sc); return status; }
    if ((scur == p7T_N || scur == p7T_C || scur == p7T_J) && scur == sprv) i--;
    sprv = scur;
    i -= c;
    if (i < 0) IH_FAIL;
  }
#undef IH_DP
#undef IH_X
#undef IH_TS
#undef IH_CHOOSE
#undef IH_FAIL
  free(sc);
  tr->M = M; tr->L = L;
  return p7_trace_fs_Reverse(tr);
}
