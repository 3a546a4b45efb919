/* impl_hip.h -- the MI355X (gfx950) implementation of BATH's impl_* directory contract.
 *
 * The reference selects its DP backend at build time: hmmer.h:1044-1052 includes "impl_xxx/impl_xxx.h" and
 * src/Makefile.in:39 links IMPLDIR.  This header is the drop-in for impl_sse/impl_sse.h: the four types hmmer.h embeds in
 * P7_PIPELINE by name (P7_OPROFILE, P7_FS_OPROFILE, P7_OIVX, P7_OMX; hmmer.h:1065-1074) and every function of
 * impl_sse.h:408-553 that p7_pipeline.c, p7_domaindef.c, bathsearch.c, p7_scoredata.c and p7_alidisplay.c reference, with
 * identical names and argument lists (tests/test_impl_hip_cpu.py checks each prototype against impl_sse.h).
 *
 * What differs is behind the types.  The striped __m128 arrays of impl_sse are an SSE detail that no code outside impl_sse
 * touches (it goes through the accessor functions kept below); here a profile is a handle to tables resident in HBM
 * (bath_hip_oprofile / bath_hip_fsprofile, include/bath_hip.h) plus the metadata the callers read, and a DP matrix is a handle
 * to the results of the device pass it belongs to.  Every function below is one call of the batched C ABI with n = 1.
 *
 * A GPU fed one target per call is latency bound; this layer exists so that libhmmer links and runs unchanged.  The
 * throughput path is the batched driver hook shown in INTEGRATION.md, which replaces the ORF loop of p7_Pipeline_BATH with
 * one bath_hip_pipeline_* call per block of targets.
 *
 * Include order: like impl_sse.h this file is included from hmmer.h, after the generic types (P7_PROFILE, P7_FS_PROFILE,
 * P7_BG, P7_GMX, P7_TRACE, P7_DOMAINDEF, P7_SCOREDATA, P7_HMM_WINDOWLIST, ESL_*) are known.
 */
#ifndef P7_IMPL_HIP_INCLUDED
#define P7_IMPL_HIP_INCLUDED

#include <stdio.h>
#include <stdint.h>
#include <sys/types.h>

#include "bath_hip.h"

/* striped-vector counts: kept because generic code sizes scratch arrays with them (p7_scoredata.c) */
#define p7O_NQB(M)   ( ESL_MAX(2, ((((M)-1) / 16) + 1)))
#define p7O_NQW(M)   ( ESL_MAX(2, ((((M)-1) / 8)  + 1)))
#define p7O_NQF(M)   ( ESL_MAX(2, ((((M)-1) / 4)  + 1)))

#define p7O_EXTRA_SB 17

#define p7O_NXSTATES  4
#define p7O_NXTRANS   2
#define p7O_NTRANS    8
enum p7o_xstates_e      { p7O_E    = 0, p7O_N    = 1,  p7O_J  = 2,  p7O_C  = 3 };
enum p7o_xtransitions_e { p7O_MOVE = 0, p7O_LOOP = 1 };
enum p7o_tsc_e          { p7O_BM   = 0, p7O_MM   = 1,  p7O_IM = 2,  p7O_DM = 3, p7O_MD   = 4, p7O_MI   = 5,  p7O_II = 6,  p7O_DD = 7 };

/*****************************************************************
 * 1. P7_OPROFILE  (impl_sse.h:75-146)
 *****************************************************************/
typedef struct p7_oprofile_s {
  /* the device-resident tables (MSV bytes, Viterbi words, Forward odds ratios, SSV costs, per-length scalars) */
  bath_hip_oprofile *dev;        /* NULL until p7_oprofile_Convert()                                  */
  bath_profile      *gm_copy;    /* the generic scores Convert() was given: Clone / Reconfig*hit rebuild from them */

  /* scalars of the limited-precision score systems for the current length configuration (read by p7_pipeline.c:451 and the
   * accessors); refreshed by every Reconfig call from bath_hip_oprofile_scalars() */
  uint8_t   tbm_b, tec_b, tjb_b;
  float     scale_b;
  uint8_t   base_b, bias_b;
  int16_t   xw[p7O_NXSTATES][p7O_NXTRANS];
  float     scale_w;
  int16_t   base_w, ddbound_w;
  float     ncj_roundoff;
  float     xf[p7O_NXSTATES][p7O_NXTRANS];

  /* unstriped host copies behind the accessor functions */
  float    *rf_host;             /* [Kp][M+1]  Forward emission odds ratios: p7_oprofile_FGetEmission            */
  float    *tf_host;             /* [M+1][8]   generic order MM IM DM BM MD DD MI II                             */
  uint8_t  *ssv_host;            /* [(M+1)*Kp] p7_oprofile_GetSSVEmissionScoreArray                              */

  off_t  offs[p7_NOFFSETS];
  off_t  roff;
  off_t  eoff;

  char  *name;
  char  *acc;
  char  *desc;
  char  *rf;
  char  *mm;
  char  *cs;
  char  *consensus;
  float  evparam[p7_NEVPARAM];
  float  cutoff[p7_NCUTOFFS];
  float  compo[p7_MAXABET];
  const ESL_ALPHABET *abc;

  int    L;
  int    M;
  int    max_length;
  int    allocM;
  int    allocQ4, allocQ8, allocQ16;     /* kept for source compatibility; nothing is striped here */
  int    mode;
  float  nj;
  int    clone;
} P7_OPROFILE;

typedef struct {
  int            count;
  int            listSize;
  P7_OPROFILE  **list;
} P7_OM_BLOCK;

/* retrieve match odds ratio [k][x] (p7_alidisplay.c) */
static inline float
p7_oprofile_FGetEmission(const P7_OPROFILE *om, int k, int x)
{
  return om->rf_host[(size_t) x * (om->M + 1) + k];
}

/*****************************************************************
 * 2. P7_FS_OPROFILE  (impl_sse.h:200-246)
 *****************************************************************/
typedef struct p7_fs_oprofile_s {
  bath_hip_fsprofile *dev;       /* codon emission table, transitions, p7_FLogsum table in HBM; NULL until Convert */
  bath_fs_profile    *gm_copy;   /* what Convert() was given (generic log scores), for Clone                        */
  float     xf[p7O_NXSTATES][p7O_NXTRANS];     /* log-space special transitions for the current (L, nj)             */

  int       codon_lengths;
  float     fsprob;

  off_t  offs[p7_NOFFSETS];
  off_t  roff;
  off_t  eoff;

  char  *name;
  char  *acc;
  char  *desc;
  char  *rf;
  char  *mm;
  char  *cs;
  char  *consensus;
  float  evparam[p7_NEVPARAM];
  float  cutoff[p7_NCUTOFFS];
  float  compo[p7_MAXABET];
  const ESL_ALPHABET *abc;

  int    L;                      /* configured target length in AMINO ACIDS (callers pass Ld/3, p7_domaindef.c:1019) */
  int    M;
  int    max_length;
  int    allocM;
  int    allocQ4;
  int    mode;
  float  nj;
  int    clone;
} P7_FS_OPROFILE;

static inline float
p7_fs_oprofile_FGetEmission(const P7_FS_OPROFILE *om_fs, int k, int c)
{
  return om_fs->gm_copy->rsc[(size_t) c * (om_fs->M + 1) + k];
}

/*****************************************************************
 * 3. P7_OIVX  (impl_sse.h:270-276): the intermediate-value rows of the frameshift kernels live in registers on the GPU;
 *    the object is kept so that p7_pipeline.c's create / grow / destroy calls link.
 *****************************************************************/
typedef struct p7_oivx_s {
  int      allocM;
  int      allocC;
  int      allocQ4;
} P7_OIVX;

/*****************************************************************
 * 4. P7_OMX  (impl_sse.h:315-348)
 *****************************************************************/
enum p7x_scells_e { p7X_M = 0, p7X_D = 1, p7X_I = 2 };
#define p7X_NSCELLS 3
enum p7x_fscells_e { p7X_FS_D = 0, p7X_FS_I = 1, p7X_FS_M = 2 };
enum p7x_fscodons_e { p7X_FS_C0 = 0, p7X_FS_C1 = 1, p7X_FS_C2 = 2, p7X_FS_C3 = 3, p7X_FS_C4 = 4, p7X_FS_C5 = 5 };
#define p7X_NSCELLS_FS 8
enum p7x_xcells_e { p7X_E = 0, p7X_N = 1, p7X_J = 2, p7X_B = 3, p7X_C = 4, p7X_SCALE = 5 };
#define p7X_NXCELLS 6

struct impl_hip_pass;            /* results of one device pass (impl_hip_domain.c) */

typedef struct p7_omx_s {
  int       M;
  int       L;
  int       nscells;

  /* The main cells never leave the device unless a caller asks for them (p7_omx_FDeconvert): <pass> refers to the device
   * pass this matrix took part in (Forward, Backward, decoding, optimal accuracy, null2 of one target), shared by the
   * matrices of that pass. */
  struct impl_hip_pass *pass;
  int       role;                /* which matrix of the pass this object stands for (impl_hip_domain.c)           */

  int       allocR;              /* rows / widths the caller asked for (p7_omx_GrowTo); bookkeeping only          */
  int       validR;
  int       allocQ4, allocQ8, allocQ16;
  size_t    ncells;

  float    *xmx;                 /* [0..L][ENJBCS], indexed [i*p7X_NXCELLS+s]: read by p7_domaindef.c through the decoding functions */
  void     *x_mem;
  int       allocXR;
  float     totscale;
  int       has_own_scales;

  int       debugging;
  FILE     *dfp;
} P7_OMX;

#define XMXo(i,s) (xmx[(i) * p7X_NXCELLS + s])

/*****************************************************************
 * 5. The external API: the subset of impl_sse.h:408-553 the bathsearch path references.
 *****************************************************************/

/* p7_omx.c */
extern P7_OMX      *p7_omx_Create   (int allocM, int allocL, int allocXL);
extern int          p7_omx_GrowTo   (P7_OMX *ox, int allocM, int allocL, int allocXL);
extern P7_OMX      *p7_omx_Create_dpf(int allocM, int allocL, int allocXL, int nscells);
extern int          p7_omx_GrowTo_dpf (P7_OMX *ox, int allocM, int allocL, int allocXL);
extern int          p7_omx_FDeconvert(P7_OMX *ox, P7_GMX *gx);
extern int          p7_omx_Reuse  (P7_OMX *ox);
extern void         p7_omx_Destroy(P7_OMX *ox);

/* p7_oprofile.c */
extern P7_OPROFILE *p7_oprofile_Create(int M, const ESL_ALPHABET *abc);
extern int          p7_oprofile_IsLocal(const P7_OPROFILE *om);
extern void         p7_oprofile_Destroy(P7_OPROFILE *om);
extern P7_OPROFILE *p7_oprofile_Clone(const P7_OPROFILE *om);

extern int          p7_oprofile_Convert    (const P7_PROFILE *gm, P7_OPROFILE *om);
extern int          p7_oprofile_ReconfigLength      (P7_OPROFILE *om, int L);
extern int          p7_oprofile_ReconfigMSVLength   (P7_OPROFILE *om, int L);
extern int          p7_oprofile_ReconfigMultihit    (P7_OPROFILE *om, int L);
extern int          p7_oprofile_ReconfigUnihit      (P7_OPROFILE *om, int L);

extern int          p7_oprofile_GetFwdTransitionArray(const P7_OPROFILE *om, int type, float *arr );
extern int          p7_oprofile_GetSSVEmissionScoreArray(const P7_OPROFILE *om, uint8_t *arr );
extern int          p7_oprofile_GetFwdEmissionScoreArray(const P7_OPROFILE *om, float *arr );

/* p7_fs_oprofile.c */
extern P7_FS_OPROFILE *p7_fs_oprofile_Create(int M, const ESL_ALPHABET *abc, int codon_lengths);
extern int             p7_fs_oprofile_IsLocal(const P7_FS_OPROFILE *om_fs);
extern void            p7_fs_oprofile_Destroy(P7_FS_OPROFILE *om_fs);
extern P7_FS_OPROFILE *p7_fs_oprofile_Clone(const P7_FS_OPROFILE *om_fs);

extern int             p7_fs_oprofile_Convert    (const P7_FS_PROFILE *gm_fs, P7_FS_OPROFILE *om_fs);

extern P7_OIVX        *p7_oivx_Create (int M_hint, int C);
extern int             p7_oivx_GrowTo (P7_OIVX *ov, int M, int C);
extern void            p7_oivx_Destroy(P7_OIVX *ov);
extern int             p7_fs_oprofile_ReconfigLength    (P7_FS_OPROFILE *om_fs, int L);
extern int             p7_fs_oprofile_ReconfigMultihit  (P7_FS_OPROFILE *om_fs, int L);
extern int             p7_fs_oprofile_ReconfigUnihit    (P7_FS_OPROFILE *om_fs, int L);

/* decoding.c */
extern int p7_Decoding      (const P7_OPROFILE *om, const P7_OMX *oxf,       P7_OMX *oxb, P7_OMX *pp);
extern int p7_DomainDecoding(const P7_OPROFILE *om, const P7_OMX *oxf, const P7_OMX *oxb, P7_DOMAINDEF *ddef);

/* decoding_fs.c */
extern int p7_Decoding_Frameshift            (const P7_FS_OPROFILE *om_fs, P7_OMX *fwd, const P7_OMX *bck);
extern int p7_DomainDecoding_Frameshift     (const P7_FS_OPROFILE *om_fs, const P7_OMX *oxf,  const P7_OMX *oxb,  P7_DOMAINDEF *ddef);

/* fwdback.c */
extern int p7_Forward       (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om,                    P7_OMX *fwd, float *opt_sc);
extern int p7_ForwardParser (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om,                    P7_OMX *fwd, float *opt_sc);
extern int p7_Backward      (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *fwd, P7_OMX *bck, float *opt_sc);
extern int p7_BackwardParser(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *fwd, P7_OMX *bck, float *opt_sc);

/* fwdback_fs.c */
extern int p7_ForwardParser_Frameshift_3Codons (const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs,                    P7_OMX *ox,  P7_OIVX *ov, float *opt_sc);
extern int p7_BackwardParser_Frameshift_3Codons(const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *fwd, P7_OMX *bck, P7_OIVX *ov, float *opt_sc);
extern int p7_Forward_Frameshift               (const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs,                    P7_OMX *ox,  P7_OIVX *ov, float *opt_sc);
extern int p7_Backward_Frameshift              (const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *fwd, P7_OMX *bck, P7_OIVX *ov, float *opt_sc);

/* ssvfilter.c */
extern int p7_SSVFilter    (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, float *ret_sc);

/* msvfilter.c */
extern int p7_MSVFilter           (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, float *ret_sc);
extern int p7_SSVFilter_BATH(const ESL_DSQ *dsq, int L, P7_OPROFILE *om, P7_OMX *ox, const P7_SCOREDATA *msvdata, P7_BG *bg, double P, P7_HMM_WINDOWLIST *windowlist);

/* null2.c */
extern int p7_Null2_ByExpectation(const P7_OPROFILE *om, const P7_OMX *pp, float *null2);
extern int p7_Null2_ByTrace      (const P7_OPROFILE *om, const P7_TRACE *tr, int zstart, int zend, P7_OMX *wrk, float *null2);

/* null2_fs.c */
extern int p7_Null2_fs_ByExpectation(const P7_FS_OPROFILE *om_fs, P7_OMX *pp, float *null2);

/* optacc.c */
extern int p7_OptimalAccuracy(const P7_OPROFILE *om, const P7_OMX *pp,       P7_OMX *ox, float *ret_e);
extern int p7_OATrace        (const P7_OPROFILE *om, const P7_OMX *pp, const P7_OMX *ox, P7_TRACE *tr);

/* optacc_fs.c */
extern int p7_OptimalAccuracy_Frameshift(const P7_FS_OPROFILE *om_fs, const P7_OMX *pp, P7_OMX *ox, float *ret_e);
extern int p7_OATrace_Frameshift(const P7_FS_OPROFILE *om_fs, const P7_OMX *pp, const P7_OMX *ox, P7_TRACE *tr);

/* stotrace.c */
extern int p7_StochasticTrace(ESL_RANDOMNESS *rng, const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, const P7_OMX *ox, P7_TRACE *tr);

/* stotrace_fs.c */
extern int p7_StochasticTrace_Frameshift(ESL_RANDOMNESS *rng, const ESL_DSQ *dsq, int L, const P7_FS_OPROFILE *om_fs, const P7_OMX *ox, P7_TRACE *tr);

/* vitfilter.c */
extern int p7_ViterbiFilter     (const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, float *ret_sc);
extern int p7_ViterbiFilter_BATH(const ESL_DSQ *dsq, int L, const P7_OPROFILE *om, P7_OMX *ox, const P7_SCOREDATA *ssvdata, float filtersc, double P, P7_HMM_WINDOWLIST *windowlist, float *ret_sc);

/*****************************************************************
 * 6. Implementation specific initialization (impl_sse.h:558-578)
 *****************************************************************/
/* Opens the GPU this thread works with (device = BATH_HIP_DEVICE or 0): one bath_hip_ctx per worker thread, the reference's
 * WORKER_INFO (bathsearch.c:34).  The SSE version sets flush-to-zero here; the kernels make no use of denormals. */
extern void impl_hip_init(void);
extern bath_hip_ctx *impl_hip_context(void);       /* the calling thread's context (created on first use) */
static inline void
impl_Init(void)
{
  impl_hip_init();
}
#endif /* P7_IMPL_HIP_INCLUDED */
