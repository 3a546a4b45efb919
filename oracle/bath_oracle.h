/* bath_oracle.h -- CPU restatement of the bathsearch DP hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity ORACLE for bath_amd. It is a plain-C, scalar restatement of the
 * reference algorithms (TravisWheelerLab/BATH, /root/reference/src). Nothing in the shipped
 * product (bath_amd/, include/) may call into it; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   The reference cannot be built in this image (it needs the un-vendored easel library,
 *   INSTALL:7-8), so there is no oracle/_ref. The oracle is pinned against the only golden
 *   vectors the reference tree holds for this path: the pipeline counters and hit tables recorded
 *   in the tutorial .out and .tbl files (tests/golden/), plus the reference unit tests' identities
 *   (MSV == GViterbi(SameAsMF), Fwd == Bwd, parser == full).  Pieces that live in easel
 *   (alphabets, esl_gencode ORF finder, esl_hmm_Forward, gumbel/exp tails) are restated from the
 *   published easel algorithms and are pinned only through those end-to-end counters.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef BATH_ORACLE_H
#define BATH_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* easel status codes the kernels return (easel.h; used at p7_pipeline.c:1471, msvfilter.c:179) */
#define BO_OK         0
#define BO_FAIL       1
#define BO_EMEM       5
#define BO_EFORMAT    7
#define BO_EINVAL    11
#define BO_ERANGE    16
#define BO_ENORESULT 19

#define BO_K_AMINO   20
#define BO_KP_AMINO  29
#define BO_K_DNA      4
#define BO_KP_DNA    18
#define BO_DSQ_SENTINEL 255

/* evparam indices, hmmer.h:67 */
enum { BO_MMU = 0, BO_MLAMBDA, BO_VMU, BO_VLAMBDA, BO_FTAU, BO_FLAMBDA, BO_FTAUFS3, BO_FTAUFS5, BO_NEVPARAM };

/* generic transition order, hmmer.h:221-231 */
enum { BO_MM = 0, BO_IM, BO_DM, BO_BM, BO_MD, BO_DD, BO_MI, BO_II, BO_NTRANS };
/* hmm file transition order, p7_hmmfile.c:1615 / hmmer.h p7H_* */
enum { BO_H_MM = 0, BO_H_MI, BO_H_MD, BO_H_IM, BO_H_II, BO_H_DM, BO_H_DD, BO_H_NTRANS };
/* special states, hmmer.h:202-218 */
enum { BO_XE = 0, BO_XN, BO_XJ, BO_XC };
enum { BO_LOOP = 0, BO_MOVE };
/* generic DP special cells xmx[i*5+s], hmmer.h:621-628 */
enum { BO_GE = 0, BO_GN, BO_GJ, BO_GB, BO_GC, BO_NXCELLS };
/* generic DP main cells, hmmer.h:604-619 */
enum { BO_GD = 0, BO_GI = 1, BO_GM = 2, BO_NSCELLS = 3, BO_NSCELLS_FS = 8 };

#define BO_MAXCODONS5 1367
#define BO_MAXCODONS3 338
#define BO_DEGEN5_C   1364
#define BO_DEGEN5_QC1 1365
#define BO_DEGEN5_QC2 1366
#define BO_DEGEN3_C   336
#define BO_DEGEN3_QC1 337

/* ---------- core HMM (p7_hmm.c / hmmer.h P7_HMM) ---------- */
typedef struct {
  int    M;
  int    max_length;
  int    ct;                       /* NCBI codon table id                      */
  float  fsprob;                   /* FRAMESHIFT PROB                          */
  float *t;                        /* [ (M+1) * 7 ]   p7H order                */
  float *mat;                      /* [ (M+1) * 20 ]                           */
  float *ins;                      /* [ (M+1) * 20 ]                           */
  float  compo[BO_K_AMINO];
  float  evparam[BO_NEVPARAM];
  char   name[128];
  char   acc[64];
  char  *consensus;                /* [M+2] */
} bo_hmm;

/* ---------- generic profile (hmmer.h:338 P7_PROFILE) ---------- */
typedef struct {
  int    M, L, max_length;
  float  nj;
  float *tsc;                      /* [ (M+1)*8 ]  (k=0..M-1 used; row M zeroed/-inf as reference leaves it) */
  float *rsc;                      /* [Kp][ (M+1)*2 ]  MSC at [x][k*2], ISC at [x][k*2+1]                   */
  float  xsc[4][2];
  float  evparam[BO_NEVPARAM];
  float  compo[BO_K_AMINO];
} bo_profile;

/* ---------- frameshift profile (hmmer.h:371 P7_FS_PROFILE) ---------- */
typedef struct {
  int    M, L, max_length, codon_lengths, maxcodons;
  float  nj, fsprob;
  float *tsc;                      /* [ (M+1)*8 ] */
  float *rsc;                      /* [ (maxcodons+Kp) * (M+1) ]  row-major rsc[c][k] */
  uint8_t *codons;                 /* [ (M+1) * maxcodons ]  best amino per (k,codon)  */
  uint8_t *indel_pos;              /* [ (M+1) * maxcodons ]                             */
  float  xsc[4][2];
  float  evparam[BO_NEVPARAM];
  float  compo[BO_K_AMINO];
} bo_fs_profile;

/* ---------- "optimized profile" scores, UNSTRIPED (impl_sse.h:75 P7_OPROFILE) ---------- */
typedef struct {
  int      M, L, max_length;
  float    nj;
  /* MSV */
  uint8_t *rb;                     /* [Kp][M+1] biased byte costs (k=1..M; [x][0] = 255)          */
  uint8_t  tbm_b, tec_b, tjb_b, base_b, bias_b;
  float    scale_b;
  /* Viterbi */
  int16_t *rw;                     /* [Kp][M+1] word scores                                       */
  int16_t *tw;                     /* [M+1][8] in generic order MM,IM,DM,BM,MD,DD,MI,II; value for
                                      "transition out of/into node k" exactly as the striped twv holds */
  int16_t  xw[4][2];
  float    scale_w;
  int16_t  base_w, ddbound_w;
  /* Forward (odds ratios) */
  float   *rf;                     /* [Kp][M+1] */
  float   *tf;                     /* [M+1][8]  */
  float    xf[4][2];
  float    evparam[BO_NEVPARAM];
  float    compo[BO_K_AMINO];
  /* log-odds match scores [Kp][M+1] and log transitions [M+1][8] of the generic profile: what p7_pli_computeAliScores_BATH
   * reads from gm_fs5 (its amino rows and tsc are the same numbers, modelconfig.c:313-318 vs :106-113) */
  float   *msc;
  float   *tsc;
} bo_oprofile;

/* ---------- background (p7_bg.c) ---------- */
typedef struct {
  float f[BO_K_AMINO];
  float p1;
  /* 2-state filter HMM (esl_hmm) */
  float t[2][3];
  float e[2][BO_K_AMINO];
  float eo[BO_KP_AMINO][2];
  float pi[3];
} bo_bg;

/* ---------- score data (p7_scoredata.c) ---------- */
typedef struct {
  int      M;
  uint8_t *ssv_scores;             /* [(M+1)*Kp] */
  float   *prefix_lengths;         /* [M+1] */
  float   *suffix_lengths;         /* [M+1] */
} bo_scoredata;

typedef struct {                   /* hmmer.h P7_HMM_WINDOW (fields used on the path) */
  int32_t id, n, k, length;
  float   score;
} bo_window;

typedef struct { bo_window *w; int count, size; } bo_windowlist;

/* ---------- ORFs (esl_gencode_Process*, easel; bathsearch.c:384-392) ---------- */
typedef struct {
  int32_t start, end;              /* 1-based nt coords on the CURRENT strand's sequence; start<end */
  int32_t n;                       /* aa length                                                    */
  int32_t frame;                   /* 0..2                                                         */
  int64_t off;                     /* offset of dsq[1] in the aa pool (sentinel at off-1... see .c) */
} bo_orf;

typedef struct { bo_orf *orf; int count, size; uint8_t *aa; int64_t aa_n, aa_size; } bo_orfblock;

/* ============ alphabet / genetic code  (abc.c) ============ */
int  bo_amino_digitize(char c);                 /* -1 if not in alphabet */
int  bo_dna_digitize(char c);
extern const char bo_amino_syms[];              /* "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~" */
extern const char bo_dna_syms[];                /* "ACGT-RYMKSWHBVDN*~"            */
int  bo_amino_degen(int x, int y);              /* does degenerate x include canonical y */
int  bo_dna_degen(int x, int y);
uint8_t bo_dna_complement(uint8_t x);
int  bo_gencode_basic(int ct, uint8_t basic[64]);              /* 16*n1+4*n2+n3 -> amino code */
uint8_t bo_gencode_translate(const uint8_t basic[64], const uint8_t *codon3); /* esl_gencode_GetTranslation */
void bo_revcomp(const uint8_t *dsq, int n, uint8_t *out);      /* dsq,out 1-based with sentinels */

/* ============ hmm file (hmmfile.c) ============ */
int  bo_hmmfile_read(const char *path, int index, bo_hmm **ret);  /* index-th model in file */
int  bo_hmmfile_count(const char *path);
void bo_hmm_free(bo_hmm *h);

/* ============ profiles (profile.c) ============ */
void bo_bg_create(bo_bg *bg);                                    /* p7_bg.c:44 */
void bo_bg_setlength(bo_bg *bg, int L);                          /* p7_bg.c:189 */
void bo_bg_setfilter(bo_bg *bg, int M, const float *compo);      /* p7_bg.c:449 */
float bo_bg_nullone(const bo_bg *bg, int L);                     /* p7_bg.c:356 */
float bo_bg_fs_nullone(const bo_bg *bg, int aminoL);             /* p7_bg.c:377 */
float bo_bg_filterscore(const bo_bg *bg, const uint8_t *dsq, int L);  /* p7_bg.c:491 */
float bo_bg_fs_filterscore(const bo_bg *bg, const uint8_t *dna, int L, const uint8_t basic[64]); /* p7_bg.c:522 */

bo_profile    *bo_profile_config(const bo_hmm *h, const bo_bg *bg, int L);           /* modelconfig.c:48 (p7_LOCAL) */
void           bo_profile_reconfig_length(bo_profile *gm, int L);                    /* modelconfig.c:723 */
void           bo_profile_free(bo_profile *gm);
bo_fs_profile *bo_fs_profile_config(const bo_hmm *h, const bo_bg *bg, const uint8_t basic[64], int codon_lengths, int L_amino); /* modelconfig.c:220 */
void           bo_fs_profile_reconfig_length(bo_fs_profile *gm, int L_amino);        /* modelconfig.c:760 */
void           bo_fs_profile_reconfig_unihit(bo_fs_profile *gm, int L_amino);        /* modelconfig.c:868 */
void           bo_fs_profile_reconfig_multihit(bo_fs_profile *gm, int L_amino);      /* modelconfig.c:825 */
void           bo_fs_profile_free(bo_fs_profile *gm);
bo_oprofile   *bo_oprofile_convert(const bo_profile *gm);                            /* p7_oprofile.c:1091 */
void           bo_oprofile_reconfig_length(bo_oprofile *om, int L);                  /* p7_oprofile.c:1261 */
void           bo_oprofile_reconfig_msv_length(bo_oprofile *om, int L);              /* p7_oprofile.c:1286 */
void           bo_oprofile_free(bo_oprofile *om);
bo_scoredata  *bo_scoredata_create(const bo_oprofile *om);                           /* p7_scoredata.c:57,314 */
void           bo_scoredata_free(bo_scoredata *sd);

/* ============ statistics (easel esl_gumbel.c / esl_exponential.c) ============ */
double bo_gumbel_surv(double x, double mu, double lambda);
double bo_gumbel_invsurv(double p, double mu, double lambda);
double bo_exp_surv(double x, double mu, double lambda);

/* ============ logsum (logsum.c:80-111) ============ */
void  bo_flogsum_init(void);
float bo_flogsum(float a, float b);
void  bo_flogsum_set_exact(int exact);  /* 1: use exact log(1+exp()) instead of table (logsum.c:109) */
const float *bo_flogsum_table(void);

/* ============ filters (filters.c) ============ */
int   bo_ssvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);  /* ssvfilter.c:876 */
int   bo_msvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);  /* msvfilter.c:74 (incl. SSV first) */
int   bo_msvfilter_noSSV(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc); /* msvfilter.c:106-207 only */
int   bo_ssvfilter_bath(const uint8_t *dsq, int L, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg, double P, bo_windowlist *wl); /* msvfilter.c:250 */
int   bo_vitfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);  /* vitfilter.c:83 */
int   bo_vitfilter_bath(const uint8_t *dsq, int L, const bo_oprofile *om, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc); /* vitfilter.c:286 */
int   bo_forward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, float *xmx6 /* (L+1)*6 or NULL */, float *ret_sc); /* fwdback.c:132,256 */
int   bo_backward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, const float *fwd_xmx6, float *bck_xmx6, float *ret_sc); /* fwdback.c:236,468 */
/* generic scalar std kernels used for the exact-emulation identities (generic_viterbi.c, generic_msv.c) */
int   bo_gviterbi(const uint8_t *dsq, int L, const bo_profile *gm, float *ret_sc);    /* generic_viterbi.c */
int   bo_gforward(const uint8_t *dsq, int L, const bo_profile *gm, float *ret_sc);    /* generic_fwdback.c */
bo_profile *bo_profile_same_as_mf(const bo_oprofile *om, const bo_profile *gm);       /* p7_oprofile.c:2141 */
bo_profile *bo_profile_same_as_vf(const bo_oprofile *om, const bo_profile *gm);       /* p7_oprofile.c:2202 */

/* kernel hooks of the cascade (oracle/sse/sse_hooks.c): the scalar restatements above, or -- after bo_pipeline_use_sse(1) -- the
 * SSE2 striped ones of oracle/sse (the impl_sse-equivalent CPU baseline) */
int   bo_k_msvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);
int   bo_k_vitfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);
int   bo_k_vitfilter_bath(const uint8_t *dsq, int L, const bo_oprofile *om, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc);
int   bo_k_forward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc);
void  bo_pipeline_use_sse(int on);
void  bo_windowlist_init(bo_windowlist *wl);
void  bo_windowlist_free(bo_windowlist *wl);

/* ============ ORF finder (translate.c) ============ */
void  bo_orfblock_init(bo_orfblock *b);
void  bo_orfblock_reuse(bo_orfblock *b);
void  bo_orfblock_free(bo_orfblock *b);
/* translate one strand of dsq[1..n] (already reverse-complemented by the caller for the bottom strand) */
int   bo_translate_orfs(const uint8_t *dsq, int n, const uint8_t basic[64], int minlen, bo_orfblock *out);
/* ... with initiation codons (bathsearch -m / -M): is_init[64] marks the canonical codons that may start an ORF
 * (gcode->is_initiator); using_initiators != 0: the initiation codon is translated as M whatever it encodes
 * (esl_gencode_WorkstateCreate: -m or -M).  NULL is_init = any codon (bo_translate_orfs). */
int   bo_translate_orfs_init(const uint8_t *dsq, int n, const uint8_t basic[64], const uint8_t *is_init, int using_initiators,
                             int minlen, bo_orfblock *out);
/* gcode->is_initiator[16*n1+4*n2+n3] for mode 0 (any: esl_gencode_SetInitiatorAny), 1 (the NCBI table's start codons, what
 * esl_gencode_Set leaves), 2 (AUG only: esl_gencode_SetInitiatorOnlyAUG) */
int   bo_gencode_initiators(int ct, int mode, uint8_t is_init[64]);
int   bo_gencode_is_initiator(const uint8_t is_init[64], const uint8_t *codon3);   /* esl_gencode_IsInitiator: every expansion of a degenerate codon must be one */
void  bo_set_seed(uint32_t seed);              /* the stochastic-trace ensembles' generator seed (pli->r), 42 by default; 0 = no reseeding */

/* ============ frameshift generic kernels (fs_*.c) ============ */
/* GMX-like matrices: dp rows of (M+1)*nscells floats, xmx (L+1)*5 floats */
typedef struct { int M, L, nrows, nscells; float *dp; float *xmx; } bo_gmx;
bo_gmx *bo_gmx_create(int M, int nrows, int L, int nscells);
void    bo_gmx_free(bo_gmx *gx);
#define BO_DP(gx,i,k,s) ((gx)->dp[((size_t)(i) * ((gx)->M+1) + (k)) * (gx)->nscells + (s)])
#define BO_X(gx,i,s)    ((gx)->xmx[(size_t)(i) * BO_NXCELLS + (s)])

/* c5_compat=1 reproduces generic_fwdback_frameshift.c:324 `(i-5)%5` aliasing; 0 uses (i-4) like fwdback_fs.c:1464 */
int bo_gforward_fs(const uint8_t *dsq, int L, const bo_fs_profile *gm5, bo_gmx *gx, int c5_compat, float *ret_sc);      /* generic_fwdback_frameshift.c:64 */
int bo_gbackward_fs(const uint8_t *dsq, int L, const bo_fs_profile *gm5, bo_gmx *gx, float *ret_sc);                    /* :1035 */
int bo_gforward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc);            /* :451 */
int bo_gbackward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc);           /* :1422 */
/* ... through the kernel hook (oracle/sse/sse_hooks.c): the scalar log-space parser, or the SSE2 striped probability-space one (bo_fs_use_sse) */
int bo_k_gforward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc);
int bo_k_gbackward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc);
int bo_gdecoding_fs(const bo_fs_profile *gm5, bo_gmx *fwd /* in: Forward; out: posteriors */, const bo_gmx *bck);       /* generic_decoding_frameshift.c:36 */
int bo_goptacc_fs(const bo_fs_profile *gm5, const bo_gmx *pp, bo_gmx *oa, float *ret_e);                               /* generic_optacc_frameshift.c:53 */
int bo_gnull2_fs(const bo_fs_profile *gm5, bo_gmx *pp /* row 0 is clobbered, as in the reference */, float *null2 /* Kp */);                                    /* generic_null2_frameshift.c:46 */

/* ============ pipeline (pipeline.c): p7_Pipeline_BATH restatement, filter cascade ============ */
typedef struct {
  double F1, F2, F3, F4;
  int    do_biasfilter, fs_pipe, minlen;
  /* counters (hmmer.h:1115-1128) */
  int64_t nres, n_orfs, n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
  int64_t pos_past_msv, pos_past_bias, pos_past_vit, pos_past_fwd;
  int64_t cells_msv, cells_vit, cells_fwd;
  double  E;                                   /* reporting E-value threshold (p7_pipeline.c:147), 10.0 */
  int32_t context;                             /* ESL_SQ.C of the window being searched: leading nucleotides the previous window
                                                * already covered (bathsearch.c:1099; p7_pipeline.c:1635-1637); 0 for whole sequences */
  /* option state of p7_pipeline_Create_BATH (p7_pipeline.c:94-234) and bathsearch.c:718-719, :831-833 that the hot path reads;
   * bo_pipeline_init sets the defaults */
  int32_t do_null2;                            /* --nonull2 clears it (:199, :213); read at :1063, :1230                              */
  int32_t std_pipe;                            /* --fsonly clears it (:107); read at :1457, :1480                                     */
  int32_t strands;                             /* 0 both, 1 top only (--strand plus), 2 bottom only (minus); bathsearch.c:1069, :1082 */
  int32_t initiator;                           /* 0 any codon (esl_gencode_SetInitiatorAny, the default), 1 the codon table's own
                                                * initiators (-M), 2 AUG only (-m); bathsearch.c:718-719                              */
  int32_t ct;                                  /* NCBI table id (for initiator == 1)                                                  */
  int32_t inc_by_E;                            /* --incT clears it (:165-175); the pipeline's early tests go by it (:1080, :1247,
                                                * p7_domaindef.c:1034), NOT by by_E                                                  */
  double  T;                                   /* pli->T: -T, 0.0 when not given (:148, :155); the early test's bit-score threshold
                                                * when inc_by_E is off                                                                */
  uint32_t seed;                               /* --seed, 42 (:98); 0: no reseeding between regions (:140-143)                        */
} bo_pipeline;

typedef struct {                 /* per-ORF cascade record (what the GPU path must reproduce) */
  int32_t strand, frame, start, end, n;
  int32_t stage;                 /* 0: failed MSV, 1: failed bias, 2: failed Vit, 3: failed Fwd(F3/F4), 4: passed */
  int32_t msv_status, vit_status;
  float   usc, nullsc, filtersc, vfsc, fwdsc;
  double  P;
} bo_orfresult;

typedef struct {                 /* per DNA-window record of p7_pli_Frameshift (p7_pipeline.c:1368-1470) */
  int32_t strand, n, length, k;  /* window start (1-based on the strand being read), length in nt, k of the seed window */
  int32_t orf_cnt, k_min, k_max;
  float   tot_orfsc, nullsc, filtersc, fwdsc;
  double  P_tot, P_min, P_fs, P_null;
  int32_t branch;                /* 1: frameshift branch (:1464), 2: standard branch (:1479) */
  int32_t ndom;                  /* domains defined in this window (frameshift branch, when a 5-codon profile is given) */
} bo_fswindow;

/* ---- domain definition of the frameshift branch (fs_domaindef.c) ---- */
enum { BO_T_S = 0, BO_T_N, BO_T_B, BO_T_M, BO_T_D, BO_T_I, BO_T_E, BO_T_J, BO_T_C, BO_T_T };   /* trace states */
typedef struct { int N, nalloc; int8_t *st; int32_t *k, *i, *c; float *pp; } bo_trace;       /* P7_TRACE with codon lengths c[] */
void bo_trace_init(bo_trace *t);
void bo_trace_free(bo_trace *t);
typedef struct {                 /* P7_DOMAIN fields the frameshift branch fills, then the hit's scores */
  int32_t ienv, jenv, iali, jali;  /* nt coordinates: in the window while defining, on the sequence after post-processing */
  int32_t ihmm, jhmm;
  float   envsc, oasc, domcorrection;      /* nats, expected residues, nats */
  float   dombias, bitscore, pre_score;    /* nats, bits, bits (p7_pipeline.c:1064-1108) */
  double  lnP;
  int32_t reported;                        /* passes E * Z <= E threshold (:1080) */
  int32_t n_shifted_codons;                /* match states emitting a quasi-codon (length != 3) */
  int32_t trace_idx;                       /* dom->tr: index for bo_traces_get */
} bo_fsdomain;
/* dom->tr as rescore_isolated_domain_frameshift / _bath keep it (p7_domaindef.c:1171, :1330), the states from the first to the last
 * match state, in the reference's conventions (p7_trace_fs_AppendWithPP, p7_trace.c:2303; p7_trace_fs_Convert, :405): st 1 M / 2 D /
 * 3 I (p7T_*), node k, i = the codon's last nucleotide in windowsq (D: envelope start - 1 in the frameshift branch, 0 in the standard
 * one), c = codon length of a match state (else 0), pp = posterior of an M / I state.  A test-side store, reset by the caller. */
typedef struct { int32_t N, win_start, orf_start, frameshift; int8_t *st; int32_t *k, *i; int8_t *c; float *pp; } bo_domtrace;
void bo_traces_reset(void);
int  bo_traces_count(void);
const bo_domtrace *bo_traces_get(int idx);
int  bo_traces_push(int N, int win_start, int orf_start, int frameshift);   /* returns the index; the arrays are allocated, the caller fills them */
void bo_gdomain_decoding_fs(const bo_fs_profile *gm5, const bo_gmx *fwd, const bo_gmx *bck, float *btot, float *etot, float *mocc); /* generic_decoding_frameshift.c:204 */
int  bo_goatrace_fs(const bo_fs_profile *gm, const bo_gmx *pp, const bo_gmx *gx, bo_trace *tr);                                  /* generic_optacc_frameshift.c:373 */
double bo_exp_logsurv(double x, double mu, double lambda);

/* ---- multi-domain regions: stochastic-trace ensemble and clustering (stotrace.c; parity unpinned, see its header) ---- */
typedef struct { uint32_t x; } bo_rng;                 /* easel's "fast" generator (esl_randomness_CreateFast) */
void   bo_rng_init(bo_rng *r, uint32_t seed);
double bo_rng_next(bo_rng *r);
int    bo_stochastic_trace(bo_rng *rng, int L, const bo_oprofile *om, const float *fwd, const float *fx, int8_t *st, int32_t *tk, int32_t *ti, int cap);
int    bo_selftest_cluster_segments(int n, const int32_t *idx, const int32_t *i, const int32_t *j, const int32_t *k, const int32_t *m,
                                    int nsamples, int fs, int *env, int max_env);
int    bo_region_trace_ensemble_fs(const bo_fs_profile *gm5, int ireg, int jreg, const bo_gmx *fwd, int *env, int max_env);
int    bo_region_trace_ensemble(const bo_oprofile *om, const uint8_t *dsq, int ireg, int jreg, const float *fwd, const float *fx,
                                float *n2sc, int *env, int max_env);

/* ---- domain definition of the standard branch (domaindef.c) ---- */
int  bo_forward_full(const uint8_t *dsq, int L, const bo_oprofile *om, float *dpf, float *xmx, float *ret_sc);                                   /* fwdback.c:94 */
int  bo_backward_full(const uint8_t *dsq, int L, const bo_oprofile *om, const float *fwd_xmx, float *dpb, float *bck_xmx, float *ret_sc, int *own_scales); /* :196 */
void bo_oprofile_reconfig_unihit(bo_oprofile *om, int L);       /* p7_oprofile.c:1418 */
void bo_oprofile_reconfig_multihit(bo_oprofile *om, int L);     /* p7_oprofile.c:1395 */
int  bo_std_envelope_trace(bo_oprofile *om, const uint8_t *dsq, int L, int *path_st, int *path_k, int *path_i, float *oasc_out);   /* test hook: optacc.c:225 on one envelope */
int  bo_domain_decoding(const bo_oprofile *om, const float *fx, const float *bx, int L, int own_scales, float *btot, float *etot, float *mocc); /* decoding.c:155 */
int  bo_domaindef_std(bo_pipeline *pli, bo_oprofile *om, bo_bg *bg, const uint8_t *dsq, int n, int orf_start, int win_start,
                      int complementarity, int seq_n, bo_fsdomain **doms, int *ndom, int *dalloc, int *nskipped, const uint8_t *strand_dsq);

void bo_pipeline_init(bo_pipeline *pli, int fs_pipe);
void  bo_local_compo(const bo_scoredata *sd, const bo_oprofile *om, const bo_bg *bg, int k_min, int k_max, float *compo); /* p7_pipeline.c:427 */
/* frameshift stage for one strand (fs_pipeline.c); dsq[1..n] is the strand being read */
int  bo_pli_frameshift(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, bo_fs_profile *gm5, const bo_scoredata *sd, bo_bg *bg, const uint8_t basic[64],
                       const bo_orfblock *blk, const double *P_orf, const float *fwdsc, const bo_windowlist *hw,
                       const uint8_t *dsq, int n, int complementarity, bo_fswindow **fw, int *nfw, int *fw_alloc,
                       bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped);
int  bo_domaindef_fs(bo_pipeline *pli, bo_fs_profile *gm3, bo_fs_profile *gm5, bo_bg *bg, const uint8_t *wdsq, int L,
                     int window_start, int complementarity, int seq_n, bo_fsdomain **doms, int *ndom, int *dalloc, int *nskipped);
int  bo_pipeline_window_hits(bo_pipeline *pli, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg,
                             const uint8_t basic[64], const uint8_t *dna, int n,
                             bo_orfresult **res, int *nres, int *res_alloc, bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped);
/* bo_pipeline_window with the frameshift stage (pli->fs_pipe must be set; gm3 = 3-codon frameshift profile) */
int  bo_pipeline_window_fs(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, const bo_scoredata *sd, bo_bg *bg,
                           const uint8_t basic[64], const uint8_t *dna, int n,
                           bo_orfresult **res, int *nres, int *res_alloc, bo_fswindow **fw, int *nfw, int *fw_alloc);
/* ... and domain definition + hit scores for the windows that take the frameshift branch (gm5 = 5-codon profile) */
int  bo_pipeline_window_fsdom(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, bo_fs_profile *gm5, const bo_scoredata *sd, bo_bg *bg,
                              const uint8_t basic[64], const uint8_t *dna, int n,
                              bo_orfresult **res, int *nres, int *res_alloc, bo_fswindow **fw, int *nfw, int *fw_alloc,
                              bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped);
/* run translate + cascade on both strands of one DNA window dsq[1..n]; results appended (realloc'd) */
int  bo_pipeline_window(bo_pipeline *pli, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg,
                        const uint8_t basic[64], const uint8_t *dna, int n,
                        bo_orfresult **res, int *nres, int *res_alloc);

#ifdef __cplusplus
}
#endif
#endif
