/* bath_hip.h -- C ABI of the MI355X (gfx950) backend for BATH's bathsearch DP hot path.
 *
 * This is the drop-in boundary: the reference selects its DP backend at build time through the
 * impl_* directory contract (src/hmmer.h:1044-1052, src/Makefile.in:39) and every kernel there has
 * the shape  int f(const ESL_DSQ *dsq, int L, const PROFILE *om, MATRIX *ox, float *opt_sc)
 * (src/impl_sse/impl_sse.h:408-553).  A GPU cannot be fed one target per call, so each entry
 * point below is the BATCHED form of the reference function named in its comment: same inputs
 * (digital sequences, 1..L, easel codes), same outputs (nat scores, easel status codes
 * eslOK=0 / eslERANGE=16 / eslENORESULT=19), one array element per target.  INTEGRATION.md shows the
 * impl_hip/ shims that bind these to the reference's single-target prototypes.
 *
 * Plain C types only.  All pointers are HOST pointers unless the name ends in _dev.
 * Every function returns 0 (BATH_OK) or an easel-compatible error code; bath_hip_last_error()
 * returns a message.  There is no CPU fallback: if no gfx950 device is usable, bath_hip_init fails.
 */
#ifndef BATH_HIP_H
#define BATH_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BATH_OK          0
#define BATH_EFAIL       1
#define BATH_EMEM        5
#define BATH_EFORMAT     7
#define BATH_EINVAL     11
#define BATH_ERANGE     16   /* eslERANGE: filter score overflow (score = +inf) / Forward range error */
#define BATH_ENORESULT  19   /* eslENORESULT: SSV cannot decide, full MSV needed                      */
#define BATH_ENODEVICE  40

#define BATH_KP_AMINO   29   /* easel amino alphabet "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~" */
#define BATH_K_AMINO    20
#define BATH_KP_DNA     18   /* easel DNA alphabet   "ACGT-RYMKSWHBVDN*~"            */
#define BATH_NEVPARAM    8   /* MMU MLAMBDA VMU VLAMBDA FTAU FLAMBDA FTAUFS3 FTAUFS5 (hmmer.h:67) */
#define BATH_FTAU        4
#define BATH_FLAMBDA     5
#define BATH_FTAUFS3     6

typedef struct bath_hip_ctx      bath_hip_ctx;      /* one GPU + one HIP stream (a reference "worker", bathsearch.c:34-52) */
typedef struct bath_hip_oprofile bath_hip_oprofile; /* device-resident P7_OPROFILE  (impl_sse.h:75)   */
typedef struct bath_hip_fsprofile bath_hip_fsprofile; /* device-resident P7_FS_OPROFILE (impl_sse.h:200) */
typedef struct bath_hip_seqs     bath_hip_seqs;     /* device-resident block of digital sequences (ESL_SQ_BLOCK) */

/* ------------------------------------------------------------------------------------------
 * Host-side model objects (generic layer the impl boundary consumes).
 * ------------------------------------------------------------------------------------------ */

/* P7_HMM as read from a BATH3/f file (p7_hmmfile.c:1342 read_asc30hmm). */
typedef struct {
  int32_t M, max_length, ct;
  float   fsprob;
  float  *t;        /* [(M+1)*7]  MM MI MD IM II DM DD   */
  float  *mat;      /* [(M+1)*20]                         */
  float  *ins;      /* [(M+1)*20]                         */
  float   compo[BATH_K_AMINO];
  float   evparam[BATH_NEVPARAM];
  char    name[128];
  char    acc[64];  /* ACC line, "" when absent                                              */
  char   *consensus;/* [M+2] consensus residue per node, consensus[0] = ' ' (P7_HMM.consensus): the file's CONS column,
                     * or p7_hmm_SetConsensus's rule when the file has none (p7_hmm.c: most probable residue,
                     * upper case if its probability is >= 0.5)                                */
  char   *rf;       /* [M+2] reference annotation per node (P7_HMM.rf, "RF yes"), rf[0] = ' '; NULL when the model has none */
  char   *cs;       /* [M+2] consensus structure annotation (P7_HMM.cs, "CS yes"); NULL when the model has none            */
} bath_hmm;

/* P7_PROFILE (hmmer.h:338): what p7_oprofile_Convert() reads. */
typedef struct {
  int32_t M, L, max_length;
  float   nj;
  float  *tsc;      /* [M*8]  order MM IM DM BM MD DD MI II (hmmer.h:221), BM stored off by one */
  float  *rsc;      /* [Kp][(M+1)*2]  match at [x][2k], insert at [x][2k+1]                     */
  float   xsc[4][2];/* [E N J C][LOOP MOVE]                                                     */
  float   evparam[BATH_NEVPARAM];
  float   compo[BATH_K_AMINO];
} bath_profile;

/* P7_FS_PROFILE (hmmer.h:371): what p7_fs_oprofile_Convert() reads. */
typedef struct {
  int32_t M, L, max_length, codon_lengths, maxcodons;
  float   nj, fsprob;
  float  *tsc;      /* [M*8]                                   */
  float  *rsc;      /* [(maxcodons+Kp)][M+1]  rsc[codon][k]    */
  uint8_t *codons;  /* [(M+1)][maxcodons] argmax amino acid    */
  uint8_t *indel_pos;
  float   xsc[4][2];
  float   evparam[BATH_NEVPARAM];
  float   compo[BATH_K_AMINO];
} bath_fs_profile;

int  bath_hmmfile_count(const char *path);
int  bath_hmmfile_read(const char *path, int index, bath_hmm **ret);            /* p7_hmmfile.c:1342 */
void bath_hmm_destroy(bath_hmm *hmm);
int  bath_gencode_basic(int ncbi_table, uint8_t basic[64]);                     /* gcode->basic[16a+4b+c] */
int  bath_profile_config(const bath_hmm *hmm, int L, bath_profile **ret);       /* p7_ProfileConfig, modelconfig.c:48 (p7_LOCAL, Swiss-Prot bg) */
void bath_profile_destroy(bath_profile *gm);
int  bath_fs_profile_config(const bath_hmm *hmm, const uint8_t basic[64], int codon_lengths, int L_amino,
                            bath_fs_profile **ret);                             /* p7_ProfileConfig_fs, modelconfig.c:220 */
void bath_fs_profile_destroy(bath_fs_profile *gm);

/* ------------------------------------------------------------------------------------------
 * Device context.
 * ------------------------------------------------------------------------------------------ */
int         bath_hip_init(int device, bath_hip_ctx **ctx);          /* impl_Init(), impl_sse.h:559 */
void        bath_hip_finalize(bath_hip_ctx *ctx);
const char *bath_hip_last_error(const bath_hip_ctx *ctx);
int         bath_hip_synchronize(bath_hip_ctx *ctx);
/* Release the lanes, side contexts and side streams the context created lazily for the concurrency inside its calls (created
 * again on demand).  For a context that will sit idle while other contexts of the process work: its streams hold hardware
 * queues.  Results of earlier calls on the context are invalid afterwards. */
int         bath_hip_trim(bath_hip_ctx *ctx);
void       *bath_hip_stream(bath_hip_ctx *ctx);                      /* hipStream_t, for event timing */
/* Frameshift recursions.  1 (the default): every sum along the model -- D(i,k), E(i), Backward's B(i) -- runs node by node in the
 * reference's order (generic_fwdback_frameshift.c:340-365, :577-590, :1279-1283), so every table log-sum has the reference's
 * operands: scores, special-state rows and matrices are BIT-IDENTICAL to the generic reference (the north star's 1e-4 with
 * room to spare) and every frameshift-branch domain has the reference's coordinates exactly.
 * 0 = "fast": the same table log-sums associated by wavefront scans in the multihit recursions (3-codon parsers, the regions'
 * Forward), ~1.3x faster on the bench's --fs pass.  WARNING: the fast mode is OUTSIDE the parity contract and CHANGES RESULTS, not
 * only the last digits of scores: window scores move by up to ~1e-3 nats (beyond 1e-4 relative for scores near zero), so a window
 * at a threshold may take the other branch, region boundaries may shift by a step and a clustered region's stochastic traces take
 * other turns -- on the bench's block 89 of 4789 domains come out with OTHER COORDINATES (bench.py: fs.fast.domains_identical_to_
 * strict_mode).  Use it for throughput estimates or pre-screening, never where hit lists are compared with the reference's.
 * Applies to the pipeline entry points and to BATH_LOGSUM_CONTEXT. */
int         bath_hip_set_fs_strict(bath_hip_ctx *ctx, int on);
/* Measurement aid: 1 = the envelope stage (bath_hip_fs5_envelopes and the domain stage's batches) runs its Backward wavefront AFTER the
 * Forward wavefront on the same stream instead of beside it, so that a kernel's HIP-event span is its time alone on the chip
 * (bench.py: fs.roofline.alone); 0 = side by side (the default); -1 = whatever BATH_HIP_FS_SERIAL says.  Results do not change. */
int         bath_hip_set_fs_serial(bath_hip_ctx *ctx, int on);

/* ------------------------------------------------------------------------------------------
 * Optimized profile (P7_OPROFILE surface).
 * ------------------------------------------------------------------------------------------ */
int  bath_hip_oprofile_convert(bath_hip_ctx *ctx, const bath_profile *gm, bath_hip_oprofile **ret); /* p7_oprofile_Convert, p7_oprofile.c:1091 */
void bath_hip_oprofile_destroy(bath_hip_oprofile *om);
/* P7_OPROFILE.consensus (impl_sse.h:128), consensus[1..M] as in bath_hmm: needed only for the hits' percent identity. */
int  bath_hip_oprofile_set_consensus(bath_hip_oprofile *om, const char *consensus);
int  bath_hip_oprofile_M(const bath_hip_oprofile *om);

/* Scalars of the limited-precision score systems (impl_sse.h:79-96), for L as configured by
 * p7_oprofile_ReconfigLength(om, L) (p7_oprofile.c:1261). */
typedef struct {
  uint8_t tbm_b, tec_b, tjb_b, base_b, bias_b;
  float   scale_b;
  int16_t xw[4][2];
  float   scale_w;
  int16_t base_w, ddbound_w;
  float   xf[4][2];
} bath_oprofile_scalars;
int  bath_hip_oprofile_scalars(const bath_hip_oprofile *om, int L, bath_oprofile_scalars *out);
/* p7_oprofile_GetSSVEmissionScoreArray (p7_oprofile.c:1507): arr[(M+1)*Kp], arr[k*Kp+x] */
int  bath_hip_oprofile_get_ssv_scores(const bath_hip_oprofile *om, uint8_t *arr);
/* unstriped views for parity tests: rw[Kp*(M+1)], tw[(M+1)*8] (MM IM DM BM MD DD MI II), rf, tf likewise */
int  bath_hip_oprofile_get_vit(const bath_hip_oprofile *om, int16_t *rw, int16_t *tw);
int  bath_hip_oprofile_get_fwd(const bath_hip_oprofile *om, float *rf, float *tf);

/* ------------------------------------------------------------------------------------------
 * Sequence blocks.  dsq holds n sequences back to back, sequence i = dsq[offsets[i] .. offsets[i+1]),
 * residues only (no sentinels), easel digital codes.
 * ------------------------------------------------------------------------------------------ */
int     bath_hip_seqs_create(bath_hip_ctx *ctx, const uint8_t *dsq, const int64_t *offsets, int64_t n, bath_hip_seqs **ret);
void    bath_hip_seqs_destroy(bath_hip_seqs *sq);
int64_t bath_hip_seqs_count(const bath_hip_seqs *sq);
/* Windows of a long target (esl_sqio_ReadWindow with context, bathsearch.c:1099): context[i] = ESL_SQ.C of window i, the
 * leading nucleotides that also ended the previous window.  ORFs inside the context are skipped (p7_pipeline.c:1635-1637)
 * and only the new residues are counted in stats->nres (bathsearch.c:1258).  NULL clears it. */
int  bath_hip_seqs_set_context(bath_hip_seqs *sq, const int32_t *context);

/* Streamed blocks: what a host that feeds the GPU block after block uses instead of bath_hip_seqs_create.  The block arrives
 * in 2 bits per nucleotide (A, C, G, T = 0..3, nucleotide j of a sequence in bits 2(j%4).. of its byte j/4; every sequence starts
 * on a byte boundary, sequences back to back), 4x less to move over PCIe than the byte-per-nucleotide form; positions holding
 * any other code (degenerate nucleotides) come as an exception list.  bath_hip_seqs_create_packed lays the block out from its
 * offsets (in nucleotides) once; bath_hip_seqs_upload_packed queues the transfer of new content on the context's copy stream
 * and returns at once when <packed> is page-locked memory (bath_hip_host_alloc), so that the upload of the next block overlaps
 * the cascade of the current one; bath_hip_seqs_upload_wait orders the context's streams after the transfer and expands the
 * block on the device to the layout every kernel reads.  The block keeps its shape (offsets, lengths, contexts) across uploads. */
void *bath_hip_host_alloc(size_t bytes);
void  bath_hip_host_free(void *p);
int   bath_hip_seqs_create_packed(bath_hip_ctx *ctx, const int64_t *offsets, int64_t n, bath_hip_seqs **ret);
int   bath_hip_seqs_upload_packed(bath_hip_seqs *sq, const uint8_t *packed, const int64_t *exc_seq, const int32_t *exc_pos,
                                  const uint8_t *exc_code, int64_t n_exc);
int   bath_hip_seqs_upload_wait(bath_hip_seqs *sq);

/* ------------------------------------------------------------------------------------------
 * Filter kernels, batched.  Each target i is scored exactly as the reference would after
 * p7_oprofile_ReconfigLength(om, L_i) (p7_pipeline.c:1644).  sc[n] nats, status[n] easel codes.
 * ------------------------------------------------------------------------------------------ */
int bath_hip_ssvfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status);      /* p7_SSVFilter, ssvfilter.c:876 */
int bath_hip_msvfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status);      /* p7_MSVFilter, msvfilter.c:74  */
int bath_hip_vitfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status);      /* p7_ViterbiFilter, vitfilter.c:83 */
int bath_hip_forward_parser(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status); /* p7_ForwardParser, fwdback.c:132 */
/* p7_bg_NullOne + p7_bg_FilterScore (p7_bg.c:356,491) with p7_bg_SetFilter(bg, M, om->compo): nullsc[n], filtersc[n] */
int bath_hip_bias_filter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *nullsc, float *filtersc);
/* p7_ForwardParser followed by p7_BackwardParser (fwdback.c:132,236) with the special-state rows p7_domaindef reads:
 * for target i, (L_i+1) rows of {E,N,J,B,C,SCALE} (P7_OMX xmx, impl_sse.h:253-262) at xmx_offsets[i] floats in fwd_xmx and
 * bck_xmx (either may be NULL).  Backward is scaled by Forward's per-row factors (fwdback.c:660-675).  Scores in nats;
 * status eslOK / eslERANGE per target (may be NULL). */
int bath_hip_fwdback_parser(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const int64_t *xmx_offsets,
                            float *fwd_sc, float *bck_sc, int32_t *fwd_status, int32_t *bck_status, float *fwd_xmx, float *bck_xmx);

/* p7_ViterbiFilter_BATH (vitfilter.c:286) and p7_SSVFilter_BATH (msvfilter.c:250) over a block: the filter plus the hit windows
 * (P7_HMM_WINDOW, hmmer.h:998: position n in the target, last model node k, length; score only from the SSV variant) that
 * p7_pli_BuildDNAWindows and the local-composition re-filter read.  filtersc[n]: the bias-filter score of every target;
 * P: the P-value threshold (pli->F2 / pli->F1).  *wins is owned by ctx (valid until the next call), ordered by target, then
 * by position.  As in the pipeline each target is scored with the profile configured for its own length. */
typedef struct {
  int64_t target;                  /* index of the sequence in the block */
  int32_t n, k, length;
  float   score;
} bath_hmm_window;
int bath_hip_vitfilter_bath(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const float *filtersc, double P,
                            float *sc, int32_t *status, const bath_hmm_window **wins, int64_t *nwins);
int bath_hip_ssvfilter_bath(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, double P,
                            const bath_hmm_window **wins, int64_t *nwins);

/* ------------------------------------------------------------------------------------------
 * The filter cascade of p7_Pipeline_BATH (p7_pipeline.c:1632-1791) over a block of DNA windows:
 * six-frame translation (esl_gencode_Process*, bathsearch.c:384-392), MSV, bias, Viterbi, Forward.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  double  F1, F2, F3, F4;         /* p7_pipeline.c:219-222 */
  int32_t do_biasfilter, fs_pipe, min_orf_len, ncbi_table;
  /* pli->nres on entry: the residues this search counted before the block's first window, both strands (bathsearch.c:1071,
   * :1084, :1258, :1268 add a window's W per strand before the window is searched).  The domain stage drops / flags a hit by
   * P * nres_running / max_length > E with the count AT THE HIT'S WINDOW AND STRAND (p7_domaindef.c:1033, p7_pipeline.c:1079,
   * :1246), so the hits of a block do not depend on how a search is cut into blocks as long as every block is told where it
   * starts; 0 for a search's first (or only) block. */
  int64_t nres_before;
  /* The rest of the option state p7_pipeline_Create_BATH keeps and the hot path reads (p7_pipeline.c:94-234; bathsearch.c:718-719,
   * :831-833).  bath_pipeline_params_default sets bathsearch's defaults; INTEGRATION.md section 4 maps each field to its pli-> member. */
  int32_t do_null2;               /* pli->do_null2: 1; --nonull2 clears it (:199, :213).  0: a hit's bias correction is 0 (:1063-1066, :1230-1233) */
  int32_t std_pipe;               /* pli->std_pipe: 1; --fsonly clears it (:107).  0: P_tot = 1 in the branch decision and a window that does not
                                   * take the frameshift branch is dropped (:1457, :1480)                                                          */
  int32_t strands;                /* pli->strands: BATH_STRAND_BOTH (0), _TOPONLY (--strand plus), _BOTTOMONLY (--strand minus); only the strands
                                   * searched are translated and counted in nres (bathsearch.c:1069, :1082, :1256, :1267)                          */
  int32_t initiator;              /* which codons may start an ORF (bathsearch.c:718-719): BATH_INIT_ANY (0, the default: esl_gencode_SetInitiatorAny),
                                   * BATH_INIT_TABLE (-M: the codon table's own start codons), BATH_INIT_AUG (-m: esl_gencode_SetInitiatorOnlyAUG).
                                   * With -m / -M the initiation codon is translated as M (esl_gencode_WorkstateCreate: using_initiators)           */
  int32_t inc_by_E;               /* pli->inc_by_E: 1; --incT clears it (:165-175).  The domain stage's early tests go by THIS flag, not by_E:
                                   * 1: drop / flag by P * Z > E (p7_domaindef.c:1034; :1080, :1247); 0: no early drop, flag by bit score >= T       */
  int32_t seed;                   /* --seed (:98): 42.  Every region's stochastic-trace ensemble starts from it (do_reseeding, p7_domaindef.c:781,
                                   * :904); 0: one arbitrary seed per process, the generator runs on from region to region (:140-143)                */
  double  T;                      /* pli->T: -T, 0.0 when not given (:148, :155); the early test's bit-score threshold when inc_by_E is 0           */
} bath_pipeline_params;
#define BATH_STRAND_BOTH       0
#define BATH_STRAND_TOPONLY    1
#define BATH_STRAND_BOTTOMONLY 2
#define BATH_INIT_ANY   0
#define BATH_INIT_TABLE 1
#define BATH_INIT_AUG   2

typedef struct {                   /* one per ORF that passed the MSV filter (P <= F1) */
  int64_t window;                  /* index of the DNA window in the block                               */
  int32_t strand, frame;           /* 0 top / 1 bottom; 0..2                                             */
  int32_t start, end;              /* 1-based nt coords on that strand (start<end), as esl_gencode gives */
  int32_t n;                       /* ORF length in aa                                                   */
  int32_t stage;                   /* 1 failed bias, 2 failed Vit, 3 failed Fwd, 4 passed (F3, or F4 if fs_pipe) */
  int32_t msv_status, vit_status;
  float   usc, nullsc, filtersc, vfsc, fwdsc;
  double  P;                       /* P-value at the stage where the ORF stopped                         */
} bath_orf_result;

typedef struct {                   /* pipeline counters, hmmer.h:1115-1128 */
  int64_t nres, n_orfs;
  int64_t n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
  int64_t pos_past_msv, pos_past_bias, pos_past_vit, pos_past_fwd;
  int64_t cells_msv, cells_vit, cells_fwd;     /* sum of L*M per stage (Mc/s numerator, msvfilter.c:563) */
} bath_pipeline_stats;

/* Six-frame translation of a block of DNA windows: esl_gencode_ProcessStart/Piece/End as driven by
 * bathsearch.c:384-392 (both strands, no initiator requirement; windows < 15 nt skipped, bathsearch.c:1066).
 * Every maximal run of non-stop codons of >= min_orf_len residues is one ORF. */
typedef struct {
  int64_t window;                  /* index in the block                                                   */
  int32_t strand, frame;           /* 0 = as given, 1 = reverse complement; frame 0..2                      */
  int32_t start, end;              /* 1-based nt coordinates on that strand (orfsq->start/end convention)   */
  int32_t n;                       /* residues                                                              */
  int64_t aa_off;                  /* residues of this ORF are aa[aa_off .. aa_off+n)                       */
} bath_orf;
/* On return *orfs (sorted by window, strand, frame, start) and *aa are owned by ctx, valid until the next call. */
int  bath_hip_translate_orfs(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int ncbi_table, int min_orf_len,
                             const bath_orf **orfs, int64_t *n_orfs, const uint8_t **aa);
/* ... for one strand only and / or with initiation codons (bath_pipeline_params.strands, .initiator) */
int  bath_hip_translate_orfs_opts(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int ncbi_table, int min_orf_len, int strands, int initiator,
                                  const bath_orf **orfs, int64_t *n_orfs, const uint8_t **aa);
/* gcode->is_initiator[16 a + 4 b + c] (easel codes A C G T = 0..3) under <initiator>; the table's own start codons are NCBI's
 * (gc.prt "sncbieaa" line); easel's copy is not in the reference tree, so BATH_INIT_TABLE is parity-unpinned */
int  bath_gencode_initiators(int ncbi_table, int initiator, uint8_t is_init[64]);

void bath_pipeline_params_default(bath_pipeline_params *p, int fs_pipe);
/* Runs the cascade on every window of <dna>, both strands.  On return *results points to an array of
 * *n_results records owned by ctx (valid until the next call); pass NULL to skip the copy-out. */
int  bath_hip_pipeline_filters(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                               const bath_pipeline_params *params, bath_pipeline_stats *stats,
                               const bath_orf_result **results, int64_t *n_results);
/* Device time of the stages of the last bath_hip_pipeline_filters call, ms (HIP events on ctx's stream).
 * names[i] are static strings. */
int  bath_hip_pipeline_timings(const bath_hip_ctx *ctx, int max, const char **names, float *ms, int64_t *launches);

/* Device time of every kernel of the stages AFTER the cascade (3-codon parsers, region heuristics, 5-codon envelope kernels,
 * tracebacks) launched by the last bath_hip_pipeline_frameshift_domains call, aggregated by kernel name: HIP events on the stream
 * each kernel ran on.  cells = DP cells (rows x nodes) the launches covered, bytes = their algorithmic HBM traffic (the matrix
 * bytes the kernel must read and write).  Returns the number of entries written (<= max). */
typedef struct {
  const char *name;                /* static string */
  float   ms;
  int64_t launches;
  double  cells, bytes;
} bath_kernel_time;
int  bath_hip_kernel_times(bath_hip_ctx *ctx, int max, bath_kernel_time *out);

/* The frameshift pipeline (bathsearch --fs) up to the decision which branch a DNA window takes:
 * the cascade with F4 at the Forward stage (p7_pipeline.c:1774-1789), then p7_pli_BuildDNAWindows (:462-572) and the
 * per-window part of p7_pli_Frameshift (:1368-1464): summed ORF score, window null / bias scores
 * (p7_bg_fs_NullOne, p7_bg_fs_FilterScore), p7_ForwardParser_Frameshift_3Codons, and the P-value comparison.
 * Domain definition after the decision is not part of this library yet. */
typedef struct {
  int64_t window;                  /* sequence index in the block                                            */
  int32_t strand;                  /* 0 = as given, 1 = reverse complement                                    */
  int32_t n, length;               /* window start (1-based on that strand) and length, nt (P7_HMM_WINDOW n, length) */
  int32_t orf_cnt, k_min, k_max;   /* ORFs with P <= F4 inside the window; model range of their hit windows   */
  float   tot_orfsc;               /* log-sum of the ORFs' (Forward - null) scores, nats (:1408)             */
  float   nullsc, filtersc, fwdsc; /* of the DNA window: null, bias-filter and frameshift Forward scores      */
  double  P_tot, P_min, P_fs, P_null;
  int32_t branch;                  /* 1: frameshift branch (:1464); 2: standard branch (:1479); 0: neither (--fsonly, :1480) */
} bath_fs_window;
/* <om_fs3> is the 3-codon frameshift profile.  stats->pos_past_fwd counts as the reference does in this mode
 * (:1468, :1490).  *fs_windows is owned by ctx, valid until the next call; results/n_results as in
 * bath_hip_pipeline_filters (may be NULL). */
int  bath_hip_pipeline_frameshift(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3,
                                  const bath_hip_seqs *dna, const bath_pipeline_params *params, bath_pipeline_stats *stats,
                                  const bath_orf_result **results, int64_t *n_results,
                                  const bath_fs_window **fs_windows, int64_t *n_fs_windows);

/* ... and what the frameshift branch does next (p7_pipeline.c:1469-1476): p7_BackwardParser_Frameshift_3Codons,
 * p7_domaindef_ByPosteriorHeuristics_Frameshift_BATH (p7_domaindef.c:301) with rescore_isolated_domain_frameshift (:993)
 * for single-domain regions, and the scores p7_pli_postDomainDef_Frameshift_BATH (p7_pipeline.c:1005) gives the hit.
 * Multi-domain regions are resolved by stochastic-trace clustering (p7_domaindef.c:396-455); *n_clustered_regions counts them
 * (ddef->nclustered; the name predates that stage). */
typedef struct {
  int64_t window;                  /* sequence index in the block                                              */
  int32_t strand;
  int32_t fs_window;               /* index into the fs_windows array                                          */
  int32_t ienv, jenv, iali, jali;  /* nt coordinates on the sequence, as P7_DOMAIN ienv/jenv/iali/jali (:1035-1049) */
  int32_t ihmm, jhmm;              /* first / last model node of the optimal-accuracy alignment                */
  float   envsc, oasc;             /* envelope Forward score (nats), expected number of correctly aligned residues */
  float   domcorrection, dombias;  /* null2 correction and the bias term derived from it, nats (:1064)          */
  float   bitscore, pre_score;     /* the hit's score in bits (hit->score) and before the null2 correction       */
  double  lnP;
  int32_t reported;                /* exp(lnP) * Z <= E_report, Z = nres of the block / max_length (:1080)      */
  int32_t n_shifted_codons;        /* match states of the alignment that emit a quasi-codon (length != 3)        */
  /* what --tblout prints of the alignment display (p7_alidisplay.c:538-935, 937-1245) */
  int32_t n_stops;                 /* aligned codons that are stop codons (ad->stops)                            */
  float   pid;                     /* percent of alignment columns whose residue is the consensus residue (ad->pid); 0 without a consensus */
  int32_t ali_columns;             /* ad->N: trace states from the first to the last match state                  */
  int64_t cigar_off;               /* the --cigar string starts at bath_hip_domain_cigars(ctx) + cigar_off        */
} bath_fs_domain;
const char *bath_hip_domain_cigars(const bath_hip_ctx *ctx);   /* NUL-terminated strings, valid until the next pipeline call */

/* P7_DOMAIN.tr: the optimal-accuracy trace of every domain of the last bath_hip_pipeline_hits / _frameshift_domains call, with its
 * posterior probabilities, as rescore_isolated_domain_frameshift / _bath keep it (p7_domaindef.c:1171, :1330) and as
 * p7_alidisplay_fs_Create / _nonfs_Create (p7_alidisplay.c:538, :937) and p7_tophits_TabularFrameshifts read it -- what the
 * alignment blocks of the default output and --fstblout are made from.  tr[d] belongs to domains[d] of that call.
 * Per trace state z = tr[d].off .. tr[d].off + tr[d].N - 1, in the reference's conventions (p7_trace_fs_AppendWithPP, p7_trace.c:2303;
 * p7_trace_fs_Convert, :405):
 *   st[z]  BATH_T_M / BATH_T_D / BATH_T_I (p7T_M, p7T_D, p7T_I; hmmer.h:487)
 *   k[z]   model node
 *   i[z]   M, I: the codon's LAST nucleotide, 1-based in windowsq -- the DNA window the domain was defined on: nucleotide
 *          win_start + i - 1 of the strand read (for a hit of the plain pipeline windowsq is the ORF's own stretch of DNA,
 *          p7_pipeline.c:1755); D: what the reference leaves there (the envelope's start - 1 in the frameshift branch, 0 in the standard one)
 *   c[z]   M: the (quasi-)codon's length 1..5 (always 3 in the standard branch); 0 otherwise
 *   pp[z]  M, I: the state's posterior probability; D: 0
 * The states are those from the first to the last match state (ad->N of them): the N / B / E / C flanks of the reference's trace carry
 * nothing any caller on this path reads (both alidisplay constructors and TabularFrameshifts start at the first M) and are not
 * materialised.  The arrays are owned by ctx and valid until its next pipeline call. */
#define BATH_T_M 1
#define BATH_T_D 2
#define BATH_T_I 3
typedef struct {
  int64_t off;                     /* first state of this trace in the arrays                                           */
  int32_t N;                       /* states (alignment columns, ad->N)                                                  */
  int32_t win_start;               /* windowsq->start on the strand read, 1-based                                        */
  int32_t orf_start;               /* standard branch: orfsq->start on the strand read (its first codon reads M under -m / -M); 0 in the frameshift branch */
  int32_t frameshift;              /* 1: a domain of the frameshift branch (p7_alidisplay_fs_Create), 0: of the standard branch (_nonfs_Create) */
} bath_domain_trace;
int bath_hip_domain_traces(bath_hip_ctx *ctx, const bath_domain_trace **tr, int64_t *n_traces,
                           const int8_t **st, const int32_t **k, const int32_t **i, const int8_t **c, const float **pp);

/* The alignment block of a hit as the default output prints it under "Alignment:" (p7_alidisplay_fs_Create / _nonfs_Create,
 * p7_alidisplay.c:538-1243, and p7_alidisplay_Print_BATH, :3757-4110, as p7_tophits_Domains calls it: p7_tophits.c:1394), from one
 * trace of bath_hip_domain_traces: the optional CS / RF lines, the model's consensus, the match line, the translation, the codons
 * (a quasi-codon's missing or extra nucleotides marked as the reference marks them), the optional frame line and the posterior
 * probability line, in blocks of (textw - names - coordinates) / 5 columns.  Host code; nothing here touches the GPU.
 *   st .. pp      the arrays of bath_hip_domain_traces, already offset to this trace (st + tr->off, ...)
 *   window_dsq    windowsq: window_dsq[i - 1] is nucleotide i of the strand read (digital), <window_len> of them -- for a domain d
 *                 of a block: the strand's nucleotides from tr->win_start on
 *   gm_fs5 / gm   the 5-codon frameshift profile (frameshift branch) / the standard profile (standard branch); the other may be NULL
 *   basic         the codon table (bath_gencode_basic), standard branch
 * Returns the text's size in bytes (without a terminating NUL) and copies at most <cap> of them to <buf>; < 0 on error. */
typedef struct {
  const char *hmm_name, *seq_name;   /* ad->hmmname, ad->sqname (the caller substitutes accessions under --acc)            */
  const char *consensus;             /* bath_hmm.consensus                                                                  */
  const char *rf, *cs;               /* bath_hmm.rf / .cs or NULL                                                           */
  int32_t M;
  int64_t sqfrom, sqto;              /* ad->sqfrom / ad->sqto: the hit's iali / jali on the sequence                       */
  int32_t textw;                     /* --textw (150); <= 0: one block of unlimited width (--notextw)                       */
  int32_t show_frameline;            /* --frameline                                                                         */
  int32_t initiator;                 /* bath_pipeline_params.initiator: under -m / -M an ORF's first codon reads M          */
} bath_alidisplay_opts;
int64_t bath_alidisplay_print(const bath_domain_trace *tr, const int8_t *st, const int32_t *k, const int32_t *i, const int8_t *c, const float *pp,
                              const uint8_t *window_dsq, int32_t window_len, const bath_fs_profile *gm_fs5, const bath_profile *gm,
                              const uint8_t basic[64], const bath_alidisplay_opts *opts, char *buf, int64_t cap);
int  bath_hip_pipeline_frameshift_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3,
                                          const bath_hip_fsprofile *om_fs5, const bath_hip_seqs *dna, const bath_pipeline_params *params,
                                          double E_report, bath_pipeline_stats *stats,
                                          const bath_fs_window **fs_windows, int64_t *n_fs_windows,
                                          const bath_fs_domain **domains, int64_t *n_domains, int64_t *n_clustered_regions);

/* The standard pipeline (bathsearch without --fs) after the Forward filter (p7_pipeline.c:1741-1771): p7_BackwardParser,
 * p7_domaindef_ByPosteriorHeuristics_BATH (p7_domaindef.c:491) with rescore_isolated_domain_bath (:1194) for single-domain
 * regions, p7_pli_postDomainDef_BATH (p7_pipeline.c:1172).  One bath_fs_domain per hit (fs_window = -1,
 * n_shifted_codons = 0).  Multi-domain regions are resolved by stochastic-trace clustering (p7_domaindef.c:539-583);
 * *n_clustered_regions counts them (ddef->nclustered). */
int  bath_hip_pipeline_hits(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                            const bath_pipeline_params *params, double E_report, bath_pipeline_stats *stats,
                            const bath_fs_domain **domains, int64_t *n_domains, int64_t *n_clustered_regions);

/* Full-matrix forms over a block of amino-acid targets (the batched twins of impl_sse's single-target functions):
 *   bath_hip_forward_full  <- p7_Forward (fwdback.c:94): Forward matrices (L_i+1) x (M+1) x {M,D,I} (odds ratios, scaled per row
 *       like the parser) back to back in <dp>, special-state rows (L_i+1) x {E,N,J,B,C,SCALE} in <xmx> (either may be NULL).
 *       cfg_len[n] (or NULL: each target's own length) is the length the profile is configured for
 *       (p7_oprofile_ReconfigLength / ReconfigMultihit, p7_domaindef.c:560 uses the ORF's saved length for a region);
 *       unihit != 0: p7_oprofile_ReconfigUnihit.  This is what p7_StochasticTrace walks.
 *   bath_hip_std_envelopes <- p7_Forward + p7_Backward + p7_Decoding + p7_OptimalAccuracy + p7_Null2_ByExpectation on envelopes
 *       (rescore_isolated_domain_bath, p7_domaindef.c:1194-1262; unihit, L = L_i): scores, null2[Kp], posterior matrices <pp>
 *       and optimal-accuracy matrices <oa> in the Forward matrix layout, posterior / OA special-state rows (L_i+1) x {E,N,J,B,C}
 *       in <ppx> / <oax> (any of the four may be NULL). */
typedef struct {
  float   fwdsc, bcksc, oasc;
  int32_t fwd_status, bck_status, ok;       /* ok == 0: numeric range error in decoding (eslERANGE), the domain is dropped */
  float   null2[BATH_KP_AMINO];
} bath_std_result;
int bath_hip_forward_full(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const int32_t *cfg_len, int unihit,
                          float *sc, int32_t *status, float *dp, float *xmx);
int bath_hip_std_envelopes(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, bath_std_result *res,
                           float *pp, float *oa, float *ppx, float *oax);

/* ------------------------------------------------------------------------------------------
 * Hit list of a search and its tabular output (P7_TOPHITS; host code).
 * What bathsearch does after its workers finish (bathsearch.c:868-921): p7_tophits_ComputeEvalues_BATH
 * (p7_tophits.c:789), SortBySeqidxAndAlipos (:379), RemoveDuplicates (:816), SortBySortkey (:345), Threshold (:914),
 * and p7_tophits_TabularTargets (:1603) for --tblout.
 * ------------------------------------------------------------------------------------------ */
typedef struct bath_tophits bath_tophits;
bath_tophits *bath_tophits_create(void);
void bath_tophits_destroy(bath_tophits *th);
/* One hit per domain with reported != 0 (the pipeline's own E-value test, p7_pipeline.c:1080,1246).  <cigars> is
 * bath_hip_domain_cigars(ctx) or NULL; the block's sequence w has global index seqidx0 + w, name seq_names[w], length
 * seq_lens[w]; seq_accs / seq_descs may be NULL. */
int  bath_tophits_add(bath_tophits *th, const bath_fs_domain *dom, int64_t n, const char *cigars, int64_t seqidx0,
                      const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs, const int64_t *seq_lens);
/* <nres>: residues searched, both strands (sum of the pipelines' nres); E: reporting threshold (-E, default 10). */
int  bath_tophits_finalize(bath_tophits *th, int64_t nres, int max_length, double E);
/* Reporting / inclusion by bit score (p7_pli_TargetReportable / p7_pli_TargetIncludable, p7_pipeline.c:583-603), before finalize:
 * by_E = 0 (-T): a hit is reported when score >= T instead of exp(lnP) <= E; inc_by_E = 0 (--incT): included when score >= incT
 * instead of exp(lnP) <= incE.  Defaults: both by E. */
void bath_tophits_set_score_thresholds(bath_tophits *th, int by_E, double T, int inc_by_E, double incT);
/* The residue count bathsearch hands p7_tophits_ComputeEvalues_BATH (bathsearch.c:868-881): with -Z <x> it is 1e6 * x, doubled when
 * both strands are searched, whatever was searched; without, the sum of the workers' pli->nres.  Pass the result to
 * bath_tophits_finalize as <nres>. */
int64_t bath_search_space_residues(int Z_is_set, double Z_megabases, int strands, int64_t nres_searched);
int64_t bath_tophits_count(const bath_tophits *th);       /* hits held, reported or not */
int64_t bath_tophits_reported(const bath_tophits *th);
#define BATH_HIT_REPORTED  1
#define BATH_HIT_DUPLICATE 4
/* The rank-th hit in the current sort order (by E-value after finalize); dom->lnP is the E-value's log after finalize. */
int  bath_tophits_get(const bath_tophits *th, int64_t rank, bath_fs_domain *dom, int64_t *seqidx, int32_t *flags);
/* Returns the table's size in bytes and copies at most <cap> of them to <buf>. */
/* p7_tophits_Targets (:1073): the "Scores for complete hits" block of the main output; textw = --textw (120; <= 0 unlimited). */
int64_t bath_tophits_targets(const bath_tophits *th, int fs_pipe, int textw, char *buf, int64_t cap);
/* Head of the rank-th hit's entry under "Annotation for each hit" (p7_tophits_Domains, :1256-1378): ">> name", the two header
 * lines and the hit's line; 0 for a hit that is not reported.  The alignment block itself is not produced. */
int64_t bath_tophits_domain_annotation(const bath_tophits *th, int64_t rank, int M, int fs_pipe, char *buf, int64_t cap);
/* p7_pli_Statistics (p7_pipeline.c:1836): the "Internal pipeline statistics summary" block without its two timing lines. */
int64_t bath_tophits_pipeline_statistics(const bath_tophits *th, const bath_pipeline_stats *stats, const bath_pipeline_params *params,
                                         int64_t nmodels, int64_t nnodes, int64_t nseqs, char *buf, int64_t cap);
void bath_tophits_set_inclusion(bath_tophits *th, double incE);   /* --incE, default 0.01; before finalize */
#define BATH_HIT_INCLUDED  2
int64_t bath_tophits_tabular_targets(const bath_tophits *th, const char *qname, const char *qacc, int M, int fs_pipe,
                                     int show_cigar, int show_header, char *buf, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * One search on several GPUs from a C host: the hit list as a byte stream and the division of the work.
 * Host code; the host's own transport (MPI, RCCL, sockets) moves the bytes.
 *
 * Hits between processes.  The reference ships a hit as p7_hit_Serialize writes it (src/p7_hit.c:174-411; read back by
 * p7_hit_Deserialize, :411-640): a self-delimiting record -- its own size first, every fixed-width field in NETWORK byte order, a
 * byte of presence flags, optional strings NUL-terminated -- followed by its domain and alignment.  The same scheme here:
 *   stream  := u32 magic "BHIT" | u32 version (1) | u64 n_hits | n_hits x hit
 *   hit     := u32 size of this record | i64 window | i32 strand, fs_window, ienv, jenv, iali, jali, ihmm, jhmm | f32 envsc, oasc,
 *              domcorrection, dombias, bitscore, pre_score | f64 lnP | i32 reported, n_shifted_codons, n_stops | f32 pid |
 *              i32 ali_columns | u8 flags (1: CIGAR present, 2: trace present) | [cigar, NUL-terminated] |
 *              [i32 N, win_start, orf_start, frameshift | N x (u8 st, u8 c, i32 k, i32 i, f32 pp)]
 * all integers and IEEE floats big-endian (esl_hton32 / esl_hton64 as p7_hit.c applies them).
 * bath_hits_serialize returns the stream's size in bytes and writes it to <buf> when <buf> is not NULL and <cap> suffices (-1: bad
 * arguments or <cap> too small); <cigars> is bath_hip_domain_cigars(ctx) or NULL; <tr> .. <pp> are bath_hip_domain_traces' outputs
 * or all NULL (hits without their traces: enough for --tblout, not for the alignment blocks).
 * ------------------------------------------------------------------------------------------ */
typedef struct bath_hits bath_hits;                 /* a deserialized hit list; owns its arrays */
int64_t bath_hits_serialize(const bath_fs_domain *dom, int64_t n, const char *cigars, const bath_domain_trace *tr,
                            const int8_t *st, const int32_t *k, const int32_t *i, const int8_t *c, const float *pp,
                            uint8_t *buf, int64_t cap);
int     bath_hits_deserialize(const uint8_t *buf, int64_t nbytes, bath_hits **ret);     /* BATH_EFORMAT for a stream that is not one */
int64_t bath_hits_stream_size(const uint8_t *buf, int64_t nbytes);   /* bytes of the stream starting at buf (streams may lie back to back in a message); -1: not a stream */
void    bath_hits_destroy(bath_hits *h);
int64_t bath_hits_count(const bath_hits *h);
bath_fs_domain *bath_hits_domains(bath_hits *h);    /* cigar_off points into bath_hits_cigars(); -1: the hit came without one */
const char *bath_hits_cigars(const bath_hits *h, int64_t *nbytes);
int     bath_hits_traces(const bath_hits *h, const bath_domain_trace **tr, const int8_t **st, const int32_t **k, const int32_t **i,
                         const int8_t **c, const float **pp);                           /* BATH_EINVAL when the stream carried none */
/* p7_tophits_Merge from a byte stream (bathsearch.c:884-888): the hits of another rank join <th>.  <window_shift> is added to every
 * hit's window index (a rank that searched windows [lo, hi) of the search numbers them from 0); <n_seqs> is the length of the
 * seq_* arrays: a hit whose shifted window is not in [0, n_seqs) -- a damaged or mismatched stream, a wrong shift -- is refused with
 * BATH_EFORMAT and nothing is added; the rest as bath_tophits_add. */
int     bath_tophits_add_serialized(bath_tophits *th, const uint8_t *buf, int64_t nbytes, int64_t window_shift, int64_t n_seqs, int64_t seqidx0,
                                    const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs, const int64_t *seq_lens);

/* Division of the work, the same on every rank without communication.
 * bath_dist_shard_range: rank r's contiguous share [lo, hi) of n units (windows of one search over the ranks: configs[1], [2], [4]).
 * bath_dist_items: a multi-query job (bathsearch's loop over an HMM database, bathsearch.c:737-844; configs[3]) as (query, window
 *   group) items: query q's windows [0, n_q) in g_q consecutive groups, g_q = round(cost_q / sum of costs x T) clipped to [1, n_q],
 *   T = max(queries, items_per_rank x world); costs NULL: g_q = ceil(items_per_rank x world / queries) for every query.  A useful
 *   cost is windows x (M + 150).  Returns the number of items; writes at most <cap>.
 * bath_dist_deal: owner rank of every item, longest processing time first onto the least loaded rank (ties: earlier item, lower rank). */
typedef struct { int32_t query; int64_t lo, hi; } bath_dist_item;
void    bath_dist_shard_range(int64_t n, int rank, int world, int64_t *lo, int64_t *hi);
int64_t bath_dist_items(const int64_t *n_windows_by_query, const double *costs_by_query, int n_queries, int world, int items_per_rank,
                        bath_dist_item *items, int64_t cap);
int     bath_dist_deal(const double *costs, int64_t n_items, int world, int32_t *owner);

/* ------------------------------------------------------------------------------------------
 * Frameshift kernels (P7_FS_OPROFILE surface), batched over DNA windows.
 * ------------------------------------------------------------------------------------------ */
#define BATH_LOGSUM_TABLE 0   /* emulate p7_FLogsum's 0.001-nat truncating table (logsum.c:105) */
#define BATH_LOGSUM_EXACT 1   /* exact log(1+exp(x))                                            */
#define BATH_LOGSUM_TABLE_SERIAL 2 /* the table, sums along the model in the reference's serial order: bit-identical to generic_fwdback_frameshift.c */
#define BATH_LOGSUM_CONTEXT 3 /* whatever bath_hip_set_fs_strict selected for the context: TABLE_SERIAL unless switched to the fast mode */

int  bath_hip_fsprofile_convert(bath_hip_ctx *ctx, const bath_fs_profile *gm_fs, bath_hip_fsprofile **ret); /* p7_fs_oprofile_Convert, p7_fs_oprofile.c:221 */
void bath_hip_fsprofile_destroy(bath_hip_fsprofile *om);

/* p7_ForwardParser_Frameshift_3Codons (fwdback_fs.c:97) / p7_GForwardParser_Frameshift_3Codons
 * (generic_fwdback_frameshift.c:451): per window i, after p7_fs_oprofile_ReconfigLength(om, L_i/3).
 * xmx (optional) receives the special-state rows: for window i, (L_i+1)*5 floats {E,N,J,B,C} at xmx_offsets[i]. */
int bath_hip_fs3_forward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, int logsum_mode,
                                float *sc, float *xmx, const int64_t *xmx_offsets);
int bath_hip_fs3_backward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, int logsum_mode,
                                 float *sc, float *xmx, const int64_t *xmx_offsets);  /* fwdback_fs.c:565 / generic :1422 */

/* Envelope rescoring (p7_domaindef.c:993-1082): p7_Forward_Frameshift + p7_Backward_Frameshift +
 * p7_Decoding_Frameshift + p7_OptimalAccuracy_Frameshift fill + p7_Null2_fs_ByExpectation, one
 * envelope per sequence of <dna>, unihit profile reconfigured to L_i/3.
 * c5_compat: 1 = generic_fwdback_frameshift.c:324 ring aliasing for 5-nt codons, 0 = fwdback_fs.c:1464. */
typedef struct {
  float fwdsc, bcksc, oasc;
  float null2[BATH_KP_AMINO];
} bath_fs5_result;
int bath_hip_fs5_envelopes(bath_hip_ctx *ctx, const bath_hip_fsprofile *om5, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                           bath_fs5_result *res,
                           float *pp /* optional: posterior matrices, (L_i+1)*(M+1)*8 floats at pp_offsets[i] */, const int64_t *pp_offsets,
                           float *oa /* optional: OA matrices (L_i+1)*(M+1)*3 */, const int64_t *oa_offsets);

/* bath_hip_fs5_envelopes with the special-state rows as well: posterior rows <ppx> and optimal-accuracy rows <oax>,
 * (L_i+1) x {E,N,J,B,C} per envelope, back to back (either may be NULL). */
int bath_hip_fs5_envelopes_x(bath_hip_ctx *ctx, const bath_hip_fsprofile *om5, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                             bath_fs5_result *res, float *pp, float *oa, float *ppx, float *oax);
/* p7_Forward_Frameshift in the MULTIHIT configuration of amino length <cfg_len_amino> (p7_domaindef.c:411-414: the model's saved
 * length): sc[n]; Forward matrices (L_i+1) x (M+1) x {D, I, M_C0, M_C1..M_C5} in <fwd>, special-state rows (L_i+1) x {E,N,J,B,C}
 * in <xmx>, back to back (either may be NULL).  This is what p7_StochasticTrace_Frameshift walks. */
int bath_hip_fs5_forward_full(bath_hip_ctx *ctx, const bath_hip_fsprofile *om5, const bath_hip_seqs *dna, int cfg_len_amino,
                              float *sc, float *fwd, float *xmx);

/* ------------------------------------------------------------------------------------------
 * Self-test hooks (host only, no GPU): the pieces of easel the multi-domain branch restates -- esl_randomness_CreateFast /
 * esl_random (p7_pipeline.c:140: the "fast" generator, x <- 69069 x + 1 on a Jenkins-mixed seed) and esl_vec_FNorm +
 * esl_rnd_FChoose as p7_StochasticTrace calls them (stotrace.c:165-300) -- so that tests can hold them against independent
 * implementations of the published algorithms.
 * ------------------------------------------------------------------------------------------ */
int bath_selftest_rng_stream(uint32_t seed, int n, double *out);                          /* the first n values of esl_random() after seeding */
int bath_selftest_fchoose(uint32_t seed, const float *p, int n, int draws, int32_t *out); /* draws x (copy p, esl_vec_FNorm, esl_rnd_FChoose) from one stream */
/* region_trace_ensemble_frameshift (p7_domaindef.c:891-958: 200 stochastic tracebacks through a region's multihit 5-codon Forward
 * matrix, single-linkage clustering) as the pipeline runs it on the host, on caller-supplied matrices: fwd (Lr+1) x (M+1) x
 * {D, I, M, C1..C5}, fx (Lr+1) x {E,N,J,B,C}, tsc the generic [M][8] log transitions; env: n_env x {i, j} in window nucleotides. */
/* p7_spensemble_Cluster + the removal of dominated clusters on caller-supplied segments (idx: the sample a segment came from) */
int bath_selftest_cluster_segments(int n, const int32_t *idx, const int32_t *i, const int32_t *j, const int32_t *k, const int32_t *m,
                                   int nsamples, int fs, int32_t *env, int max_env, int32_t *n_env);
int bath_selftest_fs_ensemble(int M, const float *tsc, float xNL, float xNM, float xE, int ireg, int Lr, const float *fwd, const float *fx,
                              int32_t *env, int max_env, int32_t *n_env);

#ifdef __cplusplus
}
#endif
#endif /* BATH_HIP_H */
