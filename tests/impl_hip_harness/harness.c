/* TEST HARNESS ONLY (see hmmer.h in this directory): small definitions of the generic helpers impl_hip/ calls, and the entry
 * points tests/test_impl_hip_gpu.py drives through ctypes.  Each hs_* function builds the reference-shaped objects from a
 * .bhmm file with the library's own host model code, then makes the calls p7_Pipeline_BATH / p7_domaindef make, in their order. */
#include <stdarg.h>
#include <string.h>
#include "hmmer.h"

/* ------------------------------------------------------------------ helpers impl_hip links against */
void esl_fatal(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); abort(); }
static char hs_last[512];
void hs_exception(int code, const char *file, int line, const char *fmt, ...)
{
  va_list ap; va_start(ap, fmt); vsnprintf(hs_last, sizeof hs_last, fmt, ap); va_end(ap);
  fprintf(stderr, "impl_hip exception %d at %s:%d: %s\n", code, file, line, hs_last);
}
double esl_random(ESL_RANDOMNESS *r) { r->x = r->x * 69069u + 1u; return (double) r->x / 4294967296.0; }
int esl_rnd_FChoose(ESL_RANDOMNESS *r, const float *p, int N)
{
  for (;;) { const float roll = (float) esl_random(r); float acc = 0.f; for (int q = 0; q < N; q++) { acc += p[q]; if (roll < acc) return q; } }
}
void esl_vec_FNorm(float *v, int n)
{
  float sum = 0.f, c = 0.f;
  for (int q = 0; q < n; q++) { const float y = v[q] - c, t = sum + y; c = (t - sum) - y; sum = t; }
  for (int q = 0; q < n; q++) v[q] = (sum != 0.0f) ? v[q] / sum : 1.0f / (float) n;
}
void esl_vec_FLogNorm(float *v, int n)
{
  float mx = v[0], denom;
  for (int q = 1; q < n; q++) if (v[q] > mx) mx = v[q];
  if (mx == INFINITY) denom = INFINITY; else if (mx == -INFINITY) denom = -INFINITY;
  else { float s = 0.f; for (int q = 0; q < n; q++) if (v[q] > mx - 50.f) s += expf(v[q] - mx); denom = logf(s) + mx; }
  for (int q = 0; q < n; q++) v[q] = expf(v[q] - denom);
  esl_vec_FNorm(v, n);
}
int esl_abc_FAvgScVec(const ESL_ALPHABET *abc, float *sc)
{
  static const int mem[5][2] = { {2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1} };     /* B=DN J=IL Z=EQ O=K U=C */
  float s = 0.f;
  for (int dx = 0; dx < 5; dx++) sc[21 + dx] = (mem[dx][0] == mem[dx][1]) ? sc[mem[dx][0]] : (sc[mem[dx][0]] + sc[mem[dx][1]]) / 2.0f;
  for (int y = 0; y < abc->K; y++) s += sc[y];
  sc[26] = s / (float) abc->K;
  return eslOK;
}
int p7_bg_SetLength(P7_BG *bg, int L) { (void) bg; (void) L; return eslOK; }
static int tr_grow(P7_TRACE *tr)
{
  if (tr->N < tr->nalloc) return eslOK;
  const int n = tr->nalloc ? tr->nalloc * 2 : 256;
  tr->st = realloc(tr->st, (size_t) n); tr->k = realloc(tr->k, sizeof(int) * (size_t) n); tr->i = realloc(tr->i, sizeof(int) * (size_t) n);
  tr->c = realloc(tr->c, sizeof(int) * (size_t) n); tr->pp = realloc(tr->pp, sizeof(float) * (size_t) n);
  tr->nalloc = n;
  return (tr->st && tr->k && tr->i && tr->c && tr->pp) ? eslOK : eslEMEM;
}
int p7_trace_fs_AppendWithPP(P7_TRACE *tr, char st, int k, int i, int c, float pp)
{
  if (tr_grow(tr) != eslOK) return eslEMEM;
  const int emits = (st == p7T_M || st == p7T_I || ((st == p7T_N || st == p7T_C || st == p7T_J) && tr->N > 0 && tr->st[tr->N - 1] == st));
  const int node = (st == p7T_M || st == p7T_D || st == p7T_I);
  tr->st[tr->N] = st; tr->k[tr->N] = node ? k : 0; tr->i[tr->N] = emits ? i : 0; tr->c[tr->N] = (st == p7T_M) ? c : 0; tr->pp[tr->N] = emits ? pp : 0.0f;
  tr->N++;
  return eslOK;
}
int p7_trace_fs_Append(P7_TRACE *tr, char st, int k, int i, int c) { return p7_trace_fs_AppendWithPP(tr, st, k, i, c, 0.0f); }
int p7_trace_AppendWithPP(P7_TRACE *tr, char st, int k, int i, float pp) { return p7_trace_fs_AppendWithPP(tr, st, k, i, 0, pp); }
int p7_trace_Append(P7_TRACE *tr, char st, int k, int i) { return p7_trace_fs_AppendWithPP(tr, st, k, i, 0, 0.0f); }
int p7_trace_fs_Reverse(P7_TRACE *tr)
{
  /* a backwards trace assigns an emitted residue to the SECOND of two identical N/C/J states; forwards it belongs to the first:
   * shift those while reversing (p7_trace.c: p7_trace_Reverse) */
  for (int z = 0; z < tr->N; z++)
    if ((tr->st[z] == p7T_N || tr->st[z] == p7T_C || tr->st[z] == p7T_J) && tr->i[z] == 0 && z > 0 && tr->i[z - 1] > 0 && tr->st[z - 1] == tr->st[z]) {
      tr->i[z] = tr->i[z - 1]; tr->i[z - 1] = 0; tr->pp[z] = tr->pp[z - 1]; tr->pp[z - 1] = 0.0f;
    }
  for (int a = 0, b = tr->N - 1; a < b; a++, b--) {
    char s = tr->st[a]; tr->st[a] = tr->st[b]; tr->st[b] = s;
    int t;
    t = tr->k[a]; tr->k[a] = tr->k[b]; tr->k[b] = t;
    t = tr->i[a]; tr->i[a] = tr->i[b]; tr->i[b] = t;
    t = tr->c[a]; tr->c[a] = tr->c[b]; tr->c[b] = t;
    float f = tr->pp[a]; tr->pp[a] = tr->pp[b]; tr->pp[b] = f;
  }
  return eslOK;
}
int p7_trace_Reverse(P7_TRACE *tr) { return p7_trace_fs_Reverse(tr); }
P7_HMM_WINDOW *p7_hmmwindow_new(P7_HMM_WINDOWLIST *l, uint32_t id, uint32_t pos, uint32_t k, uint32_t length, float score, uint8_t comp, uint32_t target_len)
{
  if (l->count == l->size) { l->size = l->size ? 2 * l->size : 16; l->windows = realloc(l->windows, sizeof(P7_HMM_WINDOW) * (size_t) l->size); }
  P7_HMM_WINDOW *w = &l->windows[l->count++];
  w->id = id; w->n = pos; w->k = (int32_t) k; w->length = (int32_t) length; w->score = score; w->complementarity = (int8_t) comp; w->target_len = target_len;
  return w;
}
static float hs_tbl[16000]; static int hs_tbl_ok = 0;
float p7_FLogsum(float a, float b)                 /* the table-driven log-sum, for p7_DomainDecoding_Frameshift's Z */
{
  if (!hs_tbl_ok) { for (int i = 0; i < 16000; i++) hs_tbl[i] = (float) log(1. + exp((double) -i / 1000.)); hs_tbl_ok = 1; }
  const float mx = a > b ? a : b, mn = a > b ? b : a;
  return (mn == -INFINITY || (mx - mn) >= 15.7f) ? mx : mx + hs_tbl[(int)((mx - mn) * 1000.f)];
}

/* ------------------------------------------------------------------ reference-shaped objects from a .bhmm file */
static const ESL_ALPHABET hs_amino = { 3, 20, 29 };

typedef struct { bath_hmm *hmm; bath_profile *bp; P7_PROFILE gm; P7_OPROFILE *om; } HS_MODEL;

static int hs_model_open(const char *path, int idx, HS_MODEL *m)
{
  memset(m, 0, sizeof *m);
  if (bath_hmmfile_read(path, idx, &m->hmm) != BATH_OK) return eslFAIL;
  if (bath_profile_config(m->hmm, 100, &m->bp) != BATH_OK) return eslFAIL;
  const int M = m->bp->M;
  P7_PROFILE *gm = &m->gm;
  gm->tsc = m->bp->tsc;
  gm->rsc = malloc(sizeof(float *) * 29);
  for (int x = 0; x < 29; x++) gm->rsc[x] = m->bp->rsc + (size_t) x * (M + 1) * p7P_NR;
  memcpy(gm->xsc, m->bp->xsc, sizeof gm->xsc);
  gm->mode = p7_LOCAL; gm->L = m->bp->L; gm->allocM = M; gm->M = M; gm->max_length = m->bp->max_length; gm->nj = m->bp->nj;
  gm->name = m->hmm->name; gm->consensus = m->hmm->consensus;
  memcpy(gm->evparam, m->bp->evparam, sizeof(float) * p7_NEVPARAM);
  memcpy(gm->compo, m->bp->compo, sizeof(float) * p7_MAXABET);
  gm->abc = &hs_amino;
  m->om = p7_oprofile_Create(M, &hs_amino);
  return p7_oprofile_Convert(gm, m->om);
}
static void hs_model_close(HS_MODEL *m)
{
  p7_oprofile_Destroy(m->om); free(m->gm.rsc);
  bath_profile_destroy(m->bp); bath_hmm_destroy(m->hmm);
}

/* p7_Pipeline_BATH's filter calls on one ORF (p7_pipeline.c:1643-1779): out = {msv, vit, vit_bath, fwd, bck}, st likewise;
 * fx / bx: the parsers' special-state rows (L+1) x 6; windows of p7_ViterbiFilter_BATH in wn/wk/wl (cap 64), count returned in *nwin */
int hs_filters(const char *path, int idx, const uint8_t *dsq1, int L, float filtersc, double P, float *out, int *st, float *fx, float *bx,
               int *wn, int *wk, int *wl, int *nwin, int *ssv_wn, int *ssv_wk, int *ssv_wl, float *ssv_wsc, int *ssv_nwin)
{
  HS_MODEL m;
  if (hs_model_open(path, idx, &m) != eslOK) return eslFAIL;
  P7_OMX *ox = p7_omx_Create(m.gm.M, 0, L), *oxb = p7_omx_Create(m.gm.M, 0, L);
  P7_HMM_WINDOWLIST wl_ = { NULL, 0, 0 }, wl2 = { NULL, 0, 0 };
  P7_BG bg = { 0 };
  p7_oprofile_ReconfigLength(m.om, L);
  p7_omx_GrowTo(ox, m.gm.M, 0, L); p7_omx_GrowTo(oxb, m.gm.M, 0, L);
  st[0] = p7_MSVFilter(dsq1, L, m.om, ox, &out[0]);
  st[1] = p7_ViterbiFilter(dsq1, L, m.om, ox, &out[1]);
  st[2] = p7_ViterbiFilter_BATH(dsq1, L, m.om, ox, NULL, filtersc, P, &wl_, &out[2]);
  p7_SSVFilter_BATH(dsq1, L, m.om, ox, NULL, &bg, P, &wl2);
  st[3] = p7_ForwardParser(dsq1, L, m.om, ox, &out[3]);
  st[4] = p7_BackwardParser(dsq1, L, m.om, ox, oxb, &out[4]);
  memcpy(fx, ox->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  memcpy(bx, oxb->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  *nwin = wl_.count;
  for (int i = 0; i < wl_.count && i < 64; i++) { wn[i] = (int) wl_.windows[i].n; wk[i] = wl_.windows[i].k; wl[i] = wl_.windows[i].length; }
  *ssv_nwin = wl2.count;
  for (int i = 0; i < wl2.count && i < 64; i++) { ssv_wn[i] = (int) wl2.windows[i].n; ssv_wk[i] = wl2.windows[i].k; ssv_wl[i] = wl2.windows[i].length; ssv_wsc[i] = wl2.windows[i].score; }
  free(wl_.windows); free(wl2.windows);
  p7_omx_Destroy(ox); p7_omx_Destroy(oxb);
  hs_model_close(&m);
  return eslOK;
}

/* p7_domaindef's parser calls and p7_DomainDecoding on one ORF (p7_domaindef.c:513-520): the rows the decoding reads and its three arrays */
int hs_std_decoding(const char *path, int idx, const uint8_t *dsq1, int L, float *fx, float *bx, float *btot, float *etot, float *mocc)
{
  HS_MODEL m;
  if (hs_model_open(path, idx, &m) != eslOK) return eslFAIL;
  P7_OMX *ox = p7_omx_Create(m.gm.M, 0, L), *oxb = p7_omx_Create(m.gm.M, 0, L);
  P7_DOMAINDEF ddef = { mocc, btot, etot, 0, L + 1, NULL };
  float f = 0.f, b = 0.f;
  p7_oprofile_ReconfigLength(m.om, L);
  int st = p7_ForwardParser(dsq1, L, m.om, ox, &f);
  if (st == eslOK) st = p7_BackwardParser(dsq1, L, m.om, ox, oxb, &b);
  if (st == eslOK) st = p7_DomainDecoding(m.om, ox, oxb, &ddef);
  memcpy(fx, ox->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  memcpy(bx, oxb->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  p7_omx_Destroy(ox); p7_omx_Destroy(oxb);
  hs_model_close(&m);
  return st;
}

/* rescore_isolated_domain_bath's calls on one envelope (p7_domaindef.c:1206-1262): out = {envsc, bcksc, oasc}; null2[29];
 * the OA trace in tst/tk/ti/tpp (cap tcap), its length returned */
int hs_std_envelope(const char *path, int idx, const uint8_t *dsq1, int L, float *out, float *null2, char *tst, int *tk, int *ti, float *tpp, int tcap)
{
  HS_MODEL m;
  if (hs_model_open(path, idx, &m) != eslOK) return -1;
  P7_OMX *ox1 = p7_omx_Create(m.gm.M, L, L), *ox2 = p7_omx_Create(m.gm.M, L, L);
  P7_TRACE tr; memset(&tr, 0, sizeof tr);
  p7_oprofile_ReconfigUnihit(m.om, L);
  p7_Forward(dsq1, L, m.om, ox1, &out[0]);
  p7_Backward(dsq1, L, m.om, ox1, ox2, &out[1]);
  int status = p7_Decoding(m.om, ox1, ox2, ox2);
  if (status == eslOK) {
    p7_OptimalAccuracy(m.om, ox2, ox1, &out[2]);
    p7_OATrace(m.om, ox2, ox1, &tr);
    p7_Null2_ByExpectation(m.om, ox2, null2);
  }
  const int n = tr.N;
  for (int z = 0; z < n && z < tcap; z++) { tst[z] = tr.st[z]; tk[z] = tr.k[z]; ti[z] = tr.i[z]; tpp[z] = tr.pp[z]; }
  free(tr.st); free(tr.k); free(tr.i); free(tr.c); free(tr.pp);
  p7_omx_Destroy(ox1); p7_omx_Destroy(ox2);
  hs_model_close(&m);
  return status == eslOK ? n : -2;
}

/* region_trace_ensemble's calls (p7_domaindef.c:557-575): multihit Forward with the ORF's length, then <ntraces> stochastic
 * tracebacks from one generator; returns for every trace its number of domains (B states) in ndom[] */
int hs_std_region(const char *path, int idx, const uint8_t *dsq1, int L, int saveL, uint32_t seed, int ntraces, float *fwdsc, int *ndom, int *first_i, int *last_i)
{
  HS_MODEL m;
  if (hs_model_open(path, idx, &m) != eslOK) return eslFAIL;
  P7_OMX *ox1 = p7_omx_Create(m.gm.M, L, L);
  ESL_RANDOMNESS rng = { seed };
  p7_oprofile_ReconfigMultihit(m.om, saveL);
  int status = p7_Forward(dsq1, L, m.om, ox1, fwdsc);
  for (int t = 0; t < ntraces && status == eslOK; t++) {
    P7_TRACE tr; memset(&tr, 0, sizeof tr);
    status = p7_StochasticTrace(&rng, dsq1, L, m.om, ox1, &tr);
    ndom[t] = 0; first_i[t] = 0; last_i[t] = 0;
    for (int z = 0; z < tr.N; z++) {
      if (tr.st[z] == p7T_B) ndom[t]++;
      if (tr.st[z] == p7T_M) { if (!first_i[t]) first_i[t] = tr.i[z]; last_i[t] = tr.i[z]; }
    }
    free(tr.st); free(tr.k); free(tr.i); free(tr.c); free(tr.pp);
  }
  p7_omx_Destroy(ox1);
  hs_model_close(&m);
  return status;
}

/* ---- frameshift side */
typedef struct { HS_MODEL m; bath_fs_profile *fp; P7_FS_PROFILE gm; P7_FS_OPROFILE *om; uint8_t basic[64]; } HS_FS;
static int hs_fs_open(const char *path, int idx, int codon_lengths, HS_FS *f)
{
  memset(f, 0, sizeof *f);
  if (bath_hmmfile_read(path, idx, &f->m.hmm) != BATH_OK) return eslFAIL;
  if (bath_gencode_basic(f->m.hmm->ct, f->basic) != BATH_OK) return eslFAIL;
  if (bath_fs_profile_config(f->m.hmm, f->basic, codon_lengths, 100, &f->fp) != BATH_OK) return eslFAIL;
  const int M = f->fp->M, nrows = f->fp->maxcodons + 29;
  P7_FS_PROFILE *gm = &f->gm;
  gm->tsc = f->fp->tsc;
  gm->rsc = malloc(sizeof(float *) * (size_t) nrows);
  for (int r = 0; r < nrows; r++) gm->rsc[r] = f->fp->rsc + (size_t) r * (M + 1);
  gm->codons = malloc(sizeof(ESL_DSQ *) * (size_t) f->fp->maxcodons); gm->indel_pos = malloc(sizeof(ESL_DSQ *) * (size_t) f->fp->maxcodons);
  for (int q = 0; q < f->fp->maxcodons; q++) {
    gm->codons[q] = malloc((size_t) M + 1); gm->indel_pos[q] = malloc((size_t) M + 1);
    for (int k = 0; k <= M; k++) { gm->codons[q][k] = f->fp->codons[(size_t) k * f->fp->maxcodons + q]; gm->indel_pos[q][k] = f->fp->indel_pos[(size_t) k * f->fp->maxcodons + q]; }
  }
  memcpy(gm->xsc, f->fp->xsc, sizeof gm->xsc);
  gm->mode = p7_LOCAL; gm->codon_lengths = codon_lengths; gm->L = f->fp->L; gm->allocM = M; gm->M = M; gm->max_length = f->fp->max_length;
  gm->nj = f->fp->nj; gm->fsprob = f->fp->fsprob; gm->name = f->m.hmm->name; gm->consensus = f->m.hmm->consensus;
  memcpy(gm->evparam, f->fp->evparam, sizeof(float) * p7_NEVPARAM);
  memcpy(gm->compo, f->fp->compo, sizeof(float) * p7_MAXABET);
  gm->abc = &hs_amino;
  f->om = p7_fs_oprofile_Create(M, &hs_amino, codon_lengths);
  return p7_fs_oprofile_Convert(gm, f->om);
}
static void hs_fs_close(HS_FS *f)
{
  p7_fs_oprofile_Destroy(f->om);
  for (int q = 0; q < f->fp->maxcodons; q++) { free(f->gm.codons[q]); free(f->gm.indel_pos[q]); }
  free(f->gm.codons); free(f->gm.indel_pos); free(f->gm.rsc);
  bath_fs_profile_destroy(f->fp); bath_hmm_destroy(f->m.hmm);
}

/* p7_pli_Frameshift's parser calls on one DNA window (p7_pipeline.c:1446-1476) and p7_domaindef's domain decoding (:320-326) */
int hs_fs_parsers(const char *path, int idx, const uint8_t *dsq1, int L, float *out, float *fx, float *bx, float *btot, float *etot, float *mocc)
{
  HS_FS f;
  if (hs_fs_open(path, idx, 3, &f) != eslOK) return eslFAIL;
  P7_OMX *oxf = p7_omx_Create(f.gm.M, 0, L), *oxb = p7_omx_Create(f.gm.M, 0, L);
  P7_OIVX *ov = p7_oivx_Create(f.gm.M, 3);
  P7_DOMAINDEF ddef = { mocc, btot, etot, 0, L + 1, NULL };
  p7_fs_oprofile_ReconfigLength(f.om, L / 3);
  const int s1 = p7_ForwardParser_Frameshift_3Codons(dsq1, L, f.om, oxf, ov, &out[0]);
  const int s2 = p7_BackwardParser_Frameshift_3Codons(dsq1, L, f.om, oxf, oxb, ov, &out[1]);
  p7_fs_oprofile_ReconfigLength(f.om, 100);          /* domain decoding runs with the model's saved length (p7_domaindef.c:318) */
  p7_DomainDecoding_Frameshift(f.om, oxf, oxb, &ddef);
  memcpy(fx, oxf->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  memcpy(bx, oxb->xmx, sizeof(float) * (size_t)(L + 1) * 6);
  p7_omx_Destroy(oxf); p7_omx_Destroy(oxb); p7_oivx_Destroy(ov);
  hs_fs_close(&f);
  return s1 != eslOK ? s1 : s2;
}

/* rescore_isolated_domain_frameshift's calls on one envelope (p7_domaindef.c:1019-1083): out = {envsc, bcksc, oasc}; null2[29];
 * the OA trace in tst/tk/ti/tc/tpp */
int hs_fs_envelope(const char *path, int idx, const uint8_t *dsq1, int L, float *out, float *null2, char *tst, int *tk, int *ti, int *tc, float *tpp, int tcap)
{
  HS_FS f;
  if (hs_fs_open(path, idx, 5, &f) != eslOK) return -1;
  P7_OMX *ox1 = p7_omx_Create_dpf(f.gm.M, L, L, p7X_NSCELLS_FS), *ox2 = p7_omx_Create_dpf(f.gm.M, L, L, p7X_NSCELLS);
  P7_OIVX *ov = p7_oivx_Create(f.gm.M, 5);
  P7_TRACE tr; memset(&tr, 0, sizeof tr);
  p7_fs_oprofile_ReconfigUnihit(f.om, L / 3);
  p7_omx_GrowTo_dpf(ox1, f.gm.M, L, L);
  int status = p7_Forward_Frameshift(dsq1, L, f.om, ox1, ov, &out[0]);
  if (status == eslOK) status = p7_Backward_Frameshift(dsq1, L, f.om, ox1, ox2, ov, &out[1]);
  if (status == eslOK) status = p7_Decoding_Frameshift(f.om, ox1, ox2);
  if (status == eslOK) {
    p7_OptimalAccuracy_Frameshift(f.om, ox1, ox2, &out[2]);
    p7_OATrace_Frameshift(f.om, ox1, ox2, &tr);
    p7_Null2_fs_ByExpectation(f.om, ox1, null2);
  }
  const int n = tr.N;
  for (int z = 0; z < n && z < tcap; z++) { tst[z] = tr.st[z]; tk[z] = tr.k[z]; ti[z] = tr.i[z]; tc[z] = tr.c[z]; tpp[z] = tr.pp[z]; }
  free(tr.st); free(tr.k); free(tr.i); free(tr.c); free(tr.pp);
  p7_omx_Destroy(ox1); p7_omx_Destroy(ox2); p7_oivx_Destroy(ov);
  hs_fs_close(&f);
  return status == eslOK ? n : -2;
}

/* region_trace_ensemble_frameshift's calls (p7_domaindef.c:411-430): multihit Forward at the saved length, stochastic tracebacks */
int hs_fs_region(const char *path, int idx, const uint8_t *dsq1, int L, uint32_t seed, int ntraces, float *fwdsc, int *ndom, int *first_i, int *last_i)
{
  HS_FS f;
  if (hs_fs_open(path, idx, 5, &f) != eslOK) return eslFAIL;
  P7_OMX *ox1 = p7_omx_Create_dpf(f.gm.M, L, L, p7X_NSCELLS_FS);
  P7_OIVX *ov = p7_oivx_Create(f.gm.M, 5);
  ESL_RANDOMNESS rng = { seed };
  p7_fs_oprofile_ReconfigMultihit(f.om, 100);
  int status = p7_Forward_Frameshift(dsq1, L, f.om, ox1, ov, fwdsc);
  for (int t = 0; t < ntraces && status == eslOK; t++) {
    P7_TRACE tr; memset(&tr, 0, sizeof tr);
    status = p7_StochasticTrace_Frameshift(&rng, dsq1, L, f.om, ox1, &tr);
    ndom[t] = 0; first_i[t] = 0; last_i[t] = 0;
    for (int z = 0; z < tr.N; z++) {
      if (tr.st[z] == p7T_B) ndom[t]++;
      if (tr.st[z] == p7T_M) { if (!first_i[t]) first_i[t] = tr.i[z] - tr.c[z] + 1; last_i[t] = tr.i[z]; }
    }
    free(tr.st); free(tr.k); free(tr.i); free(tr.c); free(tr.pp);
  }
  p7_omx_Destroy(ox1); p7_oivx_Destroy(ov);
  hs_fs_close(&f);
  return status;
}
