/* fs_domaindef.c -- ORACLE (test infrastructure): what the frameshift branch does with a DNA window after the decision
 * of p7_pli_Frameshift: domain definition and the hit's score.
 *
 * Restates, in plain scalar C (generic, log-space forms):
 *   p7_GDomainDecoding_Frameshift                        src/generic_decoding_frameshift.c:204-290
 *   p7_domaindef_ByPosteriorHeuristics_Frameshift_BATH   src/p7_domaindef.c:301-473   (single-domain regions only)
 *   is_multidomain_region_frameshift                     src/p7_domaindef.c:684-714
 *   rescore_isolated_domain_frameshift                   src/p7_domaindef.c:993-1175
 *   p7_GOATrace_Frameshift + select_*                    src/generic_optacc_frameshift.c:373-588
 *   p7_pli_postDomainDef_Frameshift_BATH                 src/p7_pipeline.c:1005-1144  (scores; no alignment display)
 * Not restated: the stochastic-trace clustering of multi-domain regions (p7_domaindef.c:396-455; such regions are
 * reported as skipped), p7_pli_computeAliScores_BATH's "aliscore < 0" garbage rule (:1070-1080), alignment display.
 * The reference keeps the length configuration of om_fs5 from whatever it scored last; here the domain decoding always
 * uses the configuration bathsearch starts with (L = 100 residues, multihit; bathsearch.c:797).
 * Pinned by tutorial/AMP_N-fs.out (tests/test_oracle_cpu.py): score 82.8 bits, bias 0.1, hmm 1..131, ali 1..402, E 1.9e-27.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529
#define NINF (-INFINITY)

/* ---- p7_GDomainDecoding_Frameshift: btot/etot/mocc[0..L] from the parsers' special-state rows (log space) */
void bo_gdomain_decoding_fs(const bo_fs_profile *gm5, const bo_gmx *fwd, const bo_gmx *bck, float *btot, float *etot, float *mocc)
{
  const int L = fwd->L;
  const float Z = bo_flogsum(BO_X(bck,0,BO_GN), bo_flogsum(BO_X(bck,1,BO_GN), BO_X(bck,2,BO_GN)));
  const float tN = gm5->xsc[BO_XN][BO_LOOP], tC = gm5->xsc[BO_XC][BO_LOOP], tJ = gm5->xsc[BO_XJ][BO_LOOP];
  for (int i = 0; i < 3 && i <= L; i++) btot[i] = etot[i] = mocc[i] = 0.f;
  for (int i = 3; i <= L; i++) {
    btot[i] = btot[i-3] + expf(BO_X(fwd,i-3,BO_GB) + BO_X(bck,i-3,BO_GB) - Z);
    etot[i] = etot[i-3] + expf(BO_X(fwd,i,BO_GE) + BO_X(bck,i,BO_GE) - Z);
  }
#define NJC(a,b) (expf(BO_X(fwd,a,BO_GN) + BO_X(bck,b,BO_GN) + tN - Z))
#define CJC(a,b) (expf(BO_X(fwd,a,BO_GC) + BO_X(bck,b,BO_GC) + tC - Z))
#define JJC(a,b) (expf(BO_X(fwd,a,BO_GJ) + BO_X(bck,b,BO_GJ) + tJ - Z))
  for (int i = 3; i < L - 1; i++) {
    float njcp = 0.0f;
    njcp += NJC(i-3,i); njcp += NJC(i-2,i+1); njcp += NJC(i-1,i+2);
    njcp += CJC(i-3,i); njcp += CJC(i-2,i+1); njcp += CJC(i-1,i+2);
    njcp += JJC(i-3,i); njcp += JJC(i-2,i+1); njcp += JJC(i-1,i+2);
    mocc[i] = (float)(1. - njcp);
  }
  if (L >= 4) {
    float njcp = 0.0f;
    njcp += NJC(L-4,L-1); njcp += NJC(L-3,L);
    njcp += CJC(L-4,L-1); njcp += CJC(L-3,L);
    njcp += JJC(L-4,L-1); njcp += JJC(L-3,L);
    mocc[L-1] = (float)(1. - njcp);
    njcp = 0.0f;
    njcp += NJC(L-3,L); njcp += CJC(L-3,L); njcp += JJC(L-3,L);
    mocc[L] = (float)(1. - njcp);
  }
#undef NJC
#undef CJC
#undef JJC
}

/* ---- the domains' traces, kept for the tests that render alignment blocks from them */
static bo_domtrace *g_traces = NULL;
static int g_ntraces = 0, g_traces_alloc = 0;
static int g_win_start = 1;                                      /* dna_window->n of the window being defined */
void bo_traces_reset(void)
{
  for (int t = 0; t < g_ntraces; t++) { free(g_traces[t].st); free(g_traces[t].k); free(g_traces[t].i); free(g_traces[t].c); free(g_traces[t].pp); }
  g_ntraces = 0;
}
int bo_traces_count(void) { return g_ntraces; }
const bo_domtrace *bo_traces_get(int idx) { return (idx >= 0 && idx < g_ntraces) ? &g_traces[idx] : NULL; }
int bo_traces_push(int N, int win_start, int orf_start, int frameshift)
{
  if (g_ntraces == g_traces_alloc) { g_traces_alloc = g_traces_alloc ? g_traces_alloc * 2 : 64; g_traces = realloc(g_traces, sizeof(bo_domtrace) * (size_t) g_traces_alloc); }
  bo_domtrace *t = &g_traces[g_ntraces];
  t->N = N; t->win_start = win_start; t->orf_start = orf_start; t->frameshift = frameshift;
  t->st = calloc((size_t) N + 1, 1); t->c = calloc((size_t) N + 1, 1);
  t->k = calloc((size_t) N + 1, sizeof(int32_t)); t->i = calloc((size_t) N + 1, sizeof(int32_t)); t->pp = calloc((size_t) N + 1, sizeof(float));
  return g_ntraces++;
}

/* ---- traces */
void bo_trace_init(bo_trace *t) { memset(t, 0, sizeof *t); }
void bo_trace_free(bo_trace *t) { free(t->st); free(t->k); free(t->i); free(t->c); free(t->pp); memset(t, 0, sizeof *t); }
static void tr_push(bo_trace *t, int st, int k, int i, int c, float pp)
{
  if (t->N == t->nalloc) {
    t->nalloc = t->nalloc ? t->nalloc * 2 : 256;
    t->st = realloc(t->st, (size_t) t->nalloc); t->k = realloc(t->k, sizeof(int32_t) * (size_t) t->nalloc);
    t->i = realloc(t->i, sizeof(int32_t) * (size_t) t->nalloc); t->c = realloc(t->c, sizeof(int32_t) * (size_t) t->nalloc);
    t->pp = realloc(t->pp, sizeof(float) * (size_t) t->nalloc);
  }
  t->st[t->N] = (int8_t) st; t->k[t->N] = k; t->i[t->N] = i; t->c[t->N] = c; t->pp[t->N] = pp; t->N++;
}

static int argmax(const float *v, int n) { int b = 0; for (int j = 1; j < n; j++) if (v[j] > v[b]) b = j; return b; }   /* esl_vec_FArgMax: first maximum */

/* ---- p7_GOATrace_Frameshift (generic_optacc_frameshift.c:373-588).  pp: posterior matrix (8 cells), gx: OA matrix. */
int bo_goatrace_fs(const bo_fs_profile *gm, const bo_gmx *pp, const bo_gmx *gx, bo_trace *tr)
{
  const float *tsc = gm->tsc;
  const int M = gm->M, L = gx->L;
#define DELTA(s,k) ((tsc[(k) * BO_NTRANS + (s)] == NINF) ? FLT_MIN : 1.0f)
#define XD(st,t)   ((gm->xsc[st][t] == NINF) ? FLT_MIN : 1.0f)
#define OM(i,k) BO_DP(gx,i,k,BO_GM)
#define OI(i,k) BO_DP(gx,i,k,BO_GI)
#define OD(i,k) BO_DP(gx,i,k,BO_GD)
  int i = L, k = 0, c = 0;
  tr->N = 0;
  tr_push(tr, BO_T_T, k, i, c, 0.0f);
  tr_push(tr, BO_T_C, k, i, c, 0.0f);
  int sprv = BO_T_C, scur = -1;
  while (sprv != BO_T_S) {
    float path[4];
    switch (sprv) {
    case BO_T_M: {                                                   /* select_m */
      static const int state[4] = { BO_T_M, BO_T_I, BO_T_D, BO_T_B };
      path[0] = DELTA(BO_MM, k-1) * OM(i,k-1); path[1] = DELTA(BO_IM, k-1) * OI(i,k-1);
      path[2] = DELTA(BO_DM, k-1) * OD(i,k-1); path[3] = DELTA(BO_BM, k-1) * BO_X(gx,i,BO_GB);
      scur = state[argmax(path, 4)]; k--; break; }
    case BO_T_D:                                                     /* select_d */
      path[0] = DELTA(BO_MD, k-1) * OM(i,k-1); path[1] = DELTA(BO_DD, k-1) * OD(i,k-1);
      scur = (path[0] >= path[1]) ? BO_T_M : BO_T_D; k--; break;
    case BO_T_I:                                                     /* select_i */
      path[0] = DELTA(BO_MI, k) * OM(i-3,k); path[1] = DELTA(BO_II, k) * OI(i-3,k);
      scur = (path[0] >= path[1]) ? BO_T_M : BO_T_I; i -= 3; break;
    case BO_T_N: scur = (i == 0) ? BO_T_S : BO_T_N; break;           /* select_n */
    case BO_T_C: {                                                   /* select_c */
      static const int state[4] = { BO_T_C, BO_T_C, BO_T_C, BO_T_E };
      if (i < 4) { scur = BO_T_E; break; }
      const float t1 = XD(BO_XC, BO_LOOP), t2 = XD(BO_XE, BO_MOVE);
      path[0] = t1 * (BO_X(gx,i-3,BO_GC) + BO_X(pp,i,BO_GC));
      path[1] = (i < L)     ? t1 * (BO_X(gx,i-2,BO_GC) + BO_X(pp,i+1,BO_GC)) : FLT_MIN;
      path[2] = (i < L - 1) ? t1 * (BO_X(gx,i-1,BO_GC) + BO_X(pp,i+2,BO_GC)) : FLT_MIN;
      path[3] = t2 * BO_X(gx,i,BO_GE);
      scur = state[argmax(path, 4)]; break; }
    case BO_T_J: {                                                   /* select_j */
      if (i <= 5) { scur = BO_T_E; break; }
      const float t1 = XD(BO_XJ, BO_LOOP), t2 = XD(BO_XE, BO_LOOP);
      path[0] = t1 * (BO_X(gx,i,BO_GJ) + BO_X(pp,i,BO_GJ)); path[1] = t2 * BO_X(gx,i,BO_GE);
      scur = (argmax(path, 2) == 0) ? BO_T_J : BO_T_E; break; }
    case BO_T_E: {                                                   /* select_e, local */
      float mx = NINF; int smax = -1, kmax = -1;
      for (int kk = 1; kk <= M; kk++) {
        if (OM(i,kk) > mx) { mx = OM(i,kk); smax = BO_T_M; kmax = kk; }
        if (OD(i,kk) > mx) { mx = OD(i,kk); smax = BO_T_D; kmax = kk; }
      }
      k = kmax; scur = smax; break; }
    case BO_T_B: {                                                   /* select_b */
      const float t1 = XD(BO_XN, BO_MOVE), t2 = XD(BO_XJ, BO_MOVE);
      scur = (t1 * BO_X(gx,i,BO_GN) > t2 * BO_X(gx,i,BO_GJ)) ? BO_T_N : BO_T_J; break; }
    default: return BO_EINVAL;
    }
    if (scur == -1) return BO_EINVAL;
    float postprob = 0.0f;                                           /* get_postprob (:425-440), including its case fall-through */
    switch (scur) {
    case BO_T_M: postprob = BO_DP(pp,i,k,BO_GM); break;
    case BO_T_I: postprob = BO_DP(pp,i,k,BO_GI); break;
    case BO_T_N: if (sprv == scur) { postprob = BO_X(pp,i,BO_GN); break; }   /* falls through in the reference when sprv != scur */
    /* fallthrough */
    case BO_T_C: if (sprv == scur) { postprob = BO_X(pp,i,BO_GC); break; }
    /* fallthrough */
    case BO_T_J: if (sprv == scur) { postprob = BO_X(pp,i,BO_GJ); break; }
    /* fallthrough */
    default: postprob = 0.0f;
    }
    if (scur == BO_T_M) {                                            /* select_codon */
      float cod[5];
      for (int q = 0; q < 5; q++) cod[q] = BO_DP(pp,i,k,BO_GM + 1 + q);
      c = argmax(cod, 5) + 1;
    } else c = 0;
    tr_push(tr, scur, k, i, c, postprob);
    if ((scur == BO_T_N || scur == BO_T_C || scur == BO_T_J) && scur == sprv) i--;
    sprv = scur;
    i -= c;
    if (tr->N > 4 * (L + M) + 64) return BO_EINVAL;                  /* cannot happen on a well-formed matrix */
  }
  /* p7_trace_fs_Reverse */
  for (int a = 0, b = tr->N - 1; a < b; a++, b--) {
    int8_t s = tr->st[a]; tr->st[a] = tr->st[b]; tr->st[b] = s;
    int32_t t;
    t = tr->k[a]; tr->k[a] = tr->k[b]; tr->k[b] = t;
    t = tr->i[a]; tr->i[a] = tr->i[b]; tr->i[b] = t;
    t = tr->c[a]; tr->c[a] = tr->c[b]; tr->c[b] = t;
    float p = tr->pp[a]; tr->pp[a] = tr->pp[b]; tr->pp[b] = p;
  }
  return BO_OK;
#undef DELTA
#undef XD
#undef OM
#undef OI
#undef OD
}

/* is_multidomain_region_frameshift, p7_domaindef.c:684-714 */
static int is_multidomain(const float *btot, const float *etot, int i, int j, float rt3)
{
  float mx = -1.0f;
  for (int ph = 0; ph < 3; ph++) {
    const int f = (j - i + 1 - ph) % 3;
    for (int z = i + 2 + ph; z <= j - f; z += 3) {
      const float a = etot[z] - etot[i - 1 + ph], b = btot[j - f] - btot[z - 3];
      const float e = a < b ? a : b;
      if (e > mx) mx = e;
    }
  }
  return mx >= rt3;
}

static void dom_push(bo_fsdomain **d, int *n, int *alloc, const bo_fsdomain *r)
{
  if (*n == *alloc) { *alloc = *alloc ? *alloc * 2 : 16; *d = realloc(*d, sizeof(bo_fsdomain) * (size_t) *alloc); }
  (*d)[(*n)++] = *r;
}

/* p7_pli_computeAliScores_BATH, p7_pipeline.c:781-979, on a frameshift trace (tr->i already window-absolute): per aligned
 * column from the first to the last match state, the amino row score of the (quasi-)codon's best amino acid plus the
 * transition that entered the state; the last match state gets no MM transition (inner loops stop at z1 < z2, :899). */
extern int bo_aliscore_drops;
extern float bo_aliscore_min;

static float fs_aliscore(const bo_fs_profile *gm, const uint8_t *wdsq, const bo_trace *tr)
{
  const size_t W = (size_t) gm->M + 1;
  int z1, z2;
  for (z1 = 0; z1 < tr->N; z1++) if (tr->st[z1] == BO_T_M) break;
  for (z2 = tr->N - 1; z2 >= 0; z2--) if (tr->st[z2] == BO_T_M) break;
  float total = 0.0f;
  for (int z = z1; z <= z2; z++) {
    const int k = tr->k[z], prev = tr->st[z - 1];
    float sc;
    if (tr->st[z] == BO_T_M) {
      const int i = tr->i[z], c = tr->c[z];
      int degen = 0, ci = 0;
      for (int q = 0; q < c; q++) if (wdsq[i - q] >= 4) degen = 1;
      /* p7P_CODON{1..5}_FS5, hmmer.h:306-310: the last nucleotide is the most significant digit */
      if      (c == 1) ci = degen ? BO_DEGEN5_QC2 : wdsq[i] * 341;
      else if (c == 2) ci = degen ? BO_DEGEN5_QC1 : wdsq[i] * 341 + wdsq[i-1] * 85 + 1;
      else if (c == 3) ci = degen ? BO_DEGEN5_C   : wdsq[i] * 341 + wdsq[i-1] * 85 + wdsq[i-2] * 21 + 2;
      else if (c == 4) ci = degen ? BO_DEGEN5_QC1 : wdsq[i] * 341 + wdsq[i-1] * 85 + wdsq[i-2] * 21 + wdsq[i-3] * 5 + 3;
      else             ci = degen ? BO_DEGEN5_QC2 : wdsq[i] * 341 + wdsq[i-1] * 85 + wdsq[i-2] * 21 + wdsq[i-3] * 5 + wdsq[i-4] + 4;
      const int amino = gm->codons[(size_t) k * gm->maxcodons + ci];
      sc = gm->rsc[(size_t)(gm->maxcodons + amino) * W + k];
      if      (prev == BO_T_I) sc += gm->tsc[(k - 1) * BO_NTRANS + BO_IM];
      else if (prev == BO_T_D) sc += gm->tsc[(k - 1) * BO_NTRANS + BO_DM];
      else if (prev == BO_T_M && z < z2) sc += gm->tsc[(k - 1) * BO_NTRANS + BO_MM];
    } else if (tr->st[z] == BO_T_I) sc = gm->tsc[k * BO_NTRANS + (prev == BO_T_I ? BO_II : BO_MI)];
    else if (tr->st[z] == BO_T_D)   sc = gm->tsc[(k - 1) * BO_NTRANS + (prev == BO_T_D ? BO_DD : BO_MD)];
    else continue;
    total += sc;
  }
  return total;
}

/* rescore_isolated_domain_frameshift (p7_domaindef.c:993-1175) on wdsq[i..j] (1-based in the window) */
static int rescore_domain(const bo_pipeline *pli, bo_fs_profile *gm5, bo_bg *bg, const uint8_t *wdsq, int i, int j,
                          bo_fsdomain **doms, int *ndom, int *dalloc, bo_trace *tr)
{
  const int Ld = j - i + 1, M = gm5->M;
  if (Ld < 15) return BO_OK;
  bo_bg_setlength(bg, Ld / 3);
  const float nullsc = bo_bg_fs_nullone(bg, Ld / 3);
  bo_fs_profile_reconfig_length(gm5, Ld / 3);
  bo_gmx *fwd = bo_gmx_create(M, Ld + 1, Ld, BO_NSCELLS_FS), *bck = bo_gmx_create(M, Ld + 1, Ld, BO_NSCELLS);
  float envsc, oasc;
  int status = BO_OK;
  if (bo_gforward_fs(wdsq + i - 1, Ld, gm5, fwd, 0, &envsc) == BO_ERANGE) goto DONE;
  {
    const float seqscore = (float)((envsc - nullsc) / LOG2C);
    const double P = bo_exp_surv(seqscore, gm5->evparam[BO_FTAUFS5], gm5->evparam[BO_FLAMBDA]);
    const double Z = (double)(float)((float) pli->nres / (float) gm5->max_length);    /* pli->Z, p7_domaindef.c:1033 */
    if (pli->inc_by_E && P * Z > pli->E) goto DONE;            /* :1034 */
  }
  if (bo_gbackward_fs(wdsq + i - 1, Ld, gm5, bck, NULL) == BO_ERANGE) goto DONE;
  if (bo_gdecoding_fs(gm5, fwd, bck) == BO_ERANGE) { status = BO_FAIL; goto DONE; }
  bo_goptacc_fs(gm5, fwd, bck, &oasc);                          /* fwd now holds posteriors, bck the OA matrix */
  if ((status = bo_goatrace_fs(gm5, fwd, bck, tr)) != BO_OK) goto DONE;
  for (int z = 0; z < tr->N; z++) if (tr->i[z] >= 0) tr->i[z] += i - 1;
  { const float alisc = fs_aliscore(gm5, wdsq, tr);
    if (alisc < bo_aliscore_min) bo_aliscore_min = alisc;
    if (alisc < 0.0f) { bo_aliscore_drops++; status = BO_FAIL; goto DONE; } }        /* p7_domaindef.c:1072: "repetitive garbage" */
  {
    float null2[BO_KP_AMINO];
    bo_gnull2_fs(gm5, fwd, null2);
    /* per-position null2 scores along the trace, p7_domaindef.c:1086-1142 */
    float *n2sc = calloc((size_t) j + 2, sizeof(float));
    int t = -1, u = -1, v = -1, w = -1, x = -1, z = 0, pos = i;
    while (pos <= j && z < tr->N) {
      x = (wdsq[pos] < 4) ? wdsq[pos] : BO_MAXCODONS5;
      switch (tr->st[z]) {
      case BO_T_N: case BO_T_C: case BO_T_J:
        n2sc[pos] = 0.0f;
        if (tr->i[z] == pos && pos > i + 1) pos++;
        z++; break;
      case BO_T_S: case BO_T_B: case BO_T_E: case BO_T_T: case BO_T_D: z++; break;
      case BO_T_M:
        if (tr->i[z] == pos) {
          int ci = 0;
#define MINIDX(a,b) ((a) < (b) ? (a) : (b))
          if      (tr->c[z] == 1) { ci = x * 341;                                              ci = MINIDX(ci, BO_DEGEN5_QC2); }
          else if (tr->c[z] == 2) { ci = x * 341 + w * 85 + 1;                                 ci = MINIDX(ci, BO_DEGEN5_QC1); }
          else if (tr->c[z] == 3) { ci = x * 341 + w * 85 + v * 21 + 2;                        ci = MINIDX(ci, BO_DEGEN5_C); }
          else if (tr->c[z] == 4) { ci = x * 341 + w * 85 + v * 21 + u * 5 + 3;                ci = MINIDX(ci, BO_DEGEN5_QC1); }
          else if (tr->c[z] == 5) { ci = x * 341 + w * 85 + v * 21 + u * 5 + t + 4;            ci = MINIDX(ci, BO_DEGEN5_QC2); }
          n2sc[pos] = logf(null2[gm5->codons[(size_t) tr->k[z] * gm5->maxcodons + ci]]);
          if (n2sc[pos] == NINF) n2sc[pos] = 0.0f;
          z++;
        } else n2sc[pos] = 0.0f;
        pos++; break;
      case BO_T_I:
        if (tr->i[z] == pos) {
          int ci = x * 341 + w * 85 + v * 21 + 2;
          ci = MINIDX(ci, BO_DEGEN5_C);
          n2sc[pos] = logf(null2[gm5->codons[(size_t) tr->k[z] * gm5->maxcodons + ci]]);
          if (n2sc[pos] == NINF) n2sc[pos] = 0.0f;
          z++;
        } else n2sc[pos] = 0.0f;
        pos++; break;
      default: z++; break;
      }
      t = u; u = v; v = w; w = x;
    }
    float domcorrection = 0.0f;
    for (pos = i; pos <= j; pos++) domcorrection += n2sc[pos];
    free(n2sc);
    int z1, z2;
    for (z1 = 0; z1 < tr->N; z1++) if (tr->st[z1] == BO_T_M) break;
    for (z2 = tr->N - 1; z2 >= 0; z2--) if (tr->st[z2] == BO_T_M) break;
    if (z1 < tr->N && z2 >= 0) {
      bo_fsdomain d;
      memset(&d, 0, sizeof d);
      d.iali = tr->i[z1] - (tr->c[z1] - 1); d.jali = tr->i[z2]; d.ienv = i; d.jenv = j;
      d.ihmm = tr->k[z1]; d.jhmm = tr->k[z2];
      d.envsc = envsc; d.oasc = oasc; d.domcorrection = domcorrection > 0.f ? domcorrection : 0.f;
      int shifts = 0;
      for (int z = 0; z < tr->N; z++) if (tr->st[z] == BO_T_M && tr->c[z] != 3) shifts++;
      d.n_shifted_codons = shifts;
      {                                                            /* dom->tr = p7_trace_fs_Clone(ddef->tr), p7_domaindef.c:1171 */
        const int tix = bo_traces_push(z2 - z1 + 1, g_win_start, 0, 1);
        const bo_domtrace *t = bo_traces_get(tix);
        for (int z = z1; z <= z2; z++) {
          const int q = z - z1;
          if (tr->st[z] == BO_T_M)      { t->st[q] = 1; t->k[q] = tr->k[z]; t->i[q] = tr->i[z]; t->c[q] = (int8_t) tr->c[z]; t->pp[q] = tr->pp[z]; }
          else if (tr->st[z] == BO_T_I) { t->st[q] = 3; t->k[q] = tr->k[z]; t->i[q] = tr->i[z]; t->c[q] = 0; t->pp[q] = tr->pp[z]; }
          else                          { t->st[q] = 2; t->k[q] = tr->k[z]; t->i[q] = i - 1;     t->c[q] = 0; t->pp[q] = 0.0f; }   /* AppendWithPP: D has i = 0; then += i - 1 (:1053) */
        }
        d.trace_idx = tix;
      }
      dom_push(doms, ndom, dalloc, &d);
    }
  }
DONE:
  bo_gmx_free(fwd); bo_gmx_free(bck);
  return status;
}

/* p7_domaindef_ByPosteriorHeuristics_Frameshift_BATH + p7_pli_postDomainDef_Frameshift_BATH for one DNA window.
 * wdsq[1..L]; gm3: 3-codon profile (parsers); gm5: 5-codon profile (envelopes); window_start: dna_window->n;
 * complementarity / seq_n for the coordinate mapping.  Appends to doms; *nskipped counts multi-domain regions. */
int bo_domaindef_fs(bo_pipeline *pli, bo_fs_profile *gm3, bo_fs_profile *gm5, bo_bg *bg, const uint8_t *wdsq, int L,
                    int window_start, int complementarity, int seq_n, bo_fsdomain **doms, int *ndom, int *dalloc, int *nskipped)
{
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  const int M = gm3->M;
  const int first = *ndom;
  g_win_start = window_start;
  bo_gmx *fx = bo_gmx_create(M, L + 1, L, 3), *bx = bo_gmx_create(M, L + 1, L, 3);
  float fsc, bsc;
  bo_fs_profile_reconfig_length(gm3, L / 3);
  bo_k_gforward_parser_fs3(wdsq, L, gm3, fx, &fsc);
  if (bo_k_gbackward_parser_fs3(wdsq, L, gm3, bx, &bsc) == BO_ERANGE) { bo_gmx_free(fx); bo_gmx_free(bx); return BO_OK; }   /* p7_pipeline.c:1471 */
  float *btot = calloc((size_t) L + 2, sizeof(float)), *etot = calloc((size_t) L + 2, sizeof(float)), *mocc = calloc((size_t) L + 2, sizeof(float));
  bo_fs_profile_reconfig_multihit(gm5, 100);                      /* the configuration bathsearch.c:797 starts with */
  bo_gdomain_decoding_fs(gm5, fx, bx, btot, etot, mocc);
  bo_gmx_free(fx); bo_gmx_free(bx);
  bo_fs_profile_reconfig_unihit(gm5, 100 / 3);                    /* p7_domaindef.c:324: unihit for every envelope */
  bo_trace tr;
  bo_trace_init(&tr);

  int i = -1, triggered = 0, start = 0, end = 0, d = 0;
  for (int j = 1; j < L; j++) {
    if (!triggered) {
      if (mocc[j] >= rt1) triggered = 1;
      d = j;
    } else {
      while (d > 1 && !start) {                                   /* :343-360: the start must be evident in all three frames */
        d--;
        if (d > 3 && mocc[d] - (btot[d] - btot[d-3]) < rt2) {
          d--;
          if (d > 3 && mocc[d] - (btot[d] - btot[d-3]) < rt2) {
            d--;
            if (d > 3 && mocc[d] - (btot[d] - btot[d-3]) < rt2) { d--; start = 1; }
          }
        }
      }
      i = (d - 3 > 1) ? d - 3 : 1;
      d = j + 1;
      while (d < L && !end) {                                     /* :365-382 */
        d++;
        if (d < L && mocc[d] - (etot[d] - etot[d-3]) < rt2) {
          d++;
          if (d < L && mocc[d] - (etot[d] - etot[d-3]) < rt2) {
            d++;
            if (d < L && mocc[d] - (etot[d] - etot[d-3]) < rt2) { d++; end = 1; }
          }
        }
      }
      j = (d + 3 < L) ? d + 3 : L;
      if (j - i + 1 >= 12) {
        if (is_multidomain(btot, etot, i, j, rt3)) {              /* p7_domaindef.c:396-455 */
          const int Lr = j - i + 1;
          int env[2 * 32], nc = 0;
          float rsc;
          (*nskipped)++;                                          /* ddef->nclustered */
          bo_fs_profile_reconfig_multihit(gm5, 100);              /* saveL */
          bo_gmx *rf = bo_gmx_create(M, Lr + 1, Lr, BO_NSCELLS_FS);
          if (bo_gforward_fs(wdsq + i - 1, Lr, gm5, rf, 0, &rsc) != BO_ERANGE && rsc > -INFINITY) nc = bo_region_trace_ensemble_fs(gm5, i, j, rf, env, 32);
          bo_gmx_free(rf);
          bo_fs_profile_reconfig_unihit(gm5, 100 / 3);
          for (int q = 0; q < nc; q++) rescore_domain(pli, gm5, bg, wdsq, env[2 * q] > 1 ? env[2 * q] : 1, env[2 * q + 1], doms, ndom, dalloc, &tr);
        } else rescore_domain(pli, gm5, bg, wdsq, i, j, doms, ndom, dalloc, &tr);
      }
      i = -1; triggered = 0; start = 0; end = 0;
    }
  }
  bo_trace_free(&tr);
  free(btot); free(etot); free(mocc);

  /* ---- p7_pli_postDomainDef_Frameshift_BATH (p7_pipeline.c:1005-1144), dnasq->start = 1 (top) or seq_n (bottom) */
  const int64_t dstart = complementarity ? seq_n : 1;
  for (int q = first; q < *ndom; q++) {
    bo_fsdomain *dm = &(*doms)[q];
    const int ali_len = dm->jali - dm->iali + 1, env_len = dm->jenv - dm->ienv + 1;
    if (ali_len < 12) { dm->reported = 0; continue; }
    if (!complementarity) {
      dm->ienv = (int32_t)(dstart + window_start + dm->ienv - 2); dm->jenv = (int32_t)(dstart + window_start + dm->jenv - 2);
      dm->iali = (int32_t)(dstart + window_start + dm->iali - 2); dm->jali = (int32_t)(dstart + window_start + dm->jali - 2);
    } else {
      dm->ienv = (int32_t)(dstart - (window_start + dm->ienv) + 2); dm->jenv = (int32_t)(dstart - (window_start + dm->jenv) + 2);
      dm->iali = (int32_t)(dstart - (window_start + dm->iali) + 2); dm->jali = (int32_t)(dstart - (window_start + dm->jali) + 2);
    }
    const int ml = gm5->max_length;
    float bitscore = dm->envsc;                                   /* :1055-1059, float accumulations of double terms */
    bitscore -= 2 * log(2. / ((env_len / 3.) + 2));
    bitscore += 2 * log(2. / (ml + 2));
    bitscore -= ((env_len - ali_len) / 3.) * log((float)(env_len / 3.) / (float)((env_len / 3.) + 2));
    bitscore += (((env_len > ml * 3 ? env_len : ml * 3) - ali_len) / 3.) * log((float) ml / (float)(ml + 2));
    const float dom_bias = pli->do_null2 ? bo_flogsum(0.0f, (float)(log(1. / 256.) + dm->domcorrection)) : 0.0f;   /* :1063-1066; bg->omega, p7_bg.c:74 */
    const int nl = (env_len / 3 > ml) ? env_len / 3 : ml;
    bo_bg_setlength(bg, nl);
    const float nullsc = bo_bg_fs_nullone(bg, nl);
    const float dom_score = (float)((bitscore - (nullsc + dom_bias)) / LOG2C);
    const double lnP = bo_exp_logsurv(dom_score, gm5->evparam[BO_FTAUFS5], gm5->evparam[BO_FLAMBDA]);
    const double Z = (double)(float)((float) pli->nres / (float) ml);
    dm->dombias = dom_bias; dm->bitscore = dom_score; dm->lnP = lnP;
    dm->pre_score = (float)(bitscore / LOG2C);
    dm->reported = (pli->inc_by_E ? (exp(lnP) * Z <= pli->E) : (dom_score >= pli->T)) ? 1 : 0;          /* :1080 */
  }
  return BO_OK;
}
