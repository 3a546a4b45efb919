/* domaindef.c -- ORACLE (test infrastructure): domain definition and hit scores of the standard (non-frameshift) branch.
 *
 * Restates, in plain scalar C, in the optimized profile's odds-ratio space with the reference's scaling:
 *   p7_DomainDecoding                         src/impl_sse/decoding.c:155-196
 *   p7_domaindef_ByPosteriorHeuristics_BATH   src/p7_domaindef.c:491-614  (single-domain regions only)
 *   is_multidomain_region                     src/p7_domaindef.c:642-654
 *   rescore_isolated_domain_bath              src/p7_domaindef.c:1194-1325
 *     p7_Forward / p7_Backward (full)         src/impl_sse/fwdback.c:94,196 (oracle/filters.c engines)
 *     p7_Decoding                             src/impl_sse/decoding.c:61-118
 *     p7_OptimalAccuracy, p7_OATrace          src/impl_sse/optacc.c:58-173, 225-430
 *     p7_Null2_ByExpectation                  src/impl_sse/null2.c:50-124
 *     p7_trace_fs_Convert                     src/p7_trace.c:405-438
 *   p7_pli_postDomainDef_BATH                 src/p7_pipeline.c:1172-1300 (scores; no alignment display)
 * Not restated: stochastic-trace clustering of multi-domain regions (:536-590), the "aliscore < 0" rule (:1253-1261).
 * Pinned by tutorial/PTH2.tbl and tutorial/AMP_N.out (tests/test_oracle_cpu.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529
#define NINF (-INFINITY)
enum { XE = 0, XN, XJ, XB, XC, XS };      /* P7_OMX xmx columns (impl_sse.h:253-262) as the oracle parsers write them */
enum { cM = 0, cD = 1, cI = 2 };          /* cells of the full matrices */

/* p7_oprofile_ReconfigUnihit / Multihit, p7_oprofile.c:1418-1430, 1395-1407 */
void bo_oprofile_reconfig_unihit(bo_oprofile *om, int L) { om->xf[BO_XE][BO_MOVE] = 1.0f; om->xf[BO_XE][BO_LOOP] = 0.0f; om->nj = 0.0f; bo_oprofile_reconfig_length(om, L); }
void bo_oprofile_reconfig_multihit(bo_oprofile *om, int L) { om->xf[BO_XE][BO_MOVE] = 0.5f; om->xf[BO_XE][BO_LOOP] = 0.5f; om->nj = 1.0f; bo_oprofile_reconfig_length(om, L); }

/* p7_DomainDecoding, decoding.c:155-196; fx/bx: (L+1) x 6 rows of the parsers */
int bo_domain_decoding(const bo_oprofile *om, const float *fx, const float *bx, int L, int own_scales, float *btot, float *etot, float *mocc)
{
  float scaleproduct = (float)(1.0 / bx[XN]);
  btot[0] = etot[0] = mocc[0] = 0.0f;
  for (int i = 1; i <= L; i++) {
    btot[i] = btot[i-1] + (fx[(i-1)*6+XB] * bx[(i-1)*6+XB] * fx[(i-1)*6+XS] * scaleproduct);
    if (own_scales) scaleproduct *= fx[(i-1)*6+XS] / bx[(i-1)*6+XS];
    etot[i] = etot[i-1] + (fx[i*6+XE] * bx[i*6+XE] * fx[i*6+XS] * scaleproduct);
    float njcp;
    njcp  = fx[(i-1)*6+XN] * bx[i*6+XN] * om->xf[BO_XN][BO_LOOP] * scaleproduct;
    njcp += fx[(i-1)*6+XJ] * bx[i*6+XJ] * om->xf[BO_XJ][BO_LOOP] * scaleproduct;
    njcp += fx[(i-1)*6+XC] * bx[i*6+XC] * om->xf[BO_XC][BO_LOOP] * scaleproduct;
    mocc[i] = (float)(1. - njcp);
  }
  return isinf(scaleproduct) ? BO_ERANGE : BO_OK;
}

/* p7_Decoding, decoding.c:61-118: pp (may alias bck) <- fwd * bck * totr; ppx rows {E,N,J,B,C} */
static int decoding(const bo_oprofile *om, int L, const float *fwd, const float *fx, float *bck, const float *bx, int own_scales, float *ppx)
{
  const int M = om->M;
  const size_t W = (size_t)(M + 1) * 3;
  float scaleproduct = (float)(1.0 / bx[XN]);
  memset(bck, 0, sizeof(float) * W);
  for (int s = 0; s < 5; s++) ppx[s] = 0.0f;
  for (int i = 1; i <= L; i++) {
    const float totr = scaleproduct * fx[i*6+XS];
    const float *f = fwd + (size_t) i * W;
    float *b = bck + (size_t) i * W;
    b[0] = b[1] = b[2] = 0.f;
    for (int k = 1; k <= M; k++) {
      b[k*3+cM] = f[k*3+cM] * (b[k*3+cM] * totr);
      b[k*3+cD] = 0.0f;                                     /* decoding.c:93: D posteriors are not kept */
      b[k*3+cI] = f[k*3+cI] * (b[k*3+cI] * totr);
    }
    ppx[i*5+XE] = 0.0f;
    ppx[i*5+XN] = fx[(i-1)*6+XN] * bx[i*6+XN] * om->xf[BO_XN][BO_LOOP] * scaleproduct;
    ppx[i*5+XJ] = fx[(i-1)*6+XJ] * bx[i*6+XJ] * om->xf[BO_XJ][BO_LOOP] * scaleproduct;
    ppx[i*5+XC] = fx[(i-1)*6+XC] * bx[i*6+XC] * om->xf[BO_XC][BO_LOOP] * scaleproduct;
    ppx[i*5+XB] = 0.0f;
    if (own_scales) scaleproduct *= fx[i*6+XS] / bx[i*6+XS];
  }
  return isinf(scaleproduct) ? BO_ERANGE : BO_OK;
}

/* p7_OptimalAccuracy, optacc.c:58-173.  "allowed transition ? value : 0" exactly as the masked SSE arithmetic does. */
static float optimal_accuracy(const bo_oprofile *om, int L, const float *pp, const float *ppx, float *oa, float *ox)
{
  const int M = om->M;
  const size_t W = (size_t)(M + 1) * 3;
  const float *tf = om->tf;
#define ALLOW(t, v) (((t) > 0.0f) ? (v) : 0.0f)
  for (int k = 0; k <= M; k++) oa[k*3+cM] = oa[k*3+cD] = oa[k*3+cI] = NINF;
  ox[XE] = NINF; ox[XN] = 0.f; ox[XJ] = NINF; ox[XB] = 0.f; ox[XC] = NINF;
  for (int i = 1; i <= L; i++) {
    const float *p = pp + (size_t) i * W, *pr = oa + (size_t)(i - 1) * W;
    float *c = oa + (size_t) i * W;
    const float xB = ox[(i-1)*5+XB];
    c[cM] = c[cD] = c[cI] = NINF;
    float xE = NINF, dcv = NINF;
    for (int k = 1; k <= M; k++) {
      const float *t = tf + k * BO_NTRANS;
      float sv = ALLOW(t[BO_BM], xB);
      float v;
      v = ALLOW(t[BO_MM], pr[(k-1)*3+cM]); if (v > sv) sv = v;
      v = ALLOW(t[BO_IM], pr[(k-1)*3+cI]); if (v > sv) sv = v;
      v = ALLOW(t[BO_DM], pr[(k-1)*3+cD]); if (v > sv) sv = v;
      sv = sv + p[k*3+cM];
      if (sv > xE) xE = sv;
      c[k*3+cM] = sv;
      c[k*3+cD] = dcv;                                      /* D(i,k) = max(allowed MD(k-1) ? M(i,k-1), allowed DD(k-1) ? D(i,k-1)) */
      dcv = ALLOW(t[BO_MD], sv);
      { const float dd = ALLOW(t[BO_DD], c[k*3+cD]); if (dd > dcv) dcv = dd; }
      float iv = ALLOW(t[BO_MI], pr[k*3+cM]);
      v = ALLOW(t[BO_II], pr[k*3+cI]); if (v > iv) iv = v;
      c[k*3+cI] = iv + p[k*3+cI];
    }
    for (int k = 1; k <= M; k++) if (c[k*3+cD] > xE) xE = c[k*3+cD];
    ox[i*5+XE] = xE;
    float t1, t2;
    t1 = (om->xf[BO_XJ][BO_LOOP] == 0.0f) ? 0.0f : ox[(i-1)*5+XJ] + ppx[i*5+XJ];
    t2 = (om->xf[BO_XE][BO_LOOP] == 0.0f) ? 0.0f : xE;
    ox[i*5+XJ] = t1 > t2 ? t1 : t2;
    t1 = (om->xf[BO_XC][BO_LOOP] == 0.0f) ? 0.0f : ox[(i-1)*5+XC] + ppx[i*5+XC];
    t2 = (om->xf[BO_XE][BO_MOVE] == 0.0f) ? 0.0f : xE;
    ox[i*5+XC] = t1 > t2 ? t1 : t2;
    ox[i*5+XN] = (om->xf[BO_XN][BO_LOOP] == 0.0f) ? 0.0f : ox[(i-1)*5+XN] + ppx[i*5+XN];
    t1 = (om->xf[BO_XN][BO_MOVE] == 0.0f) ? 0.0f : ox[i*5+XN];
    t2 = (om->xf[BO_XJ][BO_MOVE] == 0.0f) ? 0.0f : ox[i*5+XJ];
    ox[i*5+XB] = t1 > t2 ? t1 : t2;
  }
#undef ALLOW
  return ox[L*5+XC];
}

/* p7_OATrace, optacc.c:225-430: returns the first / last match state of the alignment (all the pipeline uses) */
static int oa_trace(const bo_oprofile *om, int L, const float *pp, const float *ppx, const float *oa, const float *ox,
                    int *i1, int *k1, int *i2, int *k2, int *path_st, int *path_k, int *path_i, int *path_n)
{
  const int M = om->M;
  const size_t W = (size_t)(M + 1) * 3;
  const float *tf = om->tf;
  const int Q = ((M - 1) / 4) + 1 > 2 ? ((M - 1) / 4) + 1 : 2;     /* p7O_NQF: select_e visits cells in striped order */
#define PATH(t, v) (((t) == 0.0f) ? NINF : (v))
  int i = L, k = 0, s0 = BO_T_C, n = 0;
  *i1 = *k1 = *i2 = *k2 = -1;
  while (s0 != BO_T_S) {
    int s1 = -1;
    float path[4];
    switch (s0) {
    case BO_T_M: {
      const float *t = tf + k * BO_NTRANS, *pr = oa + (size_t)(i - 1) * W;
      path[3] = PATH(t[BO_BM], ox[(i-1)*5+XB]); path[0] = PATH(t[BO_MM], pr[(k-1)*3+cM]);
      path[1] = PATH(t[BO_IM], pr[(k-1)*3+cI]); path[2] = PATH(t[BO_DM], pr[(k-1)*3+cD]);
      static const int state[4] = { BO_T_M, BO_T_I, BO_T_D, BO_T_B };
      int b = 0; for (int q = 1; q < 4; q++) if (path[q] > path[b]) b = q;
      s1 = state[b]; k--; i--; break; }
    case BO_T_D: {
      const float *t = tf + (k - 1) * BO_NTRANS, *c = oa + (size_t) i * W;     /* MD, DD out of node k-1 */
      path[0] = (k - 1 >= 1) ? PATH(t[BO_MD], c[(k-1)*3+cM]) : NINF;
      path[1] = (k - 1 >= 1) ? PATH(t[BO_DD], c[(k-1)*3+cD]) : NINF;
      s1 = (path[0] >= path[1]) ? BO_T_M : BO_T_D; k--; break; }
    case BO_T_I: {
      const float *t = tf + k * BO_NTRANS, *pr = oa + (size_t)(i - 1) * W;
      path[0] = PATH(t[BO_MI], pr[k*3+cM]); path[1] = PATH(t[BO_II], pr[k*3+cI]);
      s1 = (path[0] >= path[1]) ? BO_T_M : BO_T_I; i--; break; }
    case BO_T_N: s1 = (i == 0) ? BO_T_S : BO_T_N; break;
    case BO_T_C:
      path[0] = (om->xf[BO_XC][BO_LOOP] == 0.0f) ? NINF : ox[(i-1)*5+XC] + ppx[i*5+XC];
      path[1] = (om->xf[BO_XE][BO_MOVE] == 0.0f) ? NINF : ox[i*5+XE];
      s1 = (path[0] > path[1]) ? BO_T_C : BO_T_E; break;
    case BO_T_J:
      path[0] = (om->xf[BO_XJ][BO_LOOP] == 0.0f) ? NINF : ox[(i-1)*5+XJ] + ppx[i*5+XJ];
      path[1] = (om->xf[BO_XE][BO_LOOP] == 0.0f) ? NINF : ox[i*5+XE];
      s1 = (path[0] > path[1]) ? BO_T_J : BO_T_E; break;
    case BO_T_E: {
      const float *c = oa + (size_t) i * W;
      float mx = NINF; int smax = -1, kmax = -1;
      for (int q = 0; q < Q; q++) {
        for (int r = 0; r < 4; r++) { const int kk = r * Q + q + 1; if (kk <= M && c[kk*3+cM] >= mx) { mx = c[kk*3+cM]; smax = BO_T_M; kmax = kk; } }
        for (int r = 0; r < 4; r++) { const int kk = r * Q + q + 1; if (kk <= M && c[kk*3+cD] >  mx) { mx = c[kk*3+cD]; smax = BO_T_D; kmax = kk; } }
      }
      k = kmax; s1 = smax; break; }
    case BO_T_B:
      path[0] = (om->xf[BO_XN][BO_MOVE] == 0.0f) ? NINF : ox[i*5+XN];
      path[1] = (om->xf[BO_XJ][BO_MOVE] == 0.0f) ? NINF : ox[i*5+XJ];
      s1 = (path[0] > path[1]) ? BO_T_N : BO_T_J; break;
    default: return BO_EINVAL;
    }
    if (s1 == -1 || i < 0 || k < 0) return BO_EINVAL;
    if (s1 == BO_T_M) { if (*i2 < 0) { *i2 = i; *k2 = k; } *i1 = i; *k1 = k; }   /* walking backwards: last M first */
    if ((s1 == BO_T_M || s1 == BO_T_D || s1 == BO_T_I) && *i2 >= 0) {            /* the alignment's states, last to first */
      path_st[*path_n] = s1; path_k[*path_n] = k; path_i[*path_n] = i; (*path_n)++;
    }
    if ((s1 == BO_T_N || s1 == BO_T_J || s1 == BO_T_C) && s1 == s0) i--;
    s0 = s1;
    if (++n > 4 * (L + M) + 64) return BO_EINVAL;
  }
#undef PATH
  (void) pp;
  return BO_OK;
}

/* p7_Null2_ByExpectation, null2.c:50-124 (insert odds implicitly 1) + esl_abc_FAvgScVec for the degenerate codes */
static void null2_by_expectation(const bo_oprofile *om, int Ld, const float *pp, const float *ppx, float *null2)
{
  const int M = om->M;
  const size_t W = (size_t)(M + 1) * 3;
  float *em = calloc((size_t)(M + 1) * 2, sizeof(float));      /* expected use of M_k, I_k */
  float xN = ppx[1*5+XN], xC = ppx[1*5+XC], xJ = ppx[1*5+XJ];
  for (int k = 1; k <= M; k++) { em[k*2] = pp[W + k*3+cM]; em[k*2+1] = pp[W + k*3+cI]; }
  for (int i = 2; i <= Ld; i++) {
    const float *r = pp + (size_t) i * W;
    for (int k = 1; k <= M; k++) { em[k*2] = r[k*3+cM] + em[k*2]; em[k*2+1] = r[k*3+cI] + em[k*2+1]; }
    xN += ppx[i*5+XN]; xC += ppx[i*5+XC]; xJ += ppx[i*5+XJ];
  }
  const float norm = (float)(1.0 / (float) Ld);
  for (int k = 1; k <= M; k++) { em[k*2] *= norm; em[k*2+1] *= norm; }
  xN *= norm; xC *= norm; xJ *= norm;
  const float xfactor = xN + xC + xJ;
  for (int x = 0; x < BO_K_AMINO; x++) {
    const float *rf = om->rf + (size_t) x * (M + 1);
    float sv = 0.f;
    for (int k = 1; k <= M; k++) { sv += em[k*2] * rf[k]; sv += em[k*2+1]; }
    null2[x] = sv + xfactor;
  }
  free(em);
  /* esl_abc_FAvgScVec: a degenerate code scores the plain average of its members */
  for (int x = BO_K_AMINO + 1; x <= BO_KP_AMINO - 3; x++) {
    float sum = 0.f; int cnt = 0;
    for (int y = 0; y < BO_K_AMINO; y++) if (bo_amino_degen(x, y)) { sum += null2[y]; cnt++; }
    null2[x] = cnt ? sum / (float) cnt : 0.f;
  }
  null2[BO_K_AMINO] = 1.0f; null2[BO_KP_AMINO - 2] = 1.0f; null2[BO_KP_AMINO - 1] = 1.0f;
}

static void dom_push(bo_fsdomain **d, int *n, int *alloc, const bo_fsdomain *r)
{
  if (*n == *alloc) { *alloc = *alloc ? *alloc * 2 : 16; *d = realloc(*d, sizeof(bo_fsdomain) * (size_t) *alloc); }
  (*d)[(*n)++] = *r;
}

/* rescore_isolated_domain_bath, p7_domaindef.c:1194-1325: envelope i..j of dsq[1..n]; n2sc != NULL: null2_is_done (the region
 * went through stochastic-trace clustering and its per-residue null2 scores are already there) */
/* p7_pli_computeAliScores_BATH, p7_pipeline.c:781-979, for an amino trace (every match state has a 3-nt codon): sum over the
 * aligned columns, first to last match state, of the emission score of the codon's amino acid (X when the codon holds a
 * degenerate nucleotide: p7P_DEGEN5_C) plus the transition that entered the state.  Quirk kept: the last match state gets
 * no MM transition (the inner loops stop at z1 < z2, :899).  path_*: the columns, last to first; deg[a]: codon of ORF
 * residue a (1-based) holds a degenerate nucleotide. */
int bo_aliscore_drops = 0;
float bo_aliscore_min = 1e30f;             /* test hook: smallest aliscore seen */                /* test hook: envelopes dropped by the aliscore < 0 rule (both branches) */

static float std_aliscore(const bo_oprofile *om, const uint8_t *dsq, int off, const int *st, const int *pk, const int *pi, int n,
                          const uint8_t *strand_dsq, int orf_start)
{
  const size_t W = (size_t) om->M + 1;
  float total = 0.0f;
  int prev = BO_T_B;
  for (int z = n - 1; z >= 0; z--) {
    const int k = pk[z];
    float sc;
    if (st[z] == BO_T_M) {
      const int a = pi[z] + off;                                  /* ORF residue */
      int amino = dsq[a];
      if (strand_dsq) {
        const int p = orf_start + 3 * (a - 1);
        if (strand_dsq[p] >= 4 || strand_dsq[p + 1] >= 4 || strand_dsq[p + 2] >= 4) amino = 26;   /* codons[k][p7P_DEGEN5_C] = X */
      }
      sc = om->msc[(size_t) amino * W + k];
      if      (prev == BO_T_I) sc += om->tsc[(k - 1) * BO_NTRANS + BO_IM];
      else if (prev == BO_T_D) sc += om->tsc[(k - 1) * BO_NTRANS + BO_DM];
      else if (prev == BO_T_M && z > 0) sc += om->tsc[(k - 1) * BO_NTRANS + BO_MM];
    } else if (st[z] == BO_T_I) sc = om->tsc[k * BO_NTRANS + (prev == BO_T_I ? BO_II : BO_MI)];
    else                        sc = om->tsc[(k - 1) * BO_NTRANS + (prev == BO_T_D ? BO_DD : BO_MD)];
    total += sc;
    prev = st[z];
  }
  return total;
}

static void rescore_envelope(bo_oprofile *om, const uint8_t *dsq, int i, int j, int orf_start, int win_start, const float *n2sc,
                             bo_fsdomain **doms, int *ndom, int *dalloc, const uint8_t *strand_dsq)
{
  const int M = om->M;
  const int Ld = j - i + 1;
  const size_t W = (size_t)(M + 1) * 3;
  bo_oprofile_reconfig_length(om, Ld);
  float *fwd = calloc((size_t)(Ld + 1) * W, sizeof(float)), *bck = calloc((size_t)(Ld + 1) * W, sizeof(float));
  float *efx = calloc((size_t)(Ld + 1) * 6, sizeof(float)), *ebx = calloc((size_t)(Ld + 1) * 6, sizeof(float));
  float *ppx = calloc((size_t)(Ld + 1) * 5, sizeof(float)), *oax = calloc((size_t)(Ld + 1) * 5, sizeof(float));
  float envsc, bcksc;
  int eown = 0;
  bo_forward_full(dsq + i - 1, Ld, om, fwd, efx, &envsc);
  bo_backward_full(dsq + i - 1, Ld, om, efx, bck, ebx, &bcksc, &eown);
  if (decoding(om, Ld, fwd, efx, bck, ebx, eown, ppx) != BO_ERANGE) {
    const float oasc = optimal_accuracy(om, Ld, bck, ppx, fwd, oax);          /* fwd now holds the OA matrix */
    int i1, k1, i2, k2, pn = 0;
    int *pst = malloc(sizeof(int) * 3 * (size_t)(Ld + M + 8)), *pk = pst + (Ld + M + 8), *pi = pk + (Ld + M + 8);
    int ok = oa_trace(om, Ld, bck, ppx, fwd, oax, &i1, &k1, &i2, &k2, pst, pk, pi, &pn) == BO_OK && i1 > 0;
    const float alisc = ok ? std_aliscore(om, dsq, i - 1, pst, pk, pi, pn, strand_dsq, orf_start) : 0.0f;
    if (ok && alisc < bo_aliscore_min) bo_aliscore_min = alisc;
    if (ok && alisc < 0.0f) { ok = 0; bo_aliscore_drops++; }   /* p7_domaindef.c:1286: "repetitive garbage" */
    if (ok) {
      float domcorrection = 0.f;
      if (!n2sc) {
        float null2[BO_KP_AMINO];
        null2_by_expectation(om, Ld, bck, ppx, null2);
        for (int pos = i; pos <= j; pos++) domcorrection += logf(null2[dsq[pos]]);
      } else for (int pos = i; pos <= j; pos++) domcorrection += n2sc[pos];
      bo_fsdomain d;
      memset(&d, 0, sizeof d);
      /* trace coordinates -> ORF -> window nucleotides (p7_trace_fs_Convert: the codon's last nucleotide) */
      const int start = orf_start - win_start;
      const int a1 = i1 + i - 1, a2 = i2 + i - 1;
      d.iali = start + a1 * 3 - 2; d.jali = start + a2 * 3;
      d.ienv = i; d.jenv = j; d.ihmm = k1; d.jhmm = k2;
      d.envsc = envsc; d.oasc = oasc; d.domcorrection = domcorrection > 0.f ? domcorrection : 0.f;
      {                                                            /* dom->tr after p7_trace_fs_Convert, p7_domaindef.c:1274-1277, :1330 */
        const int tix = bo_traces_push(pn, win_start, orf_start, 0);
        const bo_domtrace *t = bo_traces_get(tix);
        for (int z = 0; z < pn; z++) {                             /* pst: last to first */
          const int q = pn - 1 - z, a = pi[z] + i - 1;
          if (pst[z] == BO_T_M)      { t->st[q] = 1; t->k[q] = pk[z]; t->i[q] = start + a * 3; t->c[q] = 3; t->pp[q] = bck[(size_t) pi[z] * W + pk[z] * 3 + cM]; }
          else if (pst[z] == BO_T_I) { t->st[q] = 3; t->k[q] = pk[z]; t->i[q] = start + a * 3; t->c[q] = 0; t->pp[q] = bck[(size_t) pi[z] * W + pk[z] * 3 + cI]; }
          else                       { t->st[q] = 2; t->k[q] = pk[z]; t->i[q] = 0;             t->c[q] = 0; t->pp[q] = 0.0f; }
        }
        d.trace_idx = tix;
      }
      dom_push(doms, ndom, dalloc, &d);
    }
    free(pst);
  }
  free(fwd); free(bck); free(efx); free(ebx); free(ppx); free(oax);
}

/* Test hook: the optimal-accuracy trace of one envelope, the calls of rescore_isolated_domain_bath (p7_domaindef.c:1206-1262):
 * p7_Forward, p7_Backward, p7_Decoding, p7_OptimalAccuracy, p7_OATrace on dsq[1..L] in the unihit configuration of length L.
 * path_*: the trace's M/D/I columns from the LAST to the first (oa_trace's order); returns their number or -1. */
int bo_std_envelope_trace(bo_oprofile *om, const uint8_t *dsq, int L, int *path_st, int *path_k, int *path_i, float *oasc_out)
{
  const int M = om->M;
  const size_t W = (size_t)(M + 1) * 3;
  bo_oprofile_reconfig_unihit(om, L);
  float *fwd = calloc((size_t)(L + 1) * W, sizeof(float)), *bck = calloc((size_t)(L + 1) * W, sizeof(float));
  float *efx = calloc((size_t)(L + 1) * 6, sizeof(float)), *ebx = calloc((size_t)(L + 1) * 6, sizeof(float));
  float *ppx = calloc((size_t)(L + 1) * 5, sizeof(float)), *oax = calloc((size_t)(L + 1) * 5, sizeof(float));
  float envsc, bcksc;
  int eown = 0, pn = -1;
  bo_forward_full(dsq, L, om, fwd, efx, &envsc);
  bo_backward_full(dsq, L, om, efx, bck, ebx, &bcksc, &eown);
  if (decoding(om, L, fwd, efx, bck, ebx, eown, ppx) != BO_ERANGE) {
    const float oasc = optimal_accuracy(om, L, bck, ppx, fwd, oax);
    int i1, k1, i2, k2;
    if (oasc_out) *oasc_out = oasc;
    pn = 0;
    if (oa_trace(om, L, bck, ppx, fwd, oax, &i1, &k1, &i2, &k2, path_st, path_k, path_i, &pn) != BO_OK) pn = -1;
  }
  free(fwd); free(bck); free(efx); free(ebx); free(ppx); free(oax);
  bo_oprofile_reconfig_multihit(om, L);
  return pn;
}

/* p7_domaindef_ByPosteriorHeuristics_BATH + p7_pli_postDomainDef_BATH for one ORF that passed the Forward filter.
 * dsq[1..n]: the ORF; orf_start: first nucleotide of the ORF on the strand being read; win_start: windowsq->start on
 * that strand (= orf_start in the plain pipeline, the DNA window's start in the frameshift pipeline's standard branch);
 * seq_n: length of the DNA sequence.  Appends bo_fsdomain records (nt coordinates on the sequence). */
int bo_domaindef_std(bo_pipeline *pli, bo_oprofile *om, bo_bg *bg, const uint8_t *dsq, int n, int orf_start, int win_start,
                     int complementarity, int seq_n, bo_fsdomain **doms, int *ndom, int *dalloc, int *nskipped, const uint8_t *strand_dsq)
{
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  const int M = om->M, first = *ndom;
  float *fx = calloc((size_t)(n + 1) * 6, sizeof(float)), *bx = calloc((size_t)(n + 1) * 6, sizeof(float));
  float fsc, bsc;
  int own = 0;
  bo_bg_setlength(bg, n);
  bo_oprofile_reconfig_multihit(om, n);                           /* the cascade's configuration, p7_pipeline.c:1644 */
  bo_forward_parser(dsq, n, om, fx, &fsc);
  bo_backward_full(dsq, n, om, fx, NULL, bx, &bsc, &own);
  float *btot = calloc((size_t) n + 2, sizeof(float)), *etot = calloc((size_t) n + 2, sizeof(float)), *mocc = calloc((size_t) n + 2, sizeof(float));
  bo_domain_decoding(om, fx, bx, n, own, btot, etot, mocc);
  free(fx); free(bx);
  bo_oprofile_reconfig_unihit(om, n);                             /* p7_domaindef.c:518 */

  int i = -1, triggered = 0;
  for (int j = 1; j <= n; j++) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j-1]) < rt2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= rt1) triggered = 1;
    } else if (mocc[j] - (etot[j] - etot[j-1]) < rt2) {
      float mx = -1.0f;                                           /* is_multidomain_region */
      for (int z = i; z <= j; z++) { const float a = etot[z] - etot[i-1], b = btot[j] - btot[z-1]; const float e = a < b ? a : b; if (e > mx) mx = e; }
      if (mx >= rt3) {                                            /* p7_domaindef.c:539-583: stochastic-trace clustering */
        const int Lr = j - i + 1;
        const size_t W = (size_t)(M + 1) * 3;
        float *rfwd = calloc((size_t)(Lr + 1) * W, sizeof(float)), *rfx = calloc((size_t)(Lr + 1) * 6, sizeof(float));
        float *n2sc = calloc((size_t) n + 2, sizeof(float)), rsc;
        int env[2 * 32];
        bo_oprofile_reconfig_multihit(om, n);                     /* saveL: the ORF's length */
        bo_forward_full(dsq + i - 1, Lr, om, rfwd, rfx, &rsc);
        const int nc = bo_region_trace_ensemble(om, dsq, i, j, rfwd, rfx, n2sc, env, 32);
        bo_oprofile_reconfig_unihit(om, n);
        (*nskipped)++;                                            /* counts clustered regions (ddef->nclustered) */
        for (int d = 0; d < nc; d++) rescore_envelope(om, dsq, env[2 * d], env[2 * d + 1], orf_start, win_start, n2sc, doms, ndom, dalloc, strand_dsq);
        free(rfwd); free(rfx); free(n2sc);
      } else rescore_envelope(om, dsq, i, j, orf_start, win_start, NULL, doms, ndom, dalloc, strand_dsq);
      i = -1; triggered = 0;
    }
  }
  free(btot); free(etot); free(mocc);
  bo_oprofile_reconfig_multihit(om, n);

  /* ---- p7_pli_postDomainDef_BATH, p7_pipeline.c:1172-1300; dnasq->start = 1 / seq_n, dnasq->end = seq_n / 1 */
  int kept = first;
  for (int q = first; q < *ndom; q++) {
    const int env_len = (*doms)[q].jenv - (*doms)[q].ienv + 1, ali_len = ((*doms)[q].jali - (*doms)[q].iali + 1) / 3;
    if (ali_len < 4) continue;                      /* :1194-1200: no hit for such a domain */
    bo_fsdomain *dm = &(*doms)[kept++];
    *dm = (*doms)[q];
    if (!complementarity) {
      dm->ienv = 1 + orf_start + dm->ienv * 3 - 4; dm->jenv = 1 + orf_start + dm->jenv * 3 - 2;
      dm->iali = 1 + win_start + dm->iali - 2;     dm->jali = 1 + win_start + dm->jali - 2;
    } else {                                        /* reference orfsq->start is the top-strand coordinate seq_n - orf_start + 1 */
      const int ostart_ref = seq_n - orf_start + 1;
      dm->ienv = 1 + ostart_ref - dm->ienv * 3 + 2; dm->jenv = 1 + ostart_ref - dm->jenv * 3;
      const int ja = seq_n - (win_start + dm->jali) + 2, ia = seq_n - (win_start + dm->iali) + 2;
      dm->jali = ja; dm->iali = ia;
    }
    const int ml = om->max_length;
    float bitscore = dm->envsc;
    bitscore -= 2 * log(2. / (env_len + 2));
    bitscore += 2 * log(2. / (ml + 2));
    bitscore -= (env_len - ali_len) * log((float) env_len / (float)(env_len + 2));
    bitscore += (ml - ali_len) * log((float) ml / (float)(ml + 2));
    const float dom_bias = pli->do_null2 ? bo_flogsum(0.0f, (float)(log(1. / 256.) + dm->domcorrection)) : 0.0f;   /* :1230-1233 */
    bo_bg_setlength(bg, ml);
    const float nullsc = bo_bg_nullone(bg, ml);
    const float dom_score = (float)((bitscore - (nullsc + dom_bias)) / LOG2C);
    const float lnP = (float) bo_exp_logsurv(dom_score, om->evparam[BO_FTAU], om->evparam[BO_FLAMBDA]);
    const double Z = (double)(float)((float) pli->nres / (float) ml);
    dm->dombias = dom_bias; dm->bitscore = dom_score; dm->lnP = lnP; dm->pre_score = (float)(bitscore / LOG2C);
    dm->reported = (pli->inc_by_E ? (exp(lnP) * Z <= pli->E) : (dom_score >= pli->T)) ? 1 : 0;          /* :1247-1248 */
  }
  *ndom = kept;
  return BO_OK;
}
