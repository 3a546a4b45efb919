/* stotrace.c -- ORACLE (test infrastructure): multi-domain regions of the standard branch, resolved by clustering an
 * ensemble of stochastic tracebacks.
 *
 * Restates, in plain scalar C on the oracle's unstriped odds-ratio matrices:
 *   region_trace_ensemble                   src/p7_domaindef.c:766-850
 *   p7_StochasticTrace, select_{m,d,i,n,c,j,e,b}   src/impl_sse/stotrace.c:71-300
 *   p7_trace_Index                          src/p7_trace.c:2592-2625 (domain segments B..E of a trace)
 *   p7_Null2_ByTrace                        src/impl_sse/null2.c:131-215 (inserts counted in the match slot, as there)
 *   p7_spensemble_Add / _Cluster, link_spsamples   src/p7_spensemble.c:100-170, 189-217, 300-440
 *   the parameters of p7_domaindef_Create_BATH     src/p7_domaindef.c:83-97
 *
 * PARITY UNPINNED for this file.  It depends on easel routines that are absent from /root/reference (easel is an un-pinned
 * submodule, branch BATH): esl_randomness_CreateFast / esl_random (the "fast" linear congruential generator, re-seeded with
 * 42 before every region, p7_pipeline.c:135-143), esl_rnd_FChoose, esl_vec_FNorm, esl_cluster_SingleLinkage.  They are
 * restated here from easel's published algorithms (esl_random.c: x = 69069 x + 1 on a Jenkins-mixed seed, u = x / 2^32;
 * FChoose: first index whose cumulative float sum exceeds the roll; single linkage = connected components of the link
 * relation).  None of the reference's recorded outputs contains a hit from a clustered region, so nothing pins the sampled
 * ensemble itself; what the tests can and do check is this restatement against the GPU path and its invariants.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

enum { XE = 0, XN, XJ, XB, XC, XS };
enum { cM = 0, cD = 1, cI = 2 };
#define MINI(a, b) ((a) < (b) ? (a) : (b))
#define MAXI(a, b) ((a) > (b) ? (a) : (b))

/* ---- easel's "fast" generator, esl_random.c */
static uint32_t jenkins_mix3(uint32_t a, uint32_t b, uint32_t c)
{
  a -= b; a -= c; a ^= (c >> 13);
  b -= c; b -= a; b ^= (a << 8);
  c -= a; c -= b; c ^= (b >> 13);
  a -= b; a -= c; a ^= (c >> 12);
  b -= c; b -= a; b ^= (a << 16);
  c -= a; c -= b; c ^= (b >> 5);
  a -= b; a -= c; a ^= (c >> 3);
  b -= c; b -= a; b ^= (a << 10);
  c -= a; c -= b; c ^= (b >> 15);
  return c;
}
/* pli->r / ddef->do_reseeding (p7_pipeline.c:140-143; p7_domaindef.c:781, :904): with a seed every region's ensemble starts
 * from it; with seed 0 the generator is seeded once (the reference takes the time of day: "arbitrary") and runs on. */
static uint32_t g_seed = 42;
static bo_rng g_running;
static int g_running_init = 0;
void bo_set_seed(uint32_t seed) { if (seed != g_seed) g_running_init = 0; g_seed = seed; }
static void region_rng(bo_rng *r)
{
  if (g_seed != 0) { bo_rng_init(r, g_seed); return; }
  if (!g_running_init) { bo_rng_init(&g_running, 20260000u); g_running_init = 1; }
  *r = g_running;
}
static void region_rng_done(const bo_rng *r) { if (g_seed == 0) g_running = *r; }
void bo_rng_init(bo_rng *r, uint32_t seed) { r->x = jenkins_mix3(seed, 87654321, 12345678); if (r->x == 0) r->x = 42; }
double bo_rng_next(bo_rng *r) { r->x *= 69069; r->x += 1; return (double) r->x / 4294967296.0; }

static void fnorm(float *p, int n)                       /* esl_vec_FNorm: compensated sum, then divide */
{
  float sum = 0.f, c = 0.f;
  for (int x = 0; x < n; x++) { float y = p[x] - c; float t = sum + y; c = (t - sum) - y; sum = t; }
  if (sum != 0.0f) for (int x = 0; x < n; x++) p[x] /= sum;
  else             for (int x = 0; x < n; x++) p[x] = 1.0f / (float) n;
}
static int fchoose(bo_rng *r, const float *p, int n)     /* esl_rnd_FChoose */
{
  for (;;) {
    float roll = (float) bo_rng_next(r), sum = 0.0f;
    for (int i = 0; i < n; i++) { sum += p[i]; if (roll < sum) return i; }
  }
}

/* p7_StochasticTrace: one sampled path through the Forward matrix, returned first state first.
 * fwd: (L+1) x (M+1) x {M,D,I}; fx: (L+1) x {E,N,J,B,C,SCALE}.  st/k/i must hold 2L + M + 8 entries... callers size it. */
int bo_stochastic_trace(bo_rng *rng, int L, const bo_oprofile *om, const float *fwd, const float *fx, int8_t *st, int32_t *tk, int32_t *ti, int cap)
{
  const int M = om->M, Q = ((M - 1) / 4) + 1 > 2 ? ((M - 1) / 4) + 1 : 2;
  const size_t W = (size_t)(M + 1) * 3;
  const float *tf = om->tf;
  int n = 0, i = L, k = 0, s0;
#define PUSH(s) do { if (n >= cap) return -1; st[n] = (int8_t)(s); tk[n] = k; ti[n] = i; n++; } while (0)
  PUSH(BO_T_T); PUSH(BO_T_C);
  s0 = BO_T_C;
  while (s0 != BO_T_S) {
    int s1 = -1;
    float path[4];
    switch (s0) {
    case BO_T_M: {
      const float *t = tf + k * BO_NTRANS, *pr = fwd + (size_t)(i - 1) * W;
      static const int state[4] = { BO_T_B, BO_T_M, BO_T_I, BO_T_D };
      path[0] = fx[(i-1)*6+XB] * t[BO_BM]; path[1] = pr[(k-1)*3+cM] * t[BO_MM]; path[2] = pr[(k-1)*3+cI] * t[BO_IM]; path[3] = pr[(k-1)*3+cD] * t[BO_DM];
      fnorm(path, 4);
      s1 = state[fchoose(rng, path, 4)]; k--; i--; break; }
    case BO_T_D: {
      const float *c = fwd + (size_t) i * W;
      path[0] = (k - 1 >= 1) ? c[(k-1)*3+cM] * tf[(k-1) * BO_NTRANS + BO_MD] : 0.f;
      path[1] = (k - 1 >= 1) ? c[(k-1)*3+cD] * tf[(k-1) * BO_NTRANS + BO_DD] : 0.f;
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? BO_T_M : BO_T_D; k--; break; }
    case BO_T_I: {
      const float *pr = fwd + (size_t)(i - 1) * W;
      path[0] = pr[k*3+cM] * tf[k * BO_NTRANS + BO_MI]; path[1] = pr[k*3+cI] * tf[k * BO_NTRANS + BO_II];
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? BO_T_M : BO_T_I; i--; break; }
    case BO_T_N: s1 = (i == 0) ? BO_T_S : BO_T_N; break;
    case BO_T_C:
      path[0] = fx[(i-1)*6+XC] * om->xf[BO_XC][BO_LOOP];
      path[1] = fx[i*6+XE] * om->xf[BO_XE][BO_MOVE] * fx[i*6+XS];
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? BO_T_C : BO_T_E; break;
    case BO_T_J:
      path[0] = fx[(i-1)*6+XJ] * om->xf[BO_XJ][BO_LOOP];
      path[1] = fx[i*6+XE] * om->xf[BO_XE][BO_LOOP] * fx[i*6+XS];
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? BO_T_J : BO_T_E; break;
    case BO_T_E: {                                        /* on-the-fly FChoose in double, cells in striped order (stotrace.c:262-281) */
      const float *c = fwd + (size_t) i * W;
      double sum = 0.0, roll = bo_rng_next(rng);
      const float norm = (float)(1.0 / fx[i*6+XE]);
      int guard = 0;
      while (s1 < 0 && guard++ < 4) {
        for (int q = 0; q < Q && s1 < 0; q++) {
          for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += (kk <= M) ? c[kk*3+cM] * norm : 0.0f; if (roll < sum) { k = kk; s1 = BO_T_M; } }
          for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += (kk <= M) ? c[kk*3+cD] * norm : 0.0f; if (roll < sum) { k = kk; s1 = BO_T_D; } }
        }
      }
      if (s1 < 0) return -1;
      break; }
    case BO_T_B:
      path[0] = fx[i*6+XN] * om->xf[BO_XN][BO_MOVE];
      path[1] = fx[i*6+XJ] * om->xf[BO_XJ][BO_MOVE];
      fnorm(path, 2);
      s1 = fchoose(rng, path, 2) == 0 ? BO_T_N : BO_T_J; break;
    default: return -1;
    }
    if (s1 < 0 || i < 0 || k < 0) return -1;
    PUSH(s1);
    if ((s1 == BO_T_N || s1 == BO_T_J || s1 == BO_T_C) && s1 == s0) i--;
    s0 = s1;
  }
#undef PUSH
  for (int a = 0, b = n - 1; a < b; a++, b--) {           /* p7_trace_Reverse */
    int8_t s = st[a]; st[a] = st[b]; st[b] = s;
    int32_t x = tk[a]; tk[a] = tk[b]; tk[b] = x;
    x = ti[a]; ti[a] = ti[b]; ti[b] = x;
  }
  return n;
}

typedef struct { int idx, i, j, k, m; float prob; } spcoord;

static int linked(const spcoord *h1, const spcoord *h2, float min_overlap, int max_diagdiff, int fs)      /* link_spsamples[_fs], of_smaller = TRUE */
{
  int nov = (h1->j < h2->j ? h1->j : h2->j) - (h1->i > h2->i ? h1->i : h2->i) + 1;
  int l1 = h1->j - h1->i + 1, l2 = h2->j - h2->i + 1, n = l1 < l2 ? l1 : l2;
  if ((float) nov / (float) n < min_overlap) return 0;
  nov = (h1->m < h2->m ? h1->m : h2->m) - (h1->k > h2->k ? h1->k : h2->k);
  l1 = h1->m - h1->k + 1; l2 = h2->m - h2->k + 1; n = l1 < l2 ? l1 : l2;
  if ((float) nov / (float) n < min_overlap) return 0;
  if (fs) {                                               /* nucleotide coordinates: diagonals in codons (p7_spensemble.c:249-250) */
    if (abs(((h1->i / 3) - h1->k) - ((h2->i / 3) - h2->k)) <= max_diagdiff) return 1;
    if (abs(((h1->j / 3) - h1->m) - ((h2->j / 3) - h2->m)) <= max_diagdiff) return 1;
    return 0;
  }
  if (abs((h1->i - h1->k) - (h2->i - h2->k)) <= max_diagdiff) return 1;
  if (abs((h1->j - h1->m) - (h2->j - h2->m)) <= max_diagdiff) return 1;
  return 0;
}

static int argmax_i(const int *v, int n) { int b = 0; for (int x = 1; x < n; x++) if (v[x] > v[b]) b = x; return b; }
static int sp_cmp(const void *a, const void *b) { const spcoord *x = a, *y = b; return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0); }

/* p7_spensemble_Cluster / p7_spensemble_fs_Cluster (p7_spensemble.c:300-440, 500-640) and the removal of dominated clusters
 * (p7_domaindef.c:815-843, 923-953).  Takes ownership of sp.  span: an upper bound on any coordinate range. */
static int cluster_segments(spcoord *sp, int nsp, int nsamples, int span, int fs, int *env, int max_env)
{
  const int max_diagdiff = 4;
  const float min_overlap = 0.8f, min_posterior = 0.25f, min_endpointp = 0.02f;
  /* ---- p7_spensemble_Cluster: single linkage = connected components */
  int *assign = malloc(sizeof(int) * (size_t)(nsp + 1)), *stack = malloc(sizeof(int) * (size_t)(nsp + 1));
  int nc = 0;
  for (int h = 0; h < nsp; h++) assign[h] = -1;
  for (int h = 0; h < nsp; h++) {
    if (assign[h] >= 0) continue;
    int top = 0;
    stack[top++] = h; assign[h] = nc;
    while (top) {
      const int a = stack[--top];
      for (int b = 0; b < nsp; b++) if (assign[b] < 0 && linked(&sp[a], &sp[b], min_overlap, max_diagdiff, fs)) { assign[b] = nc; stack[top++] = b; }
    }
    nc++;
  }
  spcoord *sig = malloc(sizeof(spcoord) * (size_t)(nc + 1));
  int nsig = 0;
  int *epc = malloc(sizeof(int) * (size_t) span);
  for (int c = 0; c < nc; c++) {
    int ninc = 0, last = -1;
    for (int h = 0; h < nsp; h++) if (assign[h] == c) { if (sp[h].idx != last) ninc++; last = sp[h].idx; }
    if ((float) ninc / (float) nsamples < min_posterior) continue;
    int imin = 0, imax = 0, jmin = 0, jmax = 0, kmin = 0, kmax = 0, mmin = 0, mmax = 0;
    for (int h = 0; h < nsp; h++) if (assign[h] == c) {
      if (imin == 0) { imin = imax = sp[h].i; jmin = jmax = sp[h].j; kmin = kmax = sp[h].k; mmin = mmax = sp[h].m; }
      else {
        imin = MINI(imin, sp[h].i); imax = MAXI(imax, sp[h].i);
        jmin = MINI(jmin, sp[h].j); jmax = MAXI(jmax, sp[h].j);
        kmin = MINI(kmin, sp[h].k); kmax = MAXI(kmax, sp[h].k);
        mmin = MINI(mmin, sp[h].m); mmax = MAXI(mmax, sp[h].m);
      }
    }
    const int thr = (int) ceilf((float) ninc * min_endpointp);
    int best_i, best_j, best_k, best_m;
#define COUNT(field, lo, hi) do { for (int x = 0; x <= (hi) - (lo); x++) epc[x] = 0; for (int h = 0; h < nsp; h++) if (assign[h] == c) epc[sp[h].field - (lo)]++; } while (0)
    COUNT(i, imin, imax); for (best_i = imin; best_i <= imax; best_i++) if (epc[best_i - imin] >= thr) break;
    if (best_i > imax) best_i = imin + argmax_i(epc, imax - imin + 1);
    COUNT(k, kmin, kmax); for (best_k = kmin; best_k <= kmax; best_k++) if (epc[best_k - kmin] >= thr) break;
    if (best_k > kmax) best_k = kmin + argmax_i(epc, kmax - kmin + 1);
    COUNT(j, jmin, jmax); for (best_j = jmax; best_j >= jmin; best_j--) if (epc[best_j - jmin] >= thr) break;
    if (best_j < jmin) best_j = jmin + argmax_i(epc, jmax - jmin + 1);
    COUNT(m, mmin, mmax); for (best_m = mmax; best_m >= mmin; best_m--) if (epc[best_m - mmin] >= thr) break;
    if (best_m < mmin) best_m = mmin + argmax_i(epc, mmax - mmin + 1);
#undef COUNT
    if (best_i > best_j || best_k > best_m) continue;
    sig[nsig].i = best_i; sig[nsig].j = best_j; sig[nsig].k = best_k; sig[nsig].m = best_m; sig[nsig].idx = c;
    sig[nsig].prob = (float) ninc / (float) nsamples; nsig++;
  }
  qsort(sig, (size_t) nsig, sizeof(spcoord), sp_cmp);
  free(epc); free(assign); free(stack); free(sp);

  /* ---- dominated clusters (p7_domaindef.c:815-843) */
  char *dominated = calloc((size_t) nsig + 1, 1);
  for (int d = 0; d < nsig; d++)
    for (int d2 = d + 1; d2 < nsig; d2++) {
      const int nov = (sig[d].j < sig[d2].j ? sig[d].j : sig[d2].j) - (sig[d].i > sig[d2].i ? sig[d].i : sig[d2].i) + 1;
      if (nov == 0) break;
      const int l1 = sig[d].j - sig[d].i + 1, l2 = sig[d2].j - sig[d2].i + 1, n = l1 < l2 ? l1 : l2;
      if ((float) nov / (float) n >= 0.8) { if (sig[d].prob > sig[d2].prob) dominated[d2] = 1; else dominated[d] = 1; }
    }
  int out = 0;
  for (int d = 0; d < nsig; d++) if (!dominated[d] && out < max_env) { env[2 * out] = sig[d].i; env[2 * out + 1] = sig[d].j; out++; }
  free(dominated); free(sig);
  return out;
}


/* test hook: the clustering step alone (the all-pairs single-linkage search above) on caller-supplied segments */
int bo_selftest_cluster_segments(int n, const int32_t *idx, const int32_t *i, const int32_t *j, const int32_t *k, const int32_t *m,
                                 int nsamples, int fs, int *env, int max_env)
{
  spcoord *sp = malloc(sizeof(spcoord) * (size_t)(n + 1));
  int span = 8;
  for (int h = 0; h < n; h++) {
    sp[h].idx = idx[h]; sp[h].i = i[h]; sp[h].j = j[h]; sp[h].k = k[h]; sp[h].m = m[h]; sp[h].prob = 0.f;
    if (j[h] + 8 > span) span = j[h] + 8;
    if (m[h] + 8 > span) span = m[h] + 8;
  }
  return cluster_segments(sp, n, nsamples, span, fs, env, max_env);
}

/* region_trace_ensemble, p7_domaindef.c:766-850.  dsq[1..n]: the ORF; region ireg..jreg; fwd/fx: p7_Forward of the region in
 * the multihit configuration of length saveL.  n2sc[ireg..jreg] receives the null2 log odds; env[2*c], env[2*c+1] the
 * envelopes of the surviving clusters (sequence coordinates of the ORF), ordered by start. */
int bo_region_trace_ensemble(const bo_oprofile *om, const uint8_t *dsq, int ireg, int jreg, const float *fwd, const float *fx,
                             float *n2sc, int *env, int max_env)
{
  const int nsamples = 200, M = om->M;
  const int Lr = jreg - ireg + 1;
  const int cap = 2 * Lr + M + 16;
  int8_t *st = malloc((size_t) cap);
  int32_t *tk = malloc(sizeof(int32_t) * (size_t) cap), *ti = malloc(sizeof(int32_t) * (size_t) cap);
  float *cnt = malloc(sizeof(float) * (size_t)(M + 1));
  spcoord *sp = NULL;
  int nsp = 0, sp_alloc = 0;
  bo_rng rng;
  region_rng(&rng);                                         /* do_reseeding: every region starts from the seed */
  for (int pos = ireg; pos <= jreg; pos++) n2sc[pos] = 0.f;
  for (int t = 0; t < nsamples; t++) {
    const int N = bo_stochastic_trace(&rng, Lr, om, fwd, fx, st, tk, ti, cap);
    if (N < 0) { free(st); free(tk); free(ti); free(cnt); free(sp); return -1; }
    int pos = 1, z = 0;
    while (z < N) {
      if (st[z] != BO_T_B) { z++; continue; }
      int zb = z, sqfrom = 0, sqto = 0, hmmfrom = 0, hmmto = 0;
      for (z = zb + 1; z < N && st[z] != BO_T_E; z++)
        if (st[z] == BO_T_M) { if (!sqfrom) sqfrom = ti[z]; if (!hmmfrom) hmmfrom = tk[z]; sqto = ti[z]; hmmto = tk[z]; }
      const int ze = z;
      if (nsp == sp_alloc) { sp_alloc = sp_alloc ? sp_alloc * 2 : 256; sp = realloc(sp, sizeof(spcoord) * (size_t) sp_alloc); }
      sp[nsp].idx = t; sp[nsp].i = sqfrom + ireg - 1; sp[nsp].j = sqto + ireg - 1; sp[nsp].k = hmmfrom; sp[nsp].m = hmmto; sp[nsp].prob = 0.f; nsp++;
      /* p7_Null2_ByTrace over zb..ze */
      float null2[BO_KP_AMINO];
      int Ld = 0;
      for (int k = 0; k <= M; k++) cnt[k] = 0.f;
      for (int y = zb; y <= ze; y++) if (st[y] == BO_T_M || st[y] == BO_T_I) { Ld++; cnt[tk[y]] += 1.0f; }
      const float norm = (float)(1.0 / (float) Ld);
      for (int k = 1; k <= M; k++) cnt[k] *= norm;
      for (int x = 0; x < BO_K_AMINO; x++) {
        const float *rf = om->rf + (size_t) x * (M + 1);
        float sv = 0.f;
        for (int k = 1; k <= M; k++) sv += cnt[k] * rf[k];
        null2[x] = sv + 0.0f;
      }
      for (int x = BO_K_AMINO + 1; x <= BO_KP_AMINO - 3; x++) {
        float sum = 0.f; int c = 0;
        for (int y = 0; y < BO_K_AMINO; y++) if (bo_amino_degen(x, y)) { sum += null2[y]; c++; }
        null2[x] = c ? sum / (float) c : 0.f;
      }
      null2[BO_K_AMINO] = 1.0f; null2[BO_KP_AMINO - 2] = 1.0f; null2[BO_KP_AMINO - 1] = 1.0f;
      for (; pos <= sqfrom; pos++) n2sc[ireg + pos - 1] += 1.0f;                 /* (sic: the first domain residue also gets +1) */
      for (; pos <= sqto; pos++)   n2sc[ireg + pos - 1] += null2[dsq[ireg + pos - 1]];
      z = ze + 1;
    }
    for (; pos <= Lr; pos++) n2sc[ireg + pos - 1] += 1.0f;
  }
  for (int pos = ireg; pos <= jreg; pos++) n2sc[pos] = logf(n2sc[pos] / (float) nsamples);
  free(st); free(tk); free(ti); free(cnt);
  region_rng_done(&rng);

  return cluster_segments(sp, nsp, nsamples, Lr + M + 4, 0, env, max_env);
}

/* ================================================================================================================
 * Frameshift branch: p7_GStochasticTrace_Frameshift (src/generic_stotrace_frameshift.c:40-215) on the generic log-space
 * Forward matrix, region_trace_ensemble_frameshift (src/p7_domaindef.c:891-958), p7_trace_fs_Index (src/p7_trace.c:2645),
 * p7_spensemble_fs_Cluster.  (bathsearch itself samples from the SSE matrix, impl_sse/stotrace_fs.c, whose E-state choice
 * walks the cells in striped order; this oracle follows the generic code like the rest of its frameshift functions.)
 * ================================================================================================================ */
static void flognorm(float *v, int n)                     /* esl_vec_FLogNorm: FLogSum, subtract, exp, FNorm */
{
  float mx = v[0];
  for (int x = 1; x < n; x++) if (v[x] > mx) mx = v[x];
  float denom;
  if (mx == INFINITY) denom = INFINITY;
  else if (mx == -INFINITY) denom = -INFINITY;
  else { float sum = 0.f; for (int x = 0; x < n; x++) if (v[x] > mx - 50.f) sum += expf(v[x] - mx); denom = logf(sum) + mx; }
  for (int x = 0; x < n; x++) v[x] = expf(v[x] - denom);
  fnorm(v, n);
}

/* one sampled path; st/k/i/c first state first.  Returns the number of states or -1. */
static int stochastic_trace_fs(bo_rng *rng, int L, const bo_fs_profile *gm, const bo_gmx *gx, float *sc /* 2M+2 */, int8_t *st, int32_t *tk, int32_t *ti, int8_t *tc, int cap)
{
  const int M = gm->M;
  const float *tsc = gm->tsc;
#define TS(s, k) (tsc[(size_t)(k) * BO_NTRANS + (s)])
#define PUSHF(s) do { if (n >= cap) return -1; st[n] = (int8_t)(s); tk[n] = k; ti[n] = i; tc[n] = (int8_t) c; n++; } while (0)
  int n = 0, i = L, k = 0, c = 0, sprv = BO_T_C;
  PUSHF(BO_T_T); PUSHF(BO_T_C);
  while (sprv != BO_T_S) {
    int scur = -1;
    switch (sprv) {
    case BO_T_C:
      if (BO_X(gx, i, BO_GC) == -INFINITY) return -1;
      if (i < 4) { scur = BO_T_E; break; }
      sc[0] = BO_X(gx, i - 3, BO_GC) + gm->xsc[BO_XC][BO_LOOP]; sc[1] = BO_X(gx, i - 2, BO_GC) + gm->xsc[BO_XC][BO_LOOP];
      sc[2] = BO_X(gx, i - 1, BO_GC) + gm->xsc[BO_XC][BO_LOOP]; sc[3] = BO_X(gx, i, BO_GE) + gm->xsc[BO_XE][BO_MOVE];
      flognorm(sc, 4);
      scur = fchoose(rng, sc, 4) < 3 ? BO_T_C : BO_T_E; break;
    case BO_T_E:
      if (BO_X(gx, i, BO_GE) == -INFINITY) return -1;
      sc[0] = sc[M + 1] = -INFINITY;
      for (k = 1; k <= M; k++) sc[k] = BO_DP(gx, i, k, BO_GM);
      for (k = 2; k <= M; k++) sc[k + M] = BO_DP(gx, i, k, BO_GD);
      flognorm(sc, 2 * M + 1);
      k = fchoose(rng, sc, 2 * M + 1);
      if (k <= M) scur = BO_T_M; else { k -= M; scur = BO_T_D; }
      break;
    case BO_T_M:
      sc[0] = BO_X(gx, i, BO_GB) + TS(BO_BM, k - 1); sc[1] = BO_DP(gx, i, k - 1, BO_GM) + TS(BO_MM, k - 1);
      sc[2] = BO_DP(gx, i, k - 1, BO_GI) + TS(BO_IM, k - 1); sc[3] = BO_DP(gx, i, k - 1, BO_GD) + TS(BO_DM, k - 1);
      flognorm(sc, 4);
      { static const int state[4] = { BO_T_B, BO_T_M, BO_T_I, BO_T_D }; scur = state[fchoose(rng, sc, 4)]; }
      k--; break;
    case BO_T_D:
      if (BO_DP(gx, i, k, BO_GD) == -INFINITY) return -1;
      sc[0] = BO_DP(gx, i, k - 1, BO_GM) + TS(BO_MD, k - 1); sc[1] = BO_DP(gx, i, k - 1, BO_GD) + TS(BO_DD, k - 1);
      flognorm(sc, 2);
      scur = fchoose(rng, sc, 2) == 0 ? BO_T_M : BO_T_D; k--; break;
    case BO_T_I:
      if (BO_DP(gx, i, k, BO_GI) == -INFINITY || i < 3) return -1;
      sc[0] = BO_DP(gx, i - 3, k, BO_GM) + TS(BO_MI, k); sc[1] = BO_DP(gx, i - 3, k, BO_GI) + TS(BO_II, k);
      flognorm(sc, 2);
      scur = fchoose(rng, sc, 2) == 0 ? BO_T_M : BO_T_I; i -= 3; break;
    case BO_T_N:
      if (BO_X(gx, i, BO_GN) == -INFINITY) return -1;
      scur = (i == 0) ? BO_T_S : BO_T_N; break;
    case BO_T_B:
      if (BO_X(gx, i, BO_GB) == -INFINITY) return -1;
      sc[0] = BO_X(gx, i, BO_GN) + gm->xsc[BO_XN][BO_MOVE]; sc[1] = BO_X(gx, i, BO_GJ) + gm->xsc[BO_XJ][BO_MOVE];
      flognorm(sc, 2);
      scur = fchoose(rng, sc, 2) == 0 ? BO_T_N : BO_T_J; break;
    case BO_T_J:
      if (BO_X(gx, i, BO_GJ) == -INFINITY) return -1;
      if (i < 4) { scur = BO_T_E; break; }
      sc[0] = BO_X(gx, i - 3, BO_GJ) + gm->xsc[BO_XJ][BO_LOOP]; sc[1] = BO_X(gx, i - 2, BO_GJ) + gm->xsc[BO_XJ][BO_LOOP];
      sc[2] = BO_X(gx, i - 1, BO_GJ) + gm->xsc[BO_XJ][BO_LOOP]; sc[3] = BO_X(gx, i, BO_GE) + gm->xsc[BO_XE][BO_LOOP];
      flognorm(sc, 4);
      scur = fchoose(rng, sc, 4) < 3 ? BO_T_J : BO_T_E; break;
    default: return -1;
    }
    if (scur == BO_T_M) {                                   /* codon length from the C1..C5 cells */
      for (int q = 0; q < 5; q++) sc[q] = BO_DP(gx, i, k, BO_GM + 1 + q);
      flognorm(sc, 5);
      c = fchoose(rng, sc, 5) + 1;
      if (i - c < 0) scur = BO_T_B;
    } else c = 0;
    if (scur < 0 || k < 0 || i < 0) return -1;
    PUSHF(scur);
    if ((scur == BO_T_N || scur == BO_T_C || scur == BO_T_J) && scur == sprv) i--;
    sprv = scur;
    i -= c;
    if (i < 0) return -1;
  }
#undef PUSHF
#undef TS
  for (int a = 0, b = n - 1; a < b; a++, b--) {
    int8_t s = st[a]; st[a] = st[b]; st[b] = s;
    s = tc[a]; tc[a] = tc[b]; tc[b] = s;
    int32_t x = tk[a]; tk[a] = tk[b]; tk[b] = x;
    x = ti[a]; ti[a] = ti[b]; ti[b] = x;
  }
  return n;
}

/* region_trace_ensemble_frameshift: fwd = p7_GForward_Frameshift of the region (multihit).  env: envelopes in window
 * coordinates (nucleotides), ordered by start.  Returns their number (0 when a traceback is impossible). */
int bo_region_trace_ensemble_fs(const bo_fs_profile *gm5, int ireg, int jreg, const bo_gmx *fwd, int *env, int max_env)
{
  const int nsamples = 200, M = gm5->M, Lr = jreg - ireg + 1;
  const int cap = 2 * Lr + M + 16;
  int8_t *st = malloc((size_t) cap), *tc = malloc((size_t) cap);
  int32_t *tk = malloc(sizeof(int32_t) * (size_t) cap), *ti = malloc(sizeof(int32_t) * (size_t) cap);
  float *sc = malloc(sizeof(float) * (size_t)(2 * M + 8));
  spcoord *sp = NULL;
  int nsp = 0, sp_alloc = 0, ok = 1;
  bo_rng rng;
  region_rng(&rng);
  for (int t = 0; t < nsamples && ok; t++) {
    const int N = stochastic_trace_fs(&rng, Lr, gm5, fwd, sc, st, tk, ti, tc, cap);
    if (N < 0) { ok = 0; break; }
    for (int z = 0; z < N; z++) {
      if (st[z] != BO_T_B) continue;
      int sqfrom = 0, sqto = 0, hmmfrom = 0, hmmto = 0;
      for (z = z + 1; z < N && st[z] != BO_T_E; z++)
        if (st[z] == BO_T_M) { if (!sqfrom) sqfrom = ti[z] - tc[z] + 1; if (!hmmfrom) hmmfrom = tk[z]; sqto = ti[z]; hmmto = tk[z]; }
      if (nsp == sp_alloc) { sp_alloc = sp_alloc ? sp_alloc * 2 : 256; sp = realloc(sp, sizeof(spcoord) * (size_t) sp_alloc); }
      sp[nsp].idx = t; sp[nsp].i = sqfrom + ireg - 1; sp[nsp].j = sqto + ireg - 1; sp[nsp].k = hmmfrom; sp[nsp].m = hmmto; sp[nsp].prob = 0.f; nsp++;
    }
  }
  free(st); free(tc); free(tk); free(ti); free(sc);
  region_rng_done(&rng);
  if (!ok) { free(sp); return 0; }
  return cluster_segments(sp, nsp, nsamples, jreg + M + 8, 1, env, max_env);
}
