/* sse_fs.c -- SSE2 (128-bit) striped, PROBABILITY-SPACE restatement of the reference's 3-codon frameshift parsers.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like everything under oracle/: the part of the "impl_sse-equivalent" CPU baseline that
 * covers the frameshift stage.  The reference's bathsearch --fs calls p7_ForwardParser_Frameshift_3Codons
 * (impl_sse/fwdback_fs.c:97-533) once per DNA window (p7_pipeline.c:1450): odds ratios instead of log-odds, four model nodes per
 * vector in HMMER's striped order (node k in vector (k-1) % Q, element (k-1) / Q: impl_sse.h:57-71), the D->D path serialised by
 * up to four passes over the row, every value of the recursion rescaled together when E(i) passes 1e4 (:472-495).  That file cannot
 * be built here (easel), so the same algorithm is written from scratch below; bench.py's fs / c5 cpu_baseline legs run the oracle's
 * --fs pipeline with it in place of the scalar log-space parser (bo_fs_use_sse).
 *
 *   bs_fsprofile_create   odds-ratio tables of a 3-codon bo_fs_profile, striped (after p7_fs_oprofile_Convert, p7_fs_oprofile.c:221)
 *   bs_fs3_forward_parser p7_ForwardParser_Frameshift_3Codons: score in nats; optionally the special-state rows, written in LOG
 *                         space (log value + the scale accumulated so far) so that the oracle's log-space domain decoding can read them
 *   bs_fs3_backward_parser p7_BackwardParser_Frameshift_3Codons (fwdback_fs.c:565-1050): the mirror image, rows L down to 0
 *
 * Recursion (generic_fwdback_frameshift.c:451-622 in odds ratios):
 *   IVX(i,k) = B(i-2) tBM(k-1) + M(i-2,k-1) tMM(k-1) + I(i-2,k-1) tIM(k-1) + D(i-2,k-1) tDM(k-1)
 *   M(i,k)   = IVX(i,k) e2(k) + IVX(i-1,k) e3(k) + IVX(i-2,k) e4(k)         e2/e3/e4: the rows of the last 2 / 3 / 4 nucleotides
 *   I(i,k)   = M(i-3,k) tMI(k) + I(i-3,k) tII(k)
 *   D(i,k)   = M(i,k-1) tMD(k-1) + D(i,k-1) tDD(k-1)
 *   E(i)     = sum_k M(i,k) + D(i,k);  N(i) = N(i-3) tNL;  J(i) = J(i-3) tJL + E(i) tEL;  C(i) = C(i-3) tCL + E(i) tEM;
 *   B(i)     = N(i) tNM + J(i) tJM;    score = log[(C(L) + C(L-1) tCL + C(L-2) tCL) tCM]
 *
 * Validated against the scalar oracle (tests/test_sse_cpu.py): the oracle's table log-sum is itself up to ~1e-3 nats from exact
 * arithmetic per window, which is the tolerance the reference's own test uses between its SSE and generic parsers when the table is
 * on (fwdback_fs.c:3189-3191: 1.0 with the table, 0.001 with exact log-sums); with the oracle switched to EXACT log-sums
 * (bo_logsum_exact) the two agree to 1e-4 relative.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_sse.h"

static void *amalloc16(size_t bytes)
{
  void *p = NULL;
  if (posix_memalign(&p, 16, bytes ? bytes : 16) != 0) return NULL;
  return p;
}

/* transitions of one stripe: what enters node k from node k-1 (BM, MM, IM, DM: tsc row k-1), what leaves node k (MD, DD, MI, II: row k) */
enum { FT_BM = 0, FT_MM, FT_IM, FT_DM, FT_MD, FT_MI, FT_II, FT_NSTRIPE };

bs_fsprofile *bs_fsprofile_create(const bo_fs_profile *gm)
{
  if (!gm || gm->codon_lengths != 3) return NULL;
  const int M = gm->M;
  const int Q = (M - 1) / 4 + 1 > 2 ? (M - 1) / 4 + 1 : 2;                  /* p7O_NQF */
  bs_fsprofile *so = calloc(1, sizeof *so);
  so->M = M; so->Q = Q; so->gm = gm; so->ncodons = BO_MAXCODONS3;
  so->rfv = amalloc16(sizeof(__m128) * (size_t) so->ncodons * Q);
  so->tfv = amalloc16(sizeof(__m128) * (size_t) Q * (FT_NSTRIPE + 1));
  float tmp[4];
  for (int c = 0; c < so->ncodons; c++) {
    const float *row = gm->rsc + (size_t) c * (M + 1);
    for (int q = 0; q < Q; q++) {
      for (int z = 0; z < 4; z++) { const int k = z * Q + q + 1; tmp[z] = (k <= M) ? expf(row[k]) : 0.0f; }
      so->rfv[(size_t) c * Q + q] = _mm_loadu_ps(tmp);
    }
  }
  const float *tsc = gm->tsc;
  for (int q = 0; q < Q; q++) {
    static const int from_prev[4] = { BO_BM, BO_MM, BO_IM, BO_DM };
    for (int t = 0; t < 4; t++) {
      for (int z = 0; z < 4; z++) { const int k = z * Q + q + 1; tmp[z] = (k <= M) ? expf(tsc[(size_t)(k - 1) * BO_NTRANS + from_prev[t]]) : 0.0f; }
      so->tfv[(size_t) q * FT_NSTRIPE + t] = _mm_loadu_ps(tmp);
    }
    static const int own[3] = { BO_MD, BO_MI, BO_II };
    for (int t = 0; t < 3; t++) {
      for (int z = 0; z < 4; z++) { const int k = z * Q + q + 1; tmp[z] = (k < M) ? expf(tsc[(size_t) k * BO_NTRANS + own[t]]) : 0.0f; }    /* node M has no I, feeds no D */
      so->tfv[(size_t) q * FT_NSTRIPE + 4 + t] = _mm_loadu_ps(tmp);
    }
    for (int z = 0; z < 4; z++) { const int k = z * Q + q + 1; tmp[z] = (k < M) ? expf(tsc[(size_t) k * BO_NTRANS + BO_DD]) : 0.0f; }
    so->tfv[(size_t) Q * FT_NSTRIPE + q] = _mm_loadu_ps(tmp);
  }
  so->rows = amalloc16(sizeof(__m128) * (size_t) Q * (4 * 3 + 3));           /* four rows of {M, D, I}, three rows of IVX */
  return so;
}

void bs_fsprofile_free(bs_fsprofile *so)
{
  if (!so) return;
  free(so->rfv); free(so->tfv); free(so->rows); free(so);
}

/* element z of every vector moves to z + 1, element 0 becomes 0: node k-1 of stripe 0 is node k of stripe Q-1 one element down */
static inline __m128 shift_up(__m128 a) { return _mm_castsi128_ps(_mm_slli_si128(_mm_castps_si128(a), 4)); }
static inline float hsum4(__m128 a)
{
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 3, 2, 1)));
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(1, 0, 3, 2)));
  float r; _mm_store_ss(&r, a); return r;
}
static inline int nuc(uint8_t d) { return d < 4 ? d : BO_MAXCODONS3; }
static inline int lower(int a, int b) { return a < b ? a : b; }

int bs_fs3_forward_parser(const uint8_t *dsq, int L, bs_fsprofile *so, float *xmx_log /* (L+1) x {E,N,J,B,C} or NULL */, float *ret_sc)
{
  const bo_fs_profile *gm = so->gm;
  const int Q = so->Q;
  if (L < 3) return BO_EINVAL;
  /* special-state transitions as probabilities: they follow the profile's CURRENT length configuration */
  const float *xs = &gm->xsc[0][0];
  const float tNL = expf(xs[BO_XN * 2 + BO_LOOP]), tNM = expf(xs[BO_XN * 2 + BO_MOVE]), tJL = expf(xs[BO_XJ * 2 + BO_LOOP]), tJM = expf(xs[BO_XJ * 2 + BO_MOVE]);
  const float tCL = expf(xs[BO_XC * 2 + BO_LOOP]), tCM = expf(xs[BO_XC * 2 + BO_MOVE]), tEL = expf(xs[BO_XE * 2 + BO_LOOP]), tEM = expf(xs[BO_XE * 2 + BO_MOVE]);
  __m128 *row[4], *iv[3];
  for (int r = 0; r < 4; r++) row[r] = so->rows + (size_t) r * 3 * Q;        /* row[r][q], [Q + q], [2Q + q] = M, D, I of stripe q */
  for (int r = 0; r < 3; r++) iv[r] = so->rows + (size_t)(12 + r) * Q;
  const __m128 zero = _mm_setzero_ps();
  for (int x = 0; x < 15 * Q; x++) so->rows[x] = zero;
  float N[4] = { 1.f, 1.f, 0.f, 0.f }, B[4] = { tNM, tNM, 0.f, 0.f }, J[4] = { 0.f, 0.f, 0.f, 0.f }, C[4] = { 0.f, 0.f, 0.f, 0.f };   /* rows i % 4 */
  double totscale = 0.0;
  if (xmx_log) for (int i = 0; i < 2; i++) { float *o = xmx_log + (size_t) i * BO_NXCELLS; o[BO_GE] = o[BO_GJ] = o[BO_GC] = -INFINITY; o[BO_GN] = 0.f; o[BO_GB] = logf(tNM); }
  int u, v = BO_MAXCODONS3, w = nuc(dsq[1]), x = nuc(dsq[2]);
  const __m128 *tdd = so->tfv + (size_t) Q * FT_NSTRIPE;
  for (int i = 2; i <= L; i++) {
    if (i > 2) { u = v; v = w; w = x; x = nuc(dsq[i]); } else u = BO_MAXCODONS3;
    const __m128 *e2 = so->rfv + (size_t) lower(x * 84 + w * 21, BO_DEGEN3_QC1) * Q;
    const __m128 *e3 = so->rfv + (size_t) lower(x * 84 + w * 21 + v * 5 + 1, BO_DEGEN3_C) * Q;
    const __m128 *e4 = so->rfv + (size_t) lower(x * 84 + w * 21 + v * 5 + u + 2, BO_DEGEN3_QC1) * Q;
    __m128 *cur = row[i & 3], *p2 = row[(i - 2) & 3], *p3 = row[(i + 1) & 3];            /* (i - 3) & 3 == (i + 1) & 3 */
    __m128 *iv0 = iv[i % 3], *iv1 = iv[(i + 2) % 3], *iv2 = iv[(i + 1) % 3];             /* IVX of rows i, i-1, i-2 */
    const int three = i >= 3, four = i >= 4;                                             /* row 2 has only the 2-nt quasi-codon (:484-517) */
    __m128 mp = shift_up(p2[Q - 1]), dp = shift_up(p2[2 * Q - 1]), ip = shift_up(p2[3 * Q - 1]);
    const __m128 b2 = _mm_set1_ps(B[(i - 2) & 3]);
    __m128 carry = zero, esum = zero;
    const __m128 *t = so->tfv;
    for (int q = 0; q < Q; q++, t += FT_NSTRIPE) {
      __m128 s = _mm_mul_ps(b2, t[FT_BM]);
      s = _mm_add_ps(s, _mm_mul_ps(mp, t[FT_MM]));
      s = _mm_add_ps(s, _mm_mul_ps(ip, t[FT_IM]));
      s = _mm_add_ps(s, _mm_mul_ps(dp, t[FT_DM]));
      iv0[q] = s;
      __m128 m = _mm_mul_ps(s, e2[q]);
      if (three) m = _mm_add_ps(m, _mm_mul_ps(iv1[q], e3[q]));
      if (four)  m = _mm_add_ps(m, _mm_mul_ps(iv2[q], e4[q]));
      esum = _mm_add_ps(esum, m);
      mp = p2[q]; dp = p2[Q + q]; ip = p2[2 * Q + q];
      cur[q] = m;
      cur[Q + q] = carry;                                                  /* M(i,k-1) tMD(k-1), the stripe before */
      carry = _mm_mul_ps(m, t[FT_MD]);
      cur[2 * Q + q] = three ? _mm_add_ps(_mm_mul_ps(p3[q], t[FT_MI]), _mm_mul_ps(p3[2 * Q + q], t[FT_II])) : zero;
    }
    /* D -> D: the carry out of stripe Q-1 enters stripe 0 one element up; a path of deletes crosses at most three such wraps */
    carry = shift_up(carry);
    for (int q = 0; q < Q; q++) { cur[Q + q] = _mm_add_ps(cur[Q + q], carry); carry = _mm_mul_ps(cur[Q + q], tdd[q]); }
    for (int pass = 1; pass < 4; pass++) {
      carry = shift_up(carry);
      __m128 grew = zero;
      for (int q = 0; q < Q; q++) {
        const __m128 s = _mm_add_ps(cur[Q + q], carry);
        grew = _mm_or_ps(grew, _mm_cmpgt_ps(s, cur[Q + q]));
        cur[Q + q] = s;
        carry = _mm_mul_ps(carry, tdd[q]);
      }
      if (!_mm_movemask_ps(grew)) break;                                   /* nothing left to add at this precision */
    }
    for (int q = 0; q < Q; q++) esum = _mm_add_ps(esum, cur[Q + q]);
    float xE = hsum4(esum);
    float xN = three ? N[(i + 1) & 3] * tNL : 1.0f;                        /* N(2) = 1 (:513) */
    float xJ = (three ? J[(i + 1) & 3] * tJL : 0.0f) + xE * tEL;
    float xC = (three ? C[(i + 1) & 3] * tCL : 0.0f) + xE * tEM;
    float xB = xN * tNM + xJ * tJM;
    if (xE > 1.0e4f) {                                                     /* everything a later row reads moves to the new scale together */
      const float f = 1.0f / xE;
      const __m128 fv = _mm_set1_ps(f);
      for (int a = 0; a < 15 * Q; a++) so->rows[a] = _mm_mul_ps(so->rows[a], fv);
      for (int r = 0; r < 4; r++) { N[r] *= f; B[r] *= f; J[r] *= f; C[r] *= f; }
      xN *= f; xJ *= f; xC *= f; xB *= f;
      totscale += log((double) xE);
      xE = 1.0f;
    }
    N[i & 3] = xN; B[i & 3] = xB; J[i & 3] = xJ; C[i & 3] = xC;
    if (xmx_log) {
      float *o = xmx_log + (size_t) i * BO_NXCELLS;
      const float ts = (float) totscale;
      o[BO_GE] = logf(xE) + ts; o[BO_GN] = logf(xN) + ts; o[BO_GJ] = logf(xJ) + ts; o[BO_GB] = logf(xB) + ts; o[BO_GC] = logf(xC) + ts;
    }
  }
  const float tot = C[L & 3] + C[(L - 1) & 3] * tCL + C[(L - 2) & 3] * tCL;
  if (isnan(tot) || isinf(tot)) { *ret_sc = -INFINITY; return BO_ERANGE; }
  if (tot == 0.0f) { *ret_sc = -INFINITY; return BO_ERANGE; }
  *ret_sc = (float)(totscale + log((double) tot * tCM));
  return BO_OK;
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * Backward (generic_fwdback_frameshift.c:1422-1737 in odds ratios; every row variant of the log-space code -- rows without a codon,
 * rows where fewer than three codon lengths fit, the main recursion -- is ONE formula once rows beyond L are zero and the emission
 * row of a codon that does not fit is zero):
 *   ivx(i,k) = M(i+2,k) e2(k) + M(i+3,k) e3(k) + M(i+4,k) e4(k)         e_c: the row of the c nucleotides x_{i+1} .. x_{i+c}
 *   B(i) = sum_k ivx(i,k) tBM(k-1);  J(i) = J(i+3) tJL + B(i) tJM;  N(i) = N(i+3) tNL + B(i) tNM;  C(i) = C(i+3) tCL
 *   (C(L) = tCM, C(L-1) = C(L-2) = tCL tCM);  E(i) = J(i) tEL + C(i) tEM
 *   D(i,k) = E(i) + D(i,k+1) tDD(k) + ivx(i,k+1) tDM(k)
 *   M(i,k) = E(i) + D(i,k+1) tMD(k) + I(i+3,k) tMI(k) + ivx(i,k+1) tMM(k);   I(i,k) = I(i+3,k) tII(k) + ivx(i,k+1) tIM(k)
 *   score = log[N(0) + N(1) + N(2)]
 * "Node k+1" of stripe q is stripe q+1, and stripe 0 one element up for the last stripe; the descending D chain is serialised like
 * Forward's, by passes that carry what crossed a wrap.  Rescaled (all ring rows and special values together) when B(i) passes 1e4.
 * ------------------------------------------------------------------------------------------------------------------------------ */
static inline __m128 shift_down(__m128 a) { return _mm_castsi128_ps(_mm_srli_si128(_mm_castps_si128(a), 4)); }

int bs_fs3_backward_parser(const uint8_t *dsq, int L, bs_fsprofile *so, float *xmx_log /* (L+1) x {E,N,J,B,C} or NULL */, float *ret_sc)
{
  const bo_fs_profile *gm = so->gm;
  const int Q = so->Q, M = so->M;
  if (L < 3) return BO_EINVAL;
  const float *xs = &gm->xsc[0][0];
  const float tNL = expf(xs[BO_XN * 2 + BO_LOOP]), tNM = expf(xs[BO_XN * 2 + BO_MOVE]), tJL = expf(xs[BO_XJ * 2 + BO_LOOP]), tJM = expf(xs[BO_XJ * 2 + BO_MOVE]);
  const float tCL = expf(xs[BO_XC * 2 + BO_LOOP]), tCM = expf(xs[BO_XC * 2 + BO_MOVE]), tEL = expf(xs[BO_XE * 2 + BO_LOOP]), tEM = expf(xs[BO_XE * 2 + BO_MOVE]);
  /* workspace: five ring rows of {M, D, I}, ivx, a zero emission row, the mask of real nodes */
  __m128 *ws = amalloc16(sizeof(__m128) * (size_t) Q * (5 * 3 + 3));
  if (!ws) return BO_EINVAL;
  const __m128 zero = _mm_setzero_ps();
  for (int a = 0; a < Q * 18; a++) ws[a] = zero;
  __m128 *ring[5];
  for (int r = 0; r < 5; r++) ring[r] = ws + (size_t) r * 3 * Q;             /* [q] M, [Q + q] D, [2Q + q] I */
  __m128 *ivx = ws + (size_t) 15 * Q, *zrow = ws + (size_t) 16 * Q, *mask = ws + (size_t) 17 * Q;
  { float tmp[4]; for (int q = 0; q < Q; q++) { for (int z = 0; z < 4; z++) tmp[z] = (z * Q + q + 1 <= M) ? 1.0f : 0.0f; mask[q] = _mm_loadu_ps(tmp); } }
  float N[5] = { 0, 0, 0, 0, 0 }, J[5] = { 0, 0, 0, 0, 0 }, Cc[5] = { 0, 0, 0, 0, 0 };                                  /* rows i % 5 */
  double totscale = 0.0;
  const __m128 *tdd = so->tfv + (size_t) Q * FT_NSTRIPE;
  int u = BO_MAXCODONS3, v = BO_MAXCODONS3, w = BO_MAXCODONS3, x = BO_MAXCODONS3;                                          /* x_{i+1} .. x_{i+4} as x, w, v, u */
  float n012[3] = { 0.f, 0.f, 0.f };
  for (int i = L; i >= 0; i--) {
    const int avail = L - i;
    if (avail >= 1) { u = v; v = w; w = x; x = nuc(dsq[i + 1]); }
    const __m128 *e2 = avail >= 2 ? so->rfv + (size_t) lower(w * 84 + x * 21, BO_DEGEN3_QC1) * Q : zrow;               /* the last nucleotide of the codon is the macro's last argument */
    const __m128 *e3 = avail >= 3 ? so->rfv + (size_t) lower(v * 84 + w * 21 + x * 5 + 1, BO_DEGEN3_C) * Q : zrow;
    const __m128 *e4 = avail >= 4 ? so->rfv + (size_t) lower(u * 84 + v * 21 + w * 5 + x + 2, BO_DEGEN3_QC1) * Q : zrow;
    __m128 *cur = ring[i % 5], *r2 = ring[(i + 2) % 5], *r3 = ring[(i + 3) % 5], *r4 = ring[(i + 4) % 5];
    __m128 bsum = zero;
    const __m128 *t = so->tfv;
    for (int q = 0; q < Q; q++, t += FT_NSTRIPE) {
      __m128 s = _mm_mul_ps(r2[q], e2[q]);
      s = _mm_add_ps(s, _mm_mul_ps(r3[q], e3[q]));
      s = _mm_add_ps(s, _mm_mul_ps(r4[q], e4[q]));
      ivx[q] = s;
      bsum = _mm_add_ps(bsum, _mm_mul_ps(s, t[FT_BM]));
    }
    float xB = hsum4(bsum);
    float xN = N[(i + 3) % 5] * tNL + xB * tNM;
    if (i == 0) {
      n012[0] = xN;
      if (xmx_log) { float *o = xmx_log; const float ts = (float) totscale; o[BO_GE] = o[BO_GJ] = o[BO_GC] = -INFINITY; o[BO_GN] = logf(xN) + ts; o[BO_GB] = logf(xB) + ts; }
      break;
    }
    float xJ = J[(i + 3) % 5] * tJL + xB * tJM;
    float xC = (i == L) ? tCM : (i >= L - 2 ? tCL * tCM : Cc[(i + 3) % 5] * tCL);
    float xE = xJ * tEL + xC * tEM;
    /* D(i,k), descending: what node k+1 hands down is in the next stripe, or in stripe 0 one element up */
    const __m128 ev = _mm_set1_ps(xE);
    const __m128 *tq = so->tfv;
    {
      __m128 carry = zero;
      for (int q = Q - 1; q >= 0; q--) {
        const __m128 nxt = (q < Q - 1) ? _mm_mul_ps(ivx[q + 1], tq[(size_t)(q + 1) * FT_NSTRIPE + FT_DM]) : shift_down(_mm_mul_ps(ivx[0], tq[FT_DM]));
        __m128 d = _mm_add_ps(_mm_mul_ps(ev, mask[q]), nxt);
        d = _mm_add_ps(d, _mm_mul_ps(carry, tdd[q]));
        cur[Q + q] = d;
        carry = d;
      }
      __m128 extra = shift_down(cur[Q]);                                   /* what stripe 0 hands to the last stripe, one element down */
      for (int pass = 1; pass < 4; pass++) {
        __m128 grew = zero;
        for (int q = Q - 1; q >= 0; q--) {
          extra = _mm_mul_ps(extra, tdd[q]);
          const __m128 d = _mm_add_ps(cur[Q + q], extra);
          grew = _mm_or_ps(grew, _mm_cmpgt_ps(d, cur[Q + q]));
          cur[Q + q] = d;
        }
        if (!_mm_movemask_ps(grew)) break;
        extra = shift_down(extra);
      }
    }
    for (int q = Q - 1; q >= 0; q--) {
      const __m128 *tn = tq + (size_t)((q < Q - 1) ? q + 1 : 0) * FT_NSTRIPE;
      const __m128 ivn = (q < Q - 1) ? ivx[q + 1] : ivx[0];
      __m128 pmm = _mm_mul_ps(ivn, tn[FT_MM]), pim = _mm_mul_ps(ivn, tn[FT_IM]);
      __m128 dn = (q < Q - 1) ? cur[Q + q + 1] : cur[Q];
      if (q == Q - 1) { pmm = shift_down(pmm); pim = shift_down(pim); dn = shift_down(dn); }
      const __m128 *tk = tq + (size_t) q * FT_NSTRIPE;
      __m128 m = _mm_mul_ps(ev, mask[q]);
      m = _mm_add_ps(m, _mm_mul_ps(dn, tk[FT_MD]));
      m = _mm_add_ps(m, _mm_mul_ps(r3[2 * Q + q], tk[FT_MI]));
      m = _mm_add_ps(m, pmm);
      cur[q] = m;
      cur[2 * Q + q] = _mm_add_ps(_mm_mul_ps(r3[2 * Q + q], tk[FT_II]), pim);
    }
    if (xB > 1.0e4f) {                                                     /* everything an earlier row reads moves to the new scale together */
      const float f = 1.0f / xB;
      const __m128 fv = _mm_set1_ps(f);
      for (int a = 0; a < 15 * Q; a++) ws[a] = _mm_mul_ps(ws[a], fv);
      for (int r = 0; r < 5; r++) { N[r] *= f; J[r] *= f; Cc[r] *= f; }
      for (int r = 0; r < 3; r++) n012[r] *= f;
      xN *= f; xJ *= f; xC *= f; xE *= f;
      totscale += log((double) xB);
      xB = 1.0f;
    }
    N[i % 5] = xN; J[i % 5] = xJ; Cc[i % 5] = xC;
    if (i <= 2) n012[i] = xN;
    if (xmx_log) {
      float *o = xmx_log + (size_t) i * BO_NXCELLS;
      const float ts = (float) totscale;
      o[BO_GE] = logf(xE) + ts; o[BO_GN] = logf(xN) + ts; o[BO_GJ] = logf(xJ) + ts; o[BO_GB] = logf(xB) + ts; o[BO_GC] = logf(xC) + ts;
    }
  }
  free(ws);
  const float tot = n012[0] + n012[1] + n012[2];
  if (isnan(tot) || isinf(tot) || tot == 0.0f) { *ret_sc = -INFINITY; return BO_ERANGE; }
  *ret_sc = (float)(totscale + log((double) tot));
  return BO_OK;
}
