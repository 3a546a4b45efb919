/* sse_filters.c -- SSE2 striped restatement of the reference's impl_sse filter kernels (see bath_sse.h).
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Written from scratch; what must coincide with the reference is the
 * arithmetic (saturating byte / word operations, operation order of the fp32 Forward), not the code.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_sse.h"

#define KP BO_KP_AMINO
enum { T_BM = 0, T_MM, T_IM, T_DM, T_MD, T_MI, T_II, T_NT };       /* striped transition order, impl_sse.h:73 */

static void *amalloc(size_t bytes)
{
  void *p = NULL;
  if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0) return NULL;
  return p;
}

/* ------------------------------------------------------------------ striping (p7_oprofile.c:667-921) */

bs_oprofile *bs_oprofile_create(const bo_oprofile *om)
{
  const int M = om->M;
  bs_oprofile *so = calloc(1, sizeof *so);
  so->M = M; so->om = om;
  so->Qb = (M - 1) / 16 + 1; if (so->Qb < 2) so->Qb = 2;       /* p7O_NQB, impl_sse.h:57 */
  so->Qw = (M - 1) / 8 + 1;  if (so->Qw < 2) so->Qw = 2;
  so->Qf = (M - 1) / 4 + 1;  if (so->Qf < 2) so->Qf = 2;
  const int Qb = so->Qb, Qw = so->Qw, Qf = so->Qf;
  const size_t W = (size_t) M + 1;
  so->sbv = so->mem[0] = amalloc(sizeof(__m128i) * KP * 2 * Qb);
  so->rbv = so->mem[1] = amalloc(sizeof(__m128i) * KP * Qb);
  so->rwv = so->mem[2] = amalloc(sizeof(__m128i) * KP * Qw);
  so->twv = so->mem[3] = amalloc(sizeof(__m128i) * 8 * Qw);
  so->rfv = so->mem[4] = amalloc(sizeof(__m128) * KP * Qf);
  so->tfv = so->mem[5] = amalloc(sizeof(__m128) * 8 * Qf);
  so->dpb = so->mem[6] = amalloc(sizeof(__m128i) * (Qb + 3 * Qw));
  so->dpw = so->dpb + Qb;
  so->dpf = so->mem[7] = amalloc(sizeof(__m128) * 3 * Qf);

  for (int x = 0; x < KP; x++) {
    uint8_t *rb = (uint8_t *)(so->rbv + (size_t) x * Qb);
    int8_t  *sb = (int8_t  *)(so->sbv + (size_t) x * 2 * Qb);
    for (int q = 0; q < Qb; q++)
      for (int z = 0; z < 16; z++) {
        const int k = z * Qb + q + 1;
        const int b = (k <= M) ? om->rb[x * W + k] : 255;
        int s = b - (int) om->bias_b; if (s > 127) s = 127;
        rb[q * 16 + z] = (uint8_t) b;
        sb[q * 16 + z] = (int8_t) s;
        sb[(Qb + q) * 16 + z] = (int8_t) s;
      }
    int16_t *rw = (int16_t *)(so->rwv + (size_t) x * Qw);
    for (int q = 0; q < Qw; q++)
      for (int z = 0; z < 8; z++) { const int k = z * Qw + q + 1; rw[q * 8 + z] = (k <= M) ? om->rw[x * W + k] : -32768; }
    float *rf = (float *)(so->rfv + (size_t) x * Qf);
    for (int q = 0; q < Qf; q++)
      for (int z = 0; z < 4; z++) { const int k = z * Qf + q + 1; rf[q * 4 + z] = (k <= M) ? om->rf[x * W + k] : 0.0f; }
  }
  static const int gen[T_NT] = { BO_BM, BO_MM, BO_IM, BO_DM, BO_MD, BO_MI, BO_II };
  int16_t *tw = (int16_t *) so->twv;
  for (int q = 0; q < Qw; q++)
    for (int z = 0; z < 8; z++) {
      const int k = z * Qw + q + 1;
      for (int t = 0; t < T_NT; t++) tw[(q * T_NT + t) * 8 + z] = (k <= M) ? om->tw[k * BO_NTRANS + gen[t]] : -32768;
      tw[(T_NT * Qw + q) * 8 + z] = (k <= M) ? om->tw[k * BO_NTRANS + BO_DD] : -32768;
    }
  float *tf = (float *) so->tfv;
  for (int q = 0; q < Qf; q++)
    for (int z = 0; z < 4; z++) {
      const int k = z * Qf + q + 1;
      for (int t = 0; t < T_NT; t++) tf[(q * T_NT + t) * 4 + z] = (k <= M) ? om->tf[k * BO_NTRANS + gen[t]] : 0.0f;
      tf[(T_NT * Qf + q) * 4 + z] = (k <= M) ? om->tf[k * BO_NTRANS + BO_DD] : 0.0f;
    }
  return so;
}

void bs_oprofile_free(bs_oprofile *so)
{
  if (!so) return;
  for (int i = 0; i < 8; i++) free(so->mem[i]);
  free(so);
}

/* ------------------------------------------------------------------ horizontal helpers (esl_sse.h equivalents) */

static inline uint8_t hmax_epu8(__m128i a)
{
  a = _mm_max_epu8(a, _mm_srli_si128(a, 8));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 4));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 2));
  a = _mm_max_epu8(a, _mm_srli_si128(a, 1));
  return (uint8_t) _mm_extract_epi16(a, 0);
}
static inline int16_t hmax_epi16(__m128i a)
{
  a = _mm_max_epi16(a, _mm_srli_si128(a, 8));
  a = _mm_max_epi16(a, _mm_srli_si128(a, 4));
  a = _mm_max_epi16(a, _mm_srli_si128(a, 2));
  return (int16_t) _mm_extract_epi16(a, 0);
}
static inline int any_gt_epi16(__m128i a, __m128i b) { return _mm_movemask_epi8(_mm_cmpgt_epi16(a, b)) != 0; }
static inline float hsum_ps(__m128 a)
{
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(0, 3, 2, 1)));
  a = _mm_add_ps(a, _mm_shuffle_ps(a, a, _MM_SHUFFLE(1, 0, 3, 2)));
  float r; _mm_store_ss(&r, a); return r;
}

/* ------------------------------------------------------------------ SSV: bands of diagonals held in registers
 * (the idea of ssvfilter.c:209-330).  A band is W adjacent striped diagonals; register j starts on vector q0+j and moves one
 * vector to the right per residue, so for Q-W residues out of Q all W registers read consecutive cost vectors; during the
 * other W residues one register per residue runs off the end of the stripe, is shifted by one element (a begin score enters)
 * and continues at vector 0.  The cost rows are stored twice back to back, so even then the W reads are consecutive.
 * One function per band width, fully unrolled by the compiler (W is a compile-time constant after inlining). */

#define SSV_MAXW 14

static inline __attribute__((always_inline)) __m128i
ssv_band(const uint8_t *dsq, int L, const __m128i *sbv, int Q, int q0, const int W, __m128i xEv)
{
  const __m128i beginv = _mm_set1_epi8(-128);
  const __m128i lowb   = _mm_set_epi8(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -128);
  __m128i r[SSV_MAXW];
#pragma GCC unroll 16
  for (int j = 0; j < W; j++) r[j] = beginv;
  int i = 1;
  int p = q0;                                   /* vector of register 0 */
  while (i <= L) {
    int plain = Q - W - p;                      /* residues before the band's last register reaches the end of the stripe */
    if (plain > L - i + 1) plain = L - i + 1;
    for (int n = 0; n < plain; n++, i++, p++) {
      const __m128i *rsc = sbv + (size_t) dsq[i] * 2 * Q + p;
#pragma GCC unroll 16
      for (int j = 0; j < W; j++) { r[j] = _mm_subs_epi8(r[j], rsc[j]); xEv = _mm_max_epu8(xEv, r[j]); }
    }
    if (i > L) break;
    /* W residues during which register W-1-t wraps after its step */
#pragma GCC unroll 16
    for (int t = 0; t < W; t++) {
      if (i > L) return xEv;
      const __m128i *rsc = sbv + (size_t) dsq[i] * 2 * Q + p;
#pragma GCC unroll 16
      for (int j = 0; j < W; j++) { r[j] = _mm_subs_epi8(r[j], rsc[j]); xEv = _mm_max_epu8(xEv, r[j]); }
      r[W - 1 - t] = _mm_or_si128(_mm_slli_si128(r[W - 1 - t], 1), lowb);
      i++; p++;
    }
    p = 0;
  }
  return xEv;
}

#define SSV_BAND_FN(Wv) \
  static __m128i ssv_band_##Wv(const uint8_t *dsq, int L, const __m128i *sbv, int Q, int q0, __m128i xEv) { return ssv_band(dsq, L, sbv, Q, q0, Wv, xEv); }
SSV_BAND_FN(1) SSV_BAND_FN(2) SSV_BAND_FN(3) SSV_BAND_FN(4) SSV_BAND_FN(5) SSV_BAND_FN(6) SSV_BAND_FN(7)
SSV_BAND_FN(8) SSV_BAND_FN(9) SSV_BAND_FN(10) SSV_BAND_FN(11) SSV_BAND_FN(12) SSV_BAND_FN(13) SSV_BAND_FN(14)
typedef __m128i (*ssv_band_fn)(const uint8_t *, int, const __m128i *, int, int, __m128i);
static const ssv_band_fn ssv_bands[SSV_MAXW + 1] = { NULL, ssv_band_1, ssv_band_2, ssv_band_3, ssv_band_4, ssv_band_5, ssv_band_6, ssv_band_7,
                                                     ssv_band_8, ssv_band_9, ssv_band_10, ssv_band_11, ssv_band_12, ssv_band_13, ssv_band_14 };

static uint8_t ssv_xE(const uint8_t *dsq, int L, const bs_oprofile *so)      /* get_xE, ssvfilter.c:832-872 */
{
  const int Q = so->Qb;
  __m128i xEv = _mm_set1_epi8(-128);
  const int nb = (Q + SSV_MAXW - 1) / SSV_MAXW;                              /* as few sweeps as the registers allow, of even width */
  int last = 0;
  for (int b = 0; b < nb; b++) {
    const int q = (Q * (b + 1)) / nb;
    xEv = ssv_bands[q - last](dsq, L, so->sbv, Q, last, xEv);
    last = q;
  }
  return hmax_epu8(xEv);
}

int bs_ssvfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc)   /* ssvfilter.c:876-925 */
{
  const bo_oprofile *om = so->om;
  uint16_t xE, xJ;
  if (om->tjb_b + om->tbm_b + om->tec_b + om->bias_b >= 127) return BO_ENORESULT;
  xE = ssv_xE(dsq, L, so);
  if (xE >= 255 - om->bias_b) {
    *ret_sc = INFINITY;
    if (om->base_b - om->tjb_b - om->tbm_b < 128) return BO_ENORESULT;
    return BO_ERANGE;
  }
  xE += om->base_b - om->tjb_b - om->tbm_b;
  xE -= 128;
  if (xE >= 255 - om->bias_b) { *ret_sc = INFINITY; return BO_ERANGE; }
  xJ = xE - om->tec_b;
  if (xJ > om->base_b) return BO_ENORESULT;
  *ret_sc = ((float) (xJ - om->tjb_b) - (float) om->base_b);
  *ret_sc /= om->scale_b;
  *ret_sc -= 3.0;
  return BO_OK;
}

/* ------------------------------------------------------------------ MSV with the J state (msvfilter.c:106-207) */

int bs_msv_full(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc)
{
  const bo_oprofile *om = so->om;
  const int Q = so->Qb;
  __m128i *dp = so->dpb;
  const __m128i biasv = _mm_set1_epi8((int8_t) om->bias_b);
  const int bias = om->bias_b, base = om->base_b, tec = om->tec_b;
  const int tjbm = (uint8_t)((int8_t) om->tjb_b + (int8_t) om->tbm_b);      /* set1_epi8 of an int8 sum, msvfilter.c:116 */
  int xJ = 0;
  __m128i xBv = _mm_set1_epi8((int8_t)(base > tjbm ? base - tjbm : 0));
  for (int q = 0; q < Q; q++) dp[q] = _mm_setzero_si128();
  for (int i = 1; i <= L; i++) {
    const __m128i *rsc = so->rbv + (size_t) dsq[i] * Q;
    __m128i xEv = _mm_setzero_si128();
    __m128i mpv = _mm_slli_si128(dp[Q - 1], 1);              /* the diagonal move across the stripe boundary; 0 enters */
    for (int q = 0; q < Q; q++) {
      __m128i sv = _mm_max_epu8(mpv, xBv);
      sv = _mm_adds_epu8(sv, biasv);
      sv = _mm_subs_epu8(sv, rsc[q]);
      xEv = _mm_max_epu8(xEv, sv);
      mpv = dp[q];
      dp[q] = sv;
    }
    int xE = hmax_epu8(xEv);                                 /* msvfilter.c:160-194 on the horizontal maximum */
    if (xE + bias >= 255) { *ret_sc = INFINITY; return BO_ERANGE; }
    xE = xE > tec ? xE - tec : 0;
    if (xE > xJ) xJ = xE;
    int xB = (base > xJ ? base : xJ) - tjbm; if (xB < 0) xB = 0;
    xBv = _mm_set1_epi8((int8_t) xB);
  }
  *ret_sc = ((float) (xJ - om->tjb_b) - (float) om->base_b);
  *ret_sc /= om->scale_b;
  *ret_sc -= 3.0;
  return BO_OK;
}

int bs_msvfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc)   /* msvfilter.c:74-104 */
{
  const int status = bs_ssvfilter(dsq, L, so, ret_sc);
  if (status != BO_ENORESULT) return status;
  return bs_msv_full(dsq, L, so, ret_sc);
}

/* ------------------------------------------------------------------ Viterbi filter (vitfilter.c:83-248, 286-465) */

#define LOG2C 0.69314718055994529
static void window_add(bo_windowlist *wl, int n, int k, int length)
{
  if (wl->count == wl->size) { wl->size *= 4; wl->w = realloc(wl->w, sizeof(bo_window) * (size_t) wl->size); }
  bo_window *w = &wl->w[wl->count++];
  w->id = 0; w->n = n; w->k = k; w->length = length; w->score = 0.0f;
}

static int vit_engine(const uint8_t *dsq, int L, bs_oprofile *so, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc)
{
  const bo_oprofile *om = so->om;
  const int Q = so->Qw, M = so->M;
  __m128i *MMX = so->dpw, *DMX = MMX + Q, *IMX = DMX + Q;
  const __m128i neginf = _mm_set1_epi16(-32768);
  const __m128i neglow = _mm_set_epi16(0, 0, 0, 0, 0, 0, 0, -32768);          /* -32768 enters at element 0 on a shift */
  int16_t xE, xB, xC, xJ, xN;
  int sc_thresh = 0, sc_ext_thresh = 0, skip_until = 0;
  if (sd) {                                                                   /* vitfilter.c:314-321 */
    float invP = (float) bo_gumbel_invsurv(P, om->evparam[BO_VMU], om->evparam[BO_VLAMBDA]);
    sc_thresh = (int16_t) ceil(((filtersc + LOG2C * invP + 3.0) * om->scale_w)
                               - (float) om->xw[BO_XE][BO_MOVE] - (float) om->xw[BO_XC][BO_MOVE] + (float) om->base_w);
    invP = (float) bo_gumbel_invsurv(P, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
    sc_ext_thresh = (int) ceil(((filtersc + LOG2C * invP + 3.0) * om->scale_b) + om->base_b + om->tec_b + om->tjb_b);
  }
  for (int q = 0; q < Q; q++) MMX[q] = DMX[q] = IMX[q] = neginf;
  xN = om->base_w;
  xB = (int16_t)(xN + om->xw[BO_XN][BO_MOVE]);
  xJ = -32768; xC = -32768; xE = -32768;
#define SHIFT16(v) _mm_or_si128(_mm_slli_si128((v), 2), neglow)
  for (int i = 1; i <= L; i++) {
    const __m128i *rsc = so->rwv + (size_t) dsq[i] * Q;
    const __m128i *tsc = so->twv;
    __m128i dcv = neginf, xEv = neginf, Dmaxv = neginf;
    const __m128i xBv = _mm_set1_epi16(xB);
    __m128i mpv = SHIFT16(MMX[Q - 1]), dpv = SHIFT16(DMX[Q - 1]), ipv = SHIFT16(IMX[Q - 1]);
    for (int q = 0; q < Q; q++, tsc += T_NT) {
      __m128i sv =            _mm_adds_epi16(xBv, tsc[T_BM]);
      sv = _mm_max_epi16(sv,  _mm_adds_epi16(mpv, tsc[T_MM]));
      sv = _mm_max_epi16(sv,  _mm_adds_epi16(ipv, tsc[T_IM]));
      sv = _mm_max_epi16(sv,  _mm_adds_epi16(dpv, tsc[T_DM]));
      sv = _mm_adds_epi16(sv, rsc[q]);
      xEv = _mm_max_epi16(xEv, sv);
      mpv = MMX[q]; dpv = DMX[q]; ipv = IMX[q];
      MMX[q] = sv;
      DMX[q] = dcv;                                        /* D(i,k) gets M(i,k-1)+tMD; D->D below */
      dcv = _mm_adds_epi16(sv, tsc[T_MD]);
      Dmaxv = _mm_max_epi16(dcv, Dmaxv);
      IMX[q] = _mm_max_epi16(_mm_adds_epi16(mpv, tsc[T_MI]), _mm_adds_epi16(ipv, tsc[T_II]));
    }
    xE = hmax_epi16(xEv);
    if (xE >= 32767) { *ret_sc = INFINITY; return BO_ERANGE; }
    xN = (int16_t)(xN + om->xw[BO_XN][BO_LOOP]);
    { int a = xC + om->xw[BO_XC][BO_LOOP], b = xE + om->xw[BO_XE][BO_MOVE]; xC = (int16_t)(a > b ? a : b); }
    { int a = xJ + om->xw[BO_XJ][BO_LOOP], b = xE + om->xw[BO_XE][BO_LOOP]; xJ = (int16_t)(a > b ? a : b); }
    { int a = xJ + om->xw[BO_XJ][BO_MOVE], b = xN + om->xw[BO_XN][BO_MOVE]; xB = (int16_t)(a > b ? a : b); }

    if (sd && i > skip_until && xE >= sc_thresh) {          /* vitfilter.c:386-424: first maximal cell in striped order */
      int k_start = 0;
      for (int q = 0; q < Q && k_start == 0; q++) {
        const int16_t *mv = (const int16_t *) &MMX[q];
        for (int z = 0; z < 8; z++) { const int k = q + Q * z + 1; if (k <= M && mv[z] == xE) { k_start = k; break; } }
      }
      int max_k_end = k_start, max_i_end = i, sc_ext = sc_ext_thresh, max_sc_ext = sc_ext, pos_since_max = 0;
      int kk = k_start + 1, nn = i + 1;
      while (kk <= M && nn <= L) {
        sc_ext += om->bias_b - sd->ssv_scores[kk * KP + dsq[nn]];
        if (sc_ext >= max_sc_ext) { max_sc_ext = sc_ext; max_k_end = kk; max_i_end = nn; pos_since_max = 0; }
        else if (++pos_since_max == 5) break;
        kk++; nn++;
      }
      window_add(wl, i, max_k_end, max_k_end - k_start + 1);
      skip_until = max_i_end;
    }

    const int16_t Dmax = hmax_epi16(Dmaxv);
    if ((int) Dmax + om->ddbound_w > xB) {                  /* lazy F (vitfilter.c:197-231): D->D only when it can matter */
      const __m128i *tdd = so->twv + (size_t) T_NT * Q;
      dcv = SHIFT16(dcv);
      for (int q = 0; q < Q; q++) { DMX[q] = _mm_max_epi16(dcv, DMX[q]); dcv = _mm_adds_epi16(DMX[q], tdd[q]); }
      int q;
      do {                                                  /* further passes until a whole stripe changes nothing */
        dcv = SHIFT16(dcv);
        for (q = 0; q < Q; q++) {
          if (!any_gt_epi16(dcv, DMX[q])) break;
          DMX[q] = _mm_max_epi16(dcv, DMX[q]);
          dcv = _mm_adds_epi16(DMX[q], tdd[q]);
        }
      } while (q == Q);
    } else DMX[0] = SHIFT16(dcv);
  }
#undef SHIFT16
  if (xC > -32768) {
    *ret_sc = (float) xC + (float) om->xw[BO_XC][BO_MOVE] - (float) om->base_w;
    *ret_sc /= om->scale_w;
    *ret_sc -= 3.0;
  } else *ret_sc = -INFINITY;
  return BO_OK;
}

int bs_vitfilter(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc) { return vit_engine(dsq, L, so, NULL, 0.f, 0., NULL, ret_sc); }
int bs_vitfilter_bath(const uint8_t *dsq, int L, bs_oprofile *so, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc)
{
  return vit_engine(dsq, L, so, sd, filtersc, P, wl, ret_sc);
}

/* ------------------------------------------------------------------ Forward parser (fwdback.c:256-463) */

int bs_forward_parser(const uint8_t *dsq, int L, bs_oprofile *so, float *ret_sc)
{
  const bo_oprofile *om = so->om;
  const int Q = so->Qf;
  __m128 *MMO = so->dpf, *DMO = MMO + Q, *IMO = DMO + Q;
  const __m128 zerov = _mm_setzero_ps();
  float xN = 1.f, xE = 0.f, xJ = 0.f, xC = 0.f, xB = om->xf[BO_XN][BO_MOVE];
  float totscale = 0.0f;
#define SHIFTF(v) _mm_castsi128_ps(_mm_slli_si128(_mm_castps_si128(v), 4))
  for (int q = 0; q < Q; q++) MMO[q] = DMO[q] = IMO[q] = zerov;
  for (int i = 1; i <= L; i++) {
    const __m128 *rsc = so->rfv + (size_t) dsq[i] * Q;
    const __m128 *tsc = so->tfv;
    __m128 dcv = zerov, xEv = zerov;
    const __m128 xBv = _mm_set1_ps(xB);
    __m128 mpv = SHIFTF(MMO[Q - 1]), dpv = SHIFTF(DMO[Q - 1]), ipv = SHIFTF(IMO[Q - 1]);
    for (int q = 0; q < Q; q++, tsc += T_NT) {
      __m128 sv =          _mm_mul_ps(xBv, tsc[T_BM]);
      sv = _mm_add_ps(sv,  _mm_mul_ps(mpv, tsc[T_MM]));
      sv = _mm_add_ps(sv,  _mm_mul_ps(ipv, tsc[T_IM]));
      sv = _mm_add_ps(sv,  _mm_mul_ps(dpv, tsc[T_DM]));
      sv = _mm_mul_ps(sv, rsc[q]);
      xEv = _mm_add_ps(xEv, sv);
      mpv = MMO[q]; dpv = DMO[q]; ipv = IMO[q];
      MMO[q] = sv;
      DMO[q] = dcv;
      dcv = _mm_mul_ps(sv, tsc[T_MD]);
      IMO[q] = _mm_add_ps(_mm_mul_ps(mpv, tsc[T_MI]), _mm_mul_ps(ipv, tsc[T_II]));
    }
    /* D->D: one pass carries every path that stays inside an element's segment; three more shifted passes carry the paths
     * that cross 1, 2, 3 segment boundaries (fwdback.c:352-404, the "fully serialized" form) */
    const __m128 *tdd = so->tfv + (size_t) T_NT * Q;
    dcv = SHIFTF(dcv);
    DMO[0] = zerov;
    for (int q = 0; q < Q; q++) { DMO[q] = _mm_add_ps(dcv, DMO[q]); dcv = _mm_mul_ps(DMO[q], tdd[q]); }
    for (int j = 1; j < 4; j++) {
      dcv = SHIFTF(dcv);
      for (int q = 0; q < Q; q++) { DMO[q] = _mm_add_ps(dcv, DMO[q]); dcv = _mm_mul_ps(dcv, tdd[q]); }
    }
    for (int q = 0; q < Q; q++) xEv = _mm_add_ps(DMO[q], xEv);
    xE = hsum_ps(xEv);
    xN = xN * om->xf[BO_XN][BO_LOOP];
    xC = (xC * om->xf[BO_XC][BO_LOOP]) + (xE * om->xf[BO_XE][BO_MOVE]);
    xJ = (xJ * om->xf[BO_XJ][BO_LOOP]) + (xE * om->xf[BO_XE][BO_LOOP]);
    xB = (xJ * om->xf[BO_XJ][BO_MOVE]) + (xN * om->xf[BO_XN][BO_MOVE]);
    if (xE > 1.0e4) {                                       /* sparse rescaling, fwdback.c:418-434 */
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      const __m128 inv = _mm_set1_ps((float)(1.0 / xE));
      for (int q = 0; q < Q; q++) { MMO[q] = _mm_mul_ps(MMO[q], inv); DMO[q] = _mm_mul_ps(DMO[q], inv); IMO[q] = _mm_mul_ps(IMO[q], inv); }
      totscale += (float) log(xE);
      xE = 1.0f;
    }
  }
#undef SHIFTF
  if (isnan(xC) || (L > 0 && xC == 0.0) || isinf(xC)) { *ret_sc = -INFINITY; return BO_ERANGE; }
  *ret_sc = (float)(totscale + log(xC * om->xf[BO_XC][BO_MOVE]));
  return BO_OK;
}
