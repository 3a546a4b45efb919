/* filters.c -- the HMMER3 filter cascade kernels, scalar and unstriped.  ORACLE (test infra only).
 *
 * Each function restates the arithmetic of its impl_sse counterpart cell by cell (same
 * saturation, same operand order), with the SIMD striping removed: model node k is an array
 * index.  Where the reference result depends on the striped visiting order (first k with
 * M(i,k)==xE in p7_ViterbiFilter_BATH, vitfilter.c:390-396) that order is reproduced explicitly.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529

static inline int sat_u8(int v)  { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int sat_i16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int imax(int a, int b) { return a > b ? a : b; }

void bo_windowlist_init(bo_windowlist *wl) { wl->size = 16; wl->count = 0; wl->w = malloc(sizeof(bo_window) * 16); }
void bo_windowlist_free(bo_windowlist *wl) { free(wl->w); wl->w = NULL; wl->count = wl->size = 0; }
static void window_new(bo_windowlist *wl, int id, int n, int k, int length, float score)   /* p7_hmmwindow.c:83 */
{
  if (wl->count == wl->size) { wl->size *= 4; wl->w = realloc(wl->w, sizeof(bo_window) * (size_t) wl->size); }
  bo_window *w = &wl->w[wl->count++];
  w->id = id; w->n = n; w->k = k; w->length = length; w->score = score;
}

/* ------------------------------------------------------------------ SSV / MSV */

/* get_xE(), ssvfilter.c:832-872, as a plain diagonal recursion.
 * The striped kernel keeps each diagonal in signed bytes starting at -128 and applies
 * sv = subs_epi8(sv, sb) with sb = min(rb - bias, 127) (sf_conversion, p7_oprofile.c:751-757);
 * xE is the unsigned max over every cell.  Returns the unsigned byte the SIMD code would return,
 * or 255 if any cell reaches the "possible overflow" zone (then the value itself is never used). */
static int ssv_get_xE(const uint8_t *dsq, int L, const bo_oprofile *om)
{
  int M = om->M;
  size_t W = (size_t) M + 1;
  int *dp = malloc(sizeof(int) * (size_t)(M + 1));
  int best = -128;
  for (int k = 0; k <= M; k++) dp[k] = -128;
  for (int i = 1; i <= L; i++) {
    const uint8_t *rb = om->rb + dsq[i] * W;
    for (int k = M; k >= 1; k--) {
      int sb = (int) rb[k] - (int) om->bias_b; if (sb > 127) sb = 127;
      int v = dp[k-1] - sb;
      if (v < -128) v = -128;
      dp[k] = v;
      if (v > best) best = v;
    }
    dp[0] = -128;
  }
  free(dp);
  if (best >= -1 - (int) om->bias_b) return 255;      /* unsigned view >= 255 - bias_b */
  return best + 256;
}

int bo_ssvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)   /* ssvfilter.c:876-925 */
{
  uint16_t xE, xJ;
  if (om->tjb_b + om->tbm_b + om->tec_b + om->bias_b >= 127) return BO_ENORESULT;
  xE = (uint16_t) ssv_get_xE(dsq, L, om);
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

int bo_msvfilter_noSSV(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)   /* msvfilter.c:106-207 */
{
  int M = om->M;
  size_t W = (size_t) M + 1;
  uint8_t *dp = calloc((size_t) M + 1, 1);
  int bias = om->bias_b, base = om->base_b;
  int tjbm = (uint8_t)((int8_t) om->tjb_b + (int8_t) om->tbm_b);   /* set1_epi8 of an int8 sum, msvfilter.c:116 */
  int tec  = om->tec_b;
  int xJ = 0;
  int xB = sat_u8(base - tjbm);
  for (int i = 1; i <= L; i++) {
    const uint8_t *rb = om->rb + dsq[i] * W;
    int xE = 0;
    for (int k = M; k >= 1; k--) {
      int sv = imax(dp[k-1], xB);           /* dp[0] stays 0: the shifted-in -infinity */
      sv = sat_u8(sv + bias);
      sv = sat_u8(sv - rb[k]);
      if (sv > xE) xE = sv;
      dp[k] = (uint8_t) sv;
    }
    if (sat_u8(xE + bias) == 255) { free(dp); *ret_sc = INFINITY; return BO_ERANGE; }
    xE = sat_u8(xE - tec);
    xJ = imax(xJ, xE);
    xB = sat_u8(imax(base, xJ) - tjbm);
  }
  free(dp);
  *ret_sc = ((float) (xJ - om->tjb_b) - (float) om->base_b);
  *ret_sc /= om->scale_b;
  *ret_sc -= 3.0;
  return BO_OK;
}

int bo_msvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)   /* msvfilter.c:74-104 */
{
  int status = bo_ssvfilter(dsq, L, om, ret_sc);
  if (status != BO_ENORESULT) return status;
  return bo_msvfilter_noSSV(dsq, L, om, ret_sc);
}

/* p7_SSVFilter_BATH, msvfilter.c:250-427.  Mutates om->tjb_b and bg->p1 exactly as the reference does
 * (msvfilter.c:308-310).  The "which state hit threshold" scan (msvfilter.c:358-368) walks striped
 * vectors q=0..Q-1, lanes 0..15 and keeps the strictly greatest byte, so ties resolve to the first
 * cell in that order. */
int bo_ssvfilter_bath(const uint8_t *dsq, int L, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg, double P, bo_windowlist *wl)
{
  int M = om->M, Kp = BO_KP_AMINO;
  size_t W = (size_t) M + 1;
  int Q = imax(2, ((M - 1) / 16) + 1);
  float invP = (float) bo_gumbel_invsurv(P, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
  bo_bg_setlength(bg, L);
  bo_oprofile_reconfig_msv_length(om, L);
  float nullsc = bo_bg_nullone(bg, L);
  uint8_t sc_thresh = (uint8_t)(int) ceil(((nullsc + (invP * LOG2C) + 3.0) * om->scale_b) + om->base_b + om->tec_b + om->tjb_b);
  int bias = om->bias_b, base = om->base_b;
  int tjbm = (uint8_t)((int8_t) om->tjb_b + (int8_t) om->tbm_b);
  int xB = sat_u8(base - tjbm);
  uint8_t *dp = calloc((size_t) Q * 16 + 1, 1);        /* covers phantom nodes up to Q*16 */
  int MP = Q * 16;

  for (int i = 1; i <= L; i++) {
    const uint8_t *rb = om->rb + dsq[i] * W;
    int xE = 0;
    for (int k = MP; k >= 1; k--) {
      int cost = (k <= M) ? rb[k] : 255;
      int sv = imax(dp[k-1], xB);
      sv = sat_u8(sv + bias);
      sv = sat_u8(sv - cost);
      if (sv > xE) xE = sv;
      dp[k] = (uint8_t) sv;
    }
    if (sat_u8(xE + (255 - sc_thresh)) == 255) {       /* xE >= sc_thresh */
      int end = -1, rem_sc = -1;
      for (int q = 0; q < Q; q++)
        for (int z = 0; z < 16; z++) {
          int k = q + Q * z + 1;
          int b = dp[k];
          if (b >= sc_thresh && b > rem_sc && k <= M) { end = k; rem_sc = b; }
        }
      for (int k = 0; k <= MP; k++) dp[k] = 0;
      int start = end, target_end = i, target_start = i;
      int sc = rem_sc;
      while (rem_sc > base - om->tjb_b - om->tbm_b) {
        rem_sc -= bias - sd->ssv_scores[start * Kp + dsq[target_start]];
        --start; --target_start;
      }
      start++; target_start++;
      int k = end + 1, n = target_end + 1, max_end = target_end, max_sc = sc, pos_since_max = 0;
      while (k < M && n <= L) {
        sc += bias - sd->ssv_scores[k * Kp + dsq[n]];
        if (sc >= max_sc) { max_sc = sc; max_end = n; pos_since_max = 0; }
        else { pos_since_max++; if (pos_since_max == 5) break; }
        k++; n++;
      }
      end += (max_end - target_end);
      target_end = max_end;
      float ret_sc = ((float) (max_sc - om->tjb_b) - (float) om->base_b);
      ret_sc /= om->scale_b;
      ret_sc -= 3.0;
      window_new(wl, 0, target_start, end, end - start + 1, ret_sc);
      i = target_end;
    }
  }
  free(dp);
  return BO_OK;
}

/* ------------------------------------------------------------------ Viterbi filter */

/* One engine for p7_ViterbiFilter (vitfilter.c:83-248) and p7_ViterbiFilter_BATH (vitfilter.c:286-465):
 * the BATH variant only adds the window bookkeeping (sd != NULL). */
static int vit_engine(const uint8_t *dsq, int L, const bo_oprofile *om, const bo_scoredata *sd,
                      float filtersc, double P, bo_windowlist *wl, float *ret_sc)
{
  int M = om->M, Kp = BO_KP_AMINO;
  size_t W = (size_t) M + 1;
  int Q = imax(2, ((M - 1) / 8) + 1);
  int16_t *Mp = malloc(sizeof(int16_t) * W * 6);
  int16_t *Ip = Mp + W, *Dp = Ip + W, *Mc = Dp + W, *Ic = Mc + W, *Dc = Ic + W;
  int16_t *alloc = Mp;
  const int16_t *tw = om->tw;
  int16_t xE, xB, xC, xJ, xN;
  int sc_thresh = 0, sc_ext_thresh = 0, skip_until = 0;

  if (sd) {                                            /* vitfilter.c:314-321 */
    float invP = (float) bo_gumbel_invsurv(P, om->evparam[BO_VMU], om->evparam[BO_VLAMBDA]);
    sc_thresh = (int16_t) ceil(((filtersc + LOG2C * invP + 3.0) * om->scale_w)
                               - (float) om->xw[BO_XE][BO_MOVE] - (float) om->xw[BO_XC][BO_MOVE] + (float) om->base_w);
    invP = (float) bo_gumbel_invsurv(P, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
    sc_ext_thresh = (int) ceil(((filtersc + LOG2C * invP + 3.0) * om->scale_b) + om->base_b + om->tec_b + om->tjb_b);
  }

  for (int k = 0; k <= M; k++) Mp[k] = Ip[k] = Dp[k] = -32768;
  xN = om->base_w;
  xB = (int16_t)(xN + om->xw[BO_XN][BO_MOVE]);
  xJ = -32768; xC = -32768; xE = -32768;

  for (int i = 1; i <= L; i++) {
    const int16_t *rw = om->rw + dsq[i] * W;
    int xEi = -32768, Dmax = -32768;
    Mc[0] = Ic[0] = Dc[0] = -32768;
    for (int k = 1; k <= M; k++) {
      const int16_t *t = tw + k * BO_NTRANS;
      int sv =          sat_i16(xB      + t[BO_BM]);
      sv = imax(sv,     sat_i16(Mp[k-1] + t[BO_MM]));
      sv = imax(sv,     sat_i16(Ip[k-1] + t[BO_IM]));
      sv = imax(sv,     sat_i16(Dp[k-1] + t[BO_DM]));
      sv = sat_i16(sv + rw[k]);
      if (sv > xEi) xEi = sv;
      Mc[k] = (int16_t) sv;
      int dcv = sat_i16(sv + t[BO_MD]);                /* partial D(i,k+1): M->D only */
      if (dcv > Dmax) Dmax = dcv;
      if (k < M) Dc[k+1] = (int16_t) dcv;
      Ic[k] = (int16_t) imax(sat_i16(Mp[k] + t[BO_MI]), sat_i16(Ip[k] + t[BO_II]));
    }
    Dc[1] = -32768;
    xE = (int16_t) xEi;
    if (xE >= 32767) { free(alloc); *ret_sc = INFINITY; return BO_ERANGE; }
    xN = (int16_t)(xN + om->xw[BO_XN][BO_LOOP]);
    xC = (int16_t) imax(xC + om->xw[BO_XC][BO_LOOP], xE + om->xw[BO_XE][BO_MOVE]);
    xJ = (int16_t) imax(xJ + om->xw[BO_XJ][BO_LOOP], xE + om->xw[BO_XE][BO_LOOP]);
    xB = (int16_t) imax(xJ + om->xw[BO_XJ][BO_MOVE], xN + om->xw[BO_XN][BO_MOVE]);

    if (sd && i > skip_until && xE >= sc_thresh) {      /* vitfilter.c:386-424 */
      int k_start = 0;
      for (int q = 0; q < Q && k_start == 0; q++)
        for (int z = 0; z < 8; z++) {
          int k = q + Q * z + 1;
          if (k <= M && Mc[k] == xE) { k_start = k; break; }
        }
      int max_k_end = k_start, max_i_end = i, sc_ext = sc_ext_thresh, max_sc_ext = sc_ext, pos_since_max = 0;
      int kk = k_start + 1, nn = i + 1;
      while (kk <= M && nn <= L) {
        sc_ext += om->bias_b - sd->ssv_scores[kk * Kp + dsq[nn]];
        if (sc_ext >= max_sc_ext) { max_sc_ext = sc_ext; max_k_end = kk; max_i_end = nn; pos_since_max = 0; }
        else if (++pos_since_max == 5) break;
        kk++; nn++;
      }
      window_new(wl, 0, i, max_k_end, max_k_end - k_start + 1, 0.0f);
      skip_until = max_i_end;
    }

    /* lazy-F (vitfilter.c:197-231): D->D paths only when they could beat B->M on the next row */
    if (Dmax + om->ddbound_w > xB) {
      for (int k = 1; k < M; k++)
        Dc[k+1] = (int16_t) imax(Dc[k+1], sat_i16(Dc[k] + tw[k * BO_NTRANS + BO_DD]));
    }
    int16_t *tmp;
    tmp = Mp; Mp = Mc; Mc = tmp;
    tmp = Ip; Ip = Ic; Ic = tmp;
    tmp = Dp; Dp = Dc; Dc = tmp;
  }
  free(alloc);
  if (xC > -32768) {
    *ret_sc = (float) xC + (float) om->xw[BO_XC][BO_MOVE] - (float) om->base_w;
    *ret_sc /= om->scale_w;
    *ret_sc -= 3.0;
  } else *ret_sc = -INFINITY;
  return BO_OK;
}

int bo_vitfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)
{
  return vit_engine(dsq, L, om, NULL, 0.f, 0., NULL, ret_sc);
}

int bo_vitfilter_bath(const uint8_t *dsq, int L, const bo_oprofile *om, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc)
{
  return vit_engine(dsq, L, om, sd, filtersc, P, wl, ret_sc);
}

/* ------------------------------------------------------------------ Forward / Backward parsers */

/* forward_engine(do_full=FALSE), fwdback.c:256-463, unstriped.  Per cell the operation order is the
 * reference's (mul,add,add,add,mul); the D row is the exact serial recurrence that the striped DD
 * passes converge to; row sums (xE) are accumulated in node order, so fp32 rounding differs from the
 * striped order at the 1e-7 level.  xmx6 (optional) receives (L+1) x {E,N,J,B,C,SCALE}. */
static int forward_engine(const uint8_t *dsq, int L, const bo_oprofile *om, float *dpf, float *xmx, float *ret_sc)
{
  int M = om->M;
  size_t W = (size_t) M + 1;
  float *Mp = calloc(W * 6, sizeof(float));
  float *Ip = Mp + W, *Dp = Ip + W, *Mc = Dp + W, *Ic = Mc + W, *Dc = Ic + W;
  float *alloc = Mp;
  const float *tf = om->tf;
  float xN = 1.f, xE = 0.f, xJ = 0.f, xC = 0.f, xB = om->xf[BO_XN][BO_MOVE];
  double totscale = 0.0;   /* ox->totscale is a float in the reference (impl_sse.h); accumulated with log() */
  float  totscale_f = 0.0f;
  if (xmx) { xmx[0] = xE; xmx[1] = xN; xmx[2] = xJ; xmx[3] = xB; xmx[4] = xC; xmx[5] = 1.0f; }
  if (dpf) memset(dpf, 0, sizeof(float) * W * 3);        /* row 0 */
  (void) totscale;

  for (int i = 1; i <= L; i++) {
    const float *rf = om->rf + dsq[i] * W;
    float sumE = 0.f;
    Mc[0] = Ic[0] = Dc[0] = 0.f;
    float dcv = 0.f;
    for (int k = 1; k <= M; k++) {
      const float *t = tf + k * BO_NTRANS;
      float sv =  xB      * t[BO_BM];
      sv = sv +   Mp[k-1] * t[BO_MM];
      sv = sv +   Ip[k-1] * t[BO_IM];
      sv = sv +   Dp[k-1] * t[BO_DM];
      sv = sv * rf[k];
      sumE += sv;
      Mc[k] = sv;
      Dc[k] = dcv;                               /* D(i,k) = M(i,k-1)*tMD + D(i,k-1)*tDD */
      dcv = sv * t[BO_MD] + Dc[k] * t[BO_DD];
      Ic[k] = Mp[k] * t[BO_MI] + Ip[k] * t[BO_II];
    }
    for (int k = 1; k <= M; k++) sumE += Dc[k];
    xE = sumE;
    xN = xN * om->xf[BO_XN][BO_LOOP];
    xC = (xC * om->xf[BO_XC][BO_LOOP]) + (xE * om->xf[BO_XE][BO_MOVE]);
    xJ = (xJ * om->xf[BO_XJ][BO_LOOP]) + (xE * om->xf[BO_XE][BO_LOOP]);
    xB = (xJ * om->xf[BO_XJ][BO_MOVE]) + (xN * om->xf[BO_XN][BO_MOVE]);
    float scale = 1.0f;
    if (xE > 1.0e4) {                            /* sparse rescaling, fwdback.c:418-434 */
      xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
      float inv = (float)(1.0 / xE);
      for (int k = 1; k <= M; k++) { Mc[k] *= inv; Dc[k] *= inv; Ic[k] *= inv; }
      scale = xE;
      totscale_f += (float) log(xE);
      xE = 1.0f;
    }
    if (xmx) { float *x = xmx + (size_t) i * 6; x[0] = xE; x[1] = xN; x[2] = xJ; x[3] = xB; x[4] = xC; x[5] = scale; }
    if (dpf) { float *r = dpf + (size_t) i * W * 3; r[0] = r[1] = r[2] = 0.f; for (int k = 1; k <= M; k++) { r[k*3+0] = Mc[k]; r[k*3+1] = Dc[k]; r[k*3+2] = Ic[k]; } }
    float *tmp;
    tmp = Mp; Mp = Mc; Mc = tmp;
    tmp = Ip; Ip = Ic; Ic = tmp;
    tmp = Dp; Dp = Dc; Dc = tmp;
  }
  free(alloc);
  if (isnan(xC) || (L > 0 && xC == 0.0) || isinf(xC)) { *ret_sc = -INFINITY; return BO_ERANGE; }
  *ret_sc = (float)(totscale_f + log(xC * om->xf[BO_XC][BO_MOVE]));
  return BO_OK;
}

int bo_forward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, float *xmx, float *ret_sc) { return forward_engine(dsq, L, om, NULL, xmx, ret_sc); }
/* p7_Forward (fwdback.c:94): full matrix dpf[(L+1)][(M+1)][3] = {M, D, I}, scaled like the parser; xmx as the parser's */
int bo_forward_full(const uint8_t *dsq, int L, const bo_oprofile *om, float *dpf, float *xmx, float *ret_sc) { return forward_engine(dsq, L, om, dpf, xmx, ret_sc); }

/* backward_engine(do_full=FALSE), fwdback.c:468-760, unstriped; uses the Forward scale factors.
 * bck_xmx6 receives (L+1) x {E,N,J,B,C,SCALE}; score = totscale + log(xN(0)). */
static int backward_engine(const uint8_t *dsq, int L, const bo_oprofile *om, const float *fwd, float *dpb, float *bck, float *ret_sc, int *ret_own_scales)
{
  int M = om->M;
  size_t W = (size_t) M + 2;
  float *Mn = calloc(W * 6, sizeof(float));
  float *In = Mn + W, *Dn = In + W, *Mc = Dn + W, *Ic = Mc + W, *Dc = Ic + W;
  float *alloc = Mn;
  const float *tf = om->tf;
  float xJ = 0.f, xB = 0.f, xN = 0.f;
  float xC = om->xf[BO_XC][BO_MOVE];
  float xE = xC * om->xf[BO_XE][BO_MOVE];
  float totscale;
  int own_scales = 0;

  /* row L: M(L,k) = D(L,k) = xE plus D->D->..->E and M->D paths (fwdback.c:499-532) */
  Dn[M+1] = 0.f; Mn[M+1] = 0.f; In[M+1] = 0.f;
  for (int k = M; k >= 1; k--) {
    float tdd = tf[k * BO_NTRANS + BO_DD], tmd = tf[k * BO_NTRANS + BO_MD];
    Dn[k] = xE + Dn[k+1] * tdd;
    Mn[k] = xE + Dn[k+1] * tmd;
    In[k] = 0.f;
  }
  float sc = fwd[(size_t) L * 6 + 5];
  if (sc > 1.0f) {
    xE /= sc; xN /= sc; xC /= sc; xJ /= sc; xB /= sc;
    float inv = (float)(1.0 / sc);
    for (int k = 1; k <= M; k++) { Mn[k] *= inv; Dn[k] *= inv; In[k] *= inv; }
  }
  totscale = (float) log(sc);
  if (bck) { float *x = bck + (size_t) L * 6; x[0] = xE; x[1] = xN; x[2] = xJ; x[3] = xB; x[4] = xC; x[5] = sc; }
  if (dpb) { float *r = dpb + (size_t) L * (M + 1) * 3; r[0] = r[1] = r[2] = 0.f; for (int k = 1; k <= M; k++) { r[k*3+0] = Mn[k]; r[k*3+1] = Dn[k]; r[k*3+2] = In[k]; } }

  for (int i = L - 1; i >= 1; i--) {
    const float *rf = om->rf + dsq[i+1] * (size_t)(M + 1);
    /* B(i) = sum_k tBM(k) * e(k,x_{i+1}) * M(i+1,k) */
    float b = 0.f;
    for (int k = 1; k <= M; k++) b += Mn[k] * rf[k] * tf[k * BO_NTRANS + BO_BM];
    xB = b;
    xC = xC * om->xf[BO_XC][BO_LOOP];
    xJ = (xB * om->xf[BO_XJ][BO_MOVE]) + (xJ * om->xf[BO_XJ][BO_LOOP]);
    xN = (xB * om->xf[BO_XN][BO_MOVE]) + (xN * om->xf[BO_XN][BO_LOOP]);
    xE = (xC * om->xf[BO_XE][BO_MOVE]) + (xJ * om->xf[BO_XE][BO_LOOP]);
    Dc[M+1] = 0.f;
    for (int k = M; k >= 1; k--) {
      float mnext = (k < M) ? Mn[k+1] * rf[k+1] : 0.f;     /* M(i+1,k+1) * e(k+1, x_{i+1}) */
      const float *t  = tf + k * BO_NTRANS;                /* out-of-k transitions MD, MI, II, DD */
      const float *t1 = tf + (k+1 <= M ? k+1 : M) * BO_NTRANS;   /* into-(k+1): MM, IM, DM stored at node k+1 */
      float tmm = (k < M) ? t1[BO_MM] : 0.f, tim = (k < M) ? t1[BO_IM] : 0.f, tdm = (k < M) ? t1[BO_DM] : 0.f;
      Ic[k] = In[k] * t[BO_II] + mnext * tim;
      Dc[k] = mnext * tdm + Dc[k+1] * t[BO_DD] + xE;
      Mc[k] = In[k] * t[BO_MI] + mnext * tmm + xE + Dc[k+1] * t[BO_MD];
    }
    float fs = fwd[(size_t) i * 6 + 5];
    if (xB > 1.0e16) own_scales = 1;
    float s = own_scales ? ((xB > 1.0e4) ? xB : 1.0f) : fs;
    if (s > 1.0f) {
      xE /= s; xN /= s; xJ /= s; xB /= s; xC /= s;
      float inv = (float)(1.0 / s);
      for (int k = 1; k <= M; k++) { Mc[k] *= inv; Dc[k] *= inv; Ic[k] *= inv; }
      totscale += (float) log(s);
    }
    if (bck) { float *x = bck + (size_t) i * 6; x[0] = xE; x[1] = xN; x[2] = xJ; x[3] = xB; x[4] = xC; x[5] = s; }
    if (dpb) { float *r = dpb + (size_t) i * (M + 1) * 3; r[0] = r[1] = r[2] = 0.f; for (int k = 1; k <= M; k++) { r[k*3+0] = Mc[k]; r[k*3+1] = Dc[k]; r[k*3+2] = Ic[k]; } }
    float *tmp;
    tmp = Mn; Mn = Mc; Mc = tmp;
    tmp = In; In = Ic; Ic = tmp;
    tmp = Dn; Dn = Dc; Dc = tmp;
  }
  /* i = 0: only N and B reachable (fwdback.c:695-740) */
  {
    const float *rf = om->rf + dsq[1] * (size_t)(M + 1);
    float b = 0.f;
    for (int k = 1; k <= M; k++) b += Mn[k] * rf[k] * tf[k * BO_NTRANS + BO_BM];
    xB = b;
    xN = (xB * om->xf[BO_XN][BO_MOVE]) + (xN * om->xf[BO_XN][BO_LOOP]);
    if (bck) { bck[0] = 0.f; bck[1] = xN; bck[2] = 0.f; bck[3] = xB; bck[4] = 0.f; bck[5] = 1.0f; }
    if (dpb) memset(dpb, 0, sizeof(float) * (size_t)(M + 1) * 3);
  }
  if (ret_own_scales) *ret_own_scales = own_scales;
  free(alloc);
  if (isnan(xN) || (L > 0 && xN == 0.0) || isinf(xN)) { if (ret_sc) *ret_sc = -INFINITY; return BO_ERANGE; }
  if (ret_sc) *ret_sc = (float)(totscale + log(xN));
  return BO_OK;
}

int bo_backward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, const float *fwd, float *bck, float *ret_sc) { return backward_engine(dsq, L, om, fwd, NULL, bck, ret_sc, NULL); }
/* p7_Backward (fwdback.c:196): full matrix, same layout as bo_forward_full */
int bo_backward_full(const uint8_t *dsq, int L, const bo_oprofile *om, const float *fwd_xmx, float *dpb, float *bck_xmx, float *ret_sc, int *own_scales) { return backward_engine(dsq, L, om, fwd_xmx, dpb, bck_xmx, ret_sc, own_scales); }

/* ------------------------------------------------------------------ generic scalar Viterbi / Forward */

#define G_TSC(gm,k,s) ((gm)->tsc[(k) * BO_NTRANS + (s)])
#define G_MSC(gm,k,x) ((gm)->rsc[(size_t)(x) * ((gm)->M + 1) * 2 + (k) * 2])
#define G_ISC(gm,k,x) ((gm)->rsc[(size_t)(x) * ((gm)->M + 1) * 2 + (k) * 2 + 1])

static float fmax2(float a, float b) { return a > b ? a : b; }

static int generic_dp(const uint8_t *dsq, int L, const bo_profile *gm, int viterbi, float *ret_sc)   /* generic_viterbi.c / generic_fwdback.c */
{
  int M = gm->M;
  size_t W = (size_t) M + 1;
  float *Mp = malloc(sizeof(float) * W * 6);
  float *Ip = Mp + W, *Dp = Ip + W, *Mc = Dp + W, *Ic = Mc + W, *Dc = Ic + W;
  float *alloc = Mp;
  float xN = 0.f, xB = gm->xsc[BO_XN][BO_MOVE], xE = -INFINITY, xJ = -INFINITY, xC = -INFINITY;
  for (int k = 0; k <= M; k++) Mp[k] = Ip[k] = Dp[k] = -INFINITY;
#define COMB(a,b) (viterbi ? fmax2((a),(b)) : bo_flogsum((a),(b)))
  for (int i = 1; i <= L; i++) {
    int x = dsq[i];
    Mc[0] = Ic[0] = Dc[0] = -INFINITY;
    xE = -INFINITY;
    for (int k = 1; k <= M; k++) {
      float sc = COMB(COMB(Mp[k-1] + G_TSC(gm, k-1, BO_MM), Ip[k-1] + G_TSC(gm, k-1, BO_IM)),
                      COMB(xB      + G_TSC(gm, k-1, BO_BM), Dp[k-1] + G_TSC(gm, k-1, BO_DM)));
      Mc[k] = sc + G_MSC(gm, k, x);
      if (k < M) Ic[k] = COMB(Mp[k] + G_TSC(gm, k, BO_MI), Ip[k] + G_TSC(gm, k, BO_II)) + G_ISC(gm, k, x);
      else       Ic[k] = -INFINITY;
      Dc[k] = COMB(Mc[k-1] + G_TSC(gm, k-1, BO_MD), Dc[k-1] + G_TSC(gm, k-1, BO_DD));
      xE = COMB(xE, Mc[k]);                   /* local mode: esc = 0 */
      if (!viterbi || k == M) xE = COMB(xE, Dc[k]);
    }
    xJ = COMB(xJ + gm->xsc[BO_XJ][BO_LOOP], xE + gm->xsc[BO_XE][BO_LOOP]);
    xC = COMB(xC + gm->xsc[BO_XC][BO_LOOP], xE + gm->xsc[BO_XE][BO_MOVE]);
    xN = xN + gm->xsc[BO_XN][BO_LOOP];
    xB = COMB(xN + gm->xsc[BO_XN][BO_MOVE], xJ + gm->xsc[BO_XJ][BO_MOVE]);
    float *tmp;
    tmp = Mp; Mp = Mc; Mc = tmp;
    tmp = Ip; Ip = Ic; Ic = tmp;
    tmp = Dp; Dp = Dc; Dc = tmp;
  }
#undef COMB
  free(alloc);
  *ret_sc = xC + gm->xsc[BO_XC][BO_MOVE];
  return BO_OK;
}

int bo_gviterbi(const uint8_t *dsq, int L, const bo_profile *gm, float *ret_sc) { return generic_dp(dsq, L, gm, 1, ret_sc); }
int bo_gforward(const uint8_t *dsq, int L, const bo_profile *gm, float *ret_sc) { return generic_dp(dsq, L, gm, 0, ret_sc); }

static bo_profile *profile_clone(const bo_profile *gm)
{
  bo_profile *c = malloc(sizeof *c);
  *c = *gm;
  size_t nt = (size_t)(gm->M + 1) * BO_NTRANS, nr = (size_t) BO_KP_AMINO * (gm->M + 1) * 2;
  c->tsc = malloc(sizeof(float) * nt); memcpy(c->tsc, gm->tsc, sizeof(float) * nt);
  c->rsc = malloc(sizeof(float) * nr); memcpy(c->rsc, gm->rsc, sizeof(float) * nr);
  return c;
}

bo_profile *bo_profile_same_as_mf(const bo_oprofile *om, const bo_profile *src)   /* p7_oprofile.c:2141-2171 */
{
  bo_profile *gm = profile_clone(src);
  int M = gm->M;
  float tbm = roundf(om->scale_b * (float)(log(2.0f / ((float) M * (float) (M + 1)))));
  for (int i = 0; i < BO_NTRANS * M; i++) gm->tsc[i] = -INFINITY;
  for (int k = 1; k < M; k++) G_TSC(gm, k, BO_MM) = 0.0f;
  for (int k = 0; k < M; k++) G_TSC(gm, k, BO_BM) = tbm;
  for (int x = 0; x < BO_KP_AMINO; x++)
    for (int k = 0; k <= M; k++) {
      float v = G_MSC(gm, k, x);
      G_MSC(gm, k, x) = (v <= -INFINITY) ? -INFINITY : roundf(om->scale_b * v);
      G_ISC(gm, k, x) = 0;
    }
  for (int k = 0; k < 4; k++) for (int x = 0; x < 2; x++)
    gm->xsc[k][x] = (gm->xsc[k][x] <= -INFINITY) ? -INFINITY : roundf(om->scale_b * gm->xsc[k][x]);
  gm->xsc[BO_XN][BO_LOOP] = gm->xsc[BO_XJ][BO_LOOP] = gm->xsc[BO_XC][BO_LOOP] = 0;
  return gm;
}

bo_profile *bo_profile_same_as_vf(const bo_oprofile *om, const bo_profile *src)   /* p7_oprofile.c:2202-2233 */
{
  bo_profile *gm = profile_clone(src);
  int M = gm->M;
  for (int x = 0; x < M * BO_NTRANS; x++)
    gm->tsc[x] = (gm->tsc[x] <= -INFINITY) ? -INFINITY : roundf(om->scale_w * gm->tsc[x]);
  for (int x = BO_II; x < M * BO_NTRANS; x += BO_NTRANS) if (gm->tsc[x] == 0.0) gm->tsc[x] = -1.0;
  for (int x = 0; x < BO_KP_AMINO; x++)
    for (int k = 0; k <= M; k++) {
      float v = G_MSC(gm, k, x);
      G_MSC(gm, k, x) = (v <= -INFINITY) ? -INFINITY : roundf(om->scale_w * v);
      G_ISC(gm, k, x) = 0.0;
    }
  for (int k = 0; k < 4; k++) for (int x = 0; x < 2; x++)
    gm->xsc[k][x] = (gm->xsc[k][x] <= -INFINITY) ? -INFINITY : roundf(om->scale_w * gm->xsc[k][x]);
  gm->xsc[BO_XN][BO_LOOP] = gm->xsc[BO_XJ][BO_LOOP] = gm->xsc[BO_XC][BO_LOOP] = 0.0;
  return gm;
}
