/* fs_fwdback.c -- frameshift-aware Forward / Backward in log space.  ORACLE (test infra only).
 *
 * Restates the scalar reference recursions of generic_fwdback_frameshift.c with the table-driven
 * p7_FLogsum (logsum.c) and, row for row, the same logsum association, because with a truncating
 * lookup table the association changes the low bits:
 *    bo_gforward_fs           <- p7_GForward_Frameshift                generic_fwdback_frameshift.c:64-413
 *    bo_gforward_parser_fs3   <- p7_GForwardParser_Frameshift_3Codons  :451-622
 *    bo_gbackward_fs          <- p7_GBackward_Frameshift               :1035-1392
 *    bo_gbackward_parser_fs3  <- p7_GBackwardParser_Frameshift_3Codons :1422-1737
 * Matrices are full (L+1 rows) here even for the "parser" variants: the ring buffers of the
 * reference (PARSER_ROWS_FWD/BWD) are a memory optimisation with no numerical effect.
 *
 * Quirk kept on request (c5_compat=1): the reference's generic 5-codon Forward reads the intermediate
 * ring slot (i-5)%5, which is the slot just written for row i (generic_fwdback_frameshift.c:324,341),
 * while the SIMD kernel the pipeline actually calls uses row i-4 (impl_sse/fwdback_fs.c:1464-1468).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_oracle.h"

#define LS bo_flogsum
#define NINF (-INFINITY)

bo_gmx *bo_gmx_create(int M, int nrows, int L, int nscells)
{
  bo_gmx *gx = calloc(1, sizeof *gx);
  gx->M = M; gx->L = L; gx->nrows = nrows; gx->nscells = nscells;
  gx->dp  = malloc(sizeof(float) * (size_t) nrows * (M + 1) * nscells);
  gx->xmx = malloc(sizeof(float) * (size_t)(L + 1) * BO_NXCELLS);
  return gx;
}
void bo_gmx_free(bo_gmx *gx) { if (gx) { free(gx->dp); free(gx->xmx); free(gx); } }

/* codon / quasi-codon row indices, hmmer.h:292-316; the LAST argument is the most recent nucleotide */
static inline int minidx(int a, int b) { return a < b ? a : b; }
#define C1_5(x)          ((x) * 341)
#define C2_5(w,x)        ((x) * 341 + (w) * 85 + 1)
#define C3_5(v,w,x)      ((x) * 341 + (w) * 85 + (v) * 21 + 2)
#define C4_5(u,v,w,x)    ((x) * 341 + (w) * 85 + (v) * 21 + (u) * 5 + 3)
#define C5_5(t,u,v,w,x)  ((x) * 341 + (w) * 85 + (v) * 21 + (u) * 5 + (t) + 4)
#define C2_3(w,x)        ((x) * 84 + (w) * 21)
#define C3_3(v,w,x)      ((x) * 84 + (w) * 21 + (v) * 5 + 1)
#define C4_3(u,v,w,x)    ((x) * 84 + (w) * 21 + (v) * 5 + (u) + 2)

#define TSC(s,k) (tsc[(k) * BO_NTRANS + (s)])
#define RSC(c)   (gm->rsc + (size_t)(c) * ((size_t) M + 1))

static inline int nuc5(uint8_t d) { return d < 4 ? d : BO_MAXCODONS5; }
static inline int nuc3(uint8_t d) { return d < 4 ? d : BO_MAXCODONS3; }

/* ------------------------------------------------------------------ Forward, 5 codon lengths, full matrix */

#define MF(i,k,c) BO_DP(gx,i,k,BO_GM + (c))       /* C0 = total, C1..C5 = per codon length */
#define IF(i,k)   BO_DP(gx,i,k,BO_GI)
#define DF(i,k)   BO_DP(gx,i,k,BO_GD)
#define XF(i,s)   BO_X(gx,i,s)

int bo_gforward_fs(const uint8_t *dsq, int L, const bo_fs_profile *gm, bo_gmx *gx, int c5_compat, float *ret_sc)
{
  const float *tsc = gm->tsc;
  int M = gm->M;
  if (gm->codon_lengths != 5 || gx->nscells != BO_NSCELLS_FS || gx->nrows < L + 1 || L < 5) return BO_EINVAL;
  float *ivx = malloc(sizeof(float) * 5 * (size_t)(M + 1));          /* ivx[slot*(M+1)+k] */
#define IV(s,k) ivx[(size_t)(s) * (M + 1) + (k)]
  for (int s = 0; s < 5; s++) for (int k = 0; k <= M; k++) IV(s,k) = NINF;
  const float *x = &gm->xsc[0][0];
  const float tNL = x[BO_XN*2+BO_LOOP], tNM = x[BO_XN*2+BO_MOVE], tJL = x[BO_XJ*2+BO_LOOP], tJM = x[BO_XJ*2+BO_MOVE];
  const float tCL = x[BO_XC*2+BO_LOOP], tCM = x[BO_XC*2+BO_MOVE], tEL = x[BO_XE*2+BO_LOOP], tEM = x[BO_XE*2+BO_MOVE];
  const float esc = 0.0f;                                             /* local mode */
  int t = BO_MAXCODONS5, u = BO_MAXCODONS5, v = BO_MAXCODONS5, w = BO_MAXCODONS5, xx = BO_MAXCODONS5;

  /* row 0 (:86-93) */
  XF(0,BO_GN) = 0.f; XF(0,BO_GB) = tNM; XF(0,BO_GE) = XF(0,BO_GJ) = XF(0,BO_GC) = NINF;
  for (int k = 0; k <= M; k++) { for (int c = 0; c < 6; c++) MF(0,k,c) = NINF; IF(0,k) = DF(0,k) = NINF; }

  for (int i = 1; i <= L; i++) {
    t = u; u = v; v = w; w = xx; xx = nuc5(dsq[i]);
    const float *r1 = RSC(minidx(C1_5(xx), BO_DEGEN5_QC2));
    const float *r2 = RSC(minidx(C2_5(w, xx), BO_DEGEN5_QC1));
    const float *r3 = RSC(minidx(C3_5(v, w, xx), BO_DEGEN5_C));
    const float *r4 = RSC(minidx(C4_5(u, v, w, xx), BO_DEGEN5_QC1));
    const float *r5 = RSC(minidx(C5_5(t, u, v, w, xx), BO_DEGEN5_QC2));
    const int s1 = i % 5, s2 = (i - 1) % 5, s3 = (i - 2) % 5, s4 = (i - 3) % 5;
    const int s5 = c5_compat ? s1 : (i - 4 + 5) % 5;
    for (int c = 0; c < 6; c++) MF(i,0,c) = NINF;
    IF(i,0) = DF(i,0) = NINF;
    XF(i,BO_GE) = NINF;

    if (i <= 2) {                                                     /* rows 1,2 (:95-167) */
      XF(i,BO_GN) = 0.f; XF(i,BO_GB) = tNM;
      for (int k = 1; k <= M; k++) {
        IV(i,k) = XF(i-1,BO_GB) + TSC(BO_BM,k-1);
        MF(i,k,1) = IV(i,k) + r1[k];
        MF(i,k,2) = (i == 2) ? IV(1,k) + r2[k] : NINF;
        MF(i,k,3) = MF(i,k,4) = MF(i,k,5) = NINF;
        MF(i,k,0) = (i == 2) ? LS(MF(i,k,1), MF(i,k,2)) : MF(i,k,1);
        IF(i,k) = NINF;
        DF(i,k) = LS(MF(i,k-1,0) + TSC(BO_MD,k-1), DF(i,k-1) + TSC(BO_DD,k-1));
        XF(i,BO_GE) = LS(MF(i,k,0) + esc, LS(DF(i,k) + esc, XF(i,BO_GE)));
      }
      XF(i,BO_GJ) = XF(i,BO_GE) + tEL;
      XF(i,BO_GC) = XF(i,BO_GE) + tEM;
      continue;
    }

    for (int k = 1; k <= M; k++) {
      IV(s1,k) = LS(MF(i-1,k-1,0) + TSC(BO_MM,k-1),
                 LS(IF(i-1,k-1)   + TSC(BO_IM,k-1),
                 LS(DF(i-1,k-1)   + TSC(BO_DM,k-1),
                    XF(i-1,BO_GB) + TSC(BO_BM,k-1))));
      MF(i,k,1) = IV(s1,k) + r1[k];
      MF(i,k,2) = IV(s2,k) + r2[k];
      MF(i,k,3) = IV(s3,k) + r3[k];
      if (i < 5) {                                                    /* rows 3,4 (:171-278) */
        MF(i,k,4) = (i == 4) ? IV(s4,k) + r4[k] : NINF;
        MF(i,k,5) = NINF;
        MF(i,k,0) = LS(MF(i,k,1), LS(MF(i,k,2), LS(MF(i,k,3), MF(i,k,4))));
      } else {                                                        /* main recursion (:281-401) */
        MF(i,k,4) = IV(s4,k) + r4[k];
        MF(i,k,5) = IV(s5,k) + r5[k];
        MF(i,k,0) = LS(LS(MF(i,k,1), LS(MF(i,k,2), MF(i,k,3))), LS(MF(i,k,4), MF(i,k,5)));
      }
      IF(i,k) = (k < M) ? LS(MF(i-3,k,0) + TSC(BO_MI,k), IF(i-3,k) + TSC(BO_II,k)) : NINF;
      DF(i,k) = LS(MF(i,k-1,0) + TSC(BO_MD,k-1), DF(i,k-1) + TSC(BO_DD,k-1));
      if (k < M) XF(i,BO_GE) = LS(MF(i,k,0) + esc, LS(DF(i,k) + esc, XF(i,BO_GE)));
      else if (i < 5) XF(i,BO_GE) = LS(MF(i,M,0), LS(DF(i,M), XF(i,BO_GE)));
      else            XF(i,BO_GE) = LS(LS(MF(i,M,0), DF(i,M)), XF(i,BO_GE));
    }
    XF(i,BO_GJ) = LS(XF(i-3,BO_GJ) + tJL, XF(i,BO_GE) + tEL);
    XF(i,BO_GC) = LS(XF(i-3,BO_GC) + tCL, XF(i,BO_GE) + tEM);
    XF(i,BO_GN) =    XF(i-3,BO_GN) + tNL;
    XF(i,BO_GB) = LS(XF(i,BO_GN) + tNM, XF(i,BO_GJ) + tJM);
  }
  if (ret_sc) *ret_sc = LS(XF(L,BO_GC), LS(XF(L-1,BO_GC) + tCL, XF(L-2,BO_GC) + tCL)) + tCM;
  free(ivx);
  gx->M = M; gx->L = L;
  return BO_OK;
#undef IV
}

/* ------------------------------------------------------------------ Forward parser, 3 codon lengths */

#define M3(i,k) BO_DP(gx,i,k,BO_GM)
#define I3(i,k) BO_DP(gx,i,k,BO_GI)
#define D3(i,k) BO_DP(gx,i,k,BO_GD)

int bo_gforward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm, bo_gmx *gx, float *ret_sc)
{
  const float *tsc = gm->tsc;
  int M = gm->M;
  if (gm->codon_lengths != 3 || gx->nscells != BO_NSCELLS || gx->nrows < L + 1 || L < 3) return BO_EINVAL;
  float *ivx = malloc(sizeof(float) * 3 * (size_t)(M + 1));
#define IV(s,k) ivx[(size_t)(s) * (M + 1) + (k)]
  for (int s = 0; s < 3; s++) for (int k = 0; k <= M; k++) IV(s,k) = NINF;
  const float *x = &gm->xsc[0][0];
  const float tNL = x[BO_XN*2+BO_LOOP], tNM = x[BO_XN*2+BO_MOVE], tJL = x[BO_XJ*2+BO_LOOP], tJM = x[BO_XJ*2+BO_MOVE];
  const float tCL = x[BO_XC*2+BO_LOOP], tCM = x[BO_XC*2+BO_MOVE], tEL = x[BO_XE*2+BO_LOOP], tEM = x[BO_XE*2+BO_MOVE];
  const float esc = 0.0f;

  for (int i = 0; i < 2; i++) {                                       /* rows 0,1 (:476-482) */
    XF(i,BO_GN) = 0.f; XF(i,BO_GB) = tNM; XF(i,BO_GE) = XF(i,BO_GJ) = XF(i,BO_GC) = NINF;
    for (int k = 0; k <= M; k++) M3(i,k) = I3(i,k) = D3(i,k) = NINF;
  }
  int u = BO_MAXCODONS3, v = BO_MAXCODONS3, w = nuc3(dsq[1]), xx = nuc3(dsq[2]);
  {                                                                   /* row 2 (:484-517) */
    const float *r2 = RSC(minidx(C2_3(w, xx), BO_DEGEN3_QC1));
    XF(2,BO_GE) = NINF;
    M3(2,0) = I3(2,0) = D3(2,0) = NINF;
    for (int k = 1; k <= M; k++) {
      IV(2,k) = XF(0,BO_GB) + TSC(BO_BM,k-1);
      M3(2,k) = IV(2,k) + r2[k];
      I3(2,k) = NINF;
      D3(2,k) = LS(M3(2,k-1) + TSC(BO_MD,k-1), D3(2,k-1) + TSC(BO_DD,k-1));
      XF(2,BO_GE) = LS(M3(2,k) + esc, LS(D3(2,k) + esc, XF(2,BO_GE)));
    }
    XF(2,BO_GN) = 0.f;
    XF(2,BO_GJ) = XF(2,BO_GE) + tEL;
    XF(2,BO_GC) = XF(2,BO_GE) + tEM;
    XF(2,BO_GB) = LS(XF(2,BO_GN) + tNM, XF(2,BO_GJ) + tJM);
  }
  for (int i = 3; i <= L; i++) {                                      /* main recursion (:520-611) */
    u = v; v = w; w = xx; xx = nuc3(dsq[i]);
    const float *r2 = RSC(minidx(C2_3(w, xx), BO_DEGEN3_QC1));
    const float *r3 = RSC(minidx(C3_3(v, w, xx), BO_DEGEN3_C));
    const float *r4 = RSC(minidx(C4_3(u, v, w, xx), BO_DEGEN3_QC1));
    const int s2 = i % 3, s3 = (i - 1) % 3, s4 = (i - 2) % 3;
    M3(i,0) = I3(i,0) = D3(i,0) = NINF;
    XF(i,BO_GE) = NINF;
    for (int k = 1; k <= M; k++) {
      IV(s2,k) = LS(M3(i-2,k-1) + TSC(BO_MM,k-1),
                 LS(I3(i-2,k-1) + TSC(BO_IM,k-1),
                 LS(D3(i-2,k-1) + TSC(BO_DM,k-1),
                    XF(i-2,BO_GB) + TSC(BO_BM,k-1))));
      M3(i,k) =             IV(s2,k) + r2[k];
      M3(i,k) = LS(M3(i,k), IV(s3,k) + r3[k]);
      M3(i,k) = LS(M3(i,k), IV(s4,k) + r4[k]);
      I3(i,k) = (k < M) ? LS(M3(i-3,k) + TSC(BO_MI,k), I3(i-3,k) + TSC(BO_II,k)) : NINF;
      D3(i,k) = LS(M3(i,k-1) + TSC(BO_MD,k-1), D3(i,k-1) + TSC(BO_DD,k-1));
      if (k < M) XF(i,BO_GE) = LS(M3(i,k) + esc, LS(D3(i,k) + esc, XF(i,BO_GE)));
      else       XF(i,BO_GE) = LS(M3(i,M), LS(D3(i,M), XF(i,BO_GE)));
    }
    XF(i,BO_GN) =    XF(i-3,BO_GN) + tNL;
    XF(i,BO_GJ) = LS(XF(i-3,BO_GJ) + tJL, XF(i,BO_GE) + tEL);
    XF(i,BO_GC) = LS(XF(i-3,BO_GC) + tCL, XF(i,BO_GE) + tEM);
    XF(i,BO_GB) = LS(XF(i,BO_GN) + tNM, XF(i,BO_GJ) + tJM);
  }
  if (ret_sc) *ret_sc = LS(XF(L,BO_GC), LS(XF(L-1,BO_GC) + tCL, XF(L-2,BO_GC) + tCL)) + tCM;
  free(ivx);
  gx->M = M; gx->L = L;
  return BO_OK;
#undef IV
}

/* ------------------------------------------------------------------ Backward (both codon systems)
 * One engine: ncod = 5 -> p7_GBackward_Frameshift, ncod = 3 -> p7_GBackwardParser_Frameshift_3Codons.
 * In Backward the rolling nucleotide window is filled from the 3' side, so the index macros are
 * called with reversed argument order (generic_fwdback_frameshift.c:1260-1270, :1614-1621).
 * The main-cell matrix has 3 cells (M,I,D) per node. */
static int backward_engine(const uint8_t *dsq, int L, const bo_fs_profile *gm, bo_gmx *gx, int ncod, float *ret_sc)
{
  const float *tsc = gm->tsc;
  int M = gm->M;
  if (gm->codon_lengths != ncod || gx->nscells != BO_NSCELLS || gx->nrows < L + 1 || L < 5) return BO_EINVAL;
  float *ivx = malloc(sizeof(float) * (size_t)(M + 2));
  const float *x = &gm->xsc[0][0];
  const float tNL = x[BO_XN*2+BO_LOOP], tNM = x[BO_XN*2+BO_MOVE], tJL = x[BO_XJ*2+BO_LOOP], tJM = x[BO_XJ*2+BO_MOVE];
  const float tCL = x[BO_XC*2+BO_LOOP], tCM = x[BO_XC*2+BO_MOVE], tEL = x[BO_XE*2+BO_LOOP], tEM = x[BO_XE*2+BO_MOVE];
  const float esc = 0.0f;
  const int five = (ncod == 5);
  const int DEG = five ? BO_MAXCODONS5 : BO_MAXCODONS3;
  for (int k = 0; k <= M + 1; k++) ivx[k] = NINF;

  /* rows with no emitted codon yet: L for fs5 (:1054-1073); L and L-1 for fs3 (:1442-1465) */
  const int first_emit = five ? L - 1 : L - 2;
  for (int i = L; i > first_emit; i--) {
    XF(i,BO_GC) = (i == L) ? tCM : tCL + tCM;
    XF(i,BO_GJ) = XF(i,BO_GB) = XF(i,BO_GN) = NINF;
    XF(i,BO_GE) = XF(i,BO_GC) + tEM;
    M3(i,M) = D3(i,M) = XF(i,BO_GE);
    I3(i,M) = NINF;
    for (int k = M - 1; k >= 1; k--) {
      M3(i,k) = LS(XF(i,BO_GE) + esc, D3(i,k+1) + TSC(BO_MD,k));
      D3(i,k) = LS(XF(i,BO_GE) + esc, D3(i,k+1) + TSC(BO_DD,k));
      I3(i,k) = NINF;
    }
    M3(i,0) = I3(i,0) = D3(i,0) = NINF;
  }

  int t = DEG, u = DEG, v = DEG, w = DEG, xx = DEG;
  if (!five) { w = dsq[L] < 4 ? dsq[L] : DEG; }                       /* fs3 primes w with x_L (:1470) */
  for (int i = first_emit; i >= 0; i--) {
    /* slide the window: the new nucleotide is x_{i+1}, the FIRST base of every codon starting at i+1 */
    if (five || i < first_emit) { t = u; u = v; v = w; w = xx; }
    xx = dsq[i+1] < 4 ? dsq[i+1] : DEG;
    if (!five && i == first_emit) { /* w already holds x_L, xx = x_{L-1} */ }
    const int avail = L - i;                                          /* nucleotides to the right of i */
    const float *r1 = NULL, *r2 = NULL, *r3 = NULL, *r4 = NULL, *r5 = NULL;
    if (five) {
      r1 = RSC(minidx(C1_5(xx), BO_DEGEN5_QC2));
      if (avail >= 2) r2 = RSC(minidx(C2_5(xx, w), BO_DEGEN5_QC1));
      if (avail >= 3) r3 = RSC(minidx(C3_5(xx, w, v), BO_DEGEN5_C));
      if (avail >= 4) r4 = RSC(minidx(C4_5(xx, w, v, u), BO_DEGEN5_QC1));
      if (avail >= 5) r5 = RSC(minidx(C5_5(xx, w, v, u, t), BO_DEGEN5_QC2));
    } else {
      if (avail >= 2) r2 = RSC(minidx(C2_3(xx, w), BO_DEGEN3_QC1));
      if (avail >= 3) r3 = RSC(minidx(C3_3(xx, w, v), BO_DEGEN3_C));
      if (avail >= 4) r4 = RSC(minidx(C4_3(xx, w, v, u), BO_DEGEN3_QC1));
    }
    const int full = (i <= L - 5);                                    /* main recursion and the i==0 row nest the logsums; rows L-1..L-4 accumulate left to right */

    /* ivx[k] = logsum_c [ M(i+c,k) + e_c(k) ]  and  B(i) */
    for (int k = 1; k <= M; k++) {
      float a;
      if (five) {
        if (full) a = LS(M3(i+1,k) + r1[k], LS(M3(i+2,k) + r2[k], LS(M3(i+3,k) + r3[k], LS(M3(i+4,k) + r4[k], M3(i+5,k) + r5[k]))));
        else {
          a = M3(i+1,k) + r1[k];
          if (r2) a = LS(a, M3(i+2,k) + r2[k]);
          if (r3) a = LS(a, M3(i+3,k) + r3[k]);
          if (r4) a = LS(a, M3(i+4,k) + r4[k]);
        }
      } else {
        if (full) a = LS(M3(i+2,k) + r2[k], LS(M3(i+3,k) + r3[k], M3(i+4,k) + r4[k]));
        else {
          a = M3(i+2,k) + r2[k];
          if (r3) a = LS(a, M3(i+3,k) + r3[k]);
          if (r4) a = LS(a, M3(i+4,k) + r4[k]);                     /* row L-4 only (:1552-1553) */
        }
      }
      ivx[k] = a;
      if (k == 1) XF(i,BO_GB) = a + TSC(BO_BM,0);
      else        XF(i,BO_GB) = LS(XF(i,BO_GB), a + TSC(BO_BM,k-1));
    }
    if (i == 0) break;

    const int tail = (avail < 3);                                     /* rows L-1, L-2: no i+3 row to loop to */
    if (tail) {
      XF(i,BO_GJ) = XF(i,BO_GB) + tJM;
      XF(i,BO_GN) = XF(i,BO_GB) + tNM;
      XF(i,BO_GC) = tCL + tCM;
    } else {
      XF(i,BO_GJ) = LS(XF(i+3,BO_GJ) + tJL, XF(i,BO_GB) + tJM);
      XF(i,BO_GC) =    XF(i+3,BO_GC) + tCL;
      XF(i,BO_GN) = LS(XF(i+3,BO_GN) + tNL, XF(i,BO_GB) + tNM);
    }
    XF(i,BO_GE) = LS(XF(i,BO_GJ) + tEL, XF(i,BO_GC) + tEM);
    M3(i,M) = D3(i,M) = XF(i,BO_GE);
    I3(i,M) = NINF;
    for (int k = M - 1; k >= 1; k--) {
      if (tail) {                                                     /* :1100-1111, :1148-1159, :1503-1514 */
        M3(i,k) = LS(D3(i,k+1) + TSC(BO_MD,k), LS(ivx[k+1] + TSC(BO_MM,k), XF(i,BO_GE) + esc));
        D3(i,k) = LS(LS(XF(i,BO_GE) + esc, D3(i,k+1) + TSC(BO_DD,k)), ivx[k+1] + TSC(BO_DM,k));
        I3(i,k) = ivx[k+1] + TSC(BO_IM,k);
      } else if (!five && !full) {                                    /* fs3 rows L-3, L-4 (:1583-1596) */
        M3(i,k) = LS(D3(i,k+1) + TSC(BO_MD,k), LS(I3(i+3,k) + TSC(BO_MI,k), LS(ivx[k+1] + TSC(BO_MM,k), XF(i,BO_GE) + esc)));
        D3(i,k) = LS(D3(i,k+1) + TSC(BO_DD,k), LS(XF(i,BO_GE) + esc, ivx[k+1] + TSC(BO_DM,k)));
        I3(i,k) = LS(I3(i+3,k) + TSC(BO_II,k), ivx[k+1] + TSC(BO_IM,k));
      } else {                                                        /* :1225-1238, :1306-1319, :1660-1673 */
        M3(i,k) = LS(LS(D3(i,k+1) + TSC(BO_MD,k), LS(I3(i+3,k) + TSC(BO_MI,k), ivx[k+1] + TSC(BO_MM,k))), XF(i,BO_GE) + esc);
        D3(i,k) = LS(LS(XF(i,BO_GE) + esc, D3(i,k+1) + TSC(BO_DD,k)), ivx[k+1] + TSC(BO_DM,k));
        I3(i,k) = LS(I3(i+3,k) + TSC(BO_II,k), ivx[k+1] + TSC(BO_IM,k));
      }
    }
    M3(i,0) = I3(i,0) = D3(i,0) = NINF;
  }
  /* i = 0 (:1375-1385, :1721-1731) */
  XF(0,BO_GJ) = XF(0,BO_GC) = XF(0,BO_GE) = NINF;
  XF(0,BO_GN) = LS(XF(3,BO_GN) + tNL, XF(0,BO_GB) + tNM);
  for (int k = 0; k <= M; k++) M3(0,k) = I3(0,k) = D3(0,k) = NINF;
  if (ret_sc) *ret_sc = LS(XF(0,BO_GN), LS(XF(1,BO_GN), XF(2,BO_GN)));
  free(ivx);
  gx->M = M; gx->L = L;
  return BO_OK;
}

int bo_gbackward_fs(const uint8_t *dsq, int L, const bo_fs_profile *gm5, bo_gmx *gx, float *ret_sc)
{
  return backward_engine(dsq, L, gm5, gx, 5, ret_sc);
}
int bo_gbackward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc)
{
  return backward_engine(dsq, L, gm3, gx, 3, ret_sc);
}
