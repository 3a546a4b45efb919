/* fs_post.c -- posterior decoding, optimal-accuracy fill and null2 for frameshift envelopes.
 * ORACLE (test infra only).  Restates
 *    bo_gdecoding_fs  <- p7_GDecoding_Frameshift        generic_decoding_frameshift.c:36-156
 *    bo_goptacc_fs    <- p7_GOptimalAccuracy_Frameshift generic_optacc_frameshift.c:53-324
 *    bo_gnull2_fs     <- p7_GNull2_fs_ByExpectation     generic_null2_frameshift.c:46-125
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

#define NINF (-INFINITY)
#define LS bo_flogsum

/* Posterior decoding IN PLACE: <fwd> (8 cells per node) is overwritten with posteriors, as the reference does. */
int bo_gdecoding_fs(const bo_fs_profile *gm, bo_gmx *fwd, const bo_gmx *bck)
{
  int L = fwd->L, M = gm->M;
  if (fwd->nscells != BO_NSCELLS_FS || bck->nscells != BO_NSCELLS) return BO_EINVAL;
  float overall = LS(BO_X(bck,0,BO_GN), LS(BO_X(bck,1,BO_GN), BO_X(bck,2,BO_GN)));
  float N0 = BO_X(fwd,0,BO_GN), J0 = BO_X(fwd,0,BO_GJ), C0 = BO_X(fwd,0,BO_GC);
  float N1 = 0, N2 = 0, N3 = 0, J1 = 0, J2 = 0, J3 = 0, C1 = 0, C2 = 0, C3 = 0;
  const float tNL = gm->xsc[BO_XN][BO_LOOP], tJL = gm->xsc[BO_XJ][BO_LOOP], tCL = gm->xsc[BO_XC][BO_LOOP];
  for (int s = 0; s < BO_NXCELLS; s++) BO_X(fwd,0,s) = 0.0f;
  for (int k = 0; k <= M; k++) for (int c = 0; c < BO_NSCELLS_FS; c++) BO_DP(fwd,0,k,c) = 0.0f;

  for (int i = 1; i <= L; i++) {
    N3 = N2; N2 = N1; N1 = N0;
    J3 = J2; J2 = J1; J1 = J0;
    C3 = C2; C2 = C1; C1 = C0;
    float denom = 0.0f;
    for (int c = 0; c < BO_NSCELLS_FS; c++) BO_DP(fwd,i,0,c) = 0.0f;
    for (int k = 1; k <= M; k++) {
      float b = BO_DP(bck,i,k,BO_GM);
      for (int c = 0; c < 6; c++) BO_DP(fwd,i,k,BO_GM + c) = expf(BO_DP(fwd,i,k,BO_GM + c) + b - overall);
      denom += BO_DP(fwd,i,k,BO_GM);
      if (k < M) { BO_DP(fwd,i,k,BO_GI) = expf(BO_DP(fwd,i,k,BO_GI) + BO_DP(bck,i,k,BO_GI) - overall); denom += BO_DP(fwd,i,k,BO_GI); }
      else         BO_DP(fwd,i,k,BO_GI) = 0.0f;
      BO_DP(fwd,i,k,BO_GD) = 0.0f;
    }
    BO_X(fwd,i,BO_GE) = 0.0f; BO_X(fwd,i,BO_GB) = 0.0f;
    N0 = BO_X(fwd,i,BO_GN); J0 = BO_X(fwd,i,BO_GJ); C0 = BO_X(fwd,i,BO_GC);
    if (i > 2) {
      BO_X(fwd,i,BO_GN) = expf(N3 + BO_X(bck,i,BO_GN) + tNL - overall);
      BO_X(fwd,i,BO_GC) = expf(C3 + BO_X(bck,i,BO_GC) + tCL - overall);
      BO_X(fwd,i,BO_GJ) = expf(J3 + BO_X(bck,i,BO_GJ) + tJL - overall);
      denom += BO_X(fwd,i,BO_GN) + BO_X(fwd,i,BO_GJ) + BO_X(fwd,i,BO_GC);
    } else {
      BO_X(fwd,i,BO_GN) = expf(BO_X(bck,i,BO_GN) - overall);
      BO_X(fwd,i,BO_GC) = 0.0f; BO_X(fwd,i,BO_GJ) = 0.0f;
      denom += BO_X(fwd,i,BO_GN);
    }
    denom = (float)(1.0 / denom);
    for (int k = 1; k <= M; k++) {
      for (int c = 0; c < 6; c++) BO_DP(fwd,i,k,BO_GM + c) *= denom;
      if (k < M) BO_DP(fwd,i,k,BO_GI) *= denom;
    }
    BO_X(fwd,i,BO_GN) *= denom; BO_X(fwd,i,BO_GC) *= denom; BO_X(fwd,i,BO_GJ) *= denom;
  }
  return BO_OK;
}

/* Optimal accuracy fill.  TSCDELTA (generic_optacc_frameshift.c:21): 1.0 for a possible transition, FLT_MIN otherwise. */
int bo_goptacc_fs(const bo_fs_profile *gm, const bo_gmx *pp, bo_gmx *gx, float *ret_e)
{
  int L = pp->L, M = gm->M;
  const float *tsc = gm->tsc;
  if (pp->nscells != BO_NSCELLS_FS || gx->nscells != BO_NSCELLS || gx->nrows < L + 1) return BO_EINVAL;
#define DELTA(s,k) ((tsc[(k) * BO_NTRANS + (s)] == NINF) ? FLT_MIN : 1.0f)
#define XD(st,tr)  ((gm->xsc[st][tr] == NINF) ? FLT_MIN : 1.0f)
#define OM(i,k) BO_DP(gx,i,k,BO_GM)
#define OI(i,k) BO_DP(gx,i,k,BO_GI)
#define OD(i,k) BO_DP(gx,i,k,BO_GD)
#define OX(i,s) BO_X(gx,i,s)
#define PPM(i,k,c) BO_DP(pp,i,k,BO_GM + (c))
#define PPI(i,k)   BO_DP(pp,i,k,BO_GI)
  const float nn = XD(BO_XN,BO_LOOP), jj = XD(BO_XJ,BO_LOOP), cc = XD(BO_XC,BO_LOOP);
  const float nb = XD(BO_XN,BO_MOVE), jb = XD(BO_XJ,BO_MOVE), ej = XD(BO_XE,BO_LOOP), ec = XD(BO_XE,BO_MOVE);
  const float esc = 1.0f;
  gx->M = M; gx->L = L;

  OX(0,BO_GN) = 0.f; OX(0,BO_GB) = 0.f; OX(0,BO_GE) = OX(0,BO_GC) = OX(0,BO_GJ) = NINF;
  for (int k = 0; k <= M; k++) OM(0,k) = OI(0,k) = OD(0,k) = NINF;

  for (int i = 1; i <= L; i++) {
    OM(i,0) = OI(i,0) = OD(i,0) = OX(i,BO_GE) = NINF;
    for (int k = 1; k <= M; k++) {
      float best;
      if (i == 1) best = DELTA(BO_BM,k-1) * PPM(1,k,1);                        /* :87 */
      else {
        /* entry from row i-c into codon length c */
        float mx[6];
        int cmax = (i >= 5) ? 5 : (i == 2 ? 2 : (i == 4 ? 4 : 3));
        for (int c = 1; c <= cmax; c++) {
          int r = i - c;
          float p = PPM(i,k,c);
          if ((i == 2 && c == 2) || (i == 4 && c == 4)) {                     /* only B(0) can precede (:113, :171) */
            mx[c] = DELTA(BO_BM,k-1) * (OX(0,BO_GB) + p);
          } else {
            float a = DELTA(BO_MM,k-1) * (OM(r,k-1) + p);
            float b = DELTA(BO_IM,k-1) * (OI(r,k-1) + p);
            float d = DELTA(BO_DM,k-1) * (OD(r,k-1) + p);
            float e = DELTA(BO_BM,k-1) * (OX(r,BO_GB) + p);
            float t3 = (d > e) ? d : e;  float t2 = (b > t3) ? b : t3;  mx[c] = (a > t2) ? a : t2;
          }
        }
        if (i == 2) best = (mx[1] > mx[2]) ? mx[1] : mx[2];
        else if (i < 5) {
          float m4 = (i == 4) ? mx[4] : NINF;
          float l = (mx[1] > mx[2]) ? mx[1] : mx[2], r2 = (mx[3] > m4) ? mx[3] : m4;
          best = (l > r2) ? l : r2;
        } else {
          float l = (mx[1] > mx[2]) ? mx[1] : mx[2];
          float r34 = (mx[3] > mx[4]) ? mx[3] : mx[4];
          float r2 = (r34 > mx[5]) ? r34 : mx[5];
          best = (l > r2) ? l : r2;
        }
      }
      OM(i,k) = best;
      if (i >= 3 && k < M) {
        float a = DELTA(BO_MI,k) * (OM(i-3,k) + PPI(i,k));
        float b = DELTA(BO_II,k) * (OI(i-3,k) + PPI(i,k));
        OI(i,k) = (a > b) ? a : b;
      } else OI(i,k) = NINF;
      {
        float a = DELTA(BO_MD,k-1) * OM(i,k-1), b = DELTA(BO_DD,k-1) * OD(i,k-1);
        OD(i,k) = (a > b) ? a : b;
      }
      if (k < M) { float v = esc * OM(i,k); if (v > OX(i,BO_GE)) OX(i,BO_GE) = v; }
      else { float v = (OM(i,M) > OD(i,M)) ? OM(i,M) : OD(i,M); if (v > OX(i,BO_GE)) OX(i,BO_GE) = v; }
    }
    if (i <= 2) {
      OX(i,BO_GJ) = ej * OX(i,BO_GE);
      OX(i,BO_GC) = ec * OX(i,BO_GE);
      OX(i,BO_GN) = nn * BO_X(pp,i,BO_GN);
    } else {
      float a = jj * (OX(i-3,BO_GJ) + BO_X(pp,i,BO_GJ)), b = ej * OX(i,BO_GE);
      OX(i,BO_GJ) = (a > b) ? a : b;
      a = cc * (OX(i-3,BO_GC) + BO_X(pp,i,BO_GC)); b = ec * OX(i,BO_GE);
      OX(i,BO_GC) = (a > b) ? a : b;
      OX(i,BO_GN) = nn * (OX(i-3,BO_GN) + BO_X(pp,i,BO_GN));
    }
    { float a = nb * OX(i,BO_GN), b = jb * OX(i,BO_GJ); OX(i,BO_GB) = (a > b) ? a : b; }
  }
  *ret_e = OX(L,BO_GC) + OX(L-1,BO_GC) + OX(L-2,BO_GC);
  return BO_OK;
}

/* esl_abc_FAvgScVec (easel): degenerate residue = plain mean of its members' scores */
static void avg_scvec(float *sc)
{
  for (int x = BO_K_AMINO + 1; x <= BO_KP_AMINO - 3; x++) {
    float result = 0.f; int n = 0;
    for (int i = 0; i < BO_K_AMINO; i++) if (bo_amino_degen(x, i)) { result += sc[i]; n++; }
    sc[x] = result / (float) n;
  }
}

/* null2 by expectation; like the reference it uses row 0 of <pp> as scratch (generic_null2_frameshift.c:62-68). */
int bo_gnull2_fs(const bo_fs_profile *gm, bo_gmx *pp, float *null2)
{
  int M = gm->M, Ld = pp->L;
  size_t W = (size_t)(M + 1) * BO_NSCELLS_FS;
  float *row0 = pp->dp, *x0 = pp->xmx;
  memcpy(row0, pp->dp + W, sizeof(float) * W);
  memcpy(x0, pp->xmx + BO_NXCELLS, sizeof(float) * BO_NXCELLS);
  for (int i = 2; i <= Ld; i++) {
    const float *r = pp->dp + (size_t) i * W;
    for (size_t j = 0; j < W; j++) row0[j] += r[j];
    for (int s = 0; s < BO_NXCELLS; s++) x0[s] += pp->xmx[(size_t) i * BO_NXCELLS + s];
  }
  float lld = (float) -log((float) Ld);
  for (size_t j = 0; j < W; j++) row0[j] = (float) log(row0[j]);          /* esl_vec_FLog: log() of each float */
  for (int s = 0; s < BO_NXCELLS; s++) x0[s] = (float) log(x0[s]);
  for (size_t j = 0; j < W; j++) row0[j] += lld;
  for (int s = 0; s < BO_NXCELLS; s++) x0[s] += lld;

  float xfactor = BO_X(pp,0,BO_GN);
  xfactor = LS(xfactor, BO_X(pp,0,BO_GC));
  xfactor = LS(xfactor, BO_X(pp,0,BO_GJ));
  const float *amino = gm->rsc + (size_t) gm->maxcodons * (M + 1);       /* p7P_MSC_AMINO5(gm,k,x) = rsc[maxcodons+x][k] */
  for (int x = 0; x < BO_K_AMINO; x++) {
    float v = NINF;
    for (int k = 1; k < M; k++) {
      v = LS(v, BO_DP(pp,0,k,BO_GM) + amino[(size_t) x * (M + 1) + k]);
      v = LS(v, BO_DP(pp,0,k,BO_GI));
    }
    v = LS(v, BO_DP(pp,0,M,BO_GM) + amino[(size_t) x * (M + 1) + M]);
    v = LS(v, xfactor);
    null2[x] = expf(v);                                                   /* esl_vec_FExp */
  }
  avg_scvec(null2);
  null2[BO_K_AMINO] = 1.0f; null2[BO_KP_AMINO - 2] = 1.0f; null2[BO_KP_AMINO - 1] = 1.0f;
  return BO_OK;
}
