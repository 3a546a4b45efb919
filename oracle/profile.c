/* profile.c -- null model, profile configuration, score quantisation.  ORACLE (test infra only).
 *
 * Restates:  p7_bg.c (null1, 2-state bias filter), modelconfig.c (p7_ProfileConfig,
 * p7_ProfileConfig_fs, length reconfiguration), impl_sse/p7_oprofile.c (byteify/wordify and the
 * mf/vf/fb conversions) and p7_scoredata.c.  Striping is dropped: every array here is indexed
 * by model node k directly; each field documents which striped value it equals.
 *
 * easel pieces restated from easel's published algorithms (easel is absent from the reference
 * tree): esl_abc_FExpectScVec, esl_hmm_Configure, esl_hmm_Forward, esl_vec_FNorm.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529   /* eslCONST_LOG2 */

/* ------------------------------------------------------------------ background */

/* p7_AminoFrequencies(), hmmer.c:161-184 (Swiss-Prot 50.8 background) */
static const float amino_f[20] = {
  0.0787945f, 0.0151600f, 0.0535222f, 0.0668298f, 0.0397062f, 0.0695071f, 0.0229198f, 0.0590092f,
  0.0594422f, 0.0963728f, 0.0237718f, 0.0414386f, 0.0482904f, 0.0395639f, 0.0540978f, 0.0683364f,
  0.0540687f, 0.0673417f, 0.0114135f, 0.0304133f };

void bo_bg_create(bo_bg *bg)            /* p7_bg_Create, p7_bg.c:44-79 */
{
  memset(bg, 0, sizeof *bg);
  memcpy(bg->f, amino_f, sizeof amino_f);
  bg->p1 = (float)(350. / 351.);
}

void bo_bg_setlength(bo_bg *bg, int L)  /* p7_bg.c:189-198 */
{
  bg->p1 = (float) L / (float) (L + 1);
  bg->t[0][0] = bg->p1;
  bg->t[0][1] = 1.0f - bg->p1;
}

float bo_bg_nullone(const bo_bg *bg, int L)   /* p7_bg.c:356-360: double log, result narrowed to float */
{
  return (float)((float) L * log(bg->p1) + log(1. - bg->p1));
}

float bo_bg_fs_nullone(const bo_bg *bg, int aminoL)  /* p7_bg.c:377-385 */
{
  float per_frame = (float)((float) aminoL * log(bg->p1) + log(1. - bg->p1));
  return (float)(per_frame + log(3.0));
}

/* esl_hmm_Configure() (easel esl_hmm.c): emission odds eo[x][k] = e[k][x]/f[x]; gap, '*', '~' -> 1;
 * degenerate residues -> sum of member emissions / sum of member background. */
static void hmm2_configure(bo_bg *bg)
{
  for (int x = 0; x < BO_K_AMINO; x++)
    for (int k = 0; k < 2; k++) bg->eo[x][k] = bg->e[k][x] / bg->f[x];
  for (int k = 0; k < 2; k++) {
    bg->eo[BO_K_AMINO][k] = 1.0f; bg->eo[BO_KP_AMINO - 2][k] = 1.0f; bg->eo[BO_KP_AMINO - 1][k] = 1.0f;
  }
  for (int x = BO_K_AMINO + 1; x <= BO_KP_AMINO - 3; x++)
    for (int k = 0; k < 2; k++) {
      float num = 0.0f, denom = 0.0f;
      for (int y = 0; y < BO_K_AMINO; y++)
        if (bo_amino_degen(x, y)) { num += bg->e[k][y]; denom += bg->f[y]; }
      bg->eo[x][k] = (denom > 0.0f) ? num / denom : 0.0f;
    }
}

void bo_bg_setfilter(bo_bg *bg, int M, const float *compo)   /* p7_bg.c:449-473 */
{
  float L0 = 400.0f;
  float L1 = (float) M / 8.0f;
  bg->t[0][0] = L0 / (L0 + 1.0f);
  bg->t[0][1] = 1.0f / (L0 + 1.0f);
  bg->t[0][2] = 1.0f;
  memcpy(bg->e[0], bg->f, sizeof(float) * BO_K_AMINO);
  bg->t[1][0] = 1.0f / (L1 + 1.0f);
  bg->t[1][1] = L1 / (L1 + 1.0f);
  bg->t[1][2] = 1.0f;
  memcpy(bg->e[1], compo, sizeof(float) * BO_K_AMINO);
  bg->pi[0] = 0.999f;
  bg->pi[1] = 0.001f;
  hmm2_configure(bg);
}

/* esl_hmm_Forward() (easel esl_hmm.c) on the 2-state filter HMM: per-row max-scaled Forward. */
static float hmm2_forward(const bo_bg *bg, const uint8_t *dsq, int L)
{
  float dp[2], nx[2], logsc = 0.0f, mx;
  if (L == 0) return (float) log(bg->pi[2]);     /* pi[M]; unused on this path (ORFs have n>0) */
  mx = 0.0f;
  for (int k = 0; k < 2; k++) { dp[k] = bg->eo[dsq[1]][k] * bg->pi[k]; if (dp[k] > mx) mx = dp[k]; }
  for (int k = 0; k < 2; k++) dp[k] /= mx;
  logsc += (float) log(mx);
  for (int i = 2; i <= L; i++) {
    mx = 0.0f;
    for (int k = 0; k < 2; k++) {
      nx[k] = 0.0f;
      for (int m = 0; m < 2; m++) nx[k] += dp[m] * bg->t[m][k];
      nx[k] *= bg->eo[dsq[i]][k];
      if (nx[k] > mx) mx = nx[k];
    }
    for (int k = 0; k < 2; k++) dp[k] = nx[k] / mx;
    logsc += (float) log(mx);
  }
  float end = 0.0f;
  for (int m = 0; m < 2; m++) end += dp[m] * bg->t[m][2];
  logsc += (float) log(end);
  return logsc;
}

float bo_bg_filterscore(const bo_bg *bg, const uint8_t *dsq, int L)   /* p7_bg.c:491-505 */
{
  float nullsc = hmm2_forward(bg, dsq, L);
  return nullsc + (float) L * logf(bg->p1) + logf((float)(1. - bg->p1));
}

/* p7_bg.c:522-568: three-frame translation, stop/degenerate codons skipped, logsum over frames */
float bo_bg_fs_filterscore(const bo_bg *bg, const uint8_t *dna, int L, const uint8_t basic[64])
{
  uint8_t *orf = malloc((size_t) L + 2);
  float sum = -INFINITY;
  orf[0] = BO_DSQ_SENTINEL;
  for (int f = 1; f <= 3; f++) {
    int j = 1;
    for (int i = f; i <= L - 2; i += 3) {
      uint8_t aa = bo_gencode_translate(basic, dna + i);
      if (aa < BO_K_AMINO) orf[j++] = aa;            /* esl_abc_XIsCanonical */
    }
    orf[j] = BO_DSQ_SENTINEL;
    float nullsc = hmm2_forward(bg, orf, j - 1);
    sum = bo_flogsum(sum, nullsc);
  }
  free(orf);
  /* p7_bg.c:561: float + (float*float + float + double) evaluated in double, stored to float */
  return (float)((double) sum + ((double)((float) (L / 3) * logf(bg->p1) + logf((float)(1. - bg->p1))) + log(3.0)));
}

/* ------------------------------------------------------------------ generic profile */

/* p7_hmm_CalculateOccupancy, p7_hmm.c:1348-1364 (mocc only) */
static void occupancy(const bo_hmm *h, float *mocc)
{
  const float *t = h->t;
  mocc[0] = 0.f;
  mocc[1] = t[BO_H_MI] + t[BO_H_MM];
  for (int k = 2; k <= h->M; k++)
    mocc[k] = (float)(mocc[k-1] * (t[(k-1)*7 + BO_H_MM] + t[(k-1)*7 + BO_H_MI]) +
                      (1.0 - mocc[k-1]) * t[(k-1)*7 + BO_H_DM]);
}

/* esl_abc_FExpectScVec (easel): degenerate residue score = f-weighted mean of member scores */
static void expect_scvec(float *sc, const float *f)
{
  for (int x = BO_K_AMINO + 1; x <= BO_KP_AMINO - 3; x++) {
    float result = 0.f, denom = 0.f;
    for (int i = 0; i < BO_K_AMINO; i++)
      if (bo_amino_degen(x, i)) { result += sc[i] * f[i]; denom += f[i]; }
    sc[x] = result / denom;
  }
}

static void set_tsc(float *tsc, const bo_hmm *h)
{
  int M = h->M;
  /* p7_profile_Create: node 0 transitions -inf (p7_profile.c:84) */
  for (int s = 0; s < BO_NTRANS; s++) tsc[s] = -INFINITY;
  /* entry: occ[k] / sum_i occ[i]*(M-i+1), stored off by one (modelconfig.c:86-100) */
  float *occ = malloc(sizeof(float) * (size_t)(M + 1));
  float Z = 0.f;
  occupancy(h, occ);
  for (int k = 1; k <= M; k++) Z += occ[k] * (float) (M - k + 1);
  for (int k = 1; k <= M; k++) tsc[(k-1) * BO_NTRANS + BO_BM] = (float) log(occ[k] / Z);
  free(occ);
  /* modelconfig.c:127-136 */
  for (int k = 1; k < M; k++) {
    float *tp = tsc + k * BO_NTRANS;
    const float *t = h->t + k * 7;
    tp[BO_MM] = (float) log(t[BO_H_MM]);
    tp[BO_MI] = (float) log(t[BO_H_MI]);
    tp[BO_MD] = (float) log(t[BO_H_MD]);
    tp[BO_IM] = (float) log(t[BO_H_IM]);
    tp[BO_II] = (float) log(t[BO_H_II]);
    tp[BO_DM] = (float) log(t[BO_H_DM]);
    tp[BO_DD] = (float) log(t[BO_H_DD]);
  }
}

static void reconfig_len(float xsc[4][2], float nj, int L)   /* modelconfig.c:730-733 */
{
  float pmove = (2.0f + nj) / ((float) L + 2.0f + nj);
  float ploop = 1.0f - pmove;
  xsc[BO_XN][BO_LOOP] = xsc[BO_XC][BO_LOOP] = xsc[BO_XJ][BO_LOOP] = (float) log(ploop);
  xsc[BO_XN][BO_MOVE] = xsc[BO_XC][BO_MOVE] = xsc[BO_XJ][BO_MOVE] = (float) log(pmove);
}

bo_profile *bo_profile_config(const bo_hmm *h, const bo_bg *bg, int L)   /* modelconfig.c:48-196, mode p7_LOCAL */
{
  int M = h->M, Kp = BO_KP_AMINO;
  bo_profile *gm = calloc(1, sizeof *gm);
  gm->M = M; gm->max_length = h->max_length;
  gm->tsc = malloc(sizeof(float) * (size_t)(M + 1) * BO_NTRANS);
  gm->rsc = malloc(sizeof(float) * (size_t) Kp * (M + 1) * 2);
  memcpy(gm->evparam, h->evparam, sizeof gm->evparam);
  memcpy(gm->compo, h->compo, sizeof gm->compo);
  for (size_t i = 0; i < (size_t)(M + 1) * BO_NTRANS; i++) gm->tsc[i] = -INFINITY;
  set_tsc(gm->tsc, h);
  gm->xsc[BO_XE][BO_MOVE] = (float) -LOG2C;       /* multihit local, modelconfig.c:116-119 */
  gm->xsc[BO_XE][BO_LOOP] = (float) -LOG2C;
  gm->nj = 1.0f;

  float sc[BO_KP_AMINO];
  size_t rs = (size_t)(M + 1) * 2;
  for (int x = 0; x < Kp; x++) { gm->rsc[x * rs + 0] = -INFINITY; gm->rsc[x * rs + 1] = -INFINITY; }
  sc[BO_K_AMINO] = -INFINITY; sc[Kp - 2] = -INFINITY; sc[Kp - 1] = -INFINITY;
  for (int k = 1; k <= M; k++) {
    for (int x = 0; x < BO_K_AMINO; x++)
      sc[x] = (float) log((double) h->mat[k * BO_K_AMINO + x] / bg->f[x]);
    expect_scvec(sc, bg->f);
    for (int x = 0; x < Kp; x++) gm->rsc[x * rs + k * 2] = sc[x];
  }
  /* inserts hard-wired to 0 (modelconfig.c:162-169) */
  for (int x = 0; x < Kp; x++) {
    for (int k = 1; k < M; k++) gm->rsc[x * rs + k * 2 + 1] = 0.0f;
    gm->rsc[x * rs + M * 2 + 1] = -INFINITY;
  }
  for (int k = 1; k <= M; k++) {
    gm->rsc[(size_t) BO_K_AMINO * rs + k * 2 + 1] = -INFINITY;
    gm->rsc[(size_t)(Kp - 2) * rs + k * 2 + 1] = -INFINITY;
    gm->rsc[(size_t)(Kp - 1) * rs + k * 2 + 1] = -INFINITY;
  }
  bo_profile_reconfig_length(gm, L);
  return gm;
}

void bo_profile_reconfig_length(bo_profile *gm, int L) { reconfig_len(gm->xsc, gm->nj, L); gm->L = L; }

void bo_profile_free(bo_profile *gm) { if (gm) { free(gm->tsc); free(gm->rsc); free(gm); } }

/* ------------------------------------------------------------------ frameshift profile */

#define C1_5(x)          ((x) * 341)
#define C2_5(w,x)        ((x) * 341 + (w) * 85 + 1)
#define C3_5(v,w,x)      ((x) * 341 + (w) * 85 + (v) * 21 + 2)
#define C4_5(u,v,w,x)    ((x) * 341 + (w) * 85 + (v) * 21 + (u) * 5 + 3)
#define C5_5(t,u,v,w,x)  ((x) * 341 + (w) * 85 + (v) * 21 + (u) * 5 + (t) + 4)
#define C2_3(w,x)        ((x) * 84 + (w) * 21)
#define C3_3(v,w,x)      ((x) * 84 + (w) * 21 + (v) * 5 + 1)
#define C4_3(u,v,w,x)    ((x) * 84 + (w) * 21 + (v) * 5 + (u) + 2)

/* indel-type labels, hmmer.h:259-276 */
enum { I___X = 0, I_X__, I_XX_, I_X_X, I__XX, I_XXX, I_XXx, I_XxX, I_xXX, I_xxx, I_XXxX, I_XxXX, I_xXXX, I_XXxxX, I_XxxXX, I_xxXXX };

typedef struct { bo_fs_profile *gm; int k; int aoff; } fsctx;

/* "if amino score beats the current quasi-codon score, take it" -- the update repeated throughout
 * modelconfig.c:369-490 */
static inline void upd(const fsctx *c, int cidx, int a, int indel)
{
  bo_fs_profile *gm = c->gm;
  size_t W = (size_t) gm->M + 1;
  float asc = gm->rsc[(size_t)(c->aoff + a) * W + c->k];
  float *dst = &gm->rsc[(size_t) cidx * W + c->k];
  if (asc > *dst) {
    *dst = asc;
    gm->codons[(size_t) c->k * gm->maxcodons + cidx] = (uint8_t) a;
    gm->indel_pos[(size_t) c->k * gm->maxcodons + cidx] = (uint8_t) indel;
  }
}
static inline void setc(const fsctx *c, int cidx, int a, int indel, float add)
{
  bo_fs_profile *gm = c->gm;
  size_t W = (size_t) gm->M + 1;
  gm->rsc[(size_t) cidx * W + c->k] = gm->rsc[(size_t)(c->aoff + a) * W + c->k] + add;
  gm->codons[(size_t) c->k * gm->maxcodons + cidx] = (uint8_t) a;
  gm->indel_pos[(size_t) c->k * gm->maxcodons + cidx] = (uint8_t) indel;
}

bo_fs_profile *bo_fs_profile_config(const bo_hmm *h, const bo_bg *bg, const uint8_t basic[64], int codon_lengths, int L_amino)
{                                                        /* modelconfig.c:220-698, mode p7_LOCAL */
  int M = h->M, Kp = BO_KP_AMINO;
  int STOP = Kp - 2, XAA = Kp - 3;
  if (codon_lengths != 3 && codon_lengths != 5) return NULL;
  bo_fs_profile *gm = calloc(1, sizeof *gm);
  gm->M = M; gm->max_length = h->max_length; gm->codon_lengths = codon_lengths;
  gm->maxcodons = (codon_lengths == 5) ? BO_MAXCODONS5 : BO_MAXCODONS3;
  gm->fsprob = h->fsprob;
  memcpy(gm->evparam, h->evparam, sizeof gm->evparam);
  memcpy(gm->compo, h->compo, sizeof gm->compo);
  float one_indel, two_indel = 0.f, no_indel, stop_codon;     /* modelconfig.c:243-254 */
  one_indel  = (float) log(h->fsprob);
  stop_codon = (float) log(h->fsprob);
  if (codon_lengths == 5) { two_indel = (float) log(h->fsprob / 2.); no_indel = (float) log(1. - h->fsprob * 4.); }
  else                    {                                          no_indel = (float) log(1. - h->fsprob * 3.); }

  size_t W = (size_t) M + 1;
  int nrows = gm->maxcodons + Kp;
  gm->tsc = malloc(sizeof(float) * W * BO_NTRANS);
  for (size_t i = 0; i < W * BO_NTRANS; i++) gm->tsc[i] = -INFINITY;
  set_tsc(gm->tsc, h);
  gm->rsc = malloc(sizeof(float) * (size_t) nrows * W);
  gm->codons    = calloc((size_t) gm->maxcodons * W, 1);
  gm->indel_pos = calloc((size_t) gm->maxcodons * W, 1);
  gm->xsc[BO_XE][BO_MOVE] = (float) -LOG2C;
  gm->xsc[BO_XE][BO_LOOP] = (float) -LOG2C;
  gm->nj = 1.0f;
  for (size_t i = 0; i < (size_t) nrows * W; i++) gm->rsc[i] = -INFINITY;   /* modelconfig.c:343-344 */

  float sc[BO_KP_AMINO];
  sc[BO_K_AMINO] = -INFINITY; sc[Kp - 2] = -INFINITY; sc[Kp - 1] = -INFINITY;
  for (int k = 1; k <= M; k++) {
    for (int x = 0; x < BO_K_AMINO; x++)
      sc[x] = (float) log((double) h->mat[k * BO_K_AMINO + x] / bg->f[x]);
    expect_scvec(sc, bg->f);
    for (int x = 0; x < Kp; x++) gm->rsc[(size_t)(gm->maxcodons + x) * W + k] = sc[x];
  }

  fsctx c = { gm, 0, gm->maxcodons };
  for (int k = 1; k <= M; k++) {
    c.k = k;
    float *col = gm->rsc;   /* rsc[cidx*W + k] */
    if (codon_lengths == 5) {
      for (int x = 0; x < 4; x++) for (int w = 0; w < 4; w++) for (int v = 0; v < 4; v++) {
        int a = basic[16 * v + 4 * w + x];
        upd(&c, C1_5(x), a, I___X);                    /* modelconfig.c:368-380 */
        upd(&c, C1_5(v), a, I_X__);
        upd(&c, C2_5(w, x), a, I__XX);                 /* :383-402 */
        upd(&c, C2_5(v, x), a, I_X_X);
        upd(&c, C2_5(v, w), a, I_XX_);
        int c3 = C3_5(v, w, x);                        /* :405-434 */
        if (a == STOP) {
          for (int s = 0; s < 4; s++) {
            upd(&c, c3, basic[16 * s + 4 * w + x], I_xXX);
            upd(&c, c3, basic[16 * v + 4 * s + x], I_XxX);
            upd(&c, c3, basic[16 * v + 4 * w + s], I_XXx);
          }
        } else setc(&c, c3, a, I_XXX, 0.f);
        for (int u = 0; u < 4; u++) {                  /* :435-491 */
          int c4 = C4_5(u, v, w, x);
          upd(&c, c4, basic[16 * u + 4 * v + x], I_XXxX);
          upd(&c, c4, basic[16 * u + 4 * w + x], I_XxXX);
          upd(&c, c4, basic[16 * v + 4 * w + x], I_xXXX);
          for (int t = 0; t < 4; t++) {
            int c5 = C5_5(t, u, v, w, x);
            upd(&c, c5, basic[16 * t + 4 * u + x], I_XXxxX);
            upd(&c, c5, basic[16 * t + 4 * w + x], I_XxxXX);
            upd(&c, c5, basic[16 * v + 4 * w + x], I_xxXXX);
          }
        }
      }
      for (int x = 0; x < 4; x++) {                    /* indel costs, :498-519 */
        col[(size_t) C1_5(x) * W + k] += two_indel;
        for (int w = 0; w < 4; w++) {
          col[(size_t) C2_5(w, x) * W + k] += one_indel;
          for (int v = 0; v < 4; v++) {
            int a = basic[16 * v + 4 * w + x];
            col[(size_t) C3_5(v, w, x) * W + k] += (a == STOP) ? stop_codon : no_indel;
            for (int u = 0; u < 4; u++) {
              col[(size_t) C4_5(u, v, w, x) * W + k] += one_indel;
              for (int t = 0; t < 4; t++) col[(size_t) C5_5(t, u, v, w, x) * W + k] += two_indel;
            }
          }
        }
      }
      setc(&c, BO_DEGEN5_C,   XAA, I_xxx, no_indel);   /* :522-537 */
      setc(&c, BO_DEGEN5_QC1, XAA, I_xxx, one_indel);
      setc(&c, BO_DEGEN5_QC2, XAA, I_xxx, two_indel);
    } else {
      for (int x = 0; x < 4; x++) for (int w = 0; w < 4; w++) for (int v = 0; v < 4; v++) {
        int a = basic[16 * v + 4 * w + x];
        upd(&c, C2_3(w, x), a, I__XX);                 /* :550-569 */
        upd(&c, C2_3(v, x), a, I_X_X);
        upd(&c, C2_3(v, w), a, I_XX_);
        int c3 = C3_3(v, w, x);                        /* :572-598 */
        if (a == STOP) {
          for (int s = 0; s < 4; s++) {
            upd(&c, c3, basic[16 * s + 4 * w + x], I_xXX);
            upd(&c, c3, basic[16 * v + 4 * s + x], I_XxX);
            upd(&c, c3, basic[16 * v + 4 * w + s], I_XXx);
          }
        } else setc(&c, c3, a, I_XXX, 0.f);
        for (int u = 0; u < 4; u++) {                  /* :600-627 */
          int c4 = C4_3(u, v, w, x);
          upd(&c, c4, basic[16 * u + 4 * v + x], I_XXxX);
          upd(&c, c4, basic[16 * u + 4 * w + x], I_XxXX);
          upd(&c, c4, basic[16 * v + 4 * w + x], I_xXXX);
        }
      }
      for (int x = 0; x < 4; x++) for (int w = 0; w < 4; w++) {   /* :633-648 */
        col[(size_t) C2_3(w, x) * W + k] += one_indel;
        for (int v = 0; v < 4; v++) {
          int a = basic[16 * v + 4 * w + x];
          col[(size_t) C3_3(v, w, x) * W + k] += (a == STOP) ? stop_codon : no_indel;
          for (int u = 0; u < 4; u++) col[(size_t) C4_3(u, v, w, x) * W + k] += one_indel;
        }
      }
      setc(&c, BO_DEGEN3_C,   XAA, I_xxx, no_indel);   /* :651-661 */
      setc(&c, BO_DEGEN3_QC1, XAA, I_xxx, one_indel);
    }
  }
  bo_fs_profile_reconfig_length(gm, L_amino);
  return gm;
}

void bo_fs_profile_reconfig_length(bo_fs_profile *gm, int L_amino) { reconfig_len(gm->xsc, gm->nj, L_amino); gm->L = L_amino; }

void bo_fs_profile_reconfig_unihit(bo_fs_profile *gm, int L_amino)    /* modelconfig.c:868-874 */
{
  gm->xsc[BO_XE][BO_MOVE] = 0.0f; gm->xsc[BO_XE][BO_LOOP] = -INFINITY; gm->nj = 0.0f;
  bo_fs_profile_reconfig_length(gm, L_amino);
}
void bo_fs_profile_reconfig_multihit(bo_fs_profile *gm, int L_amino)  /* modelconfig.c:825-831 */
{
  gm->xsc[BO_XE][BO_MOVE] = (float) -LOG2C; gm->xsc[BO_XE][BO_LOOP] = (float) -LOG2C; gm->nj = 1.0f;
  bo_fs_profile_reconfig_length(gm, L_amino);
}

void bo_fs_profile_free(bo_fs_profile *gm)
{
  if (gm) { free(gm->tsc); free(gm->rsc); free(gm->codons); free(gm->indel_pos); free(gm); }
}

/* ------------------------------------------------------------------ quantised ("optimized") profile */

static uint8_t unbiased_byteify(float scale_b, float sc)           /* p7_oprofile.c:683-690 */
{
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255.) ? 255 : (uint8_t)(int) sc;
}
static uint8_t biased_byteify(float scale_b, uint8_t bias_b, float sc)   /* p7_oprofile.c:667-674 */
{
  sc = -1.0f * roundf(scale_b * sc);
  return (sc > 255 - bias_b) ? 255 : (uint8_t)((int) sc + bias_b);
}
static int16_t wordify(float scale_w, float sc)                    /* p7_oprofile.c:699-705 */
{
  sc = roundf(scale_w * sc);
  if      (sc >=  32767.0) return  32767;
  else if (sc <= -32768.0) return -32768;
  else return (int16_t) sc;
}

#define GM_MSC(gm,k,x) ((gm)->rsc[(size_t)(x) * ((gm)->M + 1) * 2 + (k) * 2])
#define GM_TSC(gm,k,s) ((gm)->tsc[(k) * BO_NTRANS + (s)])

bo_oprofile *bo_oprofile_convert(const bo_profile *gm)             /* p7_oprofile.c:1091-1127 */
{
  int M = gm->M, Kp = BO_KP_AMINO;
  size_t W = (size_t) M + 1;
  bo_oprofile *om = calloc(1, sizeof *om);
  om->M = M; om->L = gm->L; om->nj = gm->nj; om->max_length = gm->max_length;
  memcpy(om->evparam, gm->evparam, sizeof om->evparam);
  memcpy(om->compo, gm->compo, sizeof om->compo);
  om->rb = malloc(Kp * W);
  om->rw = malloc(sizeof(int16_t) * Kp * W);
  om->tw = malloc(sizeof(int16_t) * W * BO_NTRANS);
  om->rf = malloc(sizeof(float) * Kp * W);
  om->tf = malloc(sizeof(float) * W * BO_NTRANS);
  om->msc = malloc(sizeof(float) * Kp * W);
  om->tsc = malloc(sizeof(float) * W * BO_NTRANS);
  for (int x = 0; x < Kp; x++) { om->msc[x * W] = -INFINITY; for (int k = 1; k <= M; k++) om->msc[x * W + k] = GM_MSC(gm, k, x); }
  memcpy(om->tsc, gm->tsc, sizeof(float) * W * BO_NTRANS);

  /* ---- mf_conversion, p7_oprofile.c:773-813 ---- */
  float mx = 0.0f;
  for (int x = 0; x < BO_K_AMINO; x++)
    for (size_t i = 0; i < W * 2; i++) { float v = gm->rsc[(size_t) x * W * 2 + i]; if (v > mx) mx = v; }
  om->scale_b = (float)(3.0 / LOG2C);
  om->base_b  = 190;
  om->bias_b  = unbiased_byteify(om->scale_b, (float)(-1.0 * mx));
  for (int x = 0; x < Kp; x++) {
    om->rb[x * W] = 255;
    for (int k = 1; k <= M; k++) om->rb[x * W + k] = biased_byteify(om->scale_b, om->bias_b, GM_MSC(gm, k, x));
  }
  om->tbm_b = unbiased_byteify(om->scale_b, logf(2.0f / ((float) M * (float) (M + 1))));
  om->tec_b = unbiased_byteify(om->scale_b, logf(0.5f));
  om->tjb_b = unbiased_byteify(om->scale_b, logf(3.0f / (float) (gm->L + 3)));

  /* ---- vf_conversion, p7_oprofile.c:826-921 ----
   * tw[k][BM|MM|IM|DM] = the value the striped twv holds at node k for the "into k" transitions
   * (gm index k-1); tw[k][MD|MI|II|DD] = the "out of k" transitions (gm index k, -32768 when k==M). */
  om->scale_w = (float)(500.0 / LOG2C);
  om->base_w  = 12000;
  for (int x = 0; x < Kp; x++) {
    om->rw[x * W] = -32768;
    for (int k = 1; k <= M; k++) om->rw[x * W + k] = wordify(om->scale_w, GM_MSC(gm, k, x));
  }
  for (int s = 0; s < BO_NTRANS; s++) om->tw[s] = -32768;
  for (int k = 1; k <= M; k++) {
    int16_t *t = om->tw + k * BO_NTRANS;
    int16_t v;
    v = wordify(om->scale_w, GM_TSC(gm, k-1, BO_BM)); t[BO_BM] = (v <= 0) ? v : 0;
    v = wordify(om->scale_w, GM_TSC(gm, k-1, BO_MM)); t[BO_MM] = (v <= 0) ? v : 0;
    v = wordify(om->scale_w, GM_TSC(gm, k-1, BO_IM)); t[BO_IM] = (v <= 0) ? v : 0;
    v = wordify(om->scale_w, GM_TSC(gm, k-1, BO_DM)); t[BO_DM] = (v <= 0) ? v : 0;
    v = (k < M) ? wordify(om->scale_w, GM_TSC(gm, k, BO_MD)) : -32768; t[BO_MD] = (v <= 0) ? v : 0;
    v = (k < M) ? wordify(om->scale_w, GM_TSC(gm, k, BO_MI)) : -32768; t[BO_MI] = (v <= 0) ? v : 0;
    v = (k < M) ? wordify(om->scale_w, GM_TSC(gm, k, BO_II)) : -32768; t[BO_II] = (v <= -1) ? v : -1;  /* II never 0 */
    t[BO_DD] = (k < M) ? wordify(om->scale_w, GM_TSC(gm, k, BO_DD)) : -32768;
  }
  om->xw[BO_XE][BO_LOOP] = wordify(om->scale_w, gm->xsc[BO_XE][BO_LOOP]);
  om->xw[BO_XE][BO_MOVE] = wordify(om->scale_w, gm->xsc[BO_XE][BO_MOVE]);
  om->xw[BO_XN][BO_MOVE] = wordify(om->scale_w, gm->xsc[BO_XN][BO_MOVE]);
  om->xw[BO_XN][BO_LOOP] = 0;
  om->xw[BO_XC][BO_MOVE] = wordify(om->scale_w, gm->xsc[BO_XC][BO_MOVE]);
  om->xw[BO_XC][BO_LOOP] = 0;
  om->xw[BO_XJ][BO_MOVE] = wordify(om->scale_w, gm->xsc[BO_XJ][BO_MOVE]);
  om->xw[BO_XJ][BO_LOOP] = 0;
  {
    int dd = -32768;                                   /* p7_oprofile.c:911-918 */
    for (int k = 2; k < M - 1; k++) {
      int d = (int) wordify(om->scale_w, GM_TSC(gm, k,   BO_DD))
            + (int) wordify(om->scale_w, GM_TSC(gm, k+1, BO_DM))
            - (int) wordify(om->scale_w, GM_TSC(gm, k+1, BO_BM));
      if (d > dd) dd = d;
    }
    om->ddbound_w = (int16_t) dd;
  }

  /* ---- fb_conversion, p7_oprofile.c:929-994 (odds ratios; reference uses esl_sse_expf) ---- */
  for (int x = 0; x < Kp; x++) {
    om->rf[x * W] = 0.0f;
    for (int k = 1; k <= M; k++) om->rf[x * W + k] = expf(GM_MSC(gm, k, x));
  }
  for (int s = 0; s < BO_NTRANS; s++) om->tf[s] = 0.0f;
  for (int k = 1; k <= M; k++) {
    float *t = om->tf + k * BO_NTRANS;
    t[BO_BM] = expf(GM_TSC(gm, k-1, BO_BM));
    t[BO_MM] = expf(GM_TSC(gm, k-1, BO_MM));
    t[BO_IM] = expf(GM_TSC(gm, k-1, BO_IM));
    t[BO_DM] = expf(GM_TSC(gm, k-1, BO_DM));
    t[BO_MD] = (k < M) ? expf(GM_TSC(gm, k, BO_MD)) : 0.0f;
    t[BO_MI] = (k < M) ? expf(GM_TSC(gm, k, BO_MI)) : 0.0f;
    t[BO_II] = (k < M) ? expf(GM_TSC(gm, k, BO_II)) : 0.0f;
    t[BO_DD] = (k < M) ? expf(GM_TSC(gm, k, BO_DD)) : 0.0f;
  }
  for (int s = 0; s < 4; s++) for (int t = 0; t < 2; t++) om->xf[s][t] = expf(gm->xsc[s][t]);
  return om;
}

void bo_oprofile_reconfig_msv_length(bo_oprofile *om, int L)       /* p7_oprofile.c:1286-1290 */
{
  om->tjb_b = unbiased_byteify(om->scale_b, logf(3.0f / (float) (L + 3)));
}

void bo_oprofile_reconfig_length(bo_oprofile *om, int L)           /* p7_oprofile.c:1261-1326 */
{
  bo_oprofile_reconfig_msv_length(om, L);
  float pmove = (2.0f + om->nj) / ((float) L + 2.0f + om->nj);
  float ploop = 1.0f - pmove;
  om->xf[BO_XN][BO_LOOP] = om->xf[BO_XC][BO_LOOP] = om->xf[BO_XJ][BO_LOOP] = ploop;
  om->xf[BO_XN][BO_MOVE] = om->xf[BO_XC][BO_MOVE] = om->xf[BO_XJ][BO_MOVE] = pmove;
  om->xw[BO_XN][BO_MOVE] = om->xw[BO_XC][BO_MOVE] = om->xw[BO_XJ][BO_MOVE] = wordify(om->scale_w, logf(pmove));
  om->L = L;
}

void bo_oprofile_free(bo_oprofile *om)
{
  if (om) { free(om->rb); free(om->rw); free(om->tw); free(om->rf); free(om->tf); free(om->msc); free(om->tsc); free(om); }
}

/* ------------------------------------------------------------------ score data */

bo_scoredata *bo_scoredata_create(const bo_oprofile *om)   /* p7_scoredata.c:57-70 (std) + :314-388 */
{
  int M = om->M, Kp = BO_KP_AMINO;
  size_t W = (size_t) M + 1;
  bo_scoredata *sd = calloc(1, sizeof *sd);
  sd->M = M;
  sd->ssv_scores = calloc(W * Kp, 1);
  for (int k = 1; k <= M; k++)
    for (int x = 0; x < Kp; x++) sd->ssv_scores[k * Kp + x] = om->rb[x * W + k];
  sd->prefix_lengths = calloc(W, sizeof(float));
  sd->suffix_lengths = calloc(W, sizeof(float));
  float sum = 0;
  for (int k = 1; k < M; k++) {
    float tmi = om->tf[k * BO_NTRANS + BO_MI], tii = om->tf[k * BO_NTRANS + BO_II];
    if (tmi == 0) sd->prefix_lengths[k] = 1;
    else sd->prefix_lengths[k] = (float)(1 + (int)(log(1e-7 / tmi) / log(tii)));   /* p7_DEFAULT_WINDOW_BETA */
    sum += sd->prefix_lengths[k];
  }
  sd->prefix_lengths[0] = sd->prefix_lengths[M] = 0;
  for (int k = 1; k < M; k++) sd->prefix_lengths[k] /= sum;
  sd->suffix_lengths[M] = sd->prefix_lengths[M - 1];
  for (int k = M - 1; k >= 1; k--) sd->suffix_lengths[k] = sd->suffix_lengths[k + 1] + sd->prefix_lengths[k - 1];
  for (int k = 2; k < M; k++) sd->prefix_lengths[k] += sd->prefix_lengths[k - 1];
  return sd;
}

void bo_scoredata_free(bo_scoredata *sd)
{
  if (sd) { free(sd->ssv_scores); free(sd->prefix_lengths); free(sd->suffix_lengths); free(sd); }
}
