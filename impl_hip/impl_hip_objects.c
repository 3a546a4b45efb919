/* impl_hip_objects.c -- P7_OPROFILE, P7_FS_OPROFILE, P7_OMX, P7_OIVX for the MI355X backend (see impl_hip.h).
 *
 *   p7_oprofile_*      <- impl_sse/p7_oprofile.c:90-1326      p7_fs_oprofile_* <- impl_sse/p7_fs_oprofile.c
 *   p7_omx_*           <- impl_sse/p7_omx.c                   p7_oivx_*        <- impl_sse/p7_oivx.c
 *
 * A profile object is the metadata generic code reads plus a handle to the tables in HBM; converting uploads them once
 * (bath_hip_oprofile_convert), reconfiguring the length costs a scalar refresh because the kernels take every per-length
 * quantity from tables indexed by the target's length (bath_hip_oprofile_scalars documents them).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "easel.h"
#include "esl_alphabet.h"

#include "hmmer.h"

#define IH_KP 29               /* the amino alphabet of every BATH profile (abc->Kp) */

/* ------------------------------------------------------------------ the thread's GPU context (impl_Init) */

static __thread bath_hip_ctx *ih_ctx = NULL;

bath_hip_ctx *impl_hip_context(void)
{
  if (!ih_ctx) {
    const char *e = getenv("BATH_HIP_DEVICE");
    if (bath_hip_init(e ? atoi(e) : 0, &ih_ctx) != BATH_OK) {
      /* no CPU fallback exists: without a gfx950 device nothing below can run */
      esl_fatal("impl_hip: no usable MI355X (bath_hip_init failed); this backend has no CPU path");
    }
  }
  return ih_ctx;
}
void impl_hip_init(void) { (void) impl_hip_context(); }

static char *ih_strdup(const char *s)
{
  if (!s) return NULL;
  size_t n = strlen(s) + 1;
  char *d = malloc(n);
  if (d) memcpy(d, s, n);
  return d;
}
/* annotation lines are 1..M with a sentinel at 0 and a NUL at M+1 (p7_oprofile.c:150-156) */
static char *ih_annot(const char *s, int M)
{
  char *d = calloc((size_t) M + 2, 1);
  if (d && s) memcpy(d, s, (size_t) M + 2);
  return d;
}

/* ------------------------------------------------------------------ P7_OPROFILE */

P7_OPROFILE *p7_oprofile_Create(int allocM, const ESL_ALPHABET *abc)            /* p7_oprofile.c:90 */
{
  P7_OPROFILE *om = calloc(1, sizeof *om);
  if (!om) return NULL;
  om->allocM = allocM; om->abc = abc; om->mode = p7_NO_MODE;
  om->allocQ4 = p7O_NQF(allocM); om->allocQ8 = p7O_NQW(allocM); om->allocQ16 = p7O_NQB(allocM);
  om->rf = calloc((size_t) allocM + 2, 1); om->mm = calloc((size_t) allocM + 2, 1);
  om->cs = calloc((size_t) allocM + 2, 1); om->consensus = calloc((size_t) allocM + 2, 1);
  for (int x = 0; x < p7_NEVPARAM; x++) om->evparam[x] = p7_EVPARAM_UNSET;
  for (int x = 0; x < p7_NCUTOFFS; x++) om->cutoff[x]  = p7_CUTOFF_UNSET;
  for (int x = 0; x < p7_MAXABET;  x++) om->compo[x]   = p7_COMPO_UNSET;
  for (int x = 0; x < p7_NOFFSETS; x++) om->offs[x]    = -1;
  om->roff = om->eoff = -1;
  om->nj = 0.0f;
  return om;
}

int p7_oprofile_IsLocal(const P7_OPROFILE *om)                                   /* p7_oprofile.c:177 */
{
  return (om->mode == p7_LOCAL || om->mode == p7_UNILOCAL) ? TRUE : FALSE;
}

void p7_oprofile_Destroy(P7_OPROFILE *om)                                        /* p7_oprofile.c:192 */
{
  if (!om) return;
  if (!om->clone) {
    if (om->dev) bath_hip_oprofile_destroy(om->dev);
    if (om->gm_copy) { free(om->gm_copy->tsc); free(om->gm_copy->rsc); free(om->gm_copy); }
    free(om->rf_host); free(om->tf_host); free(om->ssv_host);
    free(om->name); free(om->acc); free(om->desc);
    free(om->rf); free(om->mm); free(om->cs); free(om->consensus);
  }
  free(om);
}

/* p7_oprofile.c:376: "a shallow copy that shares the read-only score tables": exactly what a second handle to the same device
 * tables is.  The clone has its own length configuration (the scalar fields), as the reference's clones do. */
P7_OPROFILE *p7_oprofile_Clone(const P7_OPROFILE *om)
{
  P7_OPROFILE *c = malloc(sizeof *c);
  if (!c) return NULL;
  memcpy(c, om, sizeof *c);
  c->clone = 1;
  return c;
}

static int ih_refresh_scalars(P7_OPROFILE *om, int L)
{
  bath_oprofile_scalars s;
  if (bath_hip_oprofile_scalars(om->dev, L, &s) != BATH_OK) return eslFAIL;
  om->tbm_b = s.tbm_b; om->tec_b = s.tec_b; om->tjb_b = s.tjb_b; om->base_b = s.base_b; om->bias_b = s.bias_b; om->scale_b = s.scale_b;
  om->scale_w = s.scale_w; om->base_w = s.base_w; om->ddbound_w = s.ddbound_w;
  /* bath_oprofile_scalars orders special states E N J C with [LOOP, MOVE]; p7O_* orders [MOVE, LOOP] */
  for (int st = 0; st < p7O_NXSTATES; st++) {
    om->xw[st][p7O_LOOP] = s.xw[st][0]; om->xw[st][p7O_MOVE] = s.xw[st][1];
    om->xf[st][p7O_LOOP] = s.xf[st][0]; om->xf[st][p7O_MOVE] = s.xf[st][1];
  }
  om->ncj_roundoff = 0.0f;                               /* the reference also leaves it at 0 ("goes along with NCJ hack") */
  return eslOK;
}

static int ih_upload(P7_OPROFILE *om)
{
  const int M = om->M;
  if (om->dev) { bath_hip_oprofile_destroy(om->dev); om->dev = NULL; }
  if (bath_hip_oprofile_convert(impl_hip_context(), om->gm_copy, &om->dev) != BATH_OK) return eslFAIL;
  if (om->consensus && om->consensus[1]) bath_hip_oprofile_set_consensus(om->dev, om->consensus);
  free(om->rf_host); free(om->tf_host); free(om->ssv_host);
  om->rf_host  = malloc(sizeof(float) * IH_KP * (size_t)(M + 1));
  om->tf_host  = malloc(sizeof(float) * 8 * (size_t)(M + 1));
  om->ssv_host = malloc((size_t)(M + 1) * IH_KP);
  if (!om->rf_host || !om->tf_host || !om->ssv_host) return eslEMEM;
  bath_hip_oprofile_get_fwd(om->dev, om->rf_host, om->tf_host);
  bath_hip_oprofile_get_ssv_scores(om->dev, om->ssv_host);
  return ih_refresh_scalars(om, om->L);
}

/* p7_oprofile_Convert (p7_oprofile.c:1091): the three limited-precision score systems are built on the device side of the C
 * ABI (host_model.cpp restates p7_oprofile.c:773-994 bit for bit); here the generic profile is flattened and handed over. */
int p7_oprofile_Convert(const P7_PROFILE *gm, P7_OPROFILE *om)
{
  const int M = gm->M, Kp = gm->abc->Kp;
  if (gm->abc->type != om->abc->type) ESL_EXCEPTION(eslEINVAL, "alphabets of gm, om don't match");
  if (gm->M > om->allocM)             ESL_EXCEPTION(eslEINVAL, "oprofile is too small");
  if (Kp != IH_KP)                    ESL_EXCEPTION(eslEINVAL, "impl_hip scores amino-acid profiles");
  if (om->gm_copy) { free(om->gm_copy->tsc); free(om->gm_copy->rsc); free(om->gm_copy); }
  bath_profile *c = calloc(1, sizeof *c);
  if (!c) return eslEMEM;
  c->M = M; c->L = gm->L; c->max_length = gm->max_length; c->nj = gm->nj;
  c->tsc = malloc(sizeof(float) * (size_t) M * p7P_NTRANS);
  c->rsc = malloc(sizeof(float) * (size_t) Kp * (M + 1) * p7P_NR);
  if (!c->tsc || !c->rsc) return eslEMEM;
  memcpy(c->tsc, gm->tsc, sizeof(float) * (size_t) M * p7P_NTRANS);
  for (int x = 0; x < Kp; x++) memcpy(c->rsc + (size_t) x * (M + 1) * p7P_NR, gm->rsc[x], sizeof(float) * (size_t)(M + 1) * p7P_NR);
  /* bath_profile.xsc rows are E N J C x [LOOP, MOVE], the generic layout (hmmer.h:202-218) */
  for (int s = 0; s < p7P_NXSTATES; s++) for (int t = 0; t < p7P_NXTRANS; t++) c->xsc[s][t] = gm->xsc[s][t];
  for (int z = 0; z < p7_NEVPARAM && z < BATH_NEVPARAM; z++) c->evparam[z] = gm->evparam[z];
  for (int z = 0; z < BATH_K_AMINO; z++) c->compo[z] = gm->compo[z];
  om->gm_copy = c;

  free(om->name); free(om->acc); free(om->desc);
  om->name = ih_strdup(gm->name); om->acc = ih_strdup(gm->acc); om->desc = ih_strdup(gm->desc);
  free(om->rf); free(om->mm); free(om->cs); free(om->consensus);
  om->rf = ih_annot(gm->rf, M); om->mm = ih_annot(gm->mm, M); om->cs = ih_annot(gm->cs, M); om->consensus = ih_annot(gm->consensus, M);
  for (int z = 0; z < p7_NEVPARAM; z++) om->evparam[z] = gm->evparam[z];
  for (int z = 0; z < p7_NCUTOFFS; z++) om->cutoff[z]  = gm->cutoff[z];
  for (int z = 0; z < p7_MAXABET;  z++) om->compo[z]   = gm->compo[z];
  om->mode = gm->mode; om->L = gm->L; om->M = gm->M; om->max_length = gm->max_length; om->nj = gm->nj;
  return ih_upload(om);
}

/* p7_oprofile_ReconfigLength (p7_oprofile.c:1261) = ReconfigMSVLength + ReconfigRestLength.  On the device the per-length
 * scalars are table entries (tjb_b(L), xw / xf MOVE and LOOP of N, C, J) that every kernel indexes with the target's own
 * length, so "reconfiguring" is recording L and refreshing the host-visible scalars. */
int p7_oprofile_ReconfigLength(P7_OPROFILE *om, int L)
{
  om->L = L;
  return ih_refresh_scalars(om, L);
}
int p7_oprofile_ReconfigMSVLength(P7_OPROFILE *om, int L)                       /* p7_oprofile.c:1289: tjb_b only */
{
  bath_oprofile_scalars s;
  if (bath_hip_oprofile_scalars(om->dev, L, &s) != BATH_OK) return eslFAIL;
  om->tjb_b = s.tjb_b;
  return eslOK;
}
/* p7_oprofile_ReconfigMultihit / Unihit (p7_oprofile.c:1330-1400): E->J / E->C split and nj; the kernels take the hit mode as
 * an argument of the call (unihit for envelopes, multihit for parsers and regions), chosen from om->nj by the functions of
 * impl_hip_domain.c. */
int p7_oprofile_ReconfigMultihit(P7_OPROFILE *om, int L)
{
  if (p7_oprofile_ReconfigLength(om, L) != eslOK) return eslFAIL;       /* multihit is the configuration the device tables hold */
  om->nj = 1.0f;
  return eslOK;
}
int p7_oprofile_ReconfigUnihit(P7_OPROFILE *om, int L)
{
  if (p7_oprofile_ReconfigLength(om, L) != eslOK) return eslFAIL;
  om->xf[p7O_E][p7O_MOVE] = 1.0f; om->xf[p7O_E][p7O_LOOP] = 0.0f;
  om->nj = 0.0f;
  om->xw[p7O_E][p7O_MOVE] = 0; om->xw[p7O_E][p7O_LOOP] = -32768;
  /* unihit length model: pmove = 2 / (L+2) (p7_oprofile.c:1383-1390) */
  const float pmove = 2.0f / ((float) L + 2.0f), ploop = 1.0f - pmove;
  om->xf[p7O_N][p7O_LOOP] = om->xf[p7O_C][p7O_LOOP] = om->xf[p7O_J][p7O_LOOP] = ploop;
  om->xf[p7O_N][p7O_MOVE] = om->xf[p7O_C][p7O_MOVE] = om->xf[p7O_J][p7O_MOVE] = pmove;
  return eslOK;
}

/* accessors (p7_oprofile.c:1470-1600): what p7_scoredata.c builds its SSV window tables from */
int p7_oprofile_GetFwdTransitionArray(const P7_OPROFILE *om, int type, float *arr)
{
  static const int gen[p7O_NTRANS] = { 3 /*BM*/, 0 /*MM*/, 1 /*IM*/, 2 /*DM*/, 4 /*MD*/, 6 /*MI*/, 7 /*II*/, 5 /*DD*/ };   /* p7O_* -> generic column of tf_host */
  /* arr[k], k = 1..M, is the value the striped vectors hold at node k (p7_oprofile.c:1471-1477): the transition INTO k for BM, MM,
   * IM, DM and OUT of k for the others -- the convention of tf_host */
  for (int k = 1; k <= om->M; k++) arr[k] = om->tf_host[(size_t) k * 8 + gen[type]];
  return eslOK;
}
int p7_oprofile_GetSSVEmissionScoreArray(const P7_OPROFILE *om, uint8_t *arr)
{
  memcpy(arr, om->ssv_host, (size_t)(om->M + 1) * IH_KP);
  return eslOK;
}
int p7_oprofile_GetFwdEmissionScoreArray(const P7_OPROFILE *om, float *arr)      /* arr[k*Kp + x] = log odds (p7_oprofile.c:1548) */
{
  for (int k = 1; k <= om->M; k++)
    for (int x = 0; x < IH_KP; x++) arr[(size_t) k * IH_KP + x] = logf(om->rf_host[(size_t) x * (om->M + 1) + k]);
  return eslOK;
}

/* ------------------------------------------------------------------ P7_FS_OPROFILE */

P7_FS_OPROFILE *p7_fs_oprofile_Create(int allocM, const ESL_ALPHABET *abc, int codon_lengths)     /* p7_fs_oprofile.c:60 */
{
  P7_FS_OPROFILE *om = calloc(1, sizeof *om);
  if (!om) return NULL;
  om->allocM = allocM; om->abc = abc; om->mode = p7_NO_MODE; om->codon_lengths = codon_lengths;
  om->allocQ4 = p7O_NQF(allocM);
  om->rf = calloc((size_t) allocM + 2, 1); om->mm = calloc((size_t) allocM + 2, 1);
  om->cs = calloc((size_t) allocM + 2, 1); om->consensus = calloc((size_t) allocM + 2, 1);
  for (int x = 0; x < p7_NEVPARAM; x++) om->evparam[x] = p7_EVPARAM_UNSET;
  for (int x = 0; x < p7_NCUTOFFS; x++) om->cutoff[x]  = p7_CUTOFF_UNSET;
  for (int x = 0; x < p7_MAXABET;  x++) om->compo[x]   = p7_COMPO_UNSET;
  for (int x = 0; x < p7_NOFFSETS; x++) om->offs[x]    = -1;
  om->roff = om->eoff = -1;
  return om;
}
int p7_fs_oprofile_IsLocal(const P7_FS_OPROFILE *om_fs) { return (om_fs->mode == p7_LOCAL || om_fs->mode == p7_UNILOCAL) ? TRUE : FALSE; }

void p7_fs_oprofile_Destroy(P7_FS_OPROFILE *om)
{
  if (!om) return;
  if (!om->clone) {
    if (om->dev) bath_hip_fsprofile_destroy(om->dev);
    if (om->gm_copy) { free(om->gm_copy->tsc); free(om->gm_copy->rsc); free(om->gm_copy->codons); free(om->gm_copy->indel_pos); free(om->gm_copy); }
    free(om->name); free(om->acc); free(om->desc);
    free(om->rf); free(om->mm); free(om->cs); free(om->consensus);
  }
  free(om);
}
P7_FS_OPROFILE *p7_fs_oprofile_Clone(const P7_FS_OPROFILE *om)
{
  P7_FS_OPROFILE *c = malloc(sizeof *c);
  if (!c) return NULL;
  memcpy(c, om, sizeof *c);
  c->clone = 1;
  return c;
}

/* special-state transitions of the codon model for amino length L (p7_fs_ReconfigLength, modelconfig.c:767-770) in log space */
static void ih_fs_length(P7_FS_OPROFILE *om, int L)
{
  const float pmove = (2.0f + om->nj) / ((float) L + 2.0f + om->nj), ploop = 1.0f - pmove;
  om->xf[p7O_N][p7O_LOOP] = om->xf[p7O_C][p7O_LOOP] = om->xf[p7O_J][p7O_LOOP] = (float) log((double) ploop);
  om->xf[p7O_N][p7O_MOVE] = om->xf[p7O_C][p7O_MOVE] = om->xf[p7O_J][p7O_MOVE] = (float) log((double) pmove);
  om->L = L;
}

int p7_fs_oprofile_Convert(const P7_FS_PROFILE *gm, P7_FS_OPROFILE *om)         /* p7_fs_oprofile.c:221 */
{
  const int M = gm->M, Kp = gm->abc->Kp;
  const int maxcodons = (gm->codon_lengths == 5) ? p7P_MAXCODONS5 : (gm->codon_lengths == 3 ? p7P_MAXCODONS3 : p7P_MAXCODONS1);
  if (gm->M > om->allocM) ESL_EXCEPTION(eslEINVAL, "fs oprofile is too small");
  if (om->gm_copy) { free(om->gm_copy->tsc); free(om->gm_copy->rsc); free(om->gm_copy->codons); free(om->gm_copy->indel_pos); free(om->gm_copy); }
  bath_fs_profile *c = calloc(1, sizeof *c);
  if (!c) return eslEMEM;
  c->M = M; c->L = gm->L; c->max_length = gm->max_length; c->codon_lengths = gm->codon_lengths; c->maxcodons = maxcodons;
  c->nj = gm->nj; c->fsprob = gm->fsprob;
  c->tsc = malloc(sizeof(float) * (size_t) M * p7P_NTRANS);
  c->rsc = malloc(sizeof(float) * (size_t)(maxcodons + Kp) * (M + 1));
  c->codons = malloc((size_t)(M + 1) * maxcodons); c->indel_pos = malloc((size_t)(M + 1) * maxcodons);
  if (!c->tsc || !c->rsc || !c->codons || !c->indel_pos) return eslEMEM;
  memcpy(c->tsc, gm->tsc, sizeof(float) * (size_t) M * p7P_NTRANS);
  for (int r = 0; r < maxcodons + Kp; r++) memcpy(c->rsc + (size_t) r * (M + 1), gm->rsc[r], sizeof(float) * (size_t)(M + 1));
  /* gm->codons[c][k] -> [(k)*maxcodons + c] (hmmer.h:401-402 vs bath_fs_profile) */
  for (int k = 0; k <= M; k++)
    for (int q = 0; q < maxcodons; q++) {
      c->codons[(size_t) k * maxcodons + q]    = gm->codons    ? gm->codons[q][k]    : 0;
      c->indel_pos[(size_t) k * maxcodons + q] = gm->indel_pos ? gm->indel_pos[q][k] : 0;
    }
  for (int s = 0; s < p7P_NXSTATES; s++) for (int t = 0; t < p7P_NXTRANS; t++) c->xsc[s][t] = gm->xsc[s][t];
  for (int z = 0; z < p7_NEVPARAM && z < BATH_NEVPARAM; z++) c->evparam[z] = gm->evparam[z];
  for (int z = 0; z < BATH_K_AMINO; z++) c->compo[z] = gm->compo[z];
  om->gm_copy = c;
  if (om->dev) { bath_hip_fsprofile_destroy(om->dev); om->dev = NULL; }
  if (bath_hip_fsprofile_convert(impl_hip_context(), c, &om->dev) != BATH_OK) return eslFAIL;

  free(om->name); free(om->acc); free(om->desc);
  om->name = ih_strdup(gm->name); om->acc = ih_strdup(gm->acc); om->desc = ih_strdup(gm->desc);
  free(om->rf); free(om->mm); free(om->cs); free(om->consensus);
  om->rf = ih_annot(gm->rf, M); om->mm = ih_annot(gm->mm, M); om->cs = ih_annot(gm->cs, M); om->consensus = ih_annot(gm->consensus, M);
  for (int z = 0; z < p7_NEVPARAM; z++) om->evparam[z] = gm->evparam[z];
  for (int z = 0; z < p7_NCUTOFFS; z++) om->cutoff[z]  = gm->cutoff[z];
  for (int z = 0; z < p7_MAXABET;  z++) om->compo[z]   = gm->compo[z];
  om->mode = gm->mode; om->M = M; om->max_length = gm->max_length; om->nj = gm->nj;
  om->codon_lengths = gm->codon_lengths; om->fsprob = gm->fsprob;
  om->xf[p7O_E][p7O_LOOP] = gm->xsc[p7P_E][p7P_LOOP]; om->xf[p7O_E][p7O_MOVE] = gm->xsc[p7P_E][p7P_MOVE];
  ih_fs_length(om, gm->L);
  return eslOK;
}
int p7_fs_oprofile_ReconfigLength(P7_FS_OPROFILE *om, int L) { ih_fs_length(om, L); return eslOK; }       /* p7_fs_oprofile.c: ReconfigLength */
int p7_fs_oprofile_ReconfigMultihit(P7_FS_OPROFILE *om, int L)                  /* modelconfig.c:825-831 in log space */
{
  om->xf[p7O_E][p7O_MOVE] = om->xf[p7O_E][p7O_LOOP] = (float) -eslCONST_LOG2;
  om->nj = 1.0f;
  ih_fs_length(om, L);
  return eslOK;
}
int p7_fs_oprofile_ReconfigUnihit(P7_FS_OPROFILE *om, int L)                    /* modelconfig.c:862-870 */
{
  om->xf[p7O_E][p7O_MOVE] = 0.0f; om->xf[p7O_E][p7O_LOOP] = -eslINFINITY;
  om->nj = 0.0f;
  ih_fs_length(om, L);
  return eslOK;
}

/* ------------------------------------------------------------------ P7_OIVX */

P7_OIVX *p7_oivx_Create(int M_hint, int C)
{
  P7_OIVX *ov = calloc(1, sizeof *ov);
  if (ov) { ov->allocM = M_hint; ov->allocC = C; ov->allocQ4 = p7O_NQF(M_hint); }
  return ov;
}
int  p7_oivx_GrowTo(P7_OIVX *ov, int M, int C) { if (M > ov->allocM) { ov->allocM = M; ov->allocQ4 = p7O_NQF(M); } if (C > ov->allocC) ov->allocC = C; return eslOK; }
void p7_oivx_Destroy(P7_OIVX *ov) { free(ov); }

/* ------------------------------------------------------------------ P7_OMX */

extern void impl_hip_pass_release(struct impl_hip_pass *p);     /* impl_hip_domain.c */
extern int  impl_hip_pass_cells(const P7_OMX *ox, int i, int k, int s, float *ret);

static int ih_omx_rows(P7_OMX *ox, int allocXL)
{
  if (allocXL + 1 <= ox->allocXR) return eslOK;
  float *x = realloc(ox->x_mem, sizeof(float) * (size_t)(allocXL + 1) * p7X_NXCELLS);
  if (!x) return eslEMEM;
  ox->x_mem = x; ox->xmx = x; ox->allocXR = allocXL + 1;
  return eslOK;
}

P7_OMX *p7_omx_Create_dpf(int allocM, int allocL, int allocXL, int nscells)      /* p7_omx.c */
{
  P7_OMX *ox = calloc(1, sizeof *ox);
  if (!ox) return NULL;
  ox->nscells = nscells;
  ox->allocR = allocL + 1; ox->validR = allocL + 1;
  ox->allocQ4 = p7O_NQF(allocM); ox->allocQ8 = p7O_NQW(allocM); ox->allocQ16 = p7O_NQB(allocM);
  ox->ncells = (size_t)(allocL + 1) * (size_t) ox->allocQ4 * 4;
  if (ih_omx_rows(ox, allocXL) != eslOK) { free(ox); return NULL; }
  ox->M = 0; ox->L = 0; ox->totscale = 0.0f; ox->has_own_scales = TRUE;
  return ox;
}
P7_OMX *p7_omx_Create(int allocM, int allocL, int allocXL) { return p7_omx_Create_dpf(allocM, allocL, allocXL, p7X_NSCELLS); }

int p7_omx_GrowTo_dpf(P7_OMX *ox, int allocM, int allocL, int allocXL)           /* the matrices grow on the device, per pass */
{
  if (allocL + 1 > ox->allocR) { ox->allocR = allocL + 1; ox->validR = allocL + 1; }
  if (p7O_NQF(allocM) > ox->allocQ4) { ox->allocQ4 = p7O_NQF(allocM); ox->allocQ8 = p7O_NQW(allocM); ox->allocQ16 = p7O_NQB(allocM); }
  ox->ncells = (size_t) ox->allocR * (size_t) ox->allocQ4 * 4;
  return ih_omx_rows(ox, allocXL);
}
int p7_omx_GrowTo(P7_OMX *ox, int allocM, int allocL, int allocXL) { return p7_omx_GrowTo_dpf(ox, allocM, allocL, allocXL); }

int p7_omx_Reuse(P7_OMX *ox)
{
  if (ox->pass) { impl_hip_pass_release(ox->pass); ox->pass = NULL; }
  ox->M = 0; ox->L = 0; ox->totscale = 0.0f; ox->has_own_scales = TRUE;
  return eslOK;
}
void p7_omx_Destroy(P7_OMX *ox)
{
  if (!ox) return;
  if (ox->pass) impl_hip_pass_release(ox->pass);
  free(ox->x_mem);
  free(ox);
}

/* p7_omx_FDeconvert (p7_omx.c:455): the float matrix in generic P7_GMX layout, cells fetched from the pass the matrix belongs to */
int p7_omx_FDeconvert(P7_OMX *ox, P7_GMX *gx)
{
  if (!ox->pass) ESL_EXCEPTION(eslEINVAL, "matrix holds no result");
  for (int i = 0; i <= ox->L; i++) {
    gx->dp[i][0 * p7G_NSCELLS + p7G_M] = gx->dp[i][0 * p7G_NSCELLS + p7G_I] = gx->dp[i][0 * p7G_NSCELLS + p7G_D] = 0.0f;
    for (int k = 1; k <= ox->M; k++) {
      float v;
      if (impl_hip_pass_cells(ox, i, k, p7X_M, &v) != eslOK) ESL_EXCEPTION(eslEINVAL, "this matrix was not kept on the host");
      gx->dp[i][k * p7G_NSCELLS + p7G_M] = v;
      impl_hip_pass_cells(ox, i, k, p7X_D, &v); gx->dp[i][k * p7G_NSCELLS + p7G_D] = v;
      impl_hip_pass_cells(ox, i, k, p7X_I, &v); gx->dp[i][k * p7G_NSCELLS + p7G_I] = v;
    }
    gx->xmx[i * p7G_NXCELLS + p7G_E] = ox->xmx[i * p7X_NXCELLS + p7X_E];
    gx->xmx[i * p7G_NXCELLS + p7G_N] = ox->xmx[i * p7X_NXCELLS + p7X_N];
    gx->xmx[i * p7G_NXCELLS + p7G_J] = ox->xmx[i * p7X_NXCELLS + p7X_J];
    gx->xmx[i * p7G_NXCELLS + p7G_B] = ox->xmx[i * p7X_NXCELLS + p7X_B];
    gx->xmx[i * p7G_NXCELLS + p7G_C] = ox->xmx[i * p7X_NXCELLS + p7X_C];
  }
  gx->L = ox->L; gx->M = ox->M;
  return eslOK;
}
