/* pipeline.c -- the per-ORF filter cascade of p7_Pipeline_BATH().  ORACLE (test infra only).
 *
 * Restates p7_pipeline.c:1632-1791 (MSV -> bias -> Viterbi/SSV windows -> local-composition
 * re-filter -> Forward with F3, or F4 when fs_pipe) for both strands of one DNA window, with the
 * driver semantics of bathsearch.c:1053-1117 (top strand, then reverse complement).  Domain
 * definition and everything after it are outside this file.  The counters are the ones
 * p7_pli_Statistics prints (p7_pipeline.c:1851-1869) and are what tests/golden pins.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529

void bo_pipeline_init(bo_pipeline *pli, int fs_pipe)      /* p7_pipeline.c:219-222 defaults */
{
  memset(pli, 0, sizeof *pli);
  pli->F1 = 0.02; pli->F2 = 1e-3; pli->F3 = 1e-5; pli->F4 = 5e-4;
  pli->E = 10.0;
  pli->do_biasfilter = 1;
  pli->fs_pipe = fs_pipe;
  pli->minlen = 20;
  pli->do_null2 = 1; pli->std_pipe = 1; pli->strands = 0; pli->initiator = 0; pli->ct = 1;   /* p7_pipeline.c:107, :199; bathsearch.c:97, :718 */
  pli->inc_by_E = 1; pli->T = 0.0; pli->seed = 42;                                            /* p7_pipeline.c:98, :148, :165 */
}

/* p7_pli_ComputeLocalCompo, p7_pipeline.c:427-458 */
void bo_local_compo(const bo_scoredata *sd, const bo_oprofile *om, const bo_bg *bg, int k_start, int k_end, float *compo)
{
  int Kp = BO_KP_AMINO;
  int k_len = k_end - k_start + 1;
  if (k_len < 20) { k_start -= (20 - k_len) / 2; k_end += (20 - k_len) / 2; }
  if (k_start < 1) k_start = 1;
  if (k_end > om->M) k_end = om->M;
  for (int x = 0; x < BO_K_AMINO; x++) compo[x] = 0.0f;
  for (int k = k_start; k <= k_end; k++)
    for (int x = 0; x < BO_K_AMINO; x++) {
      float log_odds = ((float) om->base_b - (float) sd->ssv_scores[k * Kp + x]) / om->scale_b;
      compo[x] += bg->f[x] * expf(log_odds);
    }
  /* esl_vec_FNorm (easel: compensated sum, then divide) */
  float sum = 0.f, c = 0.f;
  for (int x = 0; x < BO_K_AMINO; x++) { float y = compo[x] - c; float t = sum + y; c = (t - sum) - y; sum = t; }
  if (sum != 0.0f) for (int x = 0; x < BO_K_AMINO; x++) compo[x] /= sum;
  else             for (int x = 0; x < BO_K_AMINO; x++) compo[x] = 1.0f / BO_K_AMINO;
}

typedef struct {                 /* what the frameshift stage needs beyond the cascade (NULL gm3: no such stage) */
  bo_fs_profile *gm3, *gm5;      /* gm5 NULL: stop at the branch decision */
  const uint8_t *basic, *dsq;    /* dsq[1..n]: the strand being read */
  int n;
  bo_fswindow **fw; int *nfw, *fw_alloc;
  bo_fsdomain **doms; int *ndom, *dom_alloc, *nskipped;
} fs_stage;

static void strand_cascade(bo_pipeline *pli, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg,
                           const bo_orfblock *blk, int strand, bo_orfresult **res, int *nres, int *res_alloc, const fs_stage *fs)
{
  bo_windowlist hw;
  bo_windowlist_init(&hw);
  double *P_orf = malloc(sizeof(double) * (size_t)(blk->count + 1));      /* p7_pipeline.c:1615-1623 */
  float *fwd_null = malloc(sizeof(float) * (size_t)(blk->count + 1));
  for (int i = 0; i < blk->count; i++) { P_orf[i] = 1.0; fwd_null[i] = -INFINITY; }
  for (int i = 0; i < blk->count; i++) {
    const bo_orf *o = &blk->orf[i];
    const uint8_t *dsq = blk->aa + o->off;
    int n = o->n;
    if (*nres == *res_alloc) { *res_alloc = *res_alloc ? *res_alloc * 2 : 256; *res = realloc(*res, sizeof(bo_orfresult) * (size_t) *res_alloc); }
    bo_orfresult *r = &(*res)[(*nres)++];
    memset(r, 0, sizeof *r);
    r->strand = strand; r->frame = o->frame; r->start = o->start; r->end = o->end; r->n = n;
    r->vfsc = -INFINITY; r->fwdsc = -INFINITY; r->P = 1.0; r->stage = 0;
    pli->n_orfs++;
    if (n <= 0) continue;
    /* p7_pipeline.c:1635-1637: the ORF showed up completely within the previous window (orfsq->start of the bottom strand is a
     * top-strand coordinate, window length - o->start + 1) */
    if (pli->context > 0 && fs && ((strand == 0 && o->end < pli->context) || (strand == 1 && fs->n - o->start + 1 < pli->context))) { r->stage = -1; continue; }

    float usc, vfsc = -INFINITY, fwdsc, filtersc, seqsc;
    double P;
    bo_bg_setlength(bg, n);                                    /* p7_pipeline.c:1643-1645 */
    bo_oprofile_reconfig_length(om, n);
    float nullsc = bo_bg_nullone(bg, n);
    r->nullsc = nullsc;

    r->msv_status = bo_k_msvfilter(dsq, n, om, &usc);            /* :1649-1652 */
    pli->cells_msv += (int64_t) n * om->M;
    r->usc = usc;
    seqsc = (float)((usc - nullsc) / LOG2C);
    P = bo_gumbel_surv(seqsc, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
    r->P = P;
    if (P > pli->F1) continue;
    pli->n_past_msv++; pli->pos_past_msv += (int64_t) n * 3;
    r->stage = 1;

    if (pli->do_biasfilter) {                                  /* :1657-1663 */
      filtersc = bo_bg_filterscore(bg, dsq, n);
      seqsc = (float)((usc - filtersc) / LOG2C);
      P = bo_gumbel_surv(seqsc, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
      r->P = P; r->filtersc = filtersc;
      if (P > pli->F1) continue;
    } else filtersc = nullsc;
    r->filtersc = filtersc;
    pli->n_past_bias++; pli->pos_past_bias += (int64_t) n * 3;
    r->stage = 2;

    int old_cnt = hw.count;
    if (P > pli->F2) {                                         /* :1669-1675 */
      r->vit_status = bo_k_vitfilter_bath(dsq, n, om, sd, filtersc, pli->F2, &hw, &vfsc);
      pli->cells_vit += (int64_t) n * om->M;
      seqsc = (float)((vfsc - filtersc) / LOG2C);
      P = bo_gumbel_surv(seqsc, om->evparam[BO_VMU], om->evparam[BO_VLAMBDA]);
      r->vfsc = vfsc; r->P = P;
      if (P > pli->F2) { hw.count = old_cnt; continue; }
    } else {
      bo_ssvfilter_bath(dsq, n, om, sd, bg, pli->F1, &hw);     /* :1676-1677 */
    }
    for (int w = old_cnt; w < hw.count; w++) hw.w[w].id = i;
    pli->n_past_vit++; pli->pos_past_vit += (int64_t) n * 3;
    r->stage = 3;

    if (pli->do_biasfilter && old_cnt < hw.count) {            /* :1684-1718 */
      int k_max = hw.w[old_cnt].k;
      int k_min = k_max - hw.w[old_cnt].length + 1;
      for (int w = old_cnt + 1; w < hw.count; w++) {
        if (hw.w[w].k > k_max) k_max = hw.w[w].k;
        if (hw.w[w].k - hw.w[w].length + 1 < k_min) k_min = hw.w[w].k - hw.w[w].length + 1;
      }
      float lc[BO_K_AMINO];
      bo_local_compo(sd, om, bg, k_min, k_max, lc);
      bo_bg_setfilter(bg, om->M, lc);
      bo_bg_setlength(bg, n);
      float local_filtersc = bo_bg_filterscore(bg, dsq, n);
      int rejected = 0;
      if (local_filtersc > filtersc) {
        filtersc = local_filtersc;
        if (vfsc == -INFINITY) {
          seqsc = (float)((usc - filtersc) / LOG2C);
          P = bo_gumbel_surv(seqsc, om->evparam[BO_MMU], om->evparam[BO_MLAMBDA]);
          if (P > pli->F2) {
            r->vit_status = bo_k_vitfilter(dsq, n, om, &vfsc);
            pli->cells_vit += (int64_t) n * om->M;
            seqsc = (float)((vfsc - filtersc) / LOG2C);
            P = bo_gumbel_surv(seqsc, om->evparam[BO_VMU], om->evparam[BO_VLAMBDA]);
            if (P > pli->F2) rejected = 1;
          }
        } else {
          seqsc = (float)((vfsc - filtersc) / LOG2C);
          P = bo_gumbel_surv(seqsc, om->evparam[BO_VMU], om->evparam[BO_VLAMBDA]);
          if (P > pli->F2) rejected = 1;
        }
      }
      bo_bg_setfilter(bg, om->M, om->compo);
      bo_bg_setlength(bg, n);
      r->filtersc = filtersc; r->vfsc = vfsc; r->P = P;
      /* NB the reference counts the ORF as past-Vit before this re-filter may reject it (:1682) */
      if (rejected) { hw.count = old_cnt; r->stage = 2; continue; }
    }

    bo_k_forward_parser(dsq, n, om, &fwdsc);                  /* :1735 / :1779 */
    pli->cells_fwd += (int64_t) n * om->M;
    seqsc = (float)((fwdsc - filtersc) / LOG2C);
    P = bo_exp_surv(seqsc, om->evparam[BO_FTAU], om->evparam[BO_FLAMBDA]);
    r->fwdsc = fwdsc; r->P = P;
    if (pli->fs_pipe) { P_orf[i] = P; fwd_null[i] = fwdsc - nullsc; }   /* :1781-1782 */
    if (P > (pli->fs_pipe ? pli->F4 : pli->F3)) continue;
    r->stage = 4;
    pli->n_past_fwd++;
    if (!pli->fs_pipe) pli->pos_past_fwd += (int64_t) n * 3;  /* :1761; the fs branch counts later (:1468,1490) */
    if (!pli->fs_pipe && fs && fs->doms && !fs->gm3)          /* :1763-1771: Backward parser, domain definition, hit scores */
      bo_domaindef_std(pli, om, bg, dsq, n, o->start, o->start, strand, fs->n, fs->doms, fs->ndom, fs->dom_alloc, fs->nskipped, fs->dsq);
  }
  if (pli->fs_pipe && fs && fs->gm3)                            /* :1793 */
    bo_pli_frameshift(pli, om, fs->gm3, fs->gm5, sd, bg, fs->basic, blk, P_orf, fwd_null, &hw, fs->dsq, fs->n, strand, fs->fw, fs->nfw, fs->fw_alloc,
                      fs->doms, fs->ndom, fs->dom_alloc, fs->nskipped);
  free(P_orf); free(fwd_null);
  bo_windowlist_free(&hw);
}

int bo_pipeline_window_fs(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, const bo_scoredata *sd, bo_bg *bg,
                          const uint8_t basic[64], const uint8_t *dna, int n,
                          bo_orfresult **res, int *nres, int *res_alloc, bo_fswindow **fw, int *nfw, int *fw_alloc)
{
  return bo_pipeline_window_fsdom(pli, om, gm3, NULL, sd, bg, basic, dna, n, res, nres, res_alloc, fw, nfw, fw_alloc, NULL, NULL, NULL, NULL);
}

int bo_pipeline_window_fsdom(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, bo_fs_profile *gm5, const bo_scoredata *sd, bo_bg *bg,
                             const uint8_t basic[64], const uint8_t *dna, int n,
                             bo_orfresult **res, int *nres, int *res_alloc, bo_fswindow **fw, int *nfw, int *fw_alloc,
                             bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped)
{
  if (n < 15) return BO_OK;                                   /* bathsearch.c:1066, p7_pipeline.c:1605 */
  bo_orfblock blk;
  bo_orfblock_init(&blk);
  bo_bg_setfilter(bg, om->M, om->compo);                      /* p7_pli_NewModel, p7_pipeline.c:635 */
  fs_stage fs = { gm3, gm5, basic, dna, n, fw, nfw, fw_alloc, doms, ndom, dom_alloc, nskipped };
  uint8_t is_init[64];                                        /* bathsearch.c:718-719: -m, -M, or any codon */
  if (pli->initiator && bo_gencode_initiators(pli->ct, pli->initiator, is_init) != BO_OK) return BO_EINVAL;
  const uint8_t *ini = pli->initiator ? is_init : NULL;
  bo_set_seed(pli->seed);

  if (pli->strands != 2) {                                    /* bathsearch.c:1069, :1256: not p7_STRAND_BOTTOMONLY */
    pli->nres += n - pli->context;                            /* top strand: dnaSeq->W, bathsearch.c:1258 */
    bo_translate_orfs_init(dna, n, basic, ini, pli->initiator != 0, pli->minlen, &blk);
    strand_cascade(pli, om, sd, bg, &blk, 0, res, nres, res_alloc, &fs);
    bo_orfblock_reuse(&blk);
  }
  if (pli->strands != 1) {                                    /* :1082, :1267: not p7_STRAND_TOPONLY */
    uint8_t *rc = malloc((size_t) n + 2);                     /* bottom strand, bathsearch.c:1084-1091 */
    bo_revcomp(dna, n, rc);
    pli->nres += n - pli->context;
    bo_translate_orfs_init(rc, n, basic, ini, pli->initiator != 0, pli->minlen, &blk);
    fs.dsq = rc;
    strand_cascade(pli, om, sd, bg, &blk, 1, res, nres, res_alloc, &fs);
    free(rc);
  }
  bo_orfblock_free(&blk);
  return BO_OK;
}

/* the plain pipeline with domain definition and hit scores for the ORFs that pass the Forward filter */
int bo_pipeline_window_hits(bo_pipeline *pli, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg,
                            const uint8_t basic[64], const uint8_t *dna, int n,
                            bo_orfresult **res, int *nres, int *res_alloc, bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped)
{
  return bo_pipeline_window_fsdom(pli, om, NULL, NULL, sd, bg, basic, dna, n, res, nres, res_alloc, NULL, NULL, NULL, doms, ndom, dom_alloc, nskipped);
}

int bo_pipeline_window(bo_pipeline *pli, bo_oprofile *om, const bo_scoredata *sd, bo_bg *bg,
                       const uint8_t basic[64], const uint8_t *dna, int n,
                       bo_orfresult **res, int *nres, int *res_alloc)
{
  return bo_pipeline_window_fs(pli, om, NULL, sd, bg, basic, dna, n, res, nres, res_alloc, NULL, NULL, NULL);
}
