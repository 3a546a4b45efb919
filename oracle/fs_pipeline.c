/* fs_pipeline.c -- ORACLE (test infrastructure): the frameshift stage of p7_Pipeline_BATH, up to the decision
 * which branch a DNA window takes.
 *
 * Restates, in plain scalar C:
 *   p7_pli_BuildDNAWindows   src/p7_pipeline.c:462-572  (ORFs with P <= F4 -> padded DNA windows -> sort -> merge)
 *   p7_pli_Frameshift        src/p7_pipeline.c:1339-1515 (per window: summed ORF score, null and bias scores of the
 *                            window, 3-codon frameshift Forward, the P-value comparison)
 * Domain definition and hit post-processing after the decision are not restated.
 * Parity: unpinned against recorded reference numbers (the reference prints nothing at this stage); the GPU path is
 * checked against this file window by window.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

#define LOG2C 0.69314718055994529

typedef struct { int64_t n; int32_t k, length; } dwin;      /* P7_HMM_WINDOW fields used here: n, k, length */

static int dwin_cmp(const void *a, const void *b)          /* p7_hmmwindow.c:133 window_pos_sorter */
{
  const dwin *x = a, *y = b;
  return (x->n > y->n) - (x->n < y->n);
}

static void fsw_push(bo_fswindow **fw, int *nfw, int *alloc, const bo_fswindow *r)
{
  if (*nfw == *alloc) { *alloc = *alloc ? *alloc * 2 : 64; *fw = realloc(*fw, sizeof(bo_fswindow) * (size_t) *alloc); }
  (*fw)[(*nfw)++] = *r;
}

/* dsq[1..n]: the strand being read (already reverse-complemented for complementarity = 1) */
int bo_pli_frameshift(bo_pipeline *pli, bo_oprofile *om, bo_fs_profile *gm3, bo_fs_profile *gm5, const bo_scoredata *sd, bo_bg *bg, const uint8_t basic[64],
                      const bo_orfblock *blk, const double *P_orf, const float *fwdsc, const bo_windowlist *hw,
                      const uint8_t *dsq, int n, int complementarity, bo_fswindow **fw, int *nfw, int *fw_alloc,
                      bo_fsdomain **doms, int *ndom, int *dom_alloc, int *nskipped)
{
  const int norf = blk->count;
  dwin *wl = malloc(sizeof(dwin) * (size_t)(norf + 1));
  int nw = 0;

  /* ---- p7_pli_BuildDNAWindows (:462), pct_overlap = 0 */
  for (int f = 0; f < norf; f++) {
    if (P_orf[f] > pli->F4) continue;
    const bo_orf *o = &blk->orf[f];
    int best = -1;
    float best_score = -INFINITY;
    for (int w = 0; w < hw->count; w++) {                     /* :486-495 best window of this ORF */
      if (hw->w[w].id != f) continue;
      if (hw->w[w].score > best_score || (hw->w[w].score == best_score && hw->w[w].length > (best >= 0 ? hw->w[best].length : 0))) {
        best_score = hw->w[w].score; best = w;
      }
    }
    int32_t cn, ck, clen;
    if (best >= 0) { cn = hw->w[best].n; ck = hw->w[best].k; clen = hw->w[best].length; }
    else if (o->n >= om->M) { cn = (o->n - om->M) / 2 + 1; ck = om->M; clen = om->M; }        /* :500-510 fallback */
    else { cn = 1; ck = om->M - ((om->M - o->n) / 2); clen = o->n; }
    /* :513-514, uint32 n promoted to double, truncated into int64 */
    int64_t ws = (int64_t)((double)(uint32_t) cn - (om->max_length * (0.1 + sd->prefix_lengths[ck - clen + 1])) + 1);
    int64_t we = (int64_t)((double)(uint32_t) cn + (double)(uint32_t) clen + (om->max_length * (0.1 + sd->suffix_lengths[ck])) - 2);
    if (ws > 0) ws = 0;                                       /* :516 ESL_MIN(0, .) */
    if (we < o->n) we = o->n;                                 /* :517 ESL_MAX(orf->n, .) */
    /* :520-527.  o->start is the ORF's first nucleotide on the strand being read: the reference's
     * (dnasq->n - curr_orf->start + 1) for the bottom strand and curr_orf->start for the top strand */
    ws = (int64_t) o->start + ws * 3; if (ws < 1) ws = 1;
    we = (int64_t) o->start + we * 3; if (we > n) we = n;
    wl[nw].n = ws; wl[nw].k = ck; wl[nw].length = (int32_t)(we - ws + 1); nw++;
  }
  if (nw == 0) { free(wl); return BO_OK; }
  char *aligned = calloc((size_t) norf + 1, 1);               /* oxf_holder[i] freed: the ORF went through the standard branch */
  qsort(wl, (size_t) nw, sizeof(dwin), dwin_cmp);
  int cnt = 0;
  for (int i = 1; i < nw; i++) {                              /* :541-566 */
    dwin *prev = &wl[cnt], *cur = &wl[i];
    int64_t ov_s = prev->n > cur->n ? prev->n : cur->n;
    int64_t pe = prev->n + prev->length - 1, ce = cur->n + cur->length - 1;
    int64_t ov_e = pe < ce ? pe : ce;
    int32_t ov_len = (int32_t)(ov_e - ov_s + 1);
    int64_t m_s = prev->n < cur->n ? prev->n : cur->n;
    int64_t m_e = pe > ce ? pe : ce;
    int32_t m_len = (int32_t)(m_e - m_s + 1);
    int32_t minlen = prev->length < cur->length ? prev->length : cur->length;
    if (((float) ov_len / minlen > 0.f) && m_len < (2 * (om->max_length * 3))) { prev->n = m_s; prev->length = m_len; }
    else { cnt++; wl[cnt] = wl[i]; }
  }
  nw = cnt + 1;

  /* ---- p7_pli_Frameshift (:1368-1470) */
  const int64_t dstart = complementarity ? n : 1;            /* dnasq->start of a whole sequence */
  for (int w = 0; w < nw; w++) {
    bo_fswindow r;
    memset(&r, 0, sizeof r);
    r.strand = complementarity; r.n = (int32_t) wl[w].n; r.length = wl[w].length; r.k = wl[w].k;
    int64_t wstart = complementarity ? dstart - (wl[w].n + wl[w].length) : dstart + wl[w].n - 1;
    int64_t wend   = complementarity ? dstart - wl[w].n + 1 : wstart + wl[w].length - 1;
    const uint8_t *wdsq = dsq + wl[w].n - 1;                  /* 1-based view, sentinel-free interior */
    int orf_cnt = 0, k_min = om->M, k_max = 0, last = 0;
    float tot = -INFINITY;
    double P_min = INFINITY;
    for (int i = 0; i < norf; i++) {
      if (P_orf[i] > pli->F4) continue;
      const bo_orf *o = &blk->orf[i];
      int64_t os, oe;
      if (complementarity) {          /* reference orfsq->start/end are top-strand coordinates: n - pos + 1 */
        int64_t rs = (int64_t) n - o->start + 1, re = (int64_t) n - o->end + 1;
        os = dstart - (n - re + 1) + 1; oe = dstart - (n - rs + 1) + 1;
      } else { os = dstart + o->start - 1; oe = dstart + o->end - 1; }
      if (os >= wstart && oe <= wend) {
        if (P_orf[i] < P_min) P_min = P_orf[i];
        tot = bo_flogsum(tot, fwdsc[i]);
        orf_cnt++;
        int h = last;
        while (h < hw->count && hw->w[h].id != i) h++;
        if (h < hw->count) {
          while (h < hw->count && hw->w[h].id == i) {
            int ks = hw->w[h].k - hw->w[h].length + 1;
            if (ks < k_min) k_min = ks;
            if (hw->w[h].k > k_max) k_max = hw->w[h].k;
            h++;
          }
          last = h;
        }
      }
    }
    double P_tot = bo_exp_surv(tot / LOG2C, om->evparam[BO_FTAU], om->evparam[BO_FLAMBDA]);
    if (!pli->std_pipe) P_tot = 1.0;                          /* :1457, --fsonly */
    const int L = wl[w].length;
    bo_bg_setlength(bg, L / 3);
    float nullsc = bo_bg_fs_nullone(bg, L / 3);
    float filtersc;
    if (pli->do_biasfilter) {
      filtersc = bo_bg_fs_filterscore(bg, wdsq, L, basic);
      if (k_min <= k_max) {
        float lc[BO_K_AMINO];
        bo_local_compo(sd, om, bg, k_min, k_max, lc);
        bo_bg_setfilter(bg, om->M, lc);
        bo_bg_setlength(bg, L / 3);
        float local = bo_bg_fs_filterscore(bg, wdsq, L, basic);
        if (local > filtersc) filtersc = local;
        bo_bg_setfilter(bg, om->M, om->compo);
        bo_bg_setlength(bg, L / 3);
      }
    } else filtersc = nullsc;
    bo_fs_profile_reconfig_length(gm3, L / 3);
    bo_gmx *gx = bo_gmx_create(gm3->M, L + 1, L, 3);
    float fsc = -INFINITY;
    bo_k_gforward_parser_fs3(wdsq, L, gm3, gx, &fsc);
    bo_gmx_free(gx);
    float seqscore = (float)((fsc - filtersc) / LOG2C);
    double P_fs = bo_exp_surv(seqscore, gm3->evparam[BO_FTAUFS3], gm3->evparam[BO_FLAMBDA]);
    double P_null = bo_exp_surv((fsc - nullsc) / LOG2C, gm3->evparam[BO_FTAUFS3], gm3->evparam[BO_FLAMBDA]);
    r.orf_cnt = orf_cnt; r.k_min = k_min; r.k_max = k_max; r.tot_orfsc = tot; r.nullsc = nullsc; r.filtersc = filtersc; r.fwdsc = fsc;
    r.P_tot = P_tot; r.P_min = P_min; r.P_fs = P_fs; r.P_null = P_null;
    if (P_fs <= pli->F3 && (P_null < P_tot || (P_null == P_tot && orf_cnt > 1) || P_min > pli->F3)) {   /* :1464 */
      r.branch = 1;
      pli->pos_past_fwd += L;
      if (gm5 && doms) {                                      /* :1469-1476 */
        const int before = *ndom;
        bo_domaindef_fs(pli, gm3, gm5, bg, wdsq, L, (int) wl[w].n, complementarity, n, doms, ndom, dom_alloc, nskipped);
        r.ndom = *ndom - before;
      }
    } else if (!pli->std_pipe) {
      r.branch = 0;                                           /* :1480: --fsonly has no standard branch, the window is dropped */
    } else {                                                  /* :1479 std_pipe */
      r.branch = 2;
      for (int i = 0; i < norf; i++) {
        const bo_orf *o = &blk->orf[i];
        if (P_orf[i] > pli->F3) continue;
        int64_t os, oe;
        if (complementarity) { int64_t rs = (int64_t) n - o->start + 1, re = (int64_t) n - o->end + 1; os = dstart - (n - re + 1) + 1; oe = dstart - (n - rs + 1) + 1; }
        else { os = dstart + o->start - 1; oe = dstart + o->end - 1; }
        if (!(os >= wstart && oe <= wend)) continue;          /* orfsq->idx == w */
        if (aligned[i]) continue;                             /* oxf_holder[i] == NULL: an overlapping window already took it */
        aligned[i] = 1;
        pli->pos_past_fwd += (int64_t) o->n * 3;
        if (doms)                                             /* :1489-1505: Backward parser, domain definition, hit scores */
          bo_domaindef_std(pli, om, bg, blk->aa + o->off, o->n, o->start, (int) wl[w].n, complementarity, n, doms, ndom, dom_alloc, nskipped, dsq);
      }
    }
    fsw_push(fw, nfw, fw_alloc, &r);
  }
  free(wl); free(aligned);
  return BO_OK;
}
