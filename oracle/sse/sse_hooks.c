/* sse_hooks.c -- lets the oracle's cascade (oracle/pipeline.c) run on the SSE2 striped kernels instead of the scalar ones, so
 * that bench.py's cpu_baseline times an impl_sse-equivalent pipeline.  TEST / MEASUREMENT INFRASTRUCTURE ONLY. */
#include <stddef.h>
#include "bath_sse.h"

static int use_sse = 0;
static __thread bs_oprofile *cached = NULL;          /* striped copy of the profile the cascade is running with */

void bo_pipeline_use_sse(int on) { use_sse = on; }

static bs_oprofile *striped(const bo_oprofile *om)
{
  if (!cached || cached->om != om || cached->M != om->M) {
    bs_oprofile_free(cached);
    cached = bs_oprofile_create(om);
  }
  return cached;
}

int bo_k_msvfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)
{
  return use_sse ? bs_msvfilter(dsq, L, striped(om), ret_sc) : bo_msvfilter(dsq, L, om, ret_sc);
}
int bo_k_vitfilter(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)
{
  return use_sse ? bs_vitfilter(dsq, L, striped(om), ret_sc) : bo_vitfilter(dsq, L, om, ret_sc);
}
int bo_k_vitfilter_bath(const uint8_t *dsq, int L, const bo_oprofile *om, const bo_scoredata *sd, float filtersc, double P, bo_windowlist *wl, float *ret_sc)
{
  return use_sse ? bs_vitfilter_bath(dsq, L, striped(om), sd, filtersc, P, wl, ret_sc) : bo_vitfilter_bath(dsq, L, om, sd, filtersc, P, wl, ret_sc);
}
int bo_k_forward_parser(const uint8_t *dsq, int L, const bo_oprofile *om, float *ret_sc)
{
  return use_sse ? bs_forward_parser(dsq, L, striped(om), ret_sc) : bo_forward_parser(dsq, L, om, NULL, ret_sc);
}

/* ---- the frameshift stage: the 3-codon Forward parser of a DNA window */
static int fs_use_sse = 0;
static __thread bs_fsprofile *fs_cached = NULL;

void bo_fs_use_sse(int on) { fs_use_sse = on; }

int bo_k_gforward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc)
{
  if (!fs_use_sse) return bo_gforward_parser_fs3(dsq, L, gm3, gx, ret_sc);
  if (!fs_cached || fs_cached->gm != gm3 || fs_cached->M != gm3->M) { bs_fsprofile_free(fs_cached); fs_cached = bs_fsprofile_create(gm3); }
  return bs_fs3_forward_parser(dsq, L, fs_cached, gx ? gx->xmx : NULL, ret_sc);
}

int bo_k_gbackward_parser_fs3(const uint8_t *dsq, int L, const bo_fs_profile *gm3, bo_gmx *gx, float *ret_sc)
{
  if (!fs_use_sse) return bo_gbackward_parser_fs3(dsq, L, gm3, gx, ret_sc);
  if (!fs_cached || fs_cached->gm != gm3 || fs_cached->M != gm3->M) { bs_fsprofile_free(fs_cached); fs_cached = bs_fsprofile_create(gm3); }
  return bs_fs3_backward_parser(dsq, L, fs_cached, gx ? gx->xmx : NULL, ret_sc);
}
