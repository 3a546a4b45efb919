// bath_profile.hip -- device context, optimized-profile construction and sequence blocks.
//
//   bath_hip_oprofile_convert  <- p7_oprofile_Convert()      src/impl_sse/p7_oprofile.c:1091
//                                 (mf_conversion :773, vf_conversion :826, fb_conversion :929, sf_conversion :721)
//   length tables              <- p7_oprofile_ReconfigLength src/impl_sse/p7_oprofile.c:1261-1326,
//                                 p7_bg_SetLength/NullOne    src/p7_bg.c:189,356
// The striped SIMD layout of the reference is an implementation detail of impl_sse; here every
// table is indexed by model node and laid out for the kernels in bath_filters.hip.
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstring>

#include "bath_common.hpp"
#include "host_model.hpp"

using namespace bath;

namespace bath {

static const double kLog2 = 0.69314718055994529;

// Tail probabilities (easel esl_gumbel.c / esl_exponential.c closed forms; call sites p7_pipeline.c:1651,1737).
double gumbel_surv(double x, double mu, double lambda) {
  double ey = -std::exp(-lambda * (x - mu));
  if (std::fabs(ey) < 5e-9) return -ey;
  return 1.0 - std::exp(ey);
}
double gumbel_invsurv(double p, double mu, double lambda) {
  double lp = (p < 5e-9) ? (std::pow(p, p) - 1) / p : std::log(-1. * std::log(1 - p));
  return mu - lp / lambda;
}
double exp_surv(double x, double mu, double lambda) { return (x < mu) ? 1.0 : std::exp(-lambda * (x - mu)); }

// Limited-precision converters (p7_oprofile.c:667-705).
static inline uint8_t cost_u8(float scale, float sc) {
  float c = -1.0f * roundf(scale * sc);
  return (c > 255.) ? 255 : (uint8_t)(int)c;
}
static inline uint8_t cost_u8_biased(float scale, uint8_t bias, float sc) {
  float c = -1.0f * roundf(scale * sc);
  return (c > 255 - bias) ? 255 : (uint8_t)((int)c + bias);
}
static inline int16_t score_i16(float scale, float sc) {
  float v = roundf(scale * sc);
  if (v >= 32767.0) return 32767;
  if (v <= -32768.0) return -32768;
  return (int16_t)v;
}

// Emission odds of the two-state bias-filter HMM (p7_bg_SetFilter p7_bg.c:449 + esl_hmm_Configure).
void bias_filter_eo(const float *compo, float eo[kKp][2]) {
  for (int x = 0; x < 20; x++) { eo[x][0] = kAminoBg[x] / kAminoBg[x]; eo[x][1] = compo[x] / kAminoBg[x]; }
  for (int x : {20, 27, 28}) eo[x][0] = eo[x][1] = 1.0f;
  for (int x = 21; x <= 26; x++)
    for (int s = 0; s < 2; s++) {
      float num = 0.f, den = 0.f;
      for (int y = 0; y < 20; y++)
        if (amino_degen_has(x, y)) { num += (s == 0 ? kAminoBg[y] : compo[y]); den += kAminoBg[y]; }
      eo[x][s] = den > 0.f ? num / den : 0.f;
    }
}

template <class T>
static hipError_t upload(T **dst, const T *src, size_t n, hipStream_t st) {
  hipError_t e = hipMalloc((void **)dst, n * sizeof(T) + 64);
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, st);
}

int build_len_tables(const bath_hip_oprofile *om, int maxL) {
  std::lock_guard<std::mutex> lock(om->grow_mu);
  LenTables &lt = om->lt;
  if (maxL <= lt.maxL) return BATH_OK;
  bath_hip_ctx *ctx = om->ctx;
  int n = std::max(maxL, 4096) + 1;
  for (void *p : {(void *)lt.d_tjb, (void *)lt.d_xwmove, (void *)lt.d_pmove, (void *)lt.d_nullsc, (void *)lt.d_lt1, (void *)lt.d_lt2, (void *)lt.d_p1})
    if (p) om->retired.push_back(p);                            // freed with the profile: another thread's kernel may be reading it
  lt.d_tjb = nullptr; lt.d_xwmove = nullptr; lt.d_pmove = nullptr; lt.d_nullsc = nullptr; lt.d_lt1 = nullptr; lt.d_lt2 = nullptr; lt.d_p1 = nullptr;
  lt.h_tjb.resize(n); lt.h_xwmove.resize(n); lt.h_pmove.resize(n); lt.h_nullsc.resize(n);
  std::vector<float> lt1(n), lt2(n), p1v(n);
  for (int L = 0; L < n; L++) {
    lt.h_tjb[L] = cost_u8(om->scale_b, logf(3.0f / (float)(L + 3)));                 // p7_oprofile.c:1288
    float pmove = (2.0f + om->nj) / ((float)L + 2.0f + om->nj);                      // p7_oprofile.c:1311
    lt.h_pmove[L] = pmove;
    lt.h_xwmove[L] = score_i16(om->scale_w, logf(pmove));                            // p7_oprofile.c:1319
    float p1 = (float)L / (float)(L + 1);                                            // p7_bg.c:191
    p1v[L] = p1;
    lt.h_nullsc[L] = (float)((float)L * std::log((double)p1) + std::log(1. - p1));           // p7_bg.c:358
    lt1[L] = (float)L * logf(p1);                                                    // p7_bg.c:501
    lt2[L] = logf((float)(1. - p1));
  }
  BATH_HIP_TRY(ctx, upload(&lt.d_tjb, lt.h_tjb.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_xwmove, lt.h_xwmove.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_pmove, lt.h_pmove.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_nullsc, lt.h_nullsc.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_lt1, lt1.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_lt2, lt2.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, upload(&lt.d_p1, p1v.data(), n, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  lt.maxL = n - 1;
  return BATH_OK;
}

}  // namespace bath

int bath_hip_oprofile::ensure_len_tables(int maxL) const { return bath::build_len_tables(this, maxL); }

// ------------------------------------------------------------------------------------------ context

namespace bath { std::atomic<int> g_ctx_total{0}, g_ctx_internal{0}; }

extern "C" int bath_hip_init(int device, bath_hip_ctx **out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BATH_ENODEVICE;
  if (device < 0 || device >= ndev) return BATH_EINVAL;
  bath_hip_ctx *ctx = new bath_hip_ctx();
  ctx->device = device;
  // BATH_HIP_SYNC=block|yield: how a host thread waits in hipStreamSynchronize (the runtime's default spins: with several worker
  // contexts, each with lanes and side threads, the waiting threads then take cores from the ensembles' host threads)
  if (const char *e = std::getenv("BATH_HIP_SYNC")) {
    static const bool once = [&] {
      (void)hipSetDevice(device);
      const unsigned fl = (e[0] == 'b') ? hipDeviceScheduleBlockingSync : (e[0] == 'y') ? hipDeviceScheduleYield : hipDeviceScheduleSpin;
      if (hipSetDeviceFlags(fl) != hipSuccess) (void)hipGetLastError();
      return true;
    }();
    (void)once;
  }
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return BATH_ENODEVICE;
  }
  bath::g_ctx_total.fetch_add(1);
  *out = ctx;
  return BATH_OK;
}

extern "C" void bath_hip_finalize(bath_hip_ctx *ctx) {
  if (!ctx) return;
  bath::g_ctx_total.fetch_sub(1);
  if (ctx->internal) bath::g_ctx_internal.fetch_sub(1);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete ctx->lane_pool; ctx->lane_pool = nullptr;                 // joins the lanes' host threads
  for (bath_hip_ctx *lane : ctx->lanes) bath_hip_finalize(lane);
  ctx->lanes.clear();
  if (ctx->ev_lanes) (void)hipEventDestroy(ctx->ev_lanes);
  if (ctx->tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
  if (ctx->aux) { bath_hip_finalize(ctx->aux); ctx->aux = nullptr; }
  if (ctx->aux2) { bath_hip_finalize(ctx->aux2); ctx->aux2 = nullptr; }
  if (ctx->aux3) { bath_hip_finalize(ctx->aux3); ctx->aux3 = nullptr; }
  for (auto &b : ctx->scratch) b.release();
  for (auto &b : ctx->pinned) b.release();
  for (auto &b : ctx->stage) b.release();
  ctx->results_pinned.release();
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  for (auto e : ctx->span_events) (void)hipEventDestroy(e);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->spec_stream) { (void)hipStreamSynchronize(ctx->spec_stream); (void)hipStreamDestroy(ctx->spec_stream); }
  if (ctx->ev_spec) (void)hipEventDestroy(ctx->ev_spec);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// Give back what the context created lazily to run parts of a call side by side: the lanes (contexts, scratch memory, host
// threads), the side contexts of the domain stages, the side / copy streams.  All of it is created again on demand.  For a context
// that will sit idle while others work (a process that switches from one big block to several worker contexts): HIP maps a
// process's streams onto a handful of hardware queues, and an idle context's dozen streams take their share of them.  Records and
// arrays returned by earlier calls on this context are invalid afterwards.
extern "C" int bath_hip_trim(bath_hip_ctx *ctx) {
  if (!ctx) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  delete ctx->lane_pool; ctx->lane_pool = nullptr;
  for (bath_hip_ctx *lane : ctx->lanes) bath_hip_finalize(lane);
  ctx->lanes.clear();
  if (ctx->aux) { bath_hip_finalize(ctx->aux); ctx->aux = nullptr; }
  if (ctx->aux2) { bath_hip_finalize(ctx->aux2); ctx->aux2 = nullptr; }
  if (ctx->aux3) { bath_hip_finalize(ctx->aux3); ctx->aux3 = nullptr; }
  if (ctx->tail_stream) { (void)hipStreamDestroy(ctx->tail_stream); ctx->tail_stream = nullptr; }
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; }
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); ctx->side_stream = nullptr;
    if (ctx->ev_fork) { (void)hipEventDestroy(ctx->ev_fork); ctx->ev_fork = nullptr; }
    if (ctx->ev_join) { (void)hipEventDestroy(ctx->ev_join); ctx->ev_join = nullptr; }
  }
  if (ctx->spec_stream) {
    (void)hipStreamSynchronize(ctx->spec_stream); (void)hipStreamDestroy(ctx->spec_stream); ctx->spec_stream = nullptr;
    if (ctx->ev_spec) { (void)hipEventDestroy(ctx->ev_spec); ctx->ev_spec = nullptr; }
  }
  ctx->fs_spec_valid = false; ctx->fs_spec_rows.clear();
  ctx->fs_std_orfs.clear(); ctx->fs_std_pool = nullptr; ctx->fs_keep_xoff.clear(); ctx->fs_regions_all.clear();
  ctx->d_records = nullptr; ctx->n_records = 0;
  return BATH_OK;
}

extern "C" const char *bath_hip_last_error(const bath_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" int bath_hip_synchronize(bath_hip_ctx *ctx) {
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}

extern "C" void *bath_hip_stream(bath_hip_ctx *ctx) { return (void *)ctx->stream; }
extern "C" int bath_hip_kernel_times(bath_hip_ctx *ctx, int max, bath_kernel_time *out) {
  if (!ctx || !out) return 0;
  int n = 0;
  for (const bath_hip_ctx *c : {(const bath_hip_ctx *)ctx, (const bath_hip_ctx *)ctx->aux2, (const bath_hip_ctx *)ctx->aux3}) {     // the regions' Forward runs on a context of its own
    if (!c) continue;
    for (const bath::KernelSpan &k : c->spans) {
      float ms = 0.f;
      if (hipEventSynchronize(k.b) != hipSuccess || hipEventElapsedTime(&ms, k.a, k.b) != hipSuccess) continue;
      int i = 0;
      while (i < n && out[i].name != k.name) i++;
      if (i == n) { if (n == max) continue; out[n++] = bath_kernel_time{k.name, 0.f, 0, 0.0, 0.0}; }
      out[i].ms += ms; out[i].launches++; out[i].cells += k.cells; out[i].bytes += k.bytes;
    }
  }
  return n;
}
extern "C" int bath_hip_set_fs_serial(bath_hip_ctx *ctx, int on) {
  if (!ctx) return BATH_EINVAL;
  ctx->fs_serial = on < 0 ? -1 : (on ? 1 : 0);
  return BATH_OK;
}

extern "C" int bath_hip_set_fs_strict(bath_hip_ctx *ctx, int on) {
  if (!ctx) return BATH_EINVAL;
  ctx->fs_strict = on ? 1 : 0;
  if (!on && ctx->aux3) { bath_hip_finalize(ctx->aux3); ctx->aux3 = nullptr; }      // the fast mode has no use for the clusters' context, and its idle streams cost it 15 ms per pass
  for (bath_hip_ctx *l : ctx->lanes) l->fs_strict = ctx->fs_strict;
  return BATH_OK;
}

// ------------------------------------------------------------------------------------------ oprofile

extern "C" void bath_hip_oprofile_destroy(bath_hip_oprofile *om) {
  if (!om) return;
  for (void *p : {(void *)om->d_emit, (void *)om->d_ssv, (void *)om->d_msv, (void *)om->d_rb, (void *)om->d_rw, (void *)om->d_tw, (void *)om->d_rf, (void *)om->d_tf,
                  (void *)om->d_bias_eo, (void *)om->d_vit_rw, (void *)om->d_vit_tw2, (void *)om->d_vit_rank, (void *)om->lt.d_tjb, (void *)om->lt.d_xwmove, (void *)om->lt.d_pmove,
                  (void *)om->lt.d_nullsc, (void *)om->lt.d_lt1, (void *)om->lt.d_lt2, (void *)om->lt.d_p1, (void *)om->d_cons, (void *)om->d_msc, (void *)om->d_tsc, (void *)om->d_rfb, (void *)om->d_tfb})
    if (p) (void)hipFree(p);
  for (void *p : om->retired) (void)hipFree(p);
  delete om;
}

// P7_OPROFILE.consensus (impl_sse.h:128; copied by p7_oprofile_Convert): the alignment display compares aligned residues with
// it for the hit table's percent identity.  Stored digitized as esl_abc_DigitizeSymbol would (case-insensitive).
extern "C" int bath_hip_oprofile_set_consensus(bath_hip_oprofile *om, const char *consensus) {
  if (!om || !consensus) return BATH_EINVAL;
  bath_hip_ctx *ctx = om->ctx;
  static const char syms[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
  om->consensus.assign((size_t)om->M + 2, ' ');
  om->cons_digital.assign((size_t)om->M + 1, 255);
  for (int k = 1; k <= om->M && consensus[k]; k++) {
    om->consensus[(size_t)k] = consensus[k];
    const char *q = std::strchr(syms, std::toupper((unsigned char)consensus[k]));
    if (q) om->cons_digital[(size_t)k] = (uint8_t)(q - syms);
  }
  if (!om->d_cons) BATH_HIP_TRY(ctx, hipMalloc(&om->d_cons, (size_t)om->M + 1));
  BATH_HIP_TRY(ctx, hipMemcpy(om->d_cons, om->cons_digital.data(), (size_t)om->M + 1, hipMemcpyHostToDevice));
  return BATH_OK;
}

extern "C" int bath_hip_oprofile_M(const bath_hip_oprofile *om) { return om->M; }

extern "C" int bath_hip_oprofile_convert(bath_hip_ctx *ctx, const bath_profile *gm, bath_hip_oprofile **ret) {
  *ret = nullptr;
  const int M = gm->M;
  if (M < 1) { ctx->set_error("profile has no nodes"); return BATH_EINVAL; }
  const size_t W = (size_t)M + 1;
  auto msc = [&](int k, int x) { return gm->rsc[(size_t)x * W * 2 + 2 * k]; };
  auto tsc = [&](int k, int s) { return gm->tsc[(size_t)k * 8 + s]; };   // valid for 0 <= k < M
  enum { MM, IM, DM, BM, MD, DD, MI, II };

  bath_hip_oprofile *om = new bath_hip_oprofile();
  { static std::atomic<uint64_t> next_uid{1}; om->uid = next_uid.fetch_add(1); }
  om->ctx = ctx; om->M = M; om->max_length = gm->max_length; om->nj = gm->nj; om->L0 = gm->L;
  std::memcpy(om->evparam, gm->evparam, sizeof om->evparam);
  std::memcpy(om->compo, gm->compo, sizeof om->compo);

  // ---- MSV bytes (mf_conversion): third-bit units, base 190, bias = -max score
  float maxsc = 0.0f;
  for (int x = 0; x < 20; x++)
    for (size_t i = 0; i < W * 2; i++) maxsc = std::max(maxsc, gm->rsc[(size_t)x * W * 2 + i]);
  om->scale_b = (float)(3.0 / kLog2);
  om->base_b = 190;
  om->bias_b = cost_u8(om->scale_b, (float)(-1.0 * maxsc));
  om->rb.assign((size_t)kKp * W, 255);
  for (int x = 0; x < kKp; x++)
    for (int k = 1; k <= M; k++) om->rb[x * W + k] = cost_u8_biased(om->scale_b, om->bias_b, msc(k, x));
  om->tbm_b = cost_u8(om->scale_b, logf(2.0f / ((float)M * (float)(M + 1))));
  om->tec_b = cost_u8(om->scale_b, logf(0.5f));

  // ---- Viterbi words (vf_conversion): 1/500 bit units; into-node transitions come from gm node k-1,
  //      out-of-node ones from gm node k (absent, i.e. -32768, at k == M); no transition above 0, II <= -1.
  om->scale_w = (float)(500.0 / kLog2);
  om->base_w = 12000;
  om->rw.assign((size_t)kKp * W, -32768);
  for (int x = 0; x < kKp; x++)
    for (int k = 1; k <= M; k++) om->rw[x * W + k] = score_i16(om->scale_w, msc(k, x));
  om->tw.assign(W * 8, -32768);
  for (int k = 1; k <= M; k++) {
    int16_t *t = &om->tw[(size_t)k * 8];
    for (int s : {BM, MM, IM, DM}) t[s] = std::min<int16_t>(score_i16(om->scale_w, tsc(k - 1, s)), 0);
    for (int s : {MD, MI}) t[s] = (k < M) ? std::min<int16_t>(score_i16(om->scale_w, tsc(k, s)), 0) : (int16_t)-32768;
    t[II] = (k < M) ? std::min<int16_t>(score_i16(om->scale_w, tsc(k, II)), -1) : (int16_t)-32768;
    t[DD] = (k < M) ? score_i16(om->scale_w, tsc(k, DD)) : (int16_t)-32768;
  }
  om->xw_E[0] = score_i16(om->scale_w, gm->xsc[0][0]);
  om->xw_E[1] = score_i16(om->scale_w, gm->xsc[0][1]);
  int ddb = -32768;
  for (int k = 2; k < M - 1; k++)
    ddb = std::max(ddb, (int)score_i16(om->scale_w, tsc(k, DD)) + (int)score_i16(om->scale_w, tsc(k + 1, DM)) -
                            (int)score_i16(om->scale_w, tsc(k + 1, BM)));
  om->ddbound_w = (int16_t)ddb;

  // ---- Forward odds ratios (fb_conversion)
  om->rf.assign((size_t)kKp * W, 0.f);
  for (int x = 0; x < kKp; x++)
    for (int k = 1; k <= M; k++) om->rf[x * W + k] = expf(msc(k, x));
  om->tf.assign(W * 8, 0.f);
  for (int k = 1; k <= M; k++) {
    float *t = &om->tf[(size_t)k * 8];
    for (int s : {BM, MM, IM, DM}) t[s] = expf(tsc(k - 1, s));
    for (int s : {MD, MI, II, DD}) t[s] = (k < M) ? expf(tsc(k, s)) : 0.f;
  }
  om->xf_E[0] = expf(gm->xsc[0][0]);
  om->xf_E[1] = expf(gm->xsc[0][1]);

  // ---- P7_SCOREDATA prefix/suffix fractions (p7_hmm_ScoreDataComputeRest, p7_scoredata.c:357-380): how much of MAXL a DNA
  // window reserves before / after a seed diagonal that starts / ends at node k.  Inputs are the Forward transition
  // odds MI, II exactly as p7_oprofile_GetFwdTransitionArray returns them.
  {
    std::vector<float> &pre = om->prefix_lengths, &suf = om->suffix_lengths;
    pre.assign(W, 0.f); suf.assign(W, 0.f);
    float total = 0;
    for (int k = 1; k < M; k++) {
      const float t_mi = om->tf[(size_t)k * 8 + MI], t_ii = om->tf[(size_t)k * 8 + II];
      pre[k] = (t_mi == 0) ? 1.f : (float)(1 + (int)(std::log(1e-7 / t_mi) / std::log((double)t_ii)));   // p7_DEFAULT_WINDOW_BETA
      total += pre[k];
    }
    pre[0] = pre[M] = 0;
    for (int k = 1; k < M; k++) pre[k] /= total;
    suf[M] = pre[M - 1];
    for (int k = M - 1; k >= 1; k--) suf[k] = suf[k + 1] + pre[k - 1];
    for (int k = 2; k < M; k++) pre[k] += pre[k - 1];
  }

  // ---- device layouts
  // SSV: signed costs sb = min(rb - bias, 127) (sf_conversion), stored as binary16 increments -sb * 2^-11, one row per
  // residue plus a "reset" row; columns 1..M real, padded with -1.0 (a full reset) up to 2*NR*G.  Row pitch is an odd multiple of 16 bytes so
  // that lanes holding different residues spread over the 16 sixteen-byte LDS slots on ds_read_b128.
  // Two lanes per target from 153 to 304 nodes: a lane's tile then stays within 76 registers and the kernel runs 4 waves per
  // SIMD, where the packed 16-bit ops issue at 4.4e11/s (with 80-152 cell registers a SIMD holds one or two waves:
  // M = 185...247: ssv 0.29-0.38 -> 0.22-0.27 ms on a 12.5 Mb genome).  Beyond 304 nodes the round-1 rule stays -- tiles of up
  // to 208 registers before more lanes are added: two lanes x 112 registers at M = 409, four / eight lanes with narrow tiles
  // at M = 409 / 1024 were measured and lose.  BATH_HIP_SSV_WIDE=1: one lane up to 416 nodes (rounds 1-2);
  // BATH_HIP_SSV_G2_MAX: the upper end of the two-lane range, for A/B runs (tools/ssv_shape_probe.py).
  static const bool wide = [] { const char *e = std::getenv("BATH_HIP_SSV_WIDE"); return e && e[0] == '1'; }();
  static const int g2max = [] { const char *e = std::getenv("BATH_HIP_SSV_G2_MAX"); return e ? std::atoi(e) : 304; }();
  int G = (!wide && M > 152 && M <= g2max) ? 2 : 1;             // (305...416 nodes: one lane with 156-208 registers is 1.2x faster than two with 112, measured at M = 409)
  while (G < 8 && M > 416 * G) G *= 2;                        // up to 208 registers (416 nodes) per lane
  if (M > 416 * G) { ctx->set_error("model longer than 3328 nodes"); delete om; return BATH_EINVAL; }
  int NR = ((M + G - 1) / G + 1) / 2;
  if (G == 1 && NR <= 112) NR = (NR + 3) / 4 * 4;             // the shapes of BATH_SSV_SHAPES
  else if (G == 2 && NR <= 76) NR = NR <= 40 ? 40 : (NR <= 72 ? (NR + 7) / 8 * 8 : 76);
  else NR = std::max((NR + 15) / 16 * 16, G > 1 ? 112 : 16);
  NR = std::max(NR, 16);
  om->NR = NR; om->G = G;
  om->ssv_row_bytes = 16 * ((NR * G / 4 + 1) | 1);            // 16-byte aligned rows, (pitch/16) odd
  {
    size_t rowsz = (size_t)om->ssv_row_bytes / 2;
    // cells are binary16 numbers in units of 2^-11 (see ssv_row): the table holds the increments -cost * 2^-11, exact
    auto half_bits = [](float v) { const _Float16 h = (_Float16)v; int16_t b; std::memcpy(&b, &h, 2); return b; };
    std::vector<int16_t> tab((size_t)kSsvRows * rowsz, half_bits(-1.0f));   // padding nodes and the reset row: back to the begin score
    // node k -> lane tile g = (k-1)/(2NR); inside the tile, register r = (k-1)%NR, half (k-1)/NR%2 (see ssv_row)
    for (int x = 0; x < kKp; x++)
      for (int k = 1; k <= M; k++) {
        const int g = (k - 1) / (2 * NR), q = (k - 1) - g * 2 * NR;
        const int cost = std::min((int)om->rb[x * W + k] - (int)om->bias_b, 127);
        tab[x * rowsz + (size_t)g * 2 * NR + 2 * (q % NR) + q / NR] = half_bits(-(float)cost / 2048.0f);
      }
    BATH_HIP_TRY(ctx, upload(&om->d_ssv, tab.data(), tab.size(), ctx->stream));
    if (G == 1 && NR <= 76) {
      // MSV (J state) with a lane per target (bath_msv_lane.hip): the same tile layout, the byte costs unclipped -- the increment of
      // a cell is bias - rb (msvfilter.c:160-162); padding nodes and the reset row: -1.0, the cell back to 0
      std::vector<int16_t> mtab((size_t)kSsvRows * rowsz, half_bits(-1.0f));
      for (int x = 0; x < kKp; x++)
        for (int k = 1; k <= M; k++) {
          const int q = k - 1;
          const int inc = (int)om->bias_b - (int)om->rb[x * W + k];
          mtab[x * rowsz + 2 * (q % NR) + q / NR] = half_bits((float)inc / 2048.0f);
        }
      BATH_HIP_TRY(ctx, upload(&om->d_msv, mtab.data(), mtab.size(), ctx->stream));
    }
  }
  // lane-per-target Viterbi tables (bath_viterbi.hip)
  {
    // pairs of nodes per lane: the kernel is instantiated in steps of 16, and in steps of 4 between 64 and 80 (every padded pair costs
    // a full pair's instructions: M = 145 needs 73, 76 instead of 80 saves 5 % of the kernel)
    int NRv = ((M + 1) / 2 + 15) / 16 * 16;
    if (NRv == 80) NRv = std::max(68, ((M + 1) / 2 + 3) / 4 * 4);
    if (NRv <= 112) {
      om->vit_NR = NRv;
      om->vit_rw_pitch = 4 * NRv + 16;
      const size_t rowsz = (size_t)om->vit_rw_pitch / 2;
      std::vector<int16_t> rwt((size_t)30 * rowsz, (int16_t)-32768);
      // register r of a lane holds node r+1 in its low half and node NRv+r+1 in its high half (bath_viterbi.hip)
      auto slot = [&](int node) { return node <= NRv ? 2 * (node - 1) : 2 * (node - NRv - 1) + 1; };
      for (int x = 0; x < kKp; x++)
        for (int k = 1; k <= M; k++) rwt[x * rowsz + slot(k)] = om->rw[x * W + k];
      std::vector<uint32_t> tw2((size_t)NRv * 8);
      std::vector<int16_t> rank((size_t)2 * NRv, (int16_t)32767);
      const int Q8 = std::max(2, ((M - 1) / 8) + 1);
      auto twv = [&](int node, int s) -> int { return node <= M ? (int)om->tw[(size_t)node * 8 + s] : -32768; };
      for (int r = 0; r < NRv; r++)
        for (int s = 0; s < 8; s++) tw2[(size_t)r * 8 + s] = ((uint32_t)(uint16_t)twv(r + 1, s)) | ((uint32_t)(uint16_t)twv(NRv + r + 1, s) << 16);
      for (int node = 1; node <= M; node++) rank[(size_t)slot(node)] = (int16_t)(((node - 1) % Q8) * 8 + (node - 1) / Q8);
      BATH_HIP_TRY(ctx, upload(&om->d_vit_rw, rwt.data(), rwt.size(), ctx->stream));
      BATH_HIP_TRY(ctx, upload(&om->d_vit_tw2, tw2.data(), tw2.size(), ctx->stream));
      BATH_HIP_TRY(ctx, upload(&om->d_vit_rank, rank.data(), rank.size(), ctx->stream));
    }
  }
  om->rb_stride = (int)W;
  BATH_HIP_TRY(ctx, upload(&om->d_rb, om->rb.data(), om->rb.size(), ctx->stream));
  BATH_HIP_TRY(ctx, upload(&om->d_rw, om->rw.data(), om->rw.size(), ctx->stream));
  BATH_HIP_TRY(ctx, upload(&om->d_tw, om->tw.data(), om->tw.size(), ctx->stream));
  BATH_HIP_TRY(ctx, upload(&om->d_rf, om->rf.data(), om->rf.size(), ctx->stream));
  BATH_HIP_TRY(ctx, upload(&om->d_tf, om->tf.data(), om->tf.size(), ctx->stream));
  {
    std::vector<float> tfb((size_t)(M + 2) * 8, 0.f), rfb((size_t)kKp * (M + 2), 0.f);
    std::memcpy(tfb.data(), om->tf.data(), sizeof(float) * (size_t)(M + 1) * 8);
    for (int x = 0; x < kKp; x++) std::memcpy(&rfb[(size_t)x * (M + 2)], &om->rf[(size_t)x * W], sizeof(float) * W);
    BATH_HIP_TRY(ctx, upload(&om->d_tfb, tfb.data(), tfb.size(), ctx->stream));
    BATH_HIP_TRY(ctx, upload(&om->d_rfb, rfb.data(), rfb.size(), ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  {
    // the generic profile's own log scores, for the per-position alignment score (p7_pli_computeAliScores_BATH reads them from
    // gm_fs5, whose amino rows and transitions are these numbers)
    std::vector<float> lm((size_t)kKp * W, -INFINITY);
    for (int x = 0; x < kKp; x++)
      for (int k = 1; k <= M; k++) lm[(size_t)x * W + k] = msc(k, x);
    BATH_HIP_TRY(ctx, upload(&om->d_msc, lm.data(), lm.size(), ctx->stream));
    BATH_HIP_TRY(ctx, upload(&om->d_tsc, gm->tsc, (size_t)M * 8, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  {
    float eo[kKp][2];
    bias_filter_eo(om->compo, eo);
    BATH_HIP_TRY(ctx, upload(&om->d_bias_eo, &eo[0][0], (size_t)kKp * 2, ctx->stream));
  }
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  int st = om->ensure_len_tables(4096);
  if (st != BATH_OK) { bath_hip_oprofile_destroy(om); return st; }
  *ret = om;
  return BATH_OK;
}

extern "C" int bath_hip_oprofile_scalars(const bath_hip_oprofile *om, int L, bath_oprofile_scalars *o) {
  int st = om->ensure_len_tables(L);
  if (st != BATH_OK) return st;
  o->tbm_b = om->tbm_b; o->tec_b = om->tec_b; o->tjb_b = om->lt.h_tjb[L]; o->base_b = om->base_b; o->bias_b = om->bias_b;
  o->scale_b = om->scale_b;
  o->xw[0][0] = om->xw_E[0]; o->xw[0][1] = om->xw_E[1];
  for (int s = 1; s < 4; s++) { o->xw[s][0] = 0; o->xw[s][1] = om->lt.h_xwmove[L]; }
  o->scale_w = om->scale_w; o->base_w = om->base_w; o->ddbound_w = om->ddbound_w;
  o->xf[0][0] = om->xf_E[0]; o->xf[0][1] = om->xf_E[1];
  for (int s = 1; s < 4; s++) { o->xf[s][1] = om->lt.h_pmove[L]; o->xf[s][0] = 1.0f - om->lt.h_pmove[L]; }
  return BATH_OK;
}

extern "C" int bath_hip_oprofile_get_ssv_scores(const bath_hip_oprofile *om, uint8_t *arr) {
  const size_t W = (size_t)om->M + 1;
  std::memset(arr, 0, W * kKp);
  for (int k = 1; k <= om->M; k++)
    for (int x = 0; x < kKp; x++) arr[(size_t)k * kKp + x] = om->rb[x * W + k];
  return BATH_OK;
}
extern "C" int bath_hip_oprofile_get_vit(const bath_hip_oprofile *om, int16_t *rw, int16_t *tw) {
  std::memcpy(rw, om->rw.data(), om->rw.size() * sizeof(int16_t));
  std::memcpy(tw, om->tw.data(), om->tw.size() * sizeof(int16_t));
  return BATH_OK;
}
extern "C" int bath_hip_oprofile_get_fwd(const bath_hip_oprofile *om, float *rf, float *tf) {
  std::memcpy(rf, om->rf.data(), om->rf.size() * sizeof(float));
  std::memcpy(tf, om->tf.data(), om->tf.size() * sizeof(float));
  return BATH_OK;
}

// ------------------------------------------------------------------------------------------ sequence blocks

extern "C" void bath_hip_seqs_destroy(bath_hip_seqs *sq) {
  if (!sq) return;
  for (bath_hip_seqs *part : sq->parts) bath_hip_seqs_destroy(part);
  sq->parts.clear();
  if (sq->is_part) { sq->d_data = nullptr; sq->d_len = nullptr; sq->d_context = nullptr; }      // borrowed from the parent block
  for (void *p : {(void *)sq->d_data, (void *)sq->d_off, (void *)sq->d_len, (void *)sq->d_context, (void *)sq->d_tile_desc, (void *)sq->d_tile_first,
                  (void *)sq->d_packed, (void *)sq->d_poff, sq->d_exc})
    if (p) (void)hipFree(p);
  if (sq->ev_upload) (void)hipEventDestroy(sq->ev_upload);
  delete sq;
}

extern "C" int64_t bath_hip_seqs_count(const bath_hip_seqs *sq) { return sq->n; }

// ESL_SQ.C of the windows of a long target read with esl_sqio_ReadWindow(dbfp, 3 * max_length, block_length, .)
// (bathsearch.c:1099): the pipeline skips ORFs that lie inside the context (p7_pipeline.c:1635-1637) and counts only the
// new residues (pli->nres += dnaSeq->W, bathsearch.c:1258,1268).
extern "C" int bath_hip_seqs_set_context(bath_hip_seqs *sq, const int32_t *context) {
  if (!sq || sq->is_part) return BATH_EINVAL;
  bath_hip_ctx *ctx = sq->ctx;
  for (bath_hip_seqs *part : sq->parts) bath_hip_seqs_destroy(part);    // parts carry a pointer into d_context: rebuild on next use
  sq->parts.clear();
  sq->cache_minlen = -1;
  if (!context) { if (sq->d_context) (void)hipFree(sq->d_context); sq->d_context = nullptr; sq->h_context.clear(); return BATH_OK; }
  for (int64_t i = 0; i < sq->n; i++) if (context[i] < 0 || context[i] > sq->h_len[(size_t)i]) { ctx->set_error("context longer than its window"); return BATH_EINVAL; }
  sq->h_context.assign(context, context + sq->n);
  if (!sq->d_context) BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_context, (size_t)std::max<int64_t>(sq->n, 1) * sizeof(int32_t)));
  if (sq->n > 0) BATH_HIP_TRY(ctx, hipMemcpy(sq->d_context, sq->h_context.data(), (size_t)sq->n * sizeof(int32_t), hipMemcpyHostToDevice));
  return BATH_OK;
}

extern "C" int bath_hip_seqs_create(bath_hip_ctx *ctx, const uint8_t *dsq, const int64_t *offsets, int64_t n, bath_hip_seqs **ret) {
  *ret = nullptr;
  if (n < 0) return BATH_EINVAL;
  bath_hip_seqs *sq = new bath_hip_seqs();
  sq->ctx = ctx; sq->n = n;
  sq->h_off.resize(n); sq->h_len.resize(n);
  // each sequence starts on a 16-byte boundary and is followed by >= 16 readable bytes
  int64_t pos = 0;
  bool contiguous_ok = true;
  for (int64_t i = 0; i < n; i++) {
    int64_t L = offsets[i + 1] - offsets[i];
    if (L < 0 || L > INT32_MAX) { delete sq; ctx->set_error("bad sequence offsets"); return BATH_EINVAL; }
    sq->h_off[i] = pos; sq->h_len[i] = (int32_t)L;
    if (pos != offsets[i] - offsets[0]) contiguous_ok = false;
    sq->maxlen = std::max(sq->maxlen, (int32_t)L);
    sq->total += L;
    pos += (L + 15) / 16 * 16;
  }
  sq->total_aligned = pos;
  size_t bytes = (size_t)pos + 64;
  BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_data, bytes));
  BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_off, (size_t)std::max<int64_t>(n, 1) * sizeof(int64_t)));
  BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_len, (size_t)std::max<int64_t>(n, 1) * sizeof(int32_t)));
  BATH_HIP_TRY(ctx, hipMemsetAsync(sq->d_data, 0x1d, bytes, ctx->stream));
  std::vector<uint8_t> staged;      // must outlive the asynchronous copy: everything here is ordered on ctx->stream
  if (n > 0) {
    if (contiguous_ok) {
      BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_data, dsq + offsets[0], (size_t)(offsets[n] - offsets[0]), hipMemcpyHostToDevice, ctx->stream));
    } else {
      staged.assign((size_t)pos, 0x1d);
      for (int64_t i = 0; i < n; i++) std::memcpy(staged.data() + sq->h_off[i], dsq + offsets[i], (size_t)sq->h_len[i]);
      BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_data, staged.data(), staged.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_off, sq->h_off.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_len, sq->h_len.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *ret = sq;
  return BATH_OK;
}


// =================================================================================================================
// Streamed blocks.  A resident block costs one byte per nucleotide over PCIe (1 GB for the bench block, ~20 ms against a 12 ms
// cascade).  A block can instead arrive in 2 bits per nucleotide (A, C, G, T; the rare other codes as an exception list), 4x
// less to move, on a copy stream of its own, so that the upload of block k+1 overlaps the cascade of block k; a kernel expands
// it to the byte-per-nucleotide layout every other kernel reads (1.25 B/nt of HBM traffic, ~0.3 ms for the bench block).
// =================================================================================================================
namespace {
struct ExcRec { int64_t seq; int32_t pos, code; };

// grid.x = sequence, grid.y = 4096-byte chunk of its packed form; a thread expands one packed byte to four nucleotide bytes
// A wave per sequence; a lane takes four packed bytes (one dword load) and writes their sixteen nucleotides with one 16-byte store
// (sequences start 16-byte aligned in <data>).  The first version ran a 256-thread block per sequence with a byte per thread:
// 10^6 blocks of one iteration each, 1.1-1.9 ms for the bench block.
__global__ __launch_bounds__(256) void unpack2_kernel(const uint8_t *__restrict__ packed, const int64_t *__restrict__ poff, const int64_t *__restrict__ off,
                                                      const int32_t *__restrict__ len, uint8_t *__restrict__ data, int64_t n) {
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n) return;
  const int lane = threadIdx.x & 63;
  const int L = len[s];
  const int nb = (L + 3) >> 2;                                               // packed bytes
  const uint8_t *src = packed + poff[s];
  uint8_t *dst = data + off[s];
  const int j_lo = blockIdx.y * 4096, j_hi = min((nb + 3) >> 2, j_lo + 4096);     // long sequences: 64 k nucleotides per wave
  for (int j = j_lo + lane; j < j_hi; j += 64) {                             // j: dword of packed bytes = 16 nucleotides
    uint32_t w;
    if (4 * j + 4 <= nb) w = *reinterpret_cast<const uint32_t *>(src + 4 * j);        // (unaligned dword loads are fine in global memory)
    else { w = 0; for (int q = 0; 4 * j + q < nb; q++) w |= (uint32_t)src[4 * j + q] << (8 * q); }
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned b = (w >> (8 * q)) & 0xffu;
      v[q] = (b & 3u) | ((b >> 2) & 3u) << 8 | ((b >> 4) & 3u) << 16 | ((b >> 6) & 3u) << 24;
    }
    if (16 * j + 16 <= L) *reinterpret_cast<uint4 *>(dst + 16 * j) = make_uint4(v[0], v[1], v[2], v[3]);
    else for (int q = 0; 16 * j + q < L; q++) dst[16 * j + q] = (uint8_t)((v[q >> 2] >> (8 * (q & 3))) & 0xffu);
  }
}
__global__ void unpack_exceptions_kernel(const ExcRec *__restrict__ exc, int64_t n, const int64_t *__restrict__ off, uint8_t *__restrict__ data) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data[off[exc[i].seq] + exc[i].pos] = (uint8_t)exc[i].code;
}
}  // namespace

extern "C" void *bath_hip_host_alloc(size_t bytes) {
  void *p = nullptr;
  return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void bath_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int bath_hip_seqs_create_packed(bath_hip_ctx *ctx, const int64_t *offsets, int64_t n, bath_hip_seqs **ret) {
  *ret = nullptr;
  if (!ctx || !offsets || n < 0) return BATH_EINVAL;
  // the block's layout from its offsets, no content yet: an all-'A' block of the same shape
  std::vector<uint8_t> zeros((size_t)std::max<int64_t>(offsets[n] - offsets[0], 1), 0);
  std::vector<int64_t> rel((size_t)n + 1);
  for (int64_t i = 0; i <= n; i++) rel[(size_t)i] = offsets[i] - offsets[0];
  bath_hip_seqs *sq = nullptr;
  int st = bath_hip_seqs_create(ctx, zeros.data(), rel.data(), n, &sq);
  if (st != BATH_OK) return st;
  std::vector<int64_t> poff((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; i++) poff[(size_t)i + 1] = poff[(size_t)i] + (sq->h_len[(size_t)i] + 3) / 4;
  sq->packed_bytes = poff[(size_t)n];
  auto packed_parts = [&]() -> int {
    BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_packed, (size_t)sq->packed_bytes + 64));
    BATH_HIP_TRY(ctx, hipMalloc((void **)&sq->d_poff, (size_t)(n + 1) * sizeof(int64_t)));
    BATH_HIP_TRY(ctx, hipMemcpy(sq->d_poff, poff.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&sq->ev_upload, hipEventDisableTiming));
    return BATH_OK;
  };
  if ((st = packed_parts()) != BATH_OK) { bath_hip_seqs_destroy(sq); return st; }      // nothing of a half-built block is left behind
  *ret = sq;
  return BATH_OK;
}

extern "C" int bath_hip_seqs_upload_packed(bath_hip_seqs *sq, const uint8_t *packed, const int64_t *exc_seq, const int32_t *exc_pos,
                                           const uint8_t *exc_code, int64_t n_exc) {
  if (!sq || !sq->d_packed || !packed || n_exc < 0) return BATH_EINVAL;
  bath_hip_ctx *ctx = sq->ctx;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->copy_stream) BATH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (n_exc > sq->exc_cap) {
    if (sq->d_exc) (void)hipFree(sq->d_exc);
    sq->exc_cap = n_exc + n_exc / 2 + 1024;
    BATH_HIP_TRY(ctx, hipMalloc(&sq->d_exc, (size_t)sq->exc_cap * sizeof(ExcRec)));
  }
  // asynchronous when <packed> is page-locked (bath_hip_host_alloc); a pageable buffer makes the call wait for its copy
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_packed, packed, (size_t)sq->packed_bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  if (n_exc > 0) {
    std::vector<ExcRec> h((size_t)n_exc);
    for (int64_t i = 0; i < n_exc; i++) {
      if (exc_seq[i] < 0 || exc_seq[i] >= sq->n || exc_pos[i] < 0 || exc_pos[i] >= sq->h_len[(size_t)exc_seq[i]]) { ctx->set_error("exception outside its sequence"); return BATH_EINVAL; }
      h[(size_t)i] = ExcRec{exc_seq[i], exc_pos[i], (int32_t)exc_code[i]};
    }
    BATH_HIP_TRY(ctx, hipMemcpyAsync(sq->d_exc, h.data(), (size_t)n_exc * sizeof(ExcRec), hipMemcpyHostToDevice, ctx->copy_stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));       // <h> is a local
  }
  sq->n_exc = n_exc;
  BATH_HIP_TRY(ctx, hipEventRecord(sq->ev_upload, ctx->copy_stream));
  sq->upload_pending = true;
  return BATH_OK;
}

extern "C" int bath_hip_seqs_upload_wait(bath_hip_seqs *sq) {
  if (!sq || !sq->d_packed) return BATH_EINVAL;
  if (!sq->upload_pending) return BATH_OK;
  bath_hip_ctx *ctx = sq->ctx;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, sq->ev_upload, 0));
  if (sq->n > 0) {
    const unsigned gy = (unsigned)std::max(1, ((sq->maxlen + 15) / 16 + 4095) / 4096);
    hipLaunchKernelGGL(unpack2_kernel, dim3((unsigned)((sq->n + 3) / 4), gy), dim3(256), 0, ctx->stream, sq->d_packed, sq->d_poff, sq->d_off, sq->d_len, sq->d_data, sq->n);
    if (sq->n_exc > 0)
      hipLaunchKernelGGL(unpack_exceptions_kernel, dim3((unsigned)((sq->n_exc + 255) / 256)), dim3(256), 0, ctx->stream, (const ExcRec *)sq->d_exc, sq->n_exc, sq->d_off, sq->d_data);
    BATH_HIP_TRY(ctx, hipGetLastError());
  }
  // the lanes' streams read d_data too: order them after the expansion
  hipEvent_t ev = sq->ev_upload;
  BATH_HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
  for (bath_hip_ctx *l : ctx->lanes) BATH_HIP_TRY(ctx, hipStreamWaitEvent(l->stream, ev, 0));
  sq->upload_pending = false;
  return BATH_OK;
}
