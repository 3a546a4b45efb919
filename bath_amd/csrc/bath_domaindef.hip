// bath_domaindef.hip -- the frameshift branch after the decision of p7_pli_Frameshift: domain definition on the DNA
// windows that take the branch, and the scores of the resulting hits.
//
// Reference (src/p7_pipeline.c:1464-1476): p7_BackwardParser_Frameshift_3Codons, then
//   p7_domaindef_ByPosteriorHeuristics_Frameshift_BATH  (src/p7_domaindef.c:301-473)
//     p7_DomainDecoding_Frameshift                      (generic form: src/generic_decoding_frameshift.c:204-290)
//     is_multidomain_region_frameshift                  (:684-714)
//     rescore_isolated_domain_frameshift                (:993-1175): Forward/Backward/decoding/OA/null2 of the envelope,
//       p7_OATrace_Frameshift (generic form: src/generic_optacc_frameshift.c:373-588), null2 along the trace
//   p7_pli_postDomainDef_Frameshift_BATH                (src/p7_pipeline.c:1005-1144): the hit's bit score, bias, P-value
//
// Division of labour.  Everything that touches a DP matrix runs on the GPU, batched over all windows / envelopes of the
// block: the 3-codon parsers (fs3_fwd_kernel, fs_bwd_kernel<.,3,.>), the five envelope kernels of bath_frameshift.hip and
// the optimal-accuracy traceback with the null2 score of the aligned residues (fs5_trace_kernel, a lane per envelope), so
// the posterior and OA matrices (>1 MB per envelope) never leave the device; the posterior sums over the parsers'
// special-state rows and the region heuristics run there too (fs_regions_kernel, a lane per window).  What remains for
// the host is bookkeeping and the score arithmetic of the hit.
// Not built: stochastic-trace clustering of multi-domain regions (:396-455; counted in *n_skipped_regions), the
// "aliscore < 0" garbage rule of p7_pli_computeAliScores_BATH (:1070-1080), alignment display.
// The reference carries om_fs5's length configuration from one window to the next; here the domain decoding always uses
// the configuration bathsearch starts with (L = 100 residues, multihit; bathsearch.c:797).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "bath_common.hpp"
#include "bath_launch.hpp"

using namespace bath;

namespace {


const double kLn2 = 0.69314718055994529;

double exp_logsurv(double x, double mu, double lambda) { return x < mu ? 0.0 : -lambda * (x - mu); }

}  // namespace

extern "C" int bath_hip_pipeline_frameshift_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3,
                                                    const bath_hip_fsprofile *om_fs5, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                                                    double E_report, bath_pipeline_stats *stats,
                                                    const bath_fs_window **fs_windows, int64_t *n_fs_windows,
                                                    const bath_fs_domain **domains, int64_t *n_domains, int64_t *n_skipped_regions) {
  if (!ctx || !om || !om_fs3 || !om_fs5 || !dna || !prm || !domains || !n_domains) return BATH_EINVAL;
  if (fsprofile_codon_lengths(om_fs5) != 5) { ctx->set_error("domain definition needs the 5-codon frameshift profile"); return BATH_EINVAL; }
  *domains = nullptr; *n_domains = 0;
  if (n_skipped_regions) *n_skipped_regions = 0;
  ctx->fs_domains.clear();
  bath_pipeline_stats st_local{};
  const bath_fs_window *fw = nullptr;
  int64_t nfw = 0;
  int st = bath_hip_pipeline_frameshift(ctx, om, om_fs3, dna, prm, &st_local, nullptr, nullptr, &fw, &nfw);
  if (st != BATH_OK) return st;
  if (stats) *stats = st_local;
  if (fs_windows) *fs_windows = fw;
  if (n_fs_windows) *n_fs_windows = nfw;
  const FsHostTables h5 = fsprofile_host(om_fs5);
  if (!h5.codons) { ctx->set_error("5-codon profile without its codon table"); return BATH_EINVAL; }

  // ---- the windows that took the frameshift branch: both 3-codon parsers with their special-state rows
  std::vector<int> sel;
  std::vector<FsWinDev> regs;
  for (int64_t i = 0; i < nfw; i++) {
    if (fw[i].branch != 1) continue;
    FsWinDev d{};
    d.src_off = dna->h_off[(size_t)fw[i].window]; d.seq_n = dna->h_len[(size_t)fw[i].window]; d.start = fw[i].n; d.len = fw[i].length; d.strand = fw[i].strand;
    sel.push_back((int)i); regs.push_back(d);
  }
  if (sel.empty()) return BATH_OK;
  OrfTablesDev tt{};
  if ((st = orf_tables_upload(ctx, prm->ncbi_table, &tt)) != BATH_OK) return st;
  const int nsel = (int)sel.size();
  const int RS = 1 + 3 * fs_max_regions();
  std::vector<int32_t> regions((size_t)nsel * RS, 0);
  const float pmove = (2.0f + 1.0f) / (100.0f + 2.0f + 1.0f);                 // p7_fs_ReconfigLength(L = 100, nj = 1), modelconfig.c:767-770
  const float loop = (float)std::log((double)(1.0f - pmove));
  {
    bath_hip_seqs view;
    if ((st = fs_gather_view(ctx, dna, regs, tt.comp, &view, nullptr)) != BATH_OK) return st;
    st = fs3_regions(ctx, om_fs3, &view, loop, regions.data());               // parsers, domain decoding, region heuristics: all on the device
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
    if (st != BATH_OK) return st;
  }
  struct Env { int sel, i, j; };
  std::vector<Env> envs;
  int n_multi = 0;
  for (int q = 0; q < nsel; q++) {
    const int32_t *r = &regions[(size_t)q * RS];
    for (int k = 0; k < r[0]; k++) {                                          // r[0] == -1: Backward underflow, the reference skips the window (:1471)
      const int i = r[1 + 3 * k], j = r[2 + 3 * k];
      if (r[3 + 3 * k]) n_multi++;
      else if (j - i + 1 >= 15) envs.push_back(Env{q, i, j});                 // rescore_isolated_domain: Ld < 15 -> nothing
    }
  }
  if (n_skipped_regions) *n_skipped_regions = n_multi;
  if (envs.empty()) return BATH_OK;

  // ---- envelopes: Forward, Backward, decoding, optimal accuracy, null2 on the GPU (unihit, length Ld/3)
  const int nenv = (int)envs.size();
  std::vector<FsWinDev> eregs((size_t)nenv);
  for (int e = 0; e < nenv; e++) {
    const FsWinDev &wr = regs[(size_t)envs[(size_t)e].sel];
    FsWinDev d = wr;
    d.start = wr.start + envs[(size_t)e].i - 1; d.len = envs[(size_t)e].j - envs[(size_t)e].i + 1;
    eregs[(size_t)e] = d;
  }
  std::vector<bath_fs5_result> res((size_t)nenv);
  std::vector<FsTraceOut> traces((size_t)nenv);
  {
    bath_hip_seqs view;
    if ((st = fs_gather_view(ctx, dna, eregs, tt.comp, &view, nullptr)) != BATH_OK) return st;
    st = fs5_envelopes_ex(ctx, om_fs5, &view, BATH_LOGSUM_TABLE, 0, res.data(), nullptr, nullptr, nullptr, nullptr, traces.data());
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
    if (st != BATH_OK) return st;
  }

  // ---- traceback, null2 along the trace, the hit's scores
  const int ml = h5.max_length;
  const float Zf = (float)st_local.nres / (float)ml;                          // pli->Z, p7_domaindef.c:1033 (here: residues of the block)
  for (int e = 0; e < nenv; e++) {
    const Env &en = envs[(size_t)e];
    const bath_fs_window &win = fw[sel[(size_t)en.sel]];
    const int Ld = eregs[(size_t)e].len;
    const float envsc = res[(size_t)e].fwdsc;
    if (!(envsc > -INFINITY) || !(res[(size_t)e].bcksc > -INFINITY)) continue;
    {
      const float p1 = (float)(Ld / 3) / (float)(Ld / 3 + 1);
      const float per_frame = (float)((float)(Ld / 3) * std::log((double)p1) + std::log(1. - p1));
      const float nullsc = (float)(per_frame + std::log(3.0));
      const float seqscore = (float)((envsc - nullsc) / kLn2);
      if (exp_surv(seqscore, h5.evparam[7], h5.evparam[BATH_FLAMBDA]) * (double)Zf > E_report) continue;    // :1034 (FTAUFS5)
    }
    const FsTraceOut &tq = traces[(size_t)e];
    if (!tq.ok) continue;
    bath_fs_domain dm{};
    dm.window = win.window; dm.strand = win.strand; dm.fs_window = sel[(size_t)en.sel];
    // window coordinates first (:1148-1163), then the sequence's (p7_pipeline.c:1035-1049); envelope i..j in the window
    const int wi = en.i, wj = en.j;
    int iali = wi - 1 + tq.iali, jali = wi - 1 + tq.jali, ienv = wi, jenv = wj;
    const int ali_len = jali - iali + 1, env_len = jenv - ienv + 1;
    dm.ihmm = tq.ihmm; dm.jhmm = tq.jhmm;
    dm.envsc = envsc; dm.oasc = res[(size_t)e].oasc; dm.domcorrection = std::max(0.f, tq.domcorrection);
    dm.n_shifted_codons = tq.nshift;
    if (ali_len >= 12) {
      const int64_t dstart = win.strand ? dna->h_len[(size_t)win.window] : 1;
      auto map = [&](int p) { return (int32_t)(win.strand ? dstart - (win.n + p) + 2 : dstart + win.n + p - 2); };
      dm.ienv = map(ienv); dm.jenv = map(jenv); dm.iali = map(iali); dm.jali = map(jali);
      float bitscore = envsc;                                                  // :1055-1059
      bitscore -= 2 * std::log(2. / ((env_len / 3.) + 2));
      bitscore += 2 * std::log(2. / (ml + 2));
      bitscore -= ((env_len - ali_len) / 3.) * std::log((double)((float)(env_len / 3.) / (float)((env_len / 3.) + 2)));
      bitscore += ((std::max(env_len, ml * 3) - ali_len) / 3.) * std::log((double)((float)ml / (float)(ml + 2)));
      const float dom_bias = flogsum_host(0.0f, (float)(std::log(1. / 256.) + dm.domcorrection));     // bg->omega = 1/256, p7_bg.c:74
      const int nl = std::max(env_len / 3, ml);
      const float p1 = (float)nl / (float)(nl + 1);
      const float per_frame = (float)((float)nl * std::log((double)p1) + std::log(1. - p1));
      const float nullsc = (float)(per_frame + std::log(3.0));
      dm.dombias = dom_bias;
      dm.bitscore = (float)((bitscore - (nullsc + dom_bias)) / kLn2);
      dm.pre_score = (float)(bitscore / kLn2);
      dm.lnP = exp_logsurv(dm.bitscore, h5.evparam[7], h5.evparam[BATH_FLAMBDA]);
      dm.reported = (std::exp(dm.lnP) * (double)Zf <= E_report) ? 1 : 0;
    } else { dm.ienv = ienv; dm.jenv = jenv; dm.iali = iali; dm.jali = jali; dm.reported = 0; }
    ctx->fs_domains.push_back(dm);
  }
  *domains = ctx->fs_domains.data(); *n_domains = (int64_t)ctx->fs_domains.size();
  return BATH_OK;
}
