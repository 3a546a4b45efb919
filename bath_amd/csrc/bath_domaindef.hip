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
// Multi-domain regions (:396-455) are resolved by stochastic-trace clustering (bath_ensemble.hip); *n_clustered_regions counts them.
// Not built: the "aliscore < 0" garbage rule of p7_pli_computeAliScores_BATH (:1070-1080), the printed alignment blocks.
// The reference carries om_fs5's length configuration from one window to the next; here the domain decoding always uses
// the configuration bathsearch starts with (L = 100 residues, multihit; bathsearch.c:797).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sched.h>
#include <thread>
#include <vector>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

namespace {


const double kLn2 = 0.69314718055994529;

double exp_logsurv(double x, double mu, double lambda) { return x < mu ? 0.0 : -lambda * (x - mu); }

// Host threads for the regions of one block (independent of each other).  work(first, step) handles items first, first + step,
// ...; the threads draw single items from a shared counter, heaviest first (<weight>), so that the threads finish together.
// Thread count: the CPUs this process may run on (its affinity mask, not the machine's thread count: a GPU box hands a job a
// slice of its cores), at most 64, divided by the ranks that share the node -- LOCAL_WORLD_SIZE, which torch.distributed.run
// exports: one process per GPU, each with its own ensembles; BATH_HIP_HOST_THREADS overrides.
// The cores a cgroup CPU quota leaves this process (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1), 0: no quota.
// A GPU box hands a job all 256 CPUs in its affinity mask and a quota of 16 cores: 64 threads that poll for their region's matrix
// then use the quota up within a scheduler period and the WHOLE process is throttled for the rest of it -- one --fs pass in four
// or five ran 10-15 ms long until round 4 found that in /sys/fs/cgroup/cpu.stat (nr_throttled).
inline int cgroup_quota_cores() {
  auto read2 = [](const char *path, long long *a, long long *b) -> bool {
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    char x[64] = {0}, y[64] = {0};
    const int n = std::fscanf(f, "%63s %63s", x, y);
    std::fclose(f);
    if (n < 1 || std::strcmp(x, "max") == 0) return false;
    *a = std::atoll(x); *b = n >= 2 ? std::atoll(y) : 0;
    return true;
  };
  long long q = 0, per = 0;
  if (read2("/sys/fs/cgroup/cpu.max", &q, &per) && q > 0 && per > 0) return (int)std::max<long long>(1, (q + per - 1) / per);
  long long q1 = 0, p1 = 0, dummy = 0;
  if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q1, &dummy) && q1 > 0 && read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p1, &dummy) && p1 > 0)
    return (int)std::max<long long>(1, (q1 + p1 - 1) / p1);
  return 0;
}
inline int host_thread_count() {
  const char *e = std::getenv("BATH_HIP_HOST_THREADS");
  if (e && std::atoi(e) > 0) return std::atoi(e);
  static const int cached = [] {
    int usable = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) usable = CPU_COUNT(&set);
    if (usable <= 0) usable = (int)std::max(1u, std::thread::hardware_concurrency());
    const int quota = cgroup_quota_cores();
    // (four times the quota: the threads sleep while they wait for their region's matrix, and 150 core-ms of tracebacks per pass are far
    // from a 16-core quota once nothing spins; 8 / 12 threads leave the ensembles on the pass's critical path: 74 / 69 ms per pass
    // against 64-65 with 16-64, and in the fast mode, where the ensembles ARE the critical path, 32 threads cost 3 ms against 64)
    if (quota > 0) usable = std::min(usable, std::max(2, quota * 4));
    int ranks = 1;
    if (const char *lw = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, std::atoi(lw));
    return std::min(64, std::max(1, usable / ranks));
  }();
  return cached;
}
// Wait for a flag another agent (a kernel writing to page-locked memory) will set: a few yields for the flag that is about to come,
// then sleeps -- a thread that polls through sched_yield for the 5-15 ms a region's matrix takes burns a core of the quota above.
inline void wait_for_flag(const int *flag) {
  for (int spins = 0; !__atomic_load_n(flag, __ATOMIC_ACQUIRE); spins++) {
    if (spins < 32) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spins < 256 ? 20 : 100));
  }
}
template <class F, class W>
void run_striped(int64_t n, F &&work, W &&weight) {
  const int T = (int)std::min<int64_t>(n, host_thread_count());
  if (T <= 1) { work(0, 1); return; }
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; i++) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return weight(a) > weight(b); });
  std::atomic<int64_t> next{0};
  std::vector<std::thread> th;
  for (int k = 0; k < T; k++)
    th.emplace_back([&] { for (int64_t j = next.fetch_add(1); j < n; j = next.fetch_add(1)) work(order[(size_t)j], n); });
  for (std::thread &t : th) t.join();
}

// The --cigar string of an alignment (p7_alidisplay_fs_Create, p7_alidisplay.c:777-815, 841-869; the non-frameshift display
// :1140-1200 is the special case "every codon has 3 nucleotides").  One code per alignment column:
// state (3 = M, 4 = D, 5 = I) | codon length << 4 | indel label << 8 (hmmer.h:259-276).
std::string cigar_from_columns(const uint16_t *S, int n) {
  enum { sM = 3, sD = 4, sI = 5 };
  enum { L___X = 0, L_X__, L_XX_, L_X_X, L__XX, L_XXX, L_XXx, L_XxX, L_xXX, L_xxx, L_XXxX, L_XxXX, L_xXXX, L_XXxxX, L_XxxXX, L_xxXXX };
  std::string out;
  int cnt = 0;
  auto emit = [&](int v, char ch) { out += std::to_string(v); out += ch; };
  for (int z = 0; z < n; z++) {
    const int s = S[z] & 0xf, c = (S[z] >> 4) & 0xf, indel = S[z] >> 8;
    const int next = (z + 1 < n) ? (S[z + 1] & 0xf) : -1;
    if (s == sM) {
      if (next != sM || c != 3) {
        if (c == 3) cnt += 3;
        else if (indel == L_XX_ || indel == L_XXxX || indel == L_XXxxX) cnt += 2;
        else if (indel == L_X_X || indel == L_X__ || indel == L_XxXX || indel == L_XxxXX) cnt += 1;
        emit(cnt, 'M');
        cnt = 0;
        if (c == 1) emit(2, 'B'); else if (c == 2) emit(1, 'B'); else if (c == 4) emit(1, 'F'); else if (c == 5) emit(2, 'F');
        if (indel == L___X || indel == L_X_X || indel == L_XXxX || indel == L_XXxxX) cnt = 1;
        if (indel == L__XX || indel == L_XxXX || indel == L_XxxXX) cnt = 2;
        if (indel == L_xXXX || indel == L_xxXXX) cnt = 3;
        if (next != sM && cnt > 0) { emit(cnt, 'M'); cnt = 0; }
      } else cnt += 3;
    } else if (s == sI || s == sD) {
      cnt += 3;
      if (next != s) { emit(cnt, s == sI ? 'I' : 'D'); cnt = 0; }
    }
  }
  return out;
}

}  // namespace

// pli->nres as the reference's serial loop has it when it searches strand s of window w: every earlier window's W on both strands, this
// window's W once (top strand) or twice (bathsearch.c:1071, :1084; windows under 15 nt are skipped before the count, :1066), on top
// of what the search counted before this block.  pli->Z = (float) nres / (float) max_length (p7_domaindef.c:1033, p7_pipeline.c:1246).
struct RunningZ {
  std::vector<int64_t> before;          // [nwin] residues counted before window w
  const bath_hip_seqs *dna;
  int strands;                          // pli->strands: a window's W is counted once per strand SEARCHED (bathsearch.c:1069-1071, :1082-1084)
  RunningZ(const bath_hip_seqs *d, int64_t nres_before, int strands_ = BATH_STRAND_BOTH) : before((size_t)d->n), dna(d), strands(strands_) {
    int64_t acc = nres_before;
    const int per_window = strands == BATH_STRAND_BOTH ? 2 : 1;
    for (int64_t w = 0; w < d->n; w++) { before[(size_t)w] = acc; acc += per_window * W(w); }
  }
  int64_t W(int64_t w) const {
    const int n = dna->h_len[(size_t)w];
    return n < 15 ? 0 : (int64_t)(n - (dna->h_context.empty() ? 0 : dna->h_context[(size_t)w]));
  }
  float Z(int64_t w, int strand, int max_length) const {
    return (float)(before[(size_t)w] + ((strand && strands == BATH_STRAND_BOTH) ? 2 : 1) * W(w)) / (float)max_length;
  }
};

// What the domain stage reads of the pipeline's option state (bath_pipeline_params + the -E argument)
struct DomOpts {
  int64_t nres_before; double E; int do_null2, inc_by_E, strands; uint32_t seed; double T;
  DomOpts(const bath_pipeline_params &p, double E_report)
      : nres_before(p.nres_before), E(E_report), do_null2(p.do_null2), inc_by_E(p.inc_by_E), strands(p.strands), seed((uint32_t)p.seed), T(p.T) {}
  // p7_pipeline.c:1080, :1247: the early reporting test goes by inc_by_E (sic), against E with the running Z or against T
  bool reportable(double lnP, float Zf, float score) const { return inc_by_E ? (std::exp(lnP) * (double)Zf <= E) : ((double)score >= T); }
};

static int std_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const std::vector<PipelineSurvivor> &surv,
                       const uint8_t *d_pool, const DomOpts &opt, int64_t *n_clustered_regions);


static int fs_branch_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3,
                             const bath_hip_fsprofile *om_fs5, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                             double E_report, bath_pipeline_stats *stats,
                             const bath_fs_window **fs_windows, int64_t *n_fs_windows,
                             const bath_fs_domain **domains, int64_t *n_domains, int64_t *n_clustered_regions, int64_t *std_clustered) {
  if (!ctx || !om || !om_fs3 || !om_fs5 || !dna || !prm || !domains || !n_domains) return BATH_EINVAL;
  if (fsprofile_codon_lengths(om_fs5) != 5) { ctx->set_error("domain definition needs the 5-codon frameshift profile"); return BATH_EINVAL; }
  *domains = nullptr; *n_domains = 0;
  if (n_clustered_regions) *n_clustered_regions = 0;
  ctx->fs_domains.clear();
  ctx->cigars.clear();
  ctx->traces_clear();
  ctx->spans_reset();
  bath_pipeline_stats st_local{};
  const bath_fs_window *fw = nullptr;
  int64_t nfw = 0;
  StageClock clk;
  ctx->fs_want_regions = true;
  int st = bath_hip_pipeline_frameshift(ctx, om, om_fs3, dna, prm, &st_local, nullptr, nullptr, &fw, &nfw);
  ctx->fs_want_regions = false;
  if (st != BATH_OK) return st;
  clk.lap("fs: cascade + windows + decision");
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
  // ---- the standard branch of the other windows (p7_pipeline.c:1479-1510) needs nothing from the frameshift branch.  The
  // stages below are chains of latency-bound kernels, a PCIe-bound one and host work that leave most of the GPU idle, so the
  // standard-branch domains run beside them: own host thread, own context (stream, scratch, page-locked buffers), reading the
  // cascade's residue pool.  Their domains are put in front of the frameshift branch's, as when they ran first.
  std::thread std_thread;
  int std_rc = BATH_OK;
  int64_t std_nclust = 0;
  const char *ser = std::getenv("BATH_HIP_FS_STD_SERIAL");
  if (std_clustered && !ctx->fs_std_orfs.empty() && !(ser && ser[0] == '1')) {
    if (!ctx->aux && (st = bath_hip_init(ctx->device, &ctx->aux)) != BATH_OK) { ctx->set_error("cannot create the context of the standard branch"); return st; }
    mark_internal(ctx->aux);
    if ((st = om->ensure_len_tables(dna->maxlen / 3 + 1)) != BATH_OK) return st;      // the profile's mutable state: fill it before the thread starts
    bath_hip_ctx *aux = ctx->aux;
    aux->fs_domains.clear(); aux->cigars.clear(); aux->traces_clear();
    const DomOpts std_opt(*prm, E_report);
    std_thread = std::thread([&, aux, std_opt] {
      if (hipSetDevice(ctx->device) != hipSuccess) { std_rc = BATH_EFAIL; return; }
      std_rc = std_domains(aux, om, dna, ctx->fs_std_orfs, ctx->fs_std_pool, std_opt, &std_nclust);
    });
  }
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } };
  Joiner std_joiner{std_thread};                                              // also on error returns
  auto finish_std = [&]() -> int {
    if (!std_thread.joinable()) return BATH_OK;
    std_thread.join();
    if (std_rc != BATH_OK) { ctx->set_error(ctx->aux->err); return std_rc; }
    const int64_t shift = (int64_t)ctx->cigars.size();
    for (bath_fs_domain dm : ctx->aux->fs_domains) { dm.cigar_off += shift; ctx->fs_domains.push_back(dm); }
    ctx->cigars += ctx->aux->cigars;
    for (const bath_hip_ctx::TraceRec &t : ctx->aux->tr_recs)
      ctx->trace_push(ctx->aux->tr_codes.data() + t.col_off, ctx->aux->tr_pp.data() + t.col_off, t.ncol, t.k1, t.i_first, t.win_start, t.orf_start, t.frameshift, t.d_i);
    *std_clustered = std_nclust;
    return BATH_OK;
  };
  OrfTablesDev tt{};
  if ((st = orf_tables_upload(ctx, prm->ncbi_table, &tt, prm->initiator)) != BATH_OK) return st;
  const int nsel = (int)sel.size();
  const int RS = 1 + 3 * fs_max_regions();
  std::vector<int32_t> regions((size_t)nsel * RS, 0);
  const float pmove = (2.0f + 1.0f) / (100.0f + 2.0f + 1.0f);                 // p7_fs_ReconfigLength(L = 100, nj = 1), modelconfig.c:767-770
  const float loop = (float)std::log((double)(1.0f - pmove));
  if (ctx->fs_regions_all.size() == (size_t)nfw * (size_t)RS) {                // the decision stage already ran them for every window
    for (int q = 0; q < nsel; q++) std::memcpy(&regions[(size_t)q * RS], &ctx->fs_regions_all[(size_t)sel[(size_t)q] * RS], sizeof(int32_t) * (size_t)RS);
  } else {
    bath_hip_seqs view;
    if ((st = fs_gather_view(ctx, dna, regs, tt.comp, &view, nullptr)) != BATH_OK) return st;
    st = fs3_regions(ctx, om_fs3, &view, loop, regions.data(), nullptr, sel.data());   // Backward parser (Forward's rows are the decision stage's), domain decoding, region heuristics: all on the device
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
    if (st != BATH_OK) return st;
    clk.lap("fs: parsers + regions");
  }
  struct Env { int sel, i, j; };
  std::vector<Env> envs, mregs;
  for (int q = 0; q < nsel; q++) {
    const int32_t *r = &regions[(size_t)q * RS];
    if (r[0] > fs_max_regions()) { ctx->set_error("a DNA window has more regions than the region buffer holds"); return BATH_ERANGE; }
    for (int k = 0; k < r[0]; k++) {                                          // r[0] == -1: Backward underflow, the reference skips the window (:1471)
      const int i = r[1 + 3 * k], j = r[2 + 3 * k];
      if (r[3 + 3 * k]) mregs.push_back(Env{q, i, j});
      else if (j - i + 1 >= 15) envs.push_back(Env{q, i, j});                 // rescore_isolated_domain: Ld < 15 -> nothing
    }
  }
  if (n_clustered_regions) *n_clustered_regions = (int64_t)mregs.size();          // regions resolved by clustering (ddef->nclustered)

  // ---- envelopes: Forward, Backward, decoding, optimal accuracy, null2 on the GPU (unihit, length Ld/3)
  std::vector<FsWinDev> eregs;
  std::vector<bath_fs5_result> res;
  std::vector<FsTraceOut> traces;
  std::vector<uint16_t> steps;
  std::vector<float> pps;                                                    // tr->pp per column, parallel to steps
  std::vector<int64_t> step_off;
  // The envelope kernels keep three matrices per envelope in HBM (Forward 32 B, Backward 12 B, optimal accuracy 12 B per
  // cell): the envelopes go through in batches of at most 24 GB of matrices (BATH_HIP_ENV_MB overrides, for tests), so a block
  // of any size fits next to the DNA and the amino-acid streams.
  size_t budget = (size_t)24 << 30;
  {                                                                          // ... and at most half of what is free right now
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 2 < budget) budget = std::max<size_t>(free_b / 2, (size_t)64 << 20);
  }
  if (const char *e = std::getenv("BATH_HIP_ENV_MB")) budget = (size_t)std::max(1, std::atoi(e)) << 20;
  // One batch of envelopes through the kernels on context <c>, results into <out> (indexed from 0 within the batch)
  struct EnvBatch {
    std::vector<FsWinDev> eregs;
    std::vector<bath_fs5_result> res;
    std::vector<FsTraceOut> traces;
    std::vector<uint16_t> steps;
    std::vector<float> pps;
    std::vector<int64_t> step_off;
  };
  auto run_env_batch = [&](bath_hip_ctx *c, const Env *list, int n, EnvBatch &out) -> int {
    out.eregs.resize((size_t)n); out.res.resize((size_t)n); out.traces.resize((size_t)n); out.step_off.assign((size_t)n + 1, 0);
    out.steps.clear(); out.pps.clear();
    for (int e = 0; e < n; e++) {
      const FsWinDev &wr = regs[(size_t)list[e].sel];
      FsWinDev d = wr;
      d.start = wr.start + list[e].i - 1; d.len = list[e].j - list[e].i + 1;
      out.eregs[(size_t)e] = d;
    }
    for (int e0 = 0; e0 < n;) {
      int e1 = e0;
      size_t bytes = 0;
      while (e1 < n) {
        const size_t b = ((size_t)out.eregs[(size_t)e1].len + 1) * (size_t)(h5.M + 1) * 56;
        if (e1 > e0 && bytes + b > budget) break;
        bytes += b; e1++;
      }
      std::vector<FsWinDev> chunk(out.eregs.begin() + e0, out.eregs.begin() + e1);
      std::vector<uint16_t> csteps;
      std::vector<float> cpps;
      std::vector<int64_t> coff;
      bath_hip_seqs view;
      // (a batch on another context runs on another host thread: its error text stays in that context until the caller has joined)
      int st2 = fs_gather_view(c, dna, chunk, tt.comp, &view, nullptr);
      if (st2 != BATH_OK) return st2;
      st2 = fs5_envelopes_ex(c, om_fs5, &view, BATH_LOGSUM_CONTEXT, 0, out.res.data() + e0, nullptr, nullptr, nullptr, nullptr, out.traces.data() + e0, om->d_cons, &csteps, &coff, &cpps);
      view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
      if (st2 != BATH_OK) return st2;
      const int64_t base = (int64_t)out.steps.size();
      for (int k = 0; k < e1 - e0; k++) out.step_off[(size_t)(e0 + k)] = base + coff[(size_t)k];
      out.steps.insert(out.steps.end(), csteps.begin(), csteps.end());
      out.pps.insert(out.pps.end(), cpps.begin(), cpps.end());
      e0 = e1;
    }
    out.step_off[(size_t)n] = (int64_t)out.steps.size();
    return BATH_OK;
  };
  auto append_batch = [&](const EnvBatch &b) {                                  // a finished batch joins the arrays the hit stage reads (envs order)
    const int64_t base = (int64_t)steps.size();
    if (step_off.empty()) step_off.push_back(0);
    step_off.pop_back();
    eregs.insert(eregs.end(), b.eregs.begin(), b.eregs.end());
    res.insert(res.end(), b.res.begin(), b.res.end());
    traces.insert(traces.end(), b.traces.begin(), b.traces.end());
    for (size_t k = 0; k + 1 < b.step_off.size(); k++) step_off.push_back(base + b.step_off[k]);
    steps.insert(steps.end(), b.steps.begin(), b.steps.end());
    pps.insert(pps.end(), b.pps.begin(), b.pps.end());
    step_off.push_back((int64_t)steps.size());
  };
  auto run_envelopes = [&](int e_begin, int e_end) -> int {                    // envs[e_begin, e_end) on this context, results appended (e_begin = what is done so far)
    EnvBatch b;
    const int st2 = run_env_batch(ctx, envs.data() + e_begin, e_end - e_begin, b);
    if (st2 != BATH_OK) return st2;
    append_batch(b);
    return BATH_OK;
  };
  const int n_single = (int)envs.size();
  const int n_single_early = n_single;
  int done = 0;

  // ---- multi-domain regions (p7_domaindef.c:396-455): Forward of the region in the multihit configuration (GPU), ensemble of
  // stochastic tracebacks and clustering (host, bath_ensemble.hip); every cluster is an envelope
  std::vector<std::vector<Env>> found;                                       // written by the ensemble threads: declared BEFORE the joiner, so it outlives the join on every return
  // Strict mode, when this is the only context the host holds (host_contexts() == 1; BATH_HIP_FS_CLUSTERS_BESIDE=1|0 forces): the
  // clusters' envelopes go through the kernels on a context of their own as soon as the last ensemble is done, from the ensembles'
  // own thread -- beside the tail of the single-domain batch (its decoding and tracebacks) instead of after it: 62.2-62.5 ms per
  // pass against 64.3-65.0.  With other contexts at work the extra context's streams cost more than that (two workers 52-59 ms per
  // block against 48.5), and an idle one costs the fast mode 15 ms per pass (bath_hip_trim releases it; bath_hip_set_fs_strict(ctx, 0)
  // does so itself).  Written by that thread, read after the join.
  EnvBatch cl_batch;
  std::vector<Env> cl_envs;
  int cl_rc = BATH_OK;
  std::string cl_err;                                                        // that thread's error text: copied into ctx->err after the join (ctx->err is the caller thread's)
  bool cl_ran = false;
  bath_hip_ctx *rctx = ctx;                                                  // where the regions' Forward runs
  std::thread ensembles;
  Joiner joiner{ensembles};                                                  // also on error returns
  if (!mregs.empty()) {
    std::vector<FsWinDev> rregs(mregs.size());
    for (size_t e = 0; e < mregs.size(); e++) {
      FsWinDev d = regs[(size_t)mregs[e].sel];
      d.start = regs[(size_t)mregs[e].sel].start + mregs[e].i - 1; d.len = mregs[e].j - mregs[e].i + 1;
      rregs[e] = d;
    }
    const float *h_f = nullptr, *h_x = nullptr;                              // pinned buffers of the context
    const int *h_done = nullptr;
    const float *h_sc_live = nullptr;
    std::vector<float> h_sc;
    std::vector<int64_t> foff, xoff;
    {
      // Strict mode: the regions' Forward is a chain of L x 2M dependent log-sums per region on a few CUs (bath_fs_chain.hip) and
      // the envelope kernels are bound by throughput (bath_fs_wavefront.hip), so the envelopes of the single-domain regions go
      // through the chip WHILE the regions' Forward runs: it gets a context of its own (stream, gather pool, offsets, job list).
      if (ctx->fs_strict && !ctx->aux2) {
        if ((st = bath_hip_init(ctx->device, &ctx->aux2)) != BATH_OK) { ctx->set_error("cannot create the context of the regions' Forward"); return st; }
        mark_internal(ctx->aux2);
        // HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams that land on one
        // queue run their kernels one after the other: with a dozen streams alive (lanes, side, copy, the standard branch) the
        // regions' Forward ended up behind the envelope kernels it is meant to run beside (+13 ms per pass in bench.py's process).
        // Streams of another priority get queues of their own; the regions' Forward is the critical path of this stage anyway.
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
          hipStream_t ps = nullptr;
          if (hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, hi) == hipSuccess) { (void)hipStreamDestroy(ctx->aux2->stream); ctx->aux2->stream = ps; }
          else (void)hipGetLastError();
        }
      }
      rctx = ctx->fs_strict ? ctx->aux2 : ctx;
      if (rctx != ctx) { rctx->fs_strict = ctx->fs_strict; rctx->spans_reset(); }
      bath_hip_seqs view;
      if ((st = fs_gather_view(rctx, dna, rregs, tt.comp, &view, nullptr)) != BATH_OK) { if (rctx != ctx) ctx->set_error(rctx->err); return st; }
      // returns after the launch: a region's ensemble starts as soon as ITS matrix has landed in host memory (h_done[e]), so the
      // tracebacks run while the kernel is still streaming the other regions over PCIe (longest regions first, on both sides)
      st = fs5_region_forward(rctx, om_fs5, &view, 100, &h_f, &foff, &h_x, &xoff, &h_sc, &h_done, &h_sc_live);      // saveL: the configuration bathsearch starts with
      view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
      if (st != BATH_OK) { if (rctx != ctx) ctx->set_error(rctx->err); return st; }
    }
    const float pm = (2.0f + 1.0f) / (100.0f + 2.0f + 1.0f);                 // p7_fs_ReconfigLength(L = 100), multihit (nj = 1)
    const float xNL = (float)std::log((double)(1.0f - pm)), xNM = (float)std::log((double)pm), xE = (float)-kLn2;
    found.assign(mregs.size(), {});
    // The ensembles are host work (200 dependent tracebacks per region from one random-number stream); the envelopes of the
    // single-domain regions do not depend on them, so their kernels run on the GPU meanwhile.
    { const char *lv = std::getenv("BATH_HIP_FS_LIVE"); if (lv && lv[0] == '0') (void)hipStreamSynchronize(rctx->stream); }
    ensembles = std::thread([&, h_f, h_x, xNL, xNM, xE, foff, xoff, h_done, h_sc_live, rregs] {
      auto work = [&](int64_t first, int64_t step) {
        std::vector<std::pair<int, int>> cl;
        for (size_t e = (size_t)first; e < mregs.size(); e += (size_t)step) {
          wait_for_flag(h_done + e);                                           // this region's matrix is still on its way
          if (!(h_sc_live[e] > -INFINITY)) continue;                          // Forward underflow: no valid traces for this region (:413)
          const int Lr = rregs[e].len;
          if (fs_region_trace_ensemble(h5.M, h5.tsc, xNL, xNM, xE, mregs[e].i, Lr, h_f + foff[e], h_x + xoff[e], &cl, (uint32_t)prm->seed) != BATH_OK) continue;
          for (const auto &c : cl) {
            const int i2 = std::max(1, c.first), j2 = c.second;               // :449
            if (j2 - i2 + 1 >= 15) found[e].push_back(Env{mregs[e].sel, i2, j2});
          }
        }
      };
      StageClock eclk;
      // shortest region first: that is the order in which their matrices finish arriving (all regions advance together, a wave
      // each, sharing the PCIe link), so a thread rarely waits for the region it drew
      run_striped((int64_t)mregs.size(), work, [&](int64_t e) { return -rregs[(size_t)e].len; });
      eclk.lap("fs:   (ensemble threads, start to end)");
      static const int beside_env = [] { const char *e = std::getenv("BATH_HIP_FS_CLUSTERS_BESIDE"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
      const bool clusters_beside = beside_env >= 0 ? beside_env == 1 : host_contexts() == 1;      // default: when this is the host's only context
      if (rctx != ctx && clusters_beside) {
        cl_ran = true;
        static const bool own_ctx = [] { const char *e = std::getenv("BATH_HIP_FS_CLUSTERS_CTX"); return !(e && e[0] == '2'); }();   // 2: on the regions' context
        if (hipSetDevice(ctx->device) != hipSuccess) { cl_rc = BATH_EFAIL; cl_err = "hipSetDevice failed on the clusters' thread"; return; }
        bath_hip_ctx *cctx = rctx;
        if (own_ctx) {
          if (!ctx->aux3 && bath_hip_init(ctx->device, &ctx->aux3) != BATH_OK) { cl_rc = BATH_EFAIL; cl_err = "cannot create the context of the clusters' envelopes"; return; }
          mark_internal(ctx->aux3);
          cctx = ctx->aux3; cctx->fs_strict = ctx->fs_strict; cctx->spans_reset();
        } else if (hipStreamSynchronize(rctx->stream) != hipSuccess) { cl_rc = BATH_EFAIL; cl_err = "the regions' stream failed"; return; }
        for (size_t e = 0; e < mregs.size(); e++) cl_envs.insert(cl_envs.end(), found[e].begin(), found[e].end());
        if (!cl_envs.empty() && (cl_rc = run_env_batch(cctx, cl_envs.data(), (int)cl_envs.size(), cl_batch)) != BATH_OK) cl_err = cctx->err;
        eclk.lap("fs:   (clusters' envelope kernels + traces, same thread)");
      }
    });
    // strict mode: the first batch of envelopes goes through now, beside the regions' Forward (which runs on its own context)
    if (rctx != ctx && n_single_early > 0) {
      if ((st = run_envelopes(0, n_single_early)) != BATH_OK) {
        (void)hipStreamSynchronize(rctx->stream);
        for (size_t e = 0; e < mregs.size(); e++) { const_cast<float *>(h_sc_live)[e] = -INFINITY; __atomic_store_n(const_cast<int *>(h_done) + e, 1, __ATOMIC_RELEASE); }
        return st;
      }
      done = n_single_early;
      clk.lap("fs: envelope kernels + traces (single-domain regions, beside the regions' Forward)");
    }
    if (hipStreamSynchronize(rctx->stream) != hipSuccess) {                   // the region Forward itself (the ensembles are already at work)
      for (size_t e = 0; e < mregs.size(); e++) {                             // release the threads waiting for matrices that will not come
        const_cast<float *>(h_sc_live)[e] = -INFINITY;
        __atomic_store_n(const_cast<int *>(h_done) + e, 1, __ATOMIC_RELEASE);
      }
      ctx->set_error("region Forward failed"); return BATH_EFAIL;
    }
    clk.lap("fs: region Forward -> host memory");
  }

  // Order of the work.  The ensembles are host threads; the standard branch of the other windows (p7_pipeline.c:1479-1510)
  // needs nothing from them, so its kernels and host work run meanwhile.  Then ALL envelopes -- of the single-domain regions
  // and of the clusters -- go through the envelope kernels as one batch: those kernels last as long as their longest
  // envelope whatever the number of envelopes (one wave each), so two batches cost two such chains (14.3 + 9.4 ms on the
  // bench block) and one batch costs one.  BATH_HIP_FS_TWO_BATCHES=1: the single-domain regions first, during the ensembles.
  const char *tb = std::getenv("BATH_HIP_FS_TWO_BATCHES");
  const bool two_batches = tb && tb[0] == '1';
  if (ensembles.joinable() && two_batches && n_single > 0 && done == 0) {
    if ((st = run_envelopes(0, n_single)) != BATH_OK) return st;
    done = n_single;
    clk.lap("fs: envelope kernels + traces (single-domain regions)");
  }
  if (ensembles.joinable()) {
    ensembles.join();
    for (size_t e = 0; e < mregs.size(); e++) envs.insert(envs.end(), found[e].begin(), found[e].end());
    clk.lap("fs: ensembles (host threads)");
  }
  if ((st = finish_std()) != BATH_OK) return st;
  clk.lap("fs: standard-branch domains (own thread and stream), remainder");
  const int nenv = (int)envs.size();
  if (nenv == 0) return BATH_OK;
  if (cl_ran && done == n_single) {                                          // the clusters' batch ran beside the first one (same order as envs)
    if (cl_rc != BATH_OK) { ctx->set_error(cl_err.empty() ? "the clusters' envelope batch failed" : cl_err.c_str()); return cl_rc; }
    if (!cl_envs.empty()) append_batch(cl_batch);
    done = nenv;
  }
  if (nenv > done && (st = run_envelopes(done, nenv)) != BATH_OK) return st;
  clk.lap("fs: envelope kernels + traces");

  // ---- traceback, null2 along the trace, the hit's scores
  const int ml = h5.max_length;
  const DomOpts opt(*prm, E_report);
  const RunningZ runZ(dna, prm->nres_before, prm->strands);                   // pli->Z, p7_domaindef.c:1033: the count at the hit's window and strand
  for (int e = 0; e < nenv; e++) {
    const Env &en = envs[(size_t)e];
    const bath_fs_window &win = fw[sel[(size_t)en.sel]];
    const int Ld = eregs[(size_t)e].len;
    const float envsc = res[(size_t)e].fwdsc;
    if (!(envsc > -INFINITY) || !(res[(size_t)e].bcksc > -INFINITY)) continue;
    const float Zf = runZ.Z(win.window, win.strand, ml);
    {
      const float p1 = (float)(Ld / 3) / (float)(Ld / 3 + 1);
      const float per_frame = (float)((float)(Ld / 3) * std::log((double)p1) + std::log(1. - p1));
      const float nullsc = (float)(per_frame + std::log(3.0));
      const float seqscore = (float)((envsc - nullsc) / kLn2);
      if (opt.inc_by_E && exp_surv(seqscore, h5.evparam[7], h5.evparam[BATH_FLAMBDA]) * (double)Zf > E_report) continue;    // :1034 (FTAUFS5)
    }
    const FsTraceOut &tq = traces[(size_t)e];
    if (!tq.ok) continue;
    if (tq.aliscore < 0.0f) continue;                                          // p7_domaindef.c:1072: "repetitive garbage", no domain
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
      const float dom_bias = opt.do_null2 ? flogsum_host(0.0f, (float)(std::log(1. / 256.) + dm.domcorrection)) : 0.0f;     // :1063-1066; bg->omega = 1/256, p7_bg.c:74
      const int nl = std::max(env_len / 3, ml);
      const float p1 = (float)nl / (float)(nl + 1);
      const float per_frame = (float)((float)nl * std::log((double)p1) + std::log(1. - p1));
      const float nullsc = (float)(per_frame + std::log(3.0));
      dm.dombias = dom_bias;
      dm.bitscore = (float)((bitscore - (nullsc + dom_bias)) / kLn2);
      dm.pre_score = (float)(bitscore / kLn2);
      dm.lnP = exp_logsurv(dm.bitscore, h5.evparam[7], h5.evparam[BATH_FLAMBDA]);
      dm.reported = opt.reportable(dm.lnP, Zf, dm.bitscore) ? 1 : 0;           // :1080
    } else { dm.ienv = ienv; dm.jenv = jenv; dm.iali = iali; dm.jali = jali; dm.reported = 0; }
    dm.n_stops = tq.nstops; dm.ali_columns = tq.ncol;
    dm.pid = tq.ncol > 0 ? ((float)tq.exact / tq.ncol) * 100 : 0.f;
    dm.cigar_off = (int64_t)ctx->cigars.size();
    ctx->cigars += cigar_from_columns(steps.data() + step_off[(size_t)e], tq.ncol);
    ctx->cigars.push_back('\0');
    ctx->fs_domains.push_back(dm);
    // dom->tr: states z1..z2 in window coordinates (the i - 1 shift of p7_domaindef.c:1053-1054 applied to every i >= 0: a D state's 0 becomes wi - 1)
    ctx->trace_push(steps.data() + step_off[(size_t)e], pps.size() == steps.size() ? pps.data() + step_off[(size_t)e] : nullptr, tq.ncol, tq.ihmm,
                    wi - 1 + tq.iali, win.n, 0, 1, wi - 1);
  }
  *domains = ctx->fs_domains.data(); *n_domains = (int64_t)ctx->fs_domains.size();
  return BATH_OK;
}

// p7_pli_Frameshift's two branches (p7_pipeline.c:1464-1510): windows that take the frameshift branch get the codon-model
// domain definition above; in the others every ORF with P <= F3 goes through the standard domain definition below, its
// alignment placed on the DNA window's coordinates.
extern "C" int bath_hip_pipeline_frameshift_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3,
                                                    const bath_hip_fsprofile *om_fs5, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                                                    double E_report, bath_pipeline_stats *stats,
                                                    const bath_fs_window **fs_windows, int64_t *n_fs_windows,
                                                    const bath_fs_domain **domains, int64_t *n_domains, int64_t *n_clustered_regions) {
  bath_pipeline_stats st_local{};
  int64_t std_clustered = -1;                                                   // >= 0: the standard branch already ran, overlapped with the ensembles
  int64_t fs_clustered = 0;
  int st = fs_branch_domains(ctx, om, om_fs3, om_fs5, dna, prm, E_report, &st_local, fs_windows, n_fs_windows, domains, n_domains, &fs_clustered, &std_clustered);
  if (st != BATH_OK) return st;
  if (stats) *stats = st_local;
  int64_t nclust = fs_clustered;
  StageClock clk;
  if (std_clustered >= 0) nclust += std_clustered;
  else {
    if ((st = std_domains(ctx, om, dna, ctx->fs_std_orfs, ctx->fs_std_pool, DomOpts(*prm, E_report), &nclust)) != BATH_OK) return st;
    clk.lap("fs: standard-branch domains");
  }
  if (n_clustered_regions) *n_clustered_regions = nclust;
  *domains = ctx->fs_domains.data(); *n_domains = (int64_t)ctx->fs_domains.size();
  return BATH_OK;
}

// =================================================================================================================
// Standard (non-frameshift) branch: what p7_Pipeline_BATH does with an ORF that passed the Forward filter
// (src/p7_pipeline.c:1741-1771): p7_BackwardParser, p7_domaindef_ByPosteriorHeuristics_BATH (src/p7_domaindef.c:491-614)
// with rescore_isolated_domain_bath (:1194-1325) for single-domain regions, p7_pli_postDomainDef_BATH (:1172-1300).
//
// The parsers and the envelopes' full Forward/Backward are the wave-per-target kernels of bath_filters.hip (the latter
// with their matrix output and the unihit length model).  The rest of an envelope -- posterior decoding, the
// optimal-accuracy fill and traceback, null2 (impl_sse/decoding.c:61, optacc.c:58,225, null2.c:50) -- is one kernel
// with a LANE per envelope working serially on the matrices in HBM: the envelopes that reach this point are ~10^-4 of
// the ORFs and a few 10^4 cells each, so round 1 keeps this stage simple and exact in operation order; it is not on the
// headline path (bench.py times the filter cascade).
// =================================================================================================================
namespace {

constexpr int kStdMaxRegions = 48;       // regions kept per ORF; an ORF with more reports the true count and the call fails loudly (the reference has no cap)

// the residues of each target of a view, to out + xoff[t] / 6 (a block per target)
// rows of target q: (len[q] + 1) x 6 floats from src + src_off[q] to dst + dst_off[q]
__global__ void copy_rows_kernel(int64_t n, const int32_t *__restrict__ len, const float *__restrict__ src, const int64_t *__restrict__ src_off,
                                 float *__restrict__ dst, const int64_t *__restrict__ dst_off) {
  for (int64_t q = blockIdx.x; q < n; q += gridDim.x) {
    const float *a = src + src_off[q];
    float *b = dst + dst_off[q];
    const int m = (len[q] + 1) * 6;
    for (int i = threadIdx.x; i < m; i += blockDim.x) b[i] = a[i];
  }
}
__global__ void gather_residues_kernel(SeqView v, const int64_t *__restrict__ xoff, uint8_t *__restrict__ out) {
  for (int64_t t = blockIdx.x; t < v.n; t += gridDim.x) {
    const uint8_t *s = v.data + v.off[t];
    uint8_t *d = out + xoff[t] / 6;
    for (int i = threadIdx.x; i < v.len[t]; i += blockDim.x) d[i] = s[i];
  }
}

// p7_DomainDecoding (impl_sse/decoding.c:155-196) + region heuristics (p7_domaindef.c:520-533, 642-654), lane per ORF
__global__ void std_regions_kernel(int64_t n, const int32_t *__restrict__ len, const float *__restrict__ fx, const float *__restrict__ bx,
                                   const int64_t *__restrict__ x_off, const float *__restrict__ pmove_tab, float *__restrict__ work /* 3 floats per row */,
                                   int32_t *__restrict__ regions /* [n][1 + 3*kStdMaxRegions] */) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  enum { XE = 0, XN, XJ, XB, XC, XS };
  const int L = len[t];
  const float *F = fx + x_off[t], *B = bx + x_off[t];
  float *btot = work + (x_off[t] / 6) * 3, *etot = btot + (L + 1), *mocc = etot + (L + 1);
  int32_t *out = regions + t * (1 + 3 * kStdMaxRegions);
  out[0] = 0;
  const float ploop = 1.0f - pmove_tab[L];                                   // xf[N|J|C][LOOP] of the multihit model at this length
  float scaleproduct = (float)(1.0 / (double)B[XN]);
  btot[0] = etot[0] = mocc[0] = 0.f;
  for (int i = 1; i <= L; i++) {
    btot[i] = btot[i - 1] + (F[(size_t)(i - 1) * 6 + XB] * B[(size_t)(i - 1) * 6 + XB] * F[(size_t)(i - 1) * 6 + XS] * scaleproduct);
    scaleproduct *= F[(size_t)(i - 1) * 6 + XS] / B[(size_t)(i - 1) * 6 + XS];              // 1 unless Backward had to use its own scale factors
    etot[i] = etot[i - 1] + (F[(size_t)i * 6 + XE] * B[(size_t)i * 6 + XE] * F[(size_t)i * 6 + XS] * scaleproduct);
    float njcp = F[(size_t)(i - 1) * 6 + XN] * B[(size_t)i * 6 + XN] * ploop * scaleproduct;
    njcp += F[(size_t)(i - 1) * 6 + XJ] * B[(size_t)i * 6 + XJ] * ploop * scaleproduct;
    njcp += F[(size_t)(i - 1) * 6 + XC] * B[(size_t)i * 6 + XC] * ploop * scaleproduct;
    mocc[i] = (float)(1. - njcp);
  }
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
  int i = -1, nreg = 0;
  bool triggered = false;
  for (int j = 1; j <= L; j++) {
    if (!triggered) {
      if (mocc[j] - (btot[j] - btot[j - 1]) < rt2) i = j;
      else if (i == -1) i = j;
      if (mocc[j] >= rt1) triggered = true;
    } else if (mocc[j] - (etot[j] - etot[j - 1]) < rt2) {
      float best = -1.0f;
      for (int z = i; z <= j; z++) best = fmaxf(best, fminf(etot[z] - etot[i - 1], btot[j] - btot[z - 1]));
      if (nreg < kStdMaxRegions) { out[1 + 3 * nreg] = i; out[2 + 3 * nreg] = j; out[3 + 3 * nreg] = best >= rt3 ? 1 : 0; }
      nreg++;
      i = -1; triggered = false;
    }
  }
  out[0] = nreg;                           // may exceed kStdMaxRegions: the host checks
}

// The same with a WAVE per ORF and the rows in LDS (ORFs up to Lcap residues; the lane kernel above takes longer ones).  The per-row
// terms are elementwise (64 lanes); what the reference accumulates in sequence -- the running scale product, btot, etot -- one lane
// accumulates in that order from LDS; the region scan runs uniformly on all lanes with the inner "best split" maximum spread over
// them (a maximum does not care about order).  Same operations in the same order as the lane kernel: identical regions.  The lane
// kernel walked its ORF alone, every row a trip to L2: 1.2 ms for the bench block's 7.6 k ORFs against ~0.1 ms.
__global__ __launch_bounds__(64) void std_regions_wave_kernel(int64_t n, const int32_t *__restrict__ len, const float *__restrict__ fx, const float *__restrict__ bx,
                                                              const int64_t *__restrict__ x_off, const float *__restrict__ pmove_tab, int Lcap,
                                                              int32_t *__restrict__ regions /* [n][1 + 3*kStdMaxRegions] */) {
  extern __shared__ float sm[];                                              // sp, btot, etot, mocc: (Lcap + 1) floats each
  enum { XE = 0, XN, XJ, XB, XC, XS };
  const int lane = threadIdx.x;
  float *ssp = sm, *sb = sm + (Lcap + 1), *se = sb + (Lcap + 1), *smo = se + (Lcap + 1);
  for (int64_t t = blockIdx.x; t < n; t += gridDim.x) {
    const int L = len[t];
    if (L > Lcap) continue;
    const float *F = fx + x_off[t], *B = bx + x_off[t];
    int32_t *out = regions + t * (1 + 3 * kStdMaxRegions);
    const float ploop = 1.0f - pmove_tab[L];
    __syncthreads();                                                         // the previous ORF's readers are done
    for (int i = 1 + lane; i <= L; i += 64) smo[i] = F[(size_t)(i - 1) * 6 + XS] / B[(size_t)(i - 1) * 6 + XS];
    __syncthreads();
    if (lane == 0) {                                                         // scaleproduct after row i (before row i+1)
      float sp = (float)(1.0 / (double)B[XN]);
      ssp[0] = sp;
      for (int i = 1; i <= L; i++) { sp *= smo[i]; ssp[i] = sp; }
    }
    __syncthreads();
    for (int i = 1 + lane; i <= L; i += 64) {
      const float spb = ssp[i - 1], spa = ssp[i];
      sb[i] = F[(size_t)(i - 1) * 6 + XB] * B[(size_t)(i - 1) * 6 + XB] * F[(size_t)(i - 1) * 6 + XS] * spb;
      se[i] = F[(size_t)i * 6 + XE] * B[(size_t)i * 6 + XE] * F[(size_t)i * 6 + XS] * spa;
      float njcp = F[(size_t)(i - 1) * 6 + XN] * B[(size_t)i * 6 + XN] * ploop * spa;
      njcp += F[(size_t)(i - 1) * 6 + XJ] * B[(size_t)i * 6 + XJ] * ploop * spa;
      njcp += F[(size_t)(i - 1) * 6 + XC] * B[(size_t)i * 6 + XC] * ploop * spa;
      smo[i] = (float)(1. - njcp);
    }
    __syncthreads();
    if (lane == 0) {                                                         // btot, etot: sums in the reference's order
      sb[0] = se[0] = smo[0] = 0.f;
      float b = 0.f, e = 0.f;
      for (int i = 1; i <= L; i++) { b = b + sb[i]; sb[i] = b; e = e + se[i]; se[i] = e; }
    }
    __syncthreads();
    const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;
    int i = -1, nreg = 0;
    bool triggered = false;
    for (int j = 1; j <= L; j++) {                                           // uniform over the wave: every lane reads the same LDS words
      if (!triggered) {
        if (smo[j] - (sb[j] - sb[j - 1]) < rt2) i = j;
        else if (i == -1) i = j;
        if (smo[j] >= rt1) triggered = true;
      } else if (smo[j] - (se[j] - se[j - 1]) < rt2) {
        float best = -1.0f;
        for (int z = i + lane; z <= j; z += 64) best = fmaxf(best, fminf(se[z] - se[i - 1], sb[j] - sb[z - 1]));
        best = wave_max_f32(best);
        if (lane == 0 && nreg < kStdMaxRegions) { out[1 + 3 * nreg] = i; out[2 + 3 * nreg] = j; out[3 + 3 * nreg] = best >= rt3 ? 1 : 0; }
        nreg++;
        i = -1; triggered = false;
      }
    }
    if (lane == 0) out[0] = nreg;                                            // may exceed kStdMaxRegions: the host checks
  }
}

struct StdEnvOut { int32_t i1, k1, i2, k2, ok; float oasc, domcorrection; int32_t ncol, exact; float aliscore; int32_t pp_off; };

// p7_Decoding + p7_OptimalAccuracy + p7_OATrace + p7_Null2_ByExpectation on one envelope per lane (unihit model).
// fwd / bck: (L+1) x (M+1) x {M, D, I}; on return bck holds the posteriors and fwd the OA matrix, as in the reference.
constexpr int kStdTraceSpread = 64;
constexpr size_t kStdFillSlack = 64 * 16 * 3 * sizeof(float);    // std_envelope_fill_kernel reads the cells of all 64 x C node slots of a row, also past node M
__global__ void std_envelope_kernel(SeqView sq, int M, const float *__restrict__ tf, const float *__restrict__ rf, float *__restrict__ fwd, float *__restrict__ bck,
                                    const int64_t *__restrict__ dp_off, const float *__restrict__ fx, const float *__restrict__ bx, const int64_t *__restrict__ x_off,
                                    float *__restrict__ ppx_all, float *__restrict__ oax_all, float *__restrict__ em_all /* [n][2*(M+1)] */, StdEnvOut *__restrict__ out,
                                    const uint8_t *__restrict__ cons /* [M+1] or null */, uint8_t *__restrict__ tbuf, const int64_t *__restrict__ t_off,
                                    int filled /* std_envelope_fill_kernel already did decoding, OA fill and null2: traceback only */,
                                    const float *__restrict__ msc /* [Kp][M+1] log-odds */, const float *__restrict__ tsc /* [M][8] log */,
                                    const uint8_t *__restrict__ nt /* the DNA block */, const int64_t *__restrict__ nt_base, const int64_t *__restrict__ nt_dir,
                                    float *__restrict__ null2_out = nullptr /* optional [n][Kp]: the null2 vector itself (filled == 0 only) */,
                                    float *__restrict__ col_pp = nullptr /* tr->pp of the alignment columns, first to last, all envelopes densely */,
                                    int *__restrict__ col_cursor = nullptr) {
  // one envelope per wave: the traceback is a state machine and lanes in different states run one after the other (see
  // fs5_trace_kernel); the chip has room for a wave per envelope
  if (threadIdx.x % kStdTraceSpread) return;
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kStdTraceSpread;
  if (t >= sq.n) return;
  enum { XE = 0, XN, XJ, XB, XC, XS };
  enum { cM = 0, cD = 1, cI = 2 };
  enum { MM = 0, IM, DM, BM, MD, DD, MI, II };
  enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC };
  const int L = sq.len[t];
  const uint8_t *dsq = sq.data + sq.off[t] - 1;                              // dsq[1..L]
  const size_t W = (size_t)(M + 1) * 3;
  float *F = fwd + dp_off[t], *Bk = bck + dp_off[t];
  const float *FX = fx + x_off[t], *BX = bx + x_off[t];
  float *PX = ppx_all + (x_off[t] / 6) * 5, *OX = oax_all + (x_off[t] / 6) * 5;
  StdEnvOut r{-1, -1, -1, -1, 0, 0.f, 0.f, 0, 0, 0.f, 0};
  const float ploop = 1.0f - 2.0f / ((float)L + 2.0f);                       // unihit: xf[N|J|C][LOOP]
  const float *P = Bk;
  float *O = F;
  if (filled) {
    const StdEnvOut pre = out[t];
    if (!pre.ok) { out[t] = r; return; }
    r.oasc = pre.oasc; r.domcorrection = pre.domcorrection;
  }
  if (!filled) {
  // ---- p7_Decoding (decoding.c:61-118): posteriors overwrite Backward
  float scaleproduct = (float)(1.0 / (double)BX[XN]);
  for (int k = 0; k <= M; k++) Bk[(size_t)k * 3] = Bk[(size_t)k * 3 + 1] = Bk[(size_t)k * 3 + 2] = 0.f;
  for (int s = 0; s < 5; s++) PX[s] = 0.f;
  for (int i = 1; i <= L; i++) {
    const float totr = scaleproduct * FX[(size_t)i * 6 + XS];
    const float *f = F + (size_t)i * W;
    float *b = Bk + (size_t)i * W;
    for (int k = 1; k <= M; k++) {
      b[(size_t)k * 3 + cM] = f[(size_t)k * 3 + cM] * (b[(size_t)k * 3 + cM] * totr);
      b[(size_t)k * 3 + cD] = 0.0f;
      b[(size_t)k * 3 + cI] = f[(size_t)k * 3 + cI] * (b[(size_t)k * 3 + cI] * totr);
    }
    float *px = PX + (size_t)i * 5;
    px[XE] = 0.f; px[XB] = 0.f;
    px[XN] = FX[(size_t)(i - 1) * 6 + XN] * BX[(size_t)i * 6 + XN] * ploop * scaleproduct;
    px[XJ] = FX[(size_t)(i - 1) * 6 + XJ] * BX[(size_t)i * 6 + XJ] * ploop * scaleproduct;
    px[XC] = FX[(size_t)(i - 1) * 6 + XC] * BX[(size_t)i * 6 + XC] * ploop * scaleproduct;
    scaleproduct *= FX[(size_t)i * 6 + XS] / BX[(size_t)i * 6 + XS];
  }
  if (isinf(scaleproduct)) { out[t] = r; return; }                           // eslERANGE: the domain is dropped (p7_domaindef.c:1214)
  // ---- p7_OptimalAccuracy (optacc.c:58-173): the OA matrix overwrites Forward
  auto allow = [](float tr, float v) { return tr > 0.0f ? v : 0.0f; };
  for (int k = 0; k <= M; k++) O[(size_t)k * 3] = O[(size_t)k * 3 + 1] = O[(size_t)k * 3 + 2] = -INFINITY;
  OX[XE] = -INFINITY; OX[XN] = 0.f; OX[XJ] = -INFINITY; OX[XB] = 0.f; OX[XC] = -INFINITY;
  for (int i = 1; i <= L; i++) {
    const float *p = P + (size_t)i * W, *pr = O + (size_t)(i - 1) * W;
    float *c = O + (size_t)i * W;
    const float xB = OX[(size_t)(i - 1) * 5 + XB];
    c[0] = c[1] = c[2] = -INFINITY;
    float xE = -INFINITY, dcv = -INFINITY;
    for (int k = 1; k <= M; k++) {
      const float *tr = tf + (size_t)k * 8;
      float sv = allow(tr[BM], xB);
      sv = fmaxf(sv, allow(tr[MM], pr[(size_t)(k - 1) * 3 + cM]));
      sv = fmaxf(sv, allow(tr[IM], pr[(size_t)(k - 1) * 3 + cI]));
      sv = fmaxf(sv, allow(tr[DM], pr[(size_t)(k - 1) * 3 + cD]));
      sv = sv + p[(size_t)k * 3 + cM];
      xE = fmaxf(xE, sv);
      c[(size_t)k * 3 + cM] = sv;
      c[(size_t)k * 3 + cD] = dcv;
      dcv = fmaxf(allow(tr[MD], sv), allow(tr[DD], dcv));
      c[(size_t)k * 3 + cI] = fmaxf(allow(tr[MI], pr[(size_t)k * 3 + cM]), allow(tr[II], pr[(size_t)k * 3 + cI])) + p[(size_t)k * 3 + cI];
    }
    for (int k = 1; k <= M; k++) xE = fmaxf(xE, c[(size_t)k * 3 + cD]);
    float *ox = OX + (size_t)i * 5;
    const float *opx = OX + (size_t)(i - 1) * 5, *px = PX + (size_t)i * 5;
    ox[XE] = xE;
    ox[XJ] = fmaxf(opx[XJ] + px[XJ], 0.0f);                                  // unihit: xf[E][LOOP] == 0 -> the E->J term is 0.0 (optacc.c:156)
    ox[XC] = fmaxf(opx[XC] + px[XC], xE);
    ox[XN] = opx[XN] + px[XN];
    ox[XB] = fmaxf(ox[XN], ox[XJ]);
  }
  r.oasc = OX[(size_t)L * 5 + XC];
  }
  // ---- p7_OATrace (optacc.c:225-430); select_e walks the cells in the reference's striped order
  {
    const int Q = max(2, (M - 1) / 4 + 1);
    auto path = [](float tr, float v) { return tr == 0.0f ? -INFINITY : v; };
    int i = L, k = 0, s0 = sC, steps = 0;
    bool bad = false;
    // the alignment columns from the last match state back to the first, for the display (p7_alidisplay_nonfs_Create)
    uint8_t *T = tbuf + t_off[t];
    const int cap = (int)(t_off[t + 1] - t_off[t]);
    int ncol = 0, exact = 0;
    while (s0 != sS && !bad) {
      int s1 = -1;
      switch (s0) {
      case sM: {
        const float *tr = tf + (size_t)k * 8, *pr = O + (size_t)(i - 1) * W;
        const float pm = path(tr[MM], pr[(size_t)(k - 1) * 3 + cM]), pi = path(tr[IM], pr[(size_t)(k - 1) * 3 + cI]);
        const float pd = path(tr[DM], pr[(size_t)(k - 1) * 3 + cD]), pb = path(tr[BM], OX[(size_t)(i - 1) * 5 + XB]);
        s1 = sM; float b = pm;
        if (pi > b) { b = pi; s1 = sI; }
        if (pd > b) { b = pd; s1 = sD; }
        if (pb > b) { b = pb; s1 = sB; }
        k--; i--; break; }
      case sD: {
        const float *tr = tf + (size_t)(k - 1) * 8, *c = O + (size_t)i * W;
        const float pm = (k - 1 >= 1) ? path(tr[MD], c[(size_t)(k - 1) * 3 + cM]) : -INFINITY;
        const float pd = (k - 1 >= 1) ? path(tr[DD], c[(size_t)(k - 1) * 3 + cD]) : -INFINITY;
        s1 = pm >= pd ? sM : sD; k--; break; }
      case sI: {
        const float *tr = tf + (size_t)k * 8, *pr = O + (size_t)(i - 1) * W;
        s1 = path(tr[MI], pr[(size_t)k * 3 + cM]) >= path(tr[II], pr[(size_t)k * 3 + cI]) ? sM : sI; i--; break; }
      case sN: s1 = (i == 0) ? sS : sN; break;
      case sC: s1 = (OX[(size_t)(i - 1) * 5 + XC] + PX[(size_t)i * 5 + XC] > OX[(size_t)i * 5 + XE]) ? sC : sE; break;
      case sJ: s1 = sJ; break;                                                // unihit: E->J impossible, path[1] = -inf (optacc.c:384)
      case sE: {
        const float *c = O + (size_t)i * W;
        float mx = -INFINITY; int smax = -1, kmax = -1;
        for (int q = 0; q < Q; q++) {
          for (int rr = 0; rr < 4; rr++) { const int kk = rr * Q + q + 1; if (kk <= M && c[(size_t)kk * 3 + cM] >= mx) { mx = c[(size_t)kk * 3 + cM]; smax = sM; kmax = kk; } }
          for (int rr = 0; rr < 4; rr++) { const int kk = rr * Q + q + 1; if (kk <= M && c[(size_t)kk * 3 + cD] > mx) { mx = c[(size_t)kk * 3 + cD]; smax = sD; kmax = kk; } }
        }
        k = kmax; s1 = smax; break; }
      case sB: s1 = (OX[(size_t)i * 5 + XN] > OX[(size_t)i * 5 + XJ]) ? sN : sJ; break;
      default: bad = true; break;
      }
      if (bad || s1 < 0 || i < 0 || k < 0) { bad = true; break; }
      if (s1 == sM) { if (r.i2 < 0) { r.i2 = i; r.k2 = k; } r.i1 = i; r.k1 = k; }
      if ((s1 == sM || s1 == sD || s1 == sI) && r.i2 >= 0) {
        if (ncol < cap) T[ncol] = (uint8_t)s1;
        ncol++;
        if (s1 == sM) { r.ncol = ncol; if (cons && min((int)dsq[i], kKp - 1) == cons[k]) exact++; r.exact = exact; }
      }
      if ((s1 == sN || s1 == sJ || s1 == sC) && s1 == s0) i--;
      s0 = s1;
      if (++steps > 4 * (L + M) + 64) bad = true;
    }
    if (bad || r.i1 <= 0) { out[t] = r; return; }
    // ---- p7_pli_computeAliScores_BATH (p7_pipeline.c:781-979) on the columns, first to last match state: the emission score of
    // each codon's amino acid (X for a codon with a degenerate nucleotide, p7P_DEGEN5_C) plus the transition that entered the
    // state; the last match state gets no MM transition (the reference's inner loops stop at z1 < z2).  Negative total: the
    // caller drops the domain (p7_domaindef.c:1286).
    if (msc && r.ncol <= cap) {
      const size_t W1 = (size_t)M + 1;
      const int64_t base = nt_base[t], dir = nt_dir[t];
      float total = 0.f;
      int prev = sB, kk = r.k1 - 1, ii = r.i1 - 1;
      for (int idx = r.ncol - 1; idx >= 0; idx--) {
        const int s = T[idx];
        float sc;
        if (s == sM) {
          kk++; ii++;
          int amino = min((int)dsq[ii], kKp - 1);
          const int64_t c0 = base + dir * (int64_t)(3 * (ii - 1));
          if (nt[c0] >= 4 || nt[c0 + dir] >= 4 || nt[c0 + 2 * dir] >= 4) amino = 26;
          sc = msc[(size_t)amino * W1 + kk];
          if (prev == sI) sc += tsc[(size_t)(kk - 1) * 8 + IM];
          else if (prev == sD) sc += tsc[(size_t)(kk - 1) * 8 + DM];
          else if (prev == sM && idx > 0) sc += tsc[(size_t)(kk - 1) * 8 + MM];
        } else if (s == sI) { ii++; sc = tsc[(size_t)kk * 8 + (prev == sI ? II : MI)]; }
        else { kk++; sc = tsc[(size_t)(kk - 1) * 8 + (prev == sD ? DD : MD)]; }
        total += sc;
        prev = s;
      }
      r.aliscore = total;
    }
    // ---- the posterior of every column's state, first to last (p7_OATrace's get_postprob, optacc.c: M and I cells of the
    // posterior matrix; a delete state has none), for bath_hip_domain_traces
    if (col_pp && r.ncol <= cap && r.ncol > 0) {
      r.pp_off = atomicAdd(col_cursor, r.ncol);
      float *out_pp = col_pp + r.pp_off;
      int kk = r.k1 - 1, ii = r.i1 - 1;
      for (int idx = r.ncol - 1; idx >= 0; idx--) {
        const int s = T[idx];
        float v = 0.0f;
        if (s == sM) { kk++; ii++; v = P[(size_t)ii * W + (size_t)kk * 3 + cM]; }
        else if (s == sI) { ii++; v = P[(size_t)ii * W + (size_t)kk * 3 + cI]; }
        else kk++;
        out_pp[r.ncol - 1 - idx] = v;
      }
    }
  }
  // ---- p7_Null2_ByExpectation (null2.c:50-124) and the correction over the envelope (p7_domaindef.c:1264-1272)
  if (!filled) {
    float *em = em_all + (size_t)t * 2 * (M + 1);
    float xN = PX[5 + XN], xC = PX[5 + XC], xJ = PX[5 + XJ];
    for (int k = 1; k <= M; k++) { em[2 * k] = P[W + (size_t)k * 3 + cM]; em[2 * k + 1] = P[W + (size_t)k * 3 + cI]; }
    for (int i = 2; i <= L; i++) {
      const float *p = P + (size_t)i * W;
      for (int k = 1; k <= M; k++) { em[2 * k] = p[(size_t)k * 3 + cM] + em[2 * k]; em[2 * k + 1] = p[(size_t)k * 3 + cI] + em[2 * k + 1]; }
      xN += PX[(size_t)i * 5 + XN]; xC += PX[(size_t)i * 5 + XC]; xJ += PX[(size_t)i * 5 + XJ];
    }
    const float norm = (float)(1.0 / (double)(float)L);
    for (int k = 1; k <= M; k++) { em[2 * k] *= norm; em[2 * k + 1] *= norm; }
    xN *= norm; xC *= norm; xJ *= norm;
    const float xfactor = xN + xC + xJ;
    float null2[kKp];
    for (int x = 0; x < 20; x++) {
      const float *e = rf + (size_t)x * (M + 1);
      float sv = 0.f;
      for (int k = 1; k <= M; k++) { sv += em[2 * k] * e[k]; sv += em[2 * k + 1]; }
      null2[x] = sv + xfactor;
    }
    const int mem[6][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}, {-1, -1}};      // B=DN J=IL Z=EQ O=K U=C X=any
    for (int dx = 0; dx < 6; dx++) {
      float sum = 0.f; int cnt = 0;
      if (dx == 5) { for (int y = 0; y < 20; y++) { sum += null2[y]; cnt++; } }
      else { const int a = min(mem[dx][0], mem[dx][1]), b = max(mem[dx][0], mem[dx][1]); sum += null2[a]; cnt++; if (b != a) { sum += null2[b]; cnt++; } }
      null2[21 + dx] = sum / (float)cnt;
    }
    null2[20] = 1.0f; null2[27] = 1.0f; null2[28] = 1.0f;
    if (null2_out) for (int x = 0; x < kKp; x++) null2_out[(size_t)t * kKp + x] = null2[x];
    float corr = 0.f;
    for (int pos = 1; pos <= L; pos++) corr += logf(null2[min((int)dsq[pos], kKp - 1)]);
    r.domcorrection = corr;
  }
  r.ok = 1;
  out[t] = r;
}

// p7_OATrace and what follows it in std_envelope_kernel (alignment score, the columns' posteriors) for an envelope whose matrices
// std_envelope_fill_kernel left behind, with the WAVE walking instead of one lane.  The walk is a chain of decisions, each on cells
// whose address the previous decision gives: a lane on its own pays a trip to memory per step (0.8 us; 900 steps for a 459-node
// model).  Here every lane follows the same state machine on the same (uniform) state and the cells come from look-ahead buffers
// filled by all 64 lanes at once:
//   * a match run walks a diagonal: on entering (i, k) lane d fetches what the step from (i - d, k - d) will need -- the three cells
//     of (i - 1 - d, k - 1 - d), B of that row, the four transitions into node k - d, the residue and the consensus residue of the
//     cell it leads to -- and the steps take them by v_readlane until the path leaves the diagonal or the 64 are used;
//   * the C flank is decided 64 rows at a time (a ballot of "stays in C"); the N flank needs no reads at all;
//   * the E state's choice among the 2 M cells of a row (the reference scans them in its striped order: ties go to the LAST match
//     cell of the scan, else to the FIRST delete cell) is a lane-parallel scan and two wave maxima over (value, rank);
//   * the columns' scores and posteriors are gathered a lane per column (prefix counts of the column kinds by ballot), and the
//     score is summed in the reference's order, first to last column.
// Same decisions on the same values: the trace is the serial kernel's, state for state (tests/test_hits_gpu.py runs both).
__global__ __launch_bounds__(64) void std_trace_wave_kernel(SeqView sq, int M, const float *__restrict__ tf, const float *__restrict__ fwd, const float *__restrict__ bck,
                                                            const int64_t *__restrict__ dp_off, const int64_t *__restrict__ x_off,
                                                            const float *__restrict__ ppx_all, const float *__restrict__ oax_all, StdEnvOut *__restrict__ out,
                                                            const uint8_t *__restrict__ cons /* [M+1] or null */, uint8_t *__restrict__ tbuf, const int64_t *__restrict__ t_off,
                                                            const float *__restrict__ msc /* [Kp][M+1] log-odds */, const float *__restrict__ tsc /* [M][8] log */,
                                                            const uint8_t *__restrict__ nt /* the DNA block */, const int64_t *__restrict__ nt_base, const int64_t *__restrict__ nt_dir,
                                                            float *__restrict__ col_pp, int *__restrict__ col_cursor) {
  enum { XE = 0, XN, XJ, XB, XC, XS };
  enum { cM = 0, cD = 1, cI = 2 };
  enum { MM = 0, IM, DM, BM, MD, DD, MI, II };
  enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC };
  const int lane = threadIdx.x;
  const int64_t t = blockIdx.x;
  const int L = sq.len[t];
  const uint8_t *dsq = sq.data + sq.off[t] - 1;                              // dsq[1..L]
  const size_t W = (size_t)(M + 1) * 3;
  const float *O = fwd + dp_off[t], *P = bck + dp_off[t];
  const float *PX = ppx_all + (x_off[t] / 6) * 5, *OX = oax_all + (x_off[t] / 6) * 5;
  StdEnvOut r{-1, -1, -1, -1, 0, 0.f, 0.f, 0, 0, 0.f, 0};
  {
    const StdEnvOut pre = out[t];
    if (!pre.ok) { if (lane == 0) out[t] = r; return; }
    r.oasc = pre.oasc; r.domcorrection = pre.domcorrection;
  }
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto rlf = [](float v, int d) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), d)); };
  auto path = [](float tr, float v) { return tr == 0.0f ? -INFINITY : v; };
  const int Q = max(2, (M - 1) / 4 + 1);
  uint8_t *T = tbuf + t_off[t];
  const int cap = (int)(t_off[t + 1] - t_off[t]);
  int i = L, k = 0, s0 = sC, steps = 0, ncol = 0, exact = 0;
  bool bad = false;
  // the diagonal look-ahead: lane d holds what the match step from (di0 - d, dk0 - d) reads
  float dM = 0.f, dI = 0.f, dD = 0.f, dXB = 0.f;
  float4 dT = make_float4(0.f, 0.f, 0.f, 0.f);
  int dRes = 0, dCons = -1, di0 = -(1 << 20), dk0 = -(1 << 20);
  while (s0 != sS && !bad) {
    i = uni(i); k = uni(k); s0 = uni(s0);
    int s1 = -1;
    bool from_diag = false; int dd = 0;
    if (s0 == sC) {
      const int rr = i - lane;
      bool stay = false;
      if (rr >= 1) stay = OX[(size_t)(rr - 1) * 5 + XC] + PX[(size_t)rr * 5 + XC] > OX[(size_t)rr * 5 + XE];
      const unsigned long long m = __ballot(stay);
      const int n = (~m == 0ull) ? 64 : (__ffsll((long long)~m) - 1);         // rows i, i-1, ... that stay in C
      if (n > 0) { i -= n; steps += n; if (steps > 4 * (L + M) + 64) bad = true; continue; }
      s1 = sE;
    } else if (s0 == sN) {                                                    // N(i) <- N(i-1) ... <- N(0) <- S: nothing to read
      steps += i + 1; i = 0; s0 = sS; continue;
    } else if (s0 == sM) {
      dd = di0 - i;
      if (dd < 0 || dd >= 64 || dk0 - k != dd) {
        const int ii = max(i - 1 - lane, 0), kk = max(k - 1 - lane, 0);
        const float *pr = O + (size_t)ii * W + (size_t)kk * 3;
        dM = pr[cM]; dI = pr[cI]; dD = pr[cD]; dXB = OX[(size_t)ii * 5 + XB];
        dT = *reinterpret_cast<const float4 *>(tf + (size_t)max(k - lane, 0) * 8);
        dRes = min((int)dsq[max(ii, 1)], kKp - 1); dCons = cons ? (int)cons[kk] : -1;
        di0 = i; dk0 = k; dd = 0;
      }
      dd = uni(dd);
      const float pm = path(rlf(dT.x, dd), rlf(dM, dd)), pi = path(rlf(dT.y, dd), rlf(dI, dd));
      const float pd = path(rlf(dT.z, dd), rlf(dD, dd)), pb = path(rlf(dT.w, dd), rlf(dXB, dd));
      s1 = sM; float b = pm;
      if (pi > b) { b = pi; s1 = sI; }
      if (pd > b) { b = pd; s1 = sD; }
      if (pb > b) { b = pb; s1 = sB; }
      k--; i--; from_diag = true;
    } else if (s0 == sD) {
      const float *tr = tf + (size_t)max(k - 1, 0) * 8, *c = O + (size_t)i * W;
      const float pm = (k - 1 >= 1) ? path(tr[MD], c[(size_t)(k - 1) * 3 + cM]) : -INFINITY;
      const float pd = (k - 1 >= 1) ? path(tr[DD], c[(size_t)(k - 1) * 3 + cD]) : -INFINITY;
      s1 = pm >= pd ? sM : sD; k--;
    } else if (s0 == sI) {
      const float *tr = tf + (size_t)k * 8, *pr = O + (size_t)(i - 1) * W;
      s1 = path(tr[MI], pr[(size_t)k * 3 + cM]) >= path(tr[II], pr[(size_t)k * 3 + cI]) ? sM : sI; i--;
    } else if (s0 == sJ) {
      s1 = sJ;                                                                // unihit: E->J impossible, path[1] = -inf (optacc.c:384)
    } else if (s0 == sE) {
      // select_e over the reference's striped order q = 0..Q-1: the four match cells rr*Q + q + 1 (>=: a later one takes a tie),
      // then the four delete cells (>: an earlier one keeps it).  rank: scan position, match cells above all delete cells.
      const float *c = O + (size_t)i * W;
      float mx = -INFINITY; int rank = -1;
      for (int q = lane; q < Q; q += 64) {
        for (int rr = 0; rr < 4; rr++) { const int kk = rr * Q + q + 1; if (kk <= M) { const float v = c[(size_t)kk * 3 + cM]; const int rk = (1 << 24) + q * 8 + rr; if (v > mx || (v == mx && rk > rank)) { mx = v; rank = rk; } } }
        for (int rr = 0; rr < 4; rr++) { const int kk = rr * Q + q + 1; if (kk <= M) { const float v = c[(size_t)kk * 3 + cD]; const int rk = (1 << 23) - (q * 8 + 4 + rr); if (v > mx || (v == mx && rk > rank)) { mx = v; rank = rk; } } }
      }
      const float best = wave_max_f32(mx);
      const int brank = wave_max_i32((mx == best) ? rank : -1);
      if (brank < 0) bad = true;
      else if (brank >= (1 << 24)) { const int pq = brank - (1 << 24); k = (pq & 7) * Q + (pq >> 3) + 1; s1 = sM; }
      else { const int pq = (1 << 23) - brank; k = ((pq & 7) - 4) * Q + (pq >> 3) + 1; s1 = sD; }
    } else if (s0 == sB) {
      s1 = (OX[(size_t)i * 5 + XN] > OX[(size_t)i * 5 + XJ]) ? sN : sJ;
    } else bad = true;
    if (bad || s1 < 0 || i < 0 || k < 0) { bad = true; break; }
    if (s1 == sM) { if (r.i2 < 0) { r.i2 = i; r.k2 = k; } r.i1 = i; r.k1 = k; }
    if ((s1 == sM || s1 == sD || s1 == sI) && r.i2 >= 0) {
      if (ncol < cap && lane == 0) T[ncol] = (uint8_t)s1;
      ncol++;
      if (s1 == sM) {
        r.ncol = ncol;
        if (cons) {
          int res, cn;
          if (from_diag) { res = __builtin_amdgcn_readlane(dRes, dd); cn = __builtin_amdgcn_readlane(dCons, dd); }
          else { res = min((int)dsq[i], kKp - 1); cn = (int)cons[k]; }
          if (res == cn) exact++;
        }
        r.exact = exact;
      }
    }
    if ((s1 == sN || s1 == sJ || s1 == sC) && s1 == s0) i--;
    s0 = s1;
    if (++steps > 4 * (L + M) + 64) bad = true;
  }
  if (bad || r.i1 <= 0) { if (lane == 0) out[t] = r; return; }
  __threadfence();                                                            // lane 0's column bytes, read below by every lane
  // ---- the columns, first to last match state (column j of the alignment = T[ncol' - 1 - j], ncol' = r.ncol), a lane per
  // column: its node and residue from prefix counts of the column kinds; p7_pli_computeAliScores_BATH (p7_pipeline.c:781-979)
  // summed in the reference's order, and the posterior of every column's state (optacc.c, get_postprob)
  const int nc = r.ncol;
  if (nc <= cap && nc > 0) {
    const size_t W1 = (size_t)M + 1;
    const int64_t base = nt_base[t], dir = nt_dir[t];
    int pp_off = 0;
    if (col_pp) { if (lane == 0) pp_off = atomicAdd(col_cursor, nc); pp_off = uni(pp_off); r.pp_off = pp_off; }
    float total = 0.f;
    int kbase = r.k1 - 1, ibase = r.i1 - 1, prev_last = sB;
    for (int jb = 0; jb < nc; jb += 64) {
      const int j = jb + lane;
      const bool in = j < nc;
      const int s = in ? (int)T[nc - 1 - j] : -1;
      const unsigned long long bMm = __ballot(s == sM), bIm = __ballot(s == sI), bDm = __ballot(s == sD);
      const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
      const int kk = kbase + __popcll((bMm | bDm) & upto), ii = ibase + __popcll((bMm | bIm) & upto);
      int prev = __shfl_up(s, 1, 64);
      if (lane == 0) prev = prev_last;
      float sc = 0.f, ppv = 0.f;
      if (in) {
        if (s == sM) {
          int amino = min((int)dsq[ii], kKp - 1);
          const int64_t c0 = base + dir * (int64_t)(3 * (ii - 1));
          if (nt[c0] >= 4 || nt[c0 + dir] >= 4 || nt[c0 + 2 * dir] >= 4) amino = 26;
          sc = msc[(size_t)amino * W1 + kk];
          if (prev == sI) sc += tsc[(size_t)(kk - 1) * 8 + IM];
          else if (prev == sD) sc += tsc[(size_t)(kk - 1) * 8 + DM];
          else if (prev == sM && j < nc - 1) sc += tsc[(size_t)(kk - 1) * 8 + MM];
          ppv = P[(size_t)ii * W + (size_t)kk * 3 + cM];
        } else if (s == sI) { sc = tsc[(size_t)kk * 8 + (prev == sI ? II : MI)]; ppv = P[(size_t)ii * W + (size_t)kk * 3 + cI]; }
        else sc = tsc[(size_t)(kk - 1) * 8 + (prev == sD ? DD : MD)];
        if (col_pp) col_pp[pp_off + j] = ppv;
      }
      const int nin = min(64, nc - jb);
      if (msc) for (int l = 0; l < nin; l++) total += rlf(sc, l);
      kbase += __popcll(bMm | bDm); ibase += __popcll(bMm | bIm);
      prev_last = __builtin_amdgcn_readlane(s, 63);
    }
    if (msc) r.aliscore = total;
  }
  r.ok = 1;
  if (lane == 0) out[t] = r;
}

// The same decoding, optimal-accuracy fill and null2 with a WAVE per envelope: lanes own C consecutive nodes each, rows are
// walked in order with the previous row in registers (neighbours by shuffle), the row's D chain D(k+1) = max(MD_k ? M_k : 0,
// DD_k ? D_k : 0) is a wavefront scan over functions x -> pass ? max(c, x) : c (closed under composition), xE a wave max.
// max / select only, so the OA matrix is bit-identical to the serial fill and the traceback (std_envelope_kernel with
// filled = 1, a lane per envelope) takes the same path.  Posteriors live in registers only: their column sums for null2 are
// accumulated on the fly in the serial kernel's order; only null2's sum over the nodes is associated differently (wave sum).
// The lane-per-envelope fill took 35 ms for 1500 envelopes (every cell a dependent trip to HBM); this one is well under 1 ms.
template <int C>
__global__ __launch_bounds__(256) void std_envelope_fill_kernel(SeqView sq, int M, const float *__restrict__ tf, const float *__restrict__ rf, float *__restrict__ fwd,
                                                                float *__restrict__ bck, const int64_t *__restrict__ dp_off, const float *__restrict__ fx,
                                                                const float *__restrict__ bx, const int64_t *__restrict__ x_off, float *__restrict__ ppx_all,
                                                                float *__restrict__ oax_all, StdEnvOut *__restrict__ out) {
  enum { XE = 0, XN, XJ, XB, XC, XS };
  enum { cM = 0, cD = 1, cI = 2 };
  enum { MM = 0, IM, DM, BM, MD, DD, MI, II };
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // What the row loop needs per node, in registers, so that it has no branch per node (a wave alone on its SIMD issues an
  // instruction every five cycles or so: with a branch, two compares and two selects per allow() the row of a 459-node model took
  // 1300 instructions, 3.6 us): the transitions as all-ones / zero masks (allow(tr, v) = v AND mask: exact, v or +0.0), and for
  // the nodes beyond M limits that turn what is computed for them into the values the guarded code left there (min(x, -inf)).
  auto band = [](float v, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };
  auto vmax = [](float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };      // (no operand is ever NaN: see bath_fs_device.hpp)
  auto vmin = [](float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  unsigned kBM[C], kMM[C], kIM[C], kDM[C], kMD[C], kDD[C], kMI[C], kII[C], vm[C];
  float lim[C], npz[C], plim[C];
  const bool full = lane * C + C <= M, partial = !full && lane * C + 1 <= M;
#pragma unroll
  for (int c = 0; c < C; c++) {
    const int node = lane * C + c + 1;
    const bool ok = node <= M;
    const float *tq = tf + (size_t)min(node, M) * 8;
    auto mk = [&](int q) { return (ok && tq[q] > 0.0f) ? 0xffffffffu : 0u; };
    kMM[c] = mk(MM); kIM[c] = mk(IM); kDM[c] = mk(DM); kBM[c] = mk(BM); kMD[c] = mk(MD); kDD[c] = mk(DD); kMI[c] = mk(MI); kII[c] = mk(II);
    vm[c] = ok ? 0xffffffffu : 0u;
    lim[c] = ok ? INFINITY : -INFINITY;
    const bool pass = !ok || tq[DD] > 0.0f;                                 // a node that is not there: the identity of the D chain
    npz[c] = pass ? -INFINITY : 0.0f;
    plim[c] = pass ? INFINITY : -INFINITY;
  }
  for (int64_t t0 = wid; t0 < sq.n; t0 += nw) {
    const int64_t t = __builtin_amdgcn_readfirstlane((int)t0);              // (the wave's envelope: uniform, so that its pointers are scalars)
    const int L = sq.len[t];
    const uint8_t *dsq = sq.data + sq.off[t] - 1;
    const size_t W = (size_t)(M + 1) * 3;
    float *F = fwd + dp_off[t];
    float *Bk = bck + dp_off[t];
    const float *FX = fx + x_off[t], *BX = bx + x_off[t];
    float *PX = ppx_all + (x_off[t] / 6) * 5, *OX = oax_all + (x_off[t] / 6) * 5;
    const float ploop = 1.0f - 2.0f / ((float)L + 2.0f);
    float pvM[C], pvI[C], pvD[C], emM[C], emI[C];
#pragma unroll
    for (int c = 0; c < C; c++) { pvM[c] = pvI[c] = pvD[c] = -INFINITY; emM[c] = emI[c] = 0.f; }
    float oxN = 0.f, oxJ = -INFINITY, oxC = -INFINITY, oxB = 0.f, sN = 0.f, sC = 0.f, sJ = 0.f;
    float scaleproduct = (float)(1.0 / (double)BX[XN]);
    for (int k = lane; k <= M; k += 64) F[(size_t)k * 3] = F[(size_t)k * 3 + 1] = F[(size_t)k * 3 + 2] = -INFINITY;      // OA row 0
    if (lane == 0) {
      for (int q = 0; q < 5; q++) PX[q] = 0.f;
      OX[XE] = -INFINITY; OX[XN] = 0.f; OX[XJ] = -INFINITY; OX[XB] = 0.f; OX[XC] = -INFINITY;
    }
    // Nothing in a row waits for memory: the Forward and Backward cells of row i + 1 are fetched while row i is worked on, and the
    // special states arrive 64 rows at a time, a lane each (lane l of a chunk: Forward's N, J, C of row r - 1 and its scale factor of
    // row r, Backward's N, J, C and scale factor of row r), the next 64 in flight.
    // (a lane reads its C nodes' cells whether they exist or not -- the buffers end in slack for that -- and masks what it read)
    float fM[C], fI[C], bM[C], bI[C];
    const size_t lane0 = (size_t)(lane * C + 1) * 3;
    auto fetch_row = [&](int r) {
      const float *fq = F + (size_t)r * W + lane0, *bq = Bk + (size_t)r * W + lane0;
#pragma unroll
      for (int c = 0; c < C; c++) { fM[c] = fq[3 * c + cM]; fI[c] = fq[3 * c + cI]; bM[c] = bq[3 * c + cM]; bI[c] = bq[3 * c + cI]; }
    };
    float xb[8], xn[8];
    auto fetch_x = [&](int r0, float (&v)[8]) {                             // rows r0 .. r0 + 63, a lane each
      const int r = min(r0 + lane, L);
      v[0] = FX[(size_t)(r - 1) * 6 + XN]; v[1] = FX[(size_t)(r - 1) * 6 + XJ]; v[2] = FX[(size_t)(r - 1) * 6 + XC]; v[3] = FX[(size_t)r * 6 + XS];
      v[4] = BX[(size_t)r * 6 + XN]; v[5] = BX[(size_t)r * 6 + XJ]; v[6] = BX[(size_t)r * 6 + XC]; v[7] = BX[(size_t)r * 6 + XS];
    };
    if (L >= 1) { fetch_row(1); fetch_x(1, xb); }
    for (int i = 1; i <= L; i++) {
      const int j = (i - 1) & 63;
      if (j == 0 && i + 64 <= L) fetch_x(i + 64, xn);
      float xv[8];
#pragma unroll
      for (int q = 0; q < 8; q++) xv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb[q]), j));
      if (j == 63) {
#pragma unroll
        for (int q = 0; q < 8; q++) xb[q] = xn[q];
      }
      const float totr = scaleproduct * xv[3];
      float *frow = F + (size_t)i * W;
      float *brow = Bk + (size_t)i * W;
      // posteriors of this row (p7_Decoding), special states included; they replace the Backward row as in the reference
      // (p7_Decoding(om, ox1, ox2, ox2), p7_domaindef.c:1249: the traceback kernel reads a column's posterior there)
      float pM[C], pI[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        pM[c] = band(fM[c] * (bM[c] * totr), vm[c]);
        pI[c] = band(fI[c] * (bI[c] * totr), vm[c]);
        if (i == 1) { emM[c] = pM[c]; emI[c] = pI[c]; } else { emM[c] = pM[c] + emM[c]; emI[c] = pI[c] + emI[c]; }
      }
      if (i < L) fetch_row(i + 1);                                          // in flight during the optimal-accuracy part of this row
      const float pxN = xv[0] * xv[4] * ploop * scaleproduct;
      const float pxJ = xv[1] * xv[5] * ploop * scaleproduct;
      const float pxC = xv[2] * xv[6] * ploop * scaleproduct;
      if (i == 1) { sN = pxN; sC = pxC; sJ = pxJ; } else { sN += pxN; sC += pxC; sJ += pxJ; }
      scaleproduct *= xv[3] / xv[7];
      // optimal-accuracy row (p7_OptimalAccuracy)
      const float mIn = wave_shr1_f32(pvM[C - 1], -INFINITY), iIn = wave_shr1_f32(pvI[C - 1], -INFINITY), dIn = wave_shr1_f32(pvD[C - 1], -INFINITY);
      float cuM[C], cuI[C], cuD[C];
      float fc = -INFINITY; unsigned fpass = 0xffffffffu;                   // the lane's composite D-chain function
      float xE = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const float m1 = c == 0 ? mIn : pvM[c - 1], i1 = c == 0 ? iIn : pvI[c - 1], d1 = c == 0 ? dIn : pvD[c - 1];
        float sv = band(oxB, kBM[c]);
        sv = vmax(sv, band(m1, kMM[c]));
        sv = vmax(sv, band(i1, kIM[c]));
        sv = vmax(sv, band(d1, kDM[c]));
        sv = vmin(sv + pM[c], lim[c]);                                      // (a node beyond M: -inf)
        cuM[c] = sv;
        xE = vmax(xE, sv);
        cuI[c] = vmin(vmax(band(pvM[c], kMI[c]), band(pvI[c], kII[c])) + pI[c], lim[c]);
        const float cst = vmin(band(sv, kMD[c]), lim[c]);                   // f_node(x) = max(cst, DD ? x : 0); beyond M: the identity (cst = -inf, pass)
        const float g = vmax(cst, npz[c]);                                  // pass ? cst : max(cst, 0)
        fc = vmax(g, vmin(fc, plim[c]));                                    // pass ? max(g, fc) : g -- f_node o (what the lane has so far)
        fpass &= kDD[c] | ~vm[c];
      }
      // inclusive scan of the lanes' functions, then the value entering each lane: D(first node of the lane)
      // by DPP; lanes without a source see the identity function (c = -inf, pass): max and select only, any scan order gives the same
      float sc_c = fc; int sc_p = fpass ? 1 : 0;
#define BATH_OAS_STEP(CTRL, MASK) { const float oc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000, __builtin_bit_cast(int, sc_c), CTRL, MASK, 0xf, false)); \
                                    const int op = __builtin_amdgcn_update_dpp(1, sc_p, CTRL, MASK, 0xf, false); \
                                    if (sc_p) { sc_c = fmaxf(sc_c, oc); sc_p = op; } }
      BATH_OAS_STEP(0x111, 0xf) BATH_OAS_STEP(0x112, 0xf) BATH_OAS_STEP(0x114, 0xf) BATH_OAS_STEP(0x118, 0xf) BATH_OAS_STEP(0x142, 0xa) BATH_OAS_STEP(0x143, 0xc)
#undef BATH_OAS_STEP
      float din = wave_shr1_f32(sc_c, -INFINITY);                           // prefix over lanes 0..lane-1 applied to D(1) = -inf
#pragma unroll
      for (int c = 0; c < C; c++) {
        cuD[c] = vmin(din, lim[c]);
        xE = vmax(xE, cuD[c]);
        din = vmax(band(cuM[c], kMD[c]), band(din, kDD[c]));
        pvM[c] = cuM[c]; pvI[c] = cuI[c]; pvD[c] = cuD[c];
      }
      // the row's posteriors over Backward's, its OA cells over Forward's: a lane whose C nodes all exist stores them without a test
      if (full) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          float *bq = brow + (size_t)(lane * C + 1) * 3 + 3 * c, *fq = frow + (size_t)(lane * C + 1) * 3 + 3 * c;
          bq[cM] = pM[c]; bq[cD] = 0.0f; bq[cI] = pI[c];
          fq[cM] = cuM[c]; fq[cD] = cuD[c]; fq[cI] = cuI[c];
        }
      } else if (partial) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          if (node <= M) {
            brow[(size_t)node * 3 + cM] = pM[c]; brow[(size_t)node * 3 + cI] = pI[c]; brow[(size_t)node * 3 + cD] = 0.0f;
            frow[(size_t)node * 3 + cM] = cuM[c]; frow[(size_t)node * 3 + cD] = cuD[c]; frow[(size_t)node * 3 + cI] = cuI[c];
          }
        }
      }
      xE = wave_max_f32(xE);
      oxJ = fmaxf(oxJ + pxJ, 0.0f);
      oxC = fmaxf(oxC + pxC, xE);
      oxN = oxN + pxN;
      oxB = fmaxf(oxN, oxJ);
      if (lane == 0) {
        frow[0] = frow[1] = frow[2] = -INFINITY;
        float *px = PX + (size_t)i * 5, *ox = OX + (size_t)i * 5;
        px[XE] = 0.f; px[XN] = pxN; px[XJ] = pxJ; px[XB] = 0.f; px[XC] = pxC;
        ox[XE] = xE; ox[XN] = oxN; ox[XJ] = oxJ; ox[XB] = oxB; ox[XC] = oxC;
      }
    }
    StdEnvOut r{-1, -1, -1, -1, 0, 0.f, 0.f, 0, 0};
    if (!isinf(scaleproduct)) {                                             // else eslERANGE: the domain is dropped
      // p7_Null2_ByExpectation and the correction over the envelope
      const float norm = (float)(1.0 / (double)(float)L);
#pragma unroll
      for (int c = 0; c < C; c++) { emM[c] *= norm; emI[c] *= norm; }
      const float xfactor = sN * norm + sC * norm + sJ * norm;
      float null2[kKp];
      for (int x = 0; x < 20; x++) {
        const float *e = rf + (size_t)x * (M + 1);
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) { const int node = lane * C + c + 1; if (node <= M) { sv += emM[c] * e[node]; sv += emI[c]; } }
        null2[x] = wave_sum_f32(sv) + xfactor;
      }
      const int mem[6][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}, {-1, -1}};      // B=DN J=IL Z=EQ O=K U=C X=any
      for (int dx = 0; dx < 6; dx++) {
        float sum = 0.f; int cnt = 0;
        if (dx == 5) { for (int y = 0; y < 20; y++) { sum += null2[y]; cnt++; } }
        else { const int a = min(mem[dx][0], mem[dx][1]), b = max(mem[dx][0], mem[dx][1]); sum += null2[a]; cnt++; if (b != a) { sum += null2[b]; cnt++; } }
        null2[21 + dx] = sum / (float)cnt;
      }
      null2[20] = 1.0f; null2[27] = 1.0f; null2[28] = 1.0f;
      float corr = 0.f;
      for (int pos = 1 + lane; pos <= L; pos += 64) {
        const int x = min((int)dsq[pos], kKp - 1);
        float v = null2[0];
#pragma unroll
        for (int q = 1; q < kKp; q++) v = (x == q) ? null2[q] : v;          // registers cannot be indexed: select
        corr += logf(v);
      }
      r.domcorrection = wave_sum_f32(corr);
      r.oasc = oxC;
      r.ok = 1;
    }
    if (lane == 0) out[t] = r;
  }
}

// The same fill with a BLOCK of four waves per envelope (round 5): models beyond 192 nodes, where a single wave holds 4-16 nodes per
// lane and a row is hundreds of instructions that one wave issues one every ~5 cycles.  Wave w owns nodes 64 w C + 1 .. 64 (w+1) C
// (C = 1, 2, 4: up to 1024 nodes), a lane C consecutive nodes as before.  What crosses a wave boundary goes through LDS: the
// previous row's last cells of wave w - 1 (M, I, D of the node to the left), the waves' composite D-chain functions (wave w applies
// those of waves 0 .. w - 1 to D(1) = -inf before its own lanes' prefixes) and the waves' row maxima for xE -- which no cell of the
// next row reads (unihit: B comes from N and J, and J does not read E), so wave 0 folds them into C and the special-state rows one
// row late.  Two LDS-only barriers per row.  max / select / AND arithmetic as in the one-wave kernel: the same OA matrix,
// posteriors and special rows bit for bit; only null2's sums over the nodes and the residues associate by wave (as they already
// differ between the one-wave kernel and the serial one).
template <int C>
__global__ __launch_bounds__(256) void std_envelope_fill_mw_kernel(SeqView sq, int M, const float *__restrict__ tf, const float *__restrict__ rf, float *__restrict__ fwd,
                                                                   float *__restrict__ bck, const int64_t *__restrict__ dp_off, const float *__restrict__ fx,
                                                                   const float *__restrict__ bx, const int64_t *__restrict__ x_off, float *__restrict__ ppx_all,
                                                                   float *__restrict__ oax_all, StdEnvOut *__restrict__ out) {
  enum { XE = 0, XN, XJ, XB, XC, XS };
  enum { cM = 0, cD = 1, cI = 2 };
  enum { MM = 0, IM, DM, BM, MD, DD, MI, II };
  constexpr int NW = 4;
  __shared__ float s_nbr[2][NW][3];        // [row parity][wave]: M, I, D of the wave's last node
  __shared__ float s_fc[NW];
  __shared__ int s_fp[NW];
  __shared__ float s_xe[2][NW];            // [row parity][wave]: the wave's maximum over the row's M and D cells
  __shared__ float s_sum[NW][20];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = wv * 64 + lane;                                              // the lane's rank among the block's 256
  auto lds_barrier = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
  auto band = [](float v, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m); };
  auto vmax = [](float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto vmin = [](float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  unsigned kBM[C], kMM[C], kIM[C], kDM[C], kMD[C], kDD[C], kMI[C], kII[C], vm[C];
  float lim[C], npz[C], plim[C];
  const bool full = gl * C + C <= M, partial = !full && gl * C + 1 <= M;
#pragma unroll
  for (int c = 0; c < C; c++) {
    const int node = gl * C + c + 1;
    const bool ok = node <= M;
    const float *tq = tf + (size_t)min(node, M) * 8;
    auto mk = [&](int q) { return (ok && tq[q] > 0.0f) ? 0xffffffffu : 0u; };
    kMM[c] = mk(MM); kIM[c] = mk(IM); kDM[c] = mk(DM); kBM[c] = mk(BM); kMD[c] = mk(MD); kDD[c] = mk(DD); kMI[c] = mk(MI); kII[c] = mk(II);
    vm[c] = ok ? 0xffffffffu : 0u;
    lim[c] = ok ? INFINITY : -INFINITY;
    const bool pass = !ok || tq[DD] > 0.0f;
    npz[c] = pass ? -INFINITY : 0.0f;
    plim[c] = pass ? INFINITY : -INFINITY;
  }
  for (int64_t t = blockIdx.x; t < sq.n; t += gridDim.x) {
    const int L = sq.len[t];
    const uint8_t *dsq = sq.data + sq.off[t] - 1;
    const size_t W = (size_t)(M + 1) * 3;
    float *F = fwd + dp_off[t];
    float *Bk = bck + dp_off[t];
    const float *FX = fx + x_off[t], *BX = bx + x_off[t];
    float *PX = ppx_all + (x_off[t] / 6) * 5, *OX = oax_all + (x_off[t] / 6) * 5;
    const float ploop = 1.0f - 2.0f / ((float)L + 2.0f);
    float pvM[C], pvI[C], pvD[C], emM[C], emI[C];
#pragma unroll
    for (int c = 0; c < C; c++) { pvM[c] = pvI[c] = pvD[c] = -INFINITY; emM[c] = emI[c] = 0.f; }
    float oxN = 0.f, oxJ = -INFINITY, oxC = -INFINITY, oxB = 0.f, sN = 0.f, sC = 0.f, sJ = 0.f;
    float scaleproduct = (float)(1.0 / (double)BX[XN]);
    for (int k = threadIdx.x; k <= M; k += blockDim.x) F[(size_t)k * 3] = F[(size_t)k * 3 + 1] = F[(size_t)k * 3 + 2] = -INFINITY;      // OA row 0
    if (threadIdx.x == 0) {
      for (int q = 0; q < 5; q++) PX[q] = 0.f;
      OX[XE] = -INFINITY; OX[XN] = 0.f; OX[XJ] = -INFINITY; OX[XB] = 0.f; OX[XC] = -INFINITY;
    }
    if (lane == 63) { s_nbr[0][wv][0] = s_nbr[0][wv][1] = s_nbr[0][wv][2] = -INFINITY; }          // "row 0" as row 1 reads it
    float fM[C], fI[C], bM[C], bI[C];
    const size_t lane0 = (size_t)(gl * C + 1) * 3;
    auto fetch_row = [&](int r) {
      const float *fq = F + (size_t)r * W + lane0, *bq = Bk + (size_t)r * W + lane0;
#pragma unroll
      for (int c = 0; c < C; c++) { fM[c] = fq[3 * c + cM]; fI[c] = fq[3 * c + cI]; bM[c] = bq[3 * c + cM]; bI[c] = bq[3 * c + cI]; }
    };
    float xb[8], xn[8];
    auto fetch_x = [&](int r0, float (&v)[8]) {                             // rows r0 .. r0 + 63, a lane each (every wave its own copy)
      const int r = min(r0 + lane, L);
      v[0] = FX[(size_t)(r - 1) * 6 + XN]; v[1] = FX[(size_t)(r - 1) * 6 + XJ]; v[2] = FX[(size_t)(r - 1) * 6 + XC]; v[3] = FX[(size_t)r * 6 + XS];
      v[4] = BX[(size_t)r * 6 + XN]; v[5] = BX[(size_t)r * 6 + XJ]; v[6] = BX[(size_t)r * 6 + XC]; v[7] = BX[(size_t)r * 6 + XS];
    };
    if (L >= 1) { fetch_row(1); fetch_x(1, xb); }
    float pxN_prev = 0.f, pxJ_prev = 0.f, pxC_prev = 0.f;
    // wave 0: C and the special-state rows of row r, once the waves' maxima of that row have crossed the barrier that ended it
    // (oxN, oxJ, oxB still hold row r's values then: they are advanced at the end of the next row)
    auto finish_row = [&](int r) {
      float xE = s_xe[r & 1][0];
#pragma unroll
      for (int w = 1; w < NW; w++) xE = vmax(xE, s_xe[r & 1][w]);
      oxC = fmaxf(oxC + pxC_prev, xE);
      if (lane == 0) {
        float *px = PX + (size_t)r * 5, *ox = OX + (size_t)r * 5;
        px[XE] = 0.f; px[XN] = pxN_prev; px[XJ] = pxJ_prev; px[XB] = 0.f; px[XC] = pxC_prev;
        ox[XE] = xE; ox[XN] = oxN; ox[XJ] = oxJ; ox[XB] = oxB; ox[XC] = oxC;
      }
    };
    lds_barrier();
    for (int i = 1; i <= L; i++) {
      const int par = i & 1;
      const int j = (i - 1) & 63;
      if (j == 0 && i + 64 <= L) fetch_x(i + 64, xn);
      float xv[8];
#pragma unroll
      for (int q = 0; q < 8; q++) xv[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb[q]), j));
      if (j == 63) {
#pragma unroll
        for (int q = 0; q < 8; q++) xb[q] = xn[q];
      }
      const float totr = scaleproduct * xv[3];
      float *frow = F + (size_t)i * W;
      float *brow = Bk + (size_t)i * W;
      float pM[C], pI[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        pM[c] = band(fM[c] * (bM[c] * totr), vm[c]);
        pI[c] = band(fI[c] * (bI[c] * totr), vm[c]);
        if (i == 1) { emM[c] = pM[c]; emI[c] = pI[c]; } else { emM[c] = pM[c] + emM[c]; emI[c] = pI[c] + emI[c]; }
      }
      if (i < L) fetch_row(i + 1);
      const float pxN = xv[0] * xv[4] * ploop * scaleproduct;
      const float pxJ = xv[1] * xv[5] * ploop * scaleproduct;
      const float pxC = xv[2] * xv[6] * ploop * scaleproduct;
      if (i == 1) { sN = pxN; sC = pxC; sJ = pxJ; } else { sN += pxN; sC += pxC; sJ += pxJ; }
      scaleproduct *= xv[3] / xv[7];
      float mIn = wave_shr1_f32(pvM[C - 1], -INFINITY), iIn = wave_shr1_f32(pvI[C - 1], -INFINITY), dIn = wave_shr1_f32(pvD[C - 1], -INFINITY);
      if (lane == 0 && wv > 0) { mIn = s_nbr[par ^ 1][wv - 1][0]; iIn = s_nbr[par ^ 1][wv - 1][1]; dIn = s_nbr[par ^ 1][wv - 1][2]; }
      float cuM[C], cuI[C], cuD[C];
      float fc = -INFINITY; unsigned fpass = 0xffffffffu;
      float xE = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const float m1 = c == 0 ? mIn : pvM[c - 1], i1 = c == 0 ? iIn : pvI[c - 1], d1 = c == 0 ? dIn : pvD[c - 1];
        float sv = band(oxB, kBM[c]);
        sv = vmax(sv, band(m1, kMM[c]));
        sv = vmax(sv, band(i1, kIM[c]));
        sv = vmax(sv, band(d1, kDM[c]));
        sv = vmin(sv + pM[c], lim[c]);
        cuM[c] = sv;
        xE = vmax(xE, sv);
        cuI[c] = vmin(vmax(band(pvM[c], kMI[c]), band(pvI[c], kII[c])) + pI[c], lim[c]);
        const float cst = vmin(band(sv, kMD[c]), lim[c]);
        const float g = vmax(cst, npz[c]);
        fc = vmax(g, vmin(fc, plim[c]));
        fpass &= kDD[c] | ~vm[c];
      }
      float sc_c = fc; int sc_p = fpass ? 1 : 0;
#define BATH_OAS_STEP(CTRL, MASK) { const float oc = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp((int)0xff800000, __builtin_bit_cast(int, sc_c), CTRL, MASK, 0xf, false)); \
                                    const int op = __builtin_amdgcn_update_dpp(1, sc_p, CTRL, MASK, 0xf, false); \
                                    if (sc_p) { sc_c = fmaxf(sc_c, oc); sc_p = op; } }
      BATH_OAS_STEP(0x111, 0xf) BATH_OAS_STEP(0x112, 0xf) BATH_OAS_STEP(0x114, 0xf) BATH_OAS_STEP(0x118, 0xf) BATH_OAS_STEP(0x142, 0xa) BATH_OAS_STEP(0x143, 0xc)
#undef BATH_OAS_STEP
      if (lane == 63) { s_fc[wv] = sc_c; s_fp[wv] = sc_p; }                   // the wave's composite function
      if (wv == 0 && i > 1) finish_row(i - 1);                                // (row i - 1's maxima are behind the barrier that ended it)
      lds_barrier();                                                           // ---- barrier 1: the waves' functions of this row
      float xin = -INFINITY;                                                   // D entering this wave: the waves before it applied to D(1) = -inf
      for (int w = 0; w < wv; w++) xin = s_fp[w] ? vmax(s_fc[w], xin) : s_fc[w];
      const float ec = wave_shr1_f32(sc_c, -INFINITY);                         // exclusive prefix of this wave's lanes (lane 0: the identity)
      int ep = __builtin_amdgcn_update_dpp(1, sc_p, 0x138, 0xf, 0xf, false);   // wave_shr:1
      if (lane == 0) ep = 1;
      float din = ep ? vmax(ec, xin) : ec;
#pragma unroll
      for (int c = 0; c < C; c++) {
        cuD[c] = vmin(din, lim[c]);
        xE = vmax(xE, cuD[c]);
        din = vmax(band(cuM[c], kMD[c]), band(din, kDD[c]));
        pvM[c] = cuM[c]; pvI[c] = cuI[c]; pvD[c] = cuD[c];
      }
      if (full) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          float *bq = brow + lane0 + 3 * c, *fq = frow + lane0 + 3 * c;
          bq[cM] = pM[c]; bq[cD] = 0.0f; bq[cI] = pI[c];
          fq[cM] = cuM[c]; fq[cD] = cuD[c]; fq[cI] = cuI[c];
        }
      } else if (partial) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = gl * C + c + 1;
          if (node <= M) {
            brow[(size_t)node * 3 + cM] = pM[c]; brow[(size_t)node * 3 + cI] = pI[c]; brow[(size_t)node * 3 + cD] = 0.0f;
            frow[(size_t)node * 3 + cM] = cuM[c]; frow[(size_t)node * 3 + cD] = cuD[c]; frow[(size_t)node * 3 + cI] = cuI[c];
          }
        }
      }
      if (threadIdx.x == 0) { frow[0] = frow[1] = frow[2] = -INFINITY; }
      xE = wave_max_f32(xE);
      if (lane == 63) { s_nbr[par][wv][0] = pvM[C - 1]; s_nbr[par][wv][1] = pvI[C - 1]; s_nbr[par][wv][2] = pvD[C - 1]; s_xe[par][wv] = xE; }
      // N, J and B of this row (every wave its own copy: none of them reads E); C waits for xE (finish_row, one row late)
      oxJ = fmaxf(oxJ + pxJ, 0.0f);
      oxN = oxN + pxN;
      oxB = fmaxf(oxN, oxJ);
      pxN_prev = pxN; pxJ_prev = pxJ; pxC_prev = pxC;
      lds_barrier();                                                           // ---- barrier 2: neighbours and maxima of row i
    }
    if (wv == 0 && L >= 1) finish_row(L);
    // oasc = C(L) lives in wave 0: hand it to everybody with the null2 sums below
    StdEnvOut r{-1, -1, -1, -1, 0, 0.f, 0.f, 0, 0};
    if (!isinf(scaleproduct)) {                                             // else eslERANGE: the domain is dropped
      const float norm = (float)(1.0 / (double)(float)L);
#pragma unroll
      for (int c = 0; c < C; c++) { emM[c] *= norm; emI[c] *= norm; }
      const float xfactor = sN * norm + sC * norm + sJ * norm;
      float null2[kKp];
      for (int x = 0; x < 20; x++) {
        const float *e = rf + (size_t)x * (M + 1);
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) { const int node = gl * C + c + 1; if (node <= M) { sv += emM[c] * e[node]; sv += emI[c]; } }
        const float ws = wave_sum_f32(sv);
        if (lane == 0) s_sum[wv][x] = ws;
      }
      lds_barrier();
      for (int x = 0; x < 20; x++) null2[x] = (((s_sum[0][x] + s_sum[1][x]) + s_sum[2][x]) + s_sum[3][x]) + xfactor;
      lds_barrier();
      const int mem[6][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}, {-1, -1}};      // B=DN J=IL Z=EQ O=K U=C X=any
      for (int dx = 0; dx < 6; dx++) {
        float sum = 0.f; int cnt = 0;
        if (dx == 5) { for (int y = 0; y < 20; y++) { sum += null2[y]; cnt++; } }
        else { const int a = min(mem[dx][0], mem[dx][1]), b = max(mem[dx][0], mem[dx][1]); sum += null2[a]; cnt++; if (b != a) { sum += null2[b]; cnt++; } }
        null2[21 + dx] = sum / (float)cnt;
      }
      null2[20] = 1.0f; null2[27] = 1.0f; null2[28] = 1.0f;
      float corr = 0.f;
      for (int pos = 1 + (int)threadIdx.x; pos <= L; pos += (int)blockDim.x) {
        const int x = min((int)dsq[pos], kKp - 1);
        float v = null2[0];
#pragma unroll
        for (int q = 1; q < kKp; q++) v = (x == q) ? null2[q] : v;
        corr += logf(v);
      }
      const float wc = wave_sum_f32(corr);
      if (lane == 0) s_sum[wv][0] = wc;
      lds_barrier();
      r.domcorrection = ((s_sum[0][0] + s_sum[1][0]) + s_sum[2][0]) + s_sum[3][0];
      r.oasc = oxC;                                                         // (wave 0's: the only reader is thread 0 below)
      r.ok = 1;
    }
    if (threadIdx.x == 0) out[t] = r;
    lds_barrier();                                                          // the LDS slots are the next envelope's
  }
}

}  // namespace

// Domain definition and hit scores for ORFs that passed the Forward filter; appends to ctx->fs_domains.
static int std_domains(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const std::vector<PipelineSurvivor> &surv,
                       const uint8_t *d_pool, const DomOpts &opt, int64_t *n_clustered_regions) {
  int st;
  const int64_t ns = (int64_t)surv.size();
  if (ns == 0) return BATH_OK;
  const int M = om->M;
  StageClock clk;

  // ---- the survivors as a sequence view into the amino-acid streams; both parsers with their special-state rows
  bath_hip_seqs view;
  view.ctx = ctx; view.n = ns; view.is_part = true;
  view.h_off.resize((size_t)ns); view.h_len.resize((size_t)ns);
  std::vector<int64_t> xoff((size_t)ns + 1, 0);
  for (int64_t i = 0; i < ns; i++) {
    view.h_off[(size_t)i] = surv[(size_t)i].aa_off; view.h_len[(size_t)i] = surv[(size_t)i].n; view.maxlen = std::max(view.maxlen, surv[(size_t)i].n);
    xoff[(size_t)i + 1] = xoff[(size_t)i] + ((int64_t)surv[(size_t)i].n + 1) * 6;
  }
  DevBuf &b_idx = ctx->scratch[0], &b_fx = ctx->scratch[6], &b_bx = ctx->scratch[7], &b_sc = ctx->scratch[2], &b_st = ctx->scratch[3], &b_work = ctx->scratch[4], &b_reg = ctx->scratch[5];
  // The small transfers of a stage go through page-locked memory (ctx->stage[4] up, [5] down): a hipMemcpyAsync from or to
  // pageable memory is staged by the runtime and holds the host ~20 us, and one query's domain stage had two dozen of them -- a
  // fifth of its time on a 12.5 Mb block (configs[3]).  up_begin() sizes the staging area for ALL uploads of a stage (it may move
  // while nothing is in flight: every stage ends with a synchronize); a sequence view's three arrays travel as ONE copy.
  size_t up_used = 0;
  auto up_begin = [&](size_t bytes) -> int { up_used = 0; BATH_HIP_TRY(ctx, ctx->stage[4].reserve(bytes + 1024)); return BATH_OK; };
  auto up = [&](void *dst, std::initializer_list<std::pair<const void *, size_t>> parts) -> int {
    size_t bytes = 0;
    for (const auto &q : parts) bytes += q.second;
    const size_t end = up_used + ((bytes + 63) & ~(size_t)63);
    if (end > ctx->stage[4].cap) { ctx->set_error("staging area of the domain stage too small"); return BATH_EFAIL; }      // before anything is written
    char *h = static_cast<char *>(ctx->stage[4].p) + up_used;
    bytes = 0;
    for (const auto &q : parts) { std::memcpy(h + bytes, q.first, q.second); bytes += q.second; }
    up_used = end;
    BATH_HIP_TRY(ctx, hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return BATH_OK;
  };
  auto view_bytes = [](int64_t n) { return (size_t)n * 12 + (size_t)(n + 1) * 8 + 256; };
  auto upload_view = [&](bath_hip_seqs &v, const std::vector<int64_t> &xo, int64_t n) -> int {
    BATH_HIP_TRY(ctx, b_idx.reserve(view_bytes(n)));
    int64_t *d_off = b_idx.as<int64_t>();
    int64_t *d_xo = d_off + n;
    int32_t *d_len = reinterpret_cast<int32_t *>(d_xo + n + 1);
    const int rc = up(d_off, {{v.h_off.data(), (size_t)n * 8}, {xo.data(), (size_t)(n + 1) * 8}, {v.h_len.data(), (size_t)n * 4}});      // contiguous on the device too
    if (rc != BATH_OK) return rc;
    v.d_data = const_cast<uint8_t *>(d_pool); v.d_off = d_off; v.d_len = d_len;
    return BATH_OK;
  };
  // downloads: into page-locked memory, then (after the stage's synchronize) a host copy into the vector the code reads
  auto down_reserve = [&](size_t bytes) -> int { BATH_HIP_TRY(ctx, ctx->stage[5].reserve(bytes + 1024)); return BATH_OK; };
  if ((st = om->ensure_len_tables(view.maxlen)) != BATH_OK) return st;
  if ((st = up_begin(view_bytes(ns) + (size_t)ns * 8 + 256)) != BATH_OK) return st;                 // (+ the kept Forward rows' offsets, below)
  if ((st = upload_view(view, xoff, ns)) != BATH_OK) return st;
  const int64_t *d_xoff = b_idx.as<int64_t>() + ns;
  BATH_HIP_TRY(ctx, b_fx.reserve((size_t)xoff[(size_t)ns] * 4 + 64)); BATH_HIP_TRY(ctx, b_bx.reserve((size_t)xoff[(size_t)ns] * 4 + 64));
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)ns * 8)); BATH_HIP_TRY(ctx, b_st.reserve((size_t)ns * 8));
  BATH_HIP_TRY(ctx, b_work.reserve((size_t)xoff[(size_t)ns] / 6 * 3 * 4 + 64));
  const int RS = 1 + 3 * kStdMaxRegions;
  BATH_HIP_TRY(ctx, b_reg.reserve((size_t)ns * RS * 4 + 64));
  // The Forward parser's rows: the reference has them from the filter (pli->oxf, p7_pipeline.c:1741-1771).  When the cascade of this
  // call kept them (ctx->fwd_rows_kept: a one-lane block through bath_hip_pipeline_hits) the survivors' rows are copied into place;
  // otherwise (the standard branch of the --fs pipeline, blocks cut into lanes, rows that did not fit) the parser runs again.
  bool kept = ctx->fwd_rows_kept != nullptr;
  for (int64_t i = 0; i < ns && kept; i++) kept = surv[(size_t)i].fx_off >= 0;
  if (kept) {
    std::vector<int64_t> src((size_t)ns);
    for (int64_t i = 0; i < ns; i++) src[(size_t)i] = surv[(size_t)i].fx_off;
    DevBuf &b_src = ctx->scratch[49];
    BATH_HIP_TRY(ctx, b_src.reserve((size_t)ns * 8 + 64));
    if ((st = up(b_src.p, {{src.data(), (size_t)ns * 8}})) != BATH_OK) return st;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)std::min<int64_t>(ns, 65535)), dim3(256), 0, ctx->stream, ns, view.d_len, ctx->fwd_rows_kept, b_src.as<int64_t>(), b_fx.as<float>(), d_xoff);
    BATH_HIP_TRY(ctx, hipGetLastError());
  } else if ((st = launch_fwd_wave(ctx, om, view.view(), nullptr, ns, b_sc.as<float>(), b_st.as<int32_t>(), nullptr, b_fx.as<float>(), d_xoff)) != BATH_OK) return st;
  if ((st = launch_bwd_wave(ctx, om, view.view(), ns, b_fx.as<float>(), d_xoff, b_sc.as<float>() + ns, b_st.as<int32_t>() + ns, b_bx.as<float>())) != BATH_OK) return st;
  {
    const char *e = std::getenv("BATH_HIP_STD_SERIAL");
    const size_t shm = (size_t)4 * ((size_t)view.maxlen + 1) * sizeof(float);
    if (shm <= (size_t)128 * 1024 && !(e && e[0] == '1')) {                     // a wave per ORF, rows in LDS
      if (shm > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)std_regions_wave_kernel));
      hipLaunchKernelGGL(std_regions_wave_kernel, dim3((unsigned)std::min<int64_t>(ns, 65535)), dim3(64), shm, ctx->stream, ns, view.d_len, b_fx.as<float>(), b_bx.as<float>(),
                         d_xoff, om->lt.d_pmove, view.maxlen, b_reg.as<int32_t>());
    } else
      hipLaunchKernelGGL(std_regions_kernel, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, ctx->stream, ns, view.d_len, b_fx.as<float>(), b_bx.as<float>(), d_xoff, om->lt.d_pmove,
                         b_work.as<float>(), b_reg.as<int32_t>());
  }
  BATH_HIP_TRY(ctx, hipGetLastError());
  std::vector<int32_t> regions((size_t)ns * RS);
  if ((st = down_reserve(regions.size() * 4)) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[5].p, b_reg.p, regions.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(regions.data(), ctx->stage[5].p, regions.size() * 4);
  view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
  clk.lap("std:   parsers + regions");

  struct Env { int s, i, j; bool clustered; float n2corr; };
  std::vector<Env> envs, mregs;
  for (int64_t q = 0; q < ns; q++) {
    const int32_t *r = &regions[(size_t)q * RS];
    if (r[0] > kStdMaxRegions) { ctx->set_error("an ORF has more regions than the region buffer holds"); return BATH_ERANGE; }
    for (int k = 0; k < r[0]; k++) (r[3 + 3 * k] ? mregs : envs).push_back(Env{(int)q, r[1 + 3 * k], r[2 + 3 * k], false, 0.f});
  }
  if (n_clustered_regions) *n_clustered_regions += (int64_t)mregs.size();      // regions resolved by clustering (ddef->nclustered)

  // ---- multi-domain regions (p7_domaindef.c:539-583): p7_Forward of the region with the ORF's multihit configuration on the
  // GPU, then the stochastic-trace ensemble and its clustering on the host (bath_ensemble.hip); every cluster is an envelope
  if (!mregs.empty()) {
    const int64_t nm = (int64_t)mregs.size();
    bath_hip_seqs mv;
    mv.ctx = ctx; mv.n = nm; mv.is_part = true;
    mv.h_off.resize((size_t)nm); mv.h_len.resize((size_t)nm);
    std::vector<int64_t> mxoff((size_t)nm + 1, 0), mdpoff((size_t)nm + 1, 0);
    std::vector<int32_t> cfg((size_t)nm);
    for (int64_t e = 0; e < nm; e++) {
      const Env &en = mregs[(size_t)e];
      const int Lr = en.j - en.i + 1;
      mv.h_off[(size_t)e] = surv[(size_t)en.s].aa_off + en.i - 1; mv.h_len[(size_t)e] = Lr; mv.maxlen = std::max(mv.maxlen, Lr);
      cfg[(size_t)e] = surv[(size_t)en.s].n;
      mxoff[(size_t)e + 1] = mxoff[(size_t)e] + ((int64_t)Lr + 1) * 6;
      mdpoff[(size_t)e + 1] = mdpoff[(size_t)e] + ((int64_t)Lr + 1) * (M + 1) * 3;
    }
    if ((st = up_begin(view_bytes(nm) + (size_t)(nm + 1) * 8 + (size_t)nm * 4 + 256)) != BATH_OK) return st;
    if ((st = upload_view(mv, mxoff, nm)) != BATH_OK) return st;
    DevBuf &b_mdpo = ctx->scratch[20], &b_cfg = ctx->scratch[5];
    BATH_HIP_TRY(ctx, b_mdpo.reserve((size_t)(nm + 1) * 8)); BATH_HIP_TRY(ctx, b_cfg.reserve((size_t)nm * 4 + 64));
    BATH_HIP_TRY(ctx, b_fx.reserve((size_t)mxoff[(size_t)nm] * 4 + 64)); BATH_HIP_TRY(ctx, b_sc.reserve((size_t)nm * 8)); BATH_HIP_TRY(ctx, b_st.reserve((size_t)nm * 8));
    if ((st = up(b_mdpo.p, {{mdpoff.data(), (size_t)(nm + 1) * 8}})) != BATH_OK) return st;
    if ((st = up(b_cfg.p, {{cfg.data(), (size_t)nm * 4}})) != BATH_OK) return st;
    // The matrices, the special-state rows and the regions' residues are for the host (the ensembles' tracebacks): the kernels
    // write them straight into page-locked host memory, the transfer rides along with the computation.  Residues of region e:
    // h_res + roff[e], one row's worth of bytes per residue row (roff = the x-row offsets / 6).
    const size_t x_bytes = ((size_t)mxoff[(size_t)nm] * 4 + 255) / 256 * 256;
    BATH_HIP_TRY(ctx, ctx->pinned[0].reserve((size_t)mdpoff[(size_t)nm] * 4 + 64)); BATH_HIP_TRY(ctx, ctx->pinned[1].reserve(x_bytes + (size_t)mxoff[(size_t)nm] / 6 + 64));
    float *h_dp = ctx->pinned[0].as<float>(), *h_x = ctx->pinned[1].as<float>();      // page-locked
    const uint8_t *h_res = reinterpret_cast<const uint8_t *>(ctx->pinned[1].p) + x_bytes;
    std::vector<int64_t> roff((size_t)nm + 1, 0);
    for (int64_t e = 0; e <= nm; e++) roff[(size_t)e] = mxoff[(size_t)e] / 6;
    void *dv_dp = nullptr, *dv_x = nullptr;
    if (hipHostGetDevicePointer(&dv_dp, ctx->pinned[0].p, 0) != hipSuccess || hipHostGetDevicePointer(&dv_x, ctx->pinned[1].p, 0) != hipSuccess) {
      ctx->set_error("page-locked host memory is not mapped into the device's address space"); return BATH_EFAIL;
    }
    if ((st = launch_fwd_wave(ctx, om, mv.view(), nullptr, nm, b_sc.as<float>(), b_st.as<int32_t>(), nullptr, static_cast<float *>(dv_x), b_idx.as<int64_t>() + nm,
                              static_cast<float *>(dv_dp), b_mdpo.as<int64_t>(), 0, b_cfg.as<int32_t>())) != BATH_OK) return st;
    hipLaunchKernelGGL(gather_residues_kernel, dim3((unsigned)std::min<int64_t>(nm, 4096)), dim3(64), 0, ctx->stream, mv.view(), b_idx.as<int64_t>() + nm,
                       static_cast<uint8_t *>(dv_x) + x_bytes);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    mv.d_data = nullptr; mv.d_off = nullptr; mv.d_len = nullptr;
    clk.lap("std:   region Forward + copy to host");
    // the regions are independent (each ensemble starts from the seed): host threads take them round-robin, results are
    // appended in region order
    if (om->ensure_len_tables(*std::max_element(cfg.begin(), cfg.end())) != BATH_OK) return BATH_EFAIL;
    std::vector<std::vector<Env>> found((size_t)nm);
    auto work = [&](int64_t first, int64_t step) {
      std::vector<float> n2sc;
      std::vector<std::pair<int, int>> cl;
      for (int64_t e = first; e < nm; e += step) {
        const Env &en = mregs[(size_t)e];
        const int Lr = en.j - en.i + 1;
        if (region_trace_ensemble(om, cfg[(size_t)e], h_res + roff[(size_t)e], Lr, h_dp + mdpoff[(size_t)e], h_x + mxoff[(size_t)e], &n2sc, &cl, opt.seed) != BATH_OK) continue;
        for (const auto &c : cl) {
          float corr = 0.f;
          for (int pos = c.first; pos <= c.second; pos++) corr += n2sc[(size_t)pos];     // null2_is_done: p7_domaindef.c:1270-1272
          found[(size_t)e].push_back(Env{en.s, en.i + c.first - 1, en.i + c.second - 1, true, corr});
        }
      }
    };
    run_striped(nm, work, [&](int64_t e) { return mv.h_len[(size_t)e]; });
    for (int64_t e = 0; e < nm; e++) envs.insert(envs.end(), found[(size_t)e].begin(), found[(size_t)e].end());
    clk.lap("std:   ensembles (host threads)");
  }
  const int64_t ne = (int64_t)envs.size();
  if (ne == 0) return BATH_OK;

  // ---- envelopes: full Forward / Backward (unihit, L = Ld), then decoding, OA, traceback, null2
  bath_hip_seqs ev;
  ev.ctx = ctx; ev.n = ne; ev.is_part = true;
  ev.h_off.resize((size_t)ne); ev.h_len.resize((size_t)ne);
  std::vector<int64_t> exoff((size_t)ne + 1, 0), dpoff((size_t)ne + 1, 0);
  for (int64_t e = 0; e < ne; e++) {
    const Env &en = envs[(size_t)e];
    const int Ld = en.j - en.i + 1;
    ev.h_off[(size_t)e] = surv[(size_t)en.s].aa_off + en.i - 1; ev.h_len[(size_t)e] = Ld; ev.maxlen = std::max(ev.maxlen, Ld);
    exoff[(size_t)e + 1] = exoff[(size_t)e] + ((int64_t)Ld + 1) * 6;
    dpoff[(size_t)e + 1] = dpoff[(size_t)e] + ((int64_t)Ld + 1) * (M + 1) * 3;
  }
  if ((st = up_begin(view_bytes(ne) + (size_t)(4 * ne + 2) * 8 + 512)) != BATH_OK) return st;      // the view, the columns' offsets (3 ne + 1), the matrices' (ne + 1)
  if ((st = upload_view(ev, exoff, ne)) != BATH_OK) return st;
  const int64_t *d_exoff = b_idx.as<int64_t>() + ne;
  DevBuf &b_f = ctx->scratch[15], &b_b = ctx->scratch[16], &b_dpo = ctx->scratch[20], &b_px = ctx->scratch[18], &b_ox = ctx->scratch[19], &b_em = ctx->scratch[22], &b_out = ctx->scratch[21];
  BATH_HIP_TRY(ctx, b_f.reserve((size_t)dpoff[(size_t)ne] * 4 + 64 + kStdFillSlack)); BATH_HIP_TRY(ctx, b_b.reserve((size_t)dpoff[(size_t)ne] * 4 + 64 + kStdFillSlack));
  BATH_HIP_TRY(ctx, b_dpo.reserve((size_t)(ne + 1) * 8)); BATH_HIP_TRY(ctx, b_fx.reserve((size_t)exoff[(size_t)ne] * 4 + 64)); BATH_HIP_TRY(ctx, b_bx.reserve((size_t)exoff[(size_t)ne] * 4 + 64));
  BATH_HIP_TRY(ctx, b_px.reserve((size_t)exoff[(size_t)ne] / 6 * 5 * 4 + 64)); BATH_HIP_TRY(ctx, b_ox.reserve((size_t)exoff[(size_t)ne] / 6 * 5 * 4 + 64));
  BATH_HIP_TRY(ctx, b_em.reserve((size_t)ne * 2 * (M + 1) * 4 + 64)); BATH_HIP_TRY(ctx, b_out.reserve((size_t)ne * sizeof(StdEnvOut) + 64));
  // toff: the envelopes' slices of the column buffer; then, per envelope, where the codon of its first residue starts in the DNA
  // block and which way the strand runs (for the degenerate-codon test of the alignment score)
  std::vector<int64_t> toff((size_t)ne + 1 + 2 * (size_t)ne, 0);
  for (int64_t e = 0; e < ne; e++) {
    toff[(size_t)e + 1] = toff[(size_t)e] + ev.h_len[(size_t)e] + M + 2;
    const Env &en = envs[(size_t)e];
    const PipelineSurvivor &o = surv[(size_t)en.s];
    const int64_t p = (int64_t)o.start + 3 * (int64_t)(en.i - 1);           // strand position (1-based) of that codon's first nucleotide
    const int64_t woff = dna->h_off[(size_t)o.window], wn = dna->h_len[(size_t)o.window];
    toff[(size_t)ne + 1 + (size_t)e] = o.strand ? woff + wn - p : woff + p - 1;
    toff[(size_t)ne + 1 + (size_t)ne + (size_t)e] = o.strand ? -1 : 1;
  }
  DevBuf &b_tb = ctx->scratch[10], &b_toff = ctx->scratch[13];
  if ((size_t)toff[(size_t)ne] >= (size_t)INT32_MAX) { ctx->set_error("too many envelopes for the trace columns' 32-bit offsets"); return BATH_ERANGE; }
  // [columns: 1 B each, per envelope] [cursor] [the columns' posteriors, densely: 4 B per column that exists]
  const size_t tb_cols = ((size_t)toff[(size_t)ne] + 255) / 256 * 256;
  BATH_HIP_TRY(ctx, b_tb.reserve(tb_cols + 256 + (size_t)toff[(size_t)ne] * sizeof(float) + 64)); BATH_HIP_TRY(ctx, b_toff.reserve(toff.size() * 8));
  int *d_col_cursor = reinterpret_cast<int *>(b_tb.as<char>() + tb_cols);
  float *d_col_pp = reinterpret_cast<float *>(b_tb.as<char>() + tb_cols + 256);
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_col_cursor, 0, sizeof(int), ctx->stream));
  if ((st = up(b_toff.p, {{toff.data(), toff.size() * 8}})) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)ne * 8)); BATH_HIP_TRY(ctx, b_st.reserve((size_t)ne * 8));
  if ((st = up(b_dpo.p, {{dpoff.data(), (size_t)(ne + 1) * 8}})) != BATH_OK) return st;
  if ((st = launch_fwd_wave(ctx, om, ev.view(), nullptr, ne, b_sc.as<float>(), b_st.as<int32_t>(), nullptr, b_fx.as<float>(), d_exoff, b_f.as<float>(), b_dpo.as<int64_t>(), 1)) != BATH_OK) return st;
  if ((st = launch_bwd_wave(ctx, om, ev.view(), ne, b_fx.as<float>(), d_exoff, b_sc.as<float>() + ne, b_st.as<int32_t>() + ne, b_bx.as<float>(), b_b.as<float>(), b_dpo.as<int64_t>(), 1)) != BATH_OK) return st;
  // decoding + OA fill + null2: a wave per envelope; then the traceback, a lane per envelope.  BATH_HIP_STD_SERIAL=1 (tests) or a
  // model longer than 1024 nodes: everything in the lane-per-envelope kernel.
  int filled = 0;
  {
    const char *e = std::getenv("BATH_HIP_STD_SERIAL");
    const int c = (M + 63) / 64;
    const unsigned grid = (unsigned)std::min<int64_t>((ne + 3) / 4, (int64_t)ctx->prop.multiProcessorCount * 8);
#define BATH_FILL(CC) hipLaunchKernelGGL(std_envelope_fill_kernel<CC>, dim3(grid), dim3(256), 0, ctx->stream, ev.view(), M, om->d_tf, om->d_rf, b_f.as<float>(), b_b.as<float>(), \
                                         b_dpo.as<int64_t>(), b_fx.as<float>(), b_bx.as<float>(), d_exoff, b_px.as<float>(), b_ox.as<float>(), b_out.as<StdEnvOut>()); filled = 1
    // a block of four waves per envelope from 4 nodes per lane on (BATH_HIP_STD_FILL_MW=0: never, =1: for every model)
    const char *mwe = std::getenv("BATH_HIP_STD_FILL_MW");
    const bool mw = M <= 1024 && (mwe ? mwe[0] == '1' : c >= 4);
    if (!(e && e[0] == '1') && mw) {
      const unsigned mgrid = (unsigned)std::min<int64_t>(ne, (int64_t)ctx->prop.multiProcessorCount * 8);
#define BATH_FILL_MW(CC) hipLaunchKernelGGL(std_envelope_fill_mw_kernel<CC>, dim3(mgrid), dim3(256), 0, ctx->stream, ev.view(), M, om->d_tf, om->d_rf, b_f.as<float>(), b_b.as<float>(), \
                                            b_dpo.as<int64_t>(), b_fx.as<float>(), b_bx.as<float>(), d_exoff, b_px.as<float>(), b_ox.as<float>(), b_out.as<StdEnvOut>()); filled = 1
      const int c4 = (M + 255) / 256;
      if (c4 <= 1) { BATH_FILL_MW(1); } else if (c4 <= 2) { BATH_FILL_MW(2); } else { BATH_FILL_MW(4); }
#undef BATH_FILL_MW
    } else
    if (!(e && e[0] == '1')) {
      if (c <= 1) { BATH_FILL(1); } else if (c <= 2) { BATH_FILL(2); } else if (c <= 3) { BATH_FILL(3); } else if (c <= 4) { BATH_FILL(4); }
      else if (c <= 6) { BATH_FILL(6); } else if (c <= 8) { BATH_FILL(8); } else if (c <= 12) { BATH_FILL(12); } else if (c <= 16) { BATH_FILL(16); }
    }
#undef BATH_FILL
    BATH_HIP_TRY(ctx, hipGetLastError());
  }
  const bool lane_trace = [] { const char *e = std::getenv("BATH_HIP_STD_TRACE_LANE"); return e && e[0] == '1'; }();             // A/B and tests: the walk by one lane
  if (filled && !lane_trace)
    hipLaunchKernelGGL(std_trace_wave_kernel, dim3((unsigned)ne), dim3(64), 0, ctx->stream, ev.view(), M, om->d_tf, b_f.as<float>(), b_b.as<float>(), b_dpo.as<int64_t>(), d_exoff,
                       b_px.as<float>(), b_ox.as<float>(), b_out.as<StdEnvOut>(), om->d_cons, b_tb.as<uint8_t>(), b_toff.as<int64_t>(),
                       om->d_msc, om->d_tsc, dna->d_data, b_toff.as<int64_t>() + ne + 1, b_toff.as<int64_t>() + 2 * ne + 1, d_col_pp, d_col_cursor);
  else
  hipLaunchKernelGGL(std_envelope_kernel, dim3((unsigned)((ne * kStdTraceSpread + 63) / 64)), dim3(64), 0, ctx->stream, ev.view(), M, om->d_tf, om->d_rf, b_f.as<float>(), b_b.as<float>(), b_dpo.as<int64_t>(),
                     b_fx.as<float>(), b_bx.as<float>(), d_exoff, b_px.as<float>(), b_ox.as<float>(), b_em.as<float>(), b_out.as<StdEnvOut>(),
                     om->d_cons, b_tb.as<uint8_t>(), b_toff.as<int64_t>(), filled,
                     om->d_msc, om->d_tsc, dna->d_data, b_toff.as<int64_t>() + ne + 1, b_toff.as<int64_t>() + 2 * ne + 1,
                     nullptr, d_col_pp, d_col_cursor);
  BATH_HIP_TRY(ctx, hipGetLastError());
  std::vector<StdEnvOut> eo((size_t)ne);
  std::vector<float> envsc((size_t)ne);
  std::vector<uint8_t> tcols((size_t)toff[(size_t)ne]);
  std::vector<float> col_pp;
  {
    // what does not depend on the number of columns first (with the count), then the columns' posteriors
    const size_t o_tc = 64, o_eo = o_tc + ((tcols.size() + 63) & ~(size_t)63), o_sc = o_eo + (((size_t)ne * sizeof(StdEnvOut) + 63) & ~(size_t)63), o_end = o_sc + (size_t)ne * 4;
    if ((st = down_reserve(o_end)) != BATH_OK) return st;
    char *h = static_cast<char *>(ctx->stage[5].p);
    BATH_HIP_TRY(ctx, hipMemcpyAsync(h, d_col_cursor, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(h + o_tc, b_tb.p, tcols.size(), hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(h + o_eo, b_out.p, (size_t)ne * sizeof(StdEnvOut), hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(h + o_sc, b_sc.p, (size_t)ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int total = 0;
    std::memcpy(&total, h, sizeof(int));
    std::memcpy(tcols.data(), h + o_tc, tcols.size());
    std::memcpy(eo.data(), h + o_eo, (size_t)ne * sizeof(StdEnvOut));
    std::memcpy(envsc.data(), h + o_sc, (size_t)ne * 4);
    col_pp.resize((size_t)total);
    if (total > 0) {
      if ((st = down_reserve((size_t)total * sizeof(float))) != BATH_OK) return st;
      BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[5].p, d_col_pp, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      std::memcpy(col_pp.data(), ctx->stage[5].p, (size_t)total * sizeof(float));
    }
  }
  ev.d_data = nullptr; ev.d_off = nullptr; ev.d_len = nullptr;
  clk.lap("std:   envelope kernels");

  // ---- p7_pli_postDomainDef_BATH: coordinates on the sequence, score corrections, P-value
  const int ml = om->max_length;
  const RunningZ runZ(dna, opt.nres_before, opt.strands);                    // pli->Z, p7_pipeline.c:1246: the count at the hit's window and strand
  for (int64_t e = 0; e < ne; e++) {
    const StdEnvOut &t = eo[(size_t)e];
    if (!t.ok) continue;
    if (t.aliscore < 0.0f) continue;                                         // p7_domaindef.c:1286: "repetitive garbage", no domain
    const Env &en = envs[(size_t)e];
    const PipelineSurvivor &o = surv[(size_t)en.s];
    const float Zf = runZ.Z(o.window, o.strand, ml);
    const int seq_n = dna->h_len[(size_t)o.window];
    bath_fs_domain dm{};
    dm.window = o.window; dm.strand = o.strand; dm.fs_window = o.fs_window;
    dm.ihmm = t.k1; dm.jhmm = t.k2; dm.envsc = envsc[(size_t)e]; dm.oasc = t.oasc;
    dm.domcorrection = std::max(0.f, en.clustered ? en.n2corr : t.domcorrection);
    // alignment in nucleotides of the window (the ORF itself in the plain pipeline): p7_trace_fs_Convert puts a residue on its
    // codon's last nucleotide, offset by where the ORF starts in the window
    const int a1 = t.i1 + en.i - 1, a2 = t.i2 + en.i - 1, shift = o.start - o.win_start;
    int iali = shift + a1 * 3 - 2, jali = shift + a2 * 3, ienv = en.i, jenv = en.j;
    const int env_len = jenv - ienv + 1, ali_len = (jali - iali + 1) / 3;
    if (ali_len < 4) continue;                                               // p7_pipeline.c:1197
    if (!o.strand) {
      dm.ienv = 1 + o.start + ienv * 3 - 4; dm.jenv = 1 + o.start + jenv * 3 - 2;
      dm.iali = 1 + o.win_start + iali - 2; dm.jali = 1 + o.win_start + jali - 2;
    } else {                                                                 // the reference's orfsq->start is the top-strand coordinate
      const int ostart_ref = seq_n - o.start + 1;
      dm.ienv = 1 + ostart_ref - ienv * 3 + 2; dm.jenv = 1 + ostart_ref - jenv * 3;
      dm.jali = seq_n - (o.win_start + jali) + 2; dm.iali = seq_n - (o.win_start + iali) + 2;
    }
    float bitscore = dm.envsc;                                               // :1222-1226
    bitscore -= 2 * std::log(2. / (env_len + 2));
    bitscore += 2 * std::log(2. / (ml + 2));
    bitscore -= (env_len - ali_len) * std::log((double)((float)env_len / (float)(env_len + 2)));
    bitscore += (ml - ali_len) * std::log((double)((float)ml / (float)(ml + 2)));
    const float dom_bias = opt.do_null2 ? flogsum_host(0.0f, (float)(std::log(1. / 256.) + dm.domcorrection)) : 0.0f;   // :1230-1233
    const float p1 = (float)ml / (float)(ml + 1);
    const float nullsc = (float)((float)ml * std::log((double)p1) + std::log(1. - p1));      // p7_bg_NullOne at max_length
    dm.dombias = dom_bias;
    dm.bitscore = (float)((bitscore - (nullsc + dom_bias)) / kLn2);
    dm.pre_score = (float)(bitscore / kLn2);
    dm.lnP = (double)(float)exp_logsurv(dm.bitscore, om->evparam[BATH_FTAU], om->evparam[BATH_FLAMBDA]);
    dm.reported = opt.reportable(dm.lnP, Zf, dm.bitscore) ? 1 : 0;           // :1247-1248
    {                                                                        // columns were written last to first; every codon has 3 nucleotides
      const int nc = std::min<int>(t.ncol, (int)(toff[(size_t)e + 1] - toff[(size_t)e]));
      std::vector<uint16_t> cols((size_t)nc);
      for (int z = 0; z < nc; z++) cols[(size_t)z] = (uint16_t)(tcols[(size_t)toff[(size_t)e] + (size_t)(nc - 1 - z)] | (3u << 4) | (5u << 8));
      dm.ali_columns = nc; dm.n_stops = 0;
      dm.pid = nc > 0 ? ((float)t.exact / nc) * 100 : 0.f;
      dm.cigar_off = (int64_t)ctx->cigars.size();
      ctx->cigars += cigar_from_columns(cols.data(), nc);
      ctx->cigars.push_back('\0');
      // dom->tr after p7_trace_fs_Convert (p7_trace.c:405): a residue sits on its codon's last nucleotide, start = orf_start - window_start
      const bool have_pp = (size_t)t.pp_off + (size_t)nc <= col_pp.size() && nc == t.ncol;
      ctx->trace_push(cols.data(), have_pp ? col_pp.data() + t.pp_off : nullptr, nc, t.k1, shift + a1 * 3 - 2, o.win_start, o.start, 0, 0);
    }
    ctx->fs_domains.push_back(dm);
  }
  clk.lap("std:   post-processing (host)");
  return BATH_OK;
}

extern "C" int bath_hip_pipeline_hits(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const bath_pipeline_params *prm_in,
                                      double E_report, bath_pipeline_stats *stats, const bath_fs_domain **domains, int64_t *n_domains,
                                      int64_t *n_clustered_regions) {
  if (!ctx || !om || !dna || !prm_in || !domains || !n_domains) return BATH_EINVAL;
  *domains = nullptr; *n_domains = 0;
  int64_t nclust = 0;
  ctx->fs_domains.clear();
  ctx->cigars.clear();
  ctx->traces_clear();
  bath_pipeline_params prm = *prm_in;
  prm.fs_pipe = 0;
  bath_pipeline_stats st_local{};
  std::vector<PipelineSurvivor> surv;
  const uint8_t *d_pool = nullptr;
  StageClock clk;
  // the cascade's Forward parser leaves its special-state rows for the domain stage (BATH_HIP_KEEP_FWD=0: the domain stage runs the
  // parser again, as it did until round 5)
  const bool keep = [] { const char *e = std::getenv("BATH_HIP_KEEP_FWD"); return !(e && e[0] == '0'); }();
  ctx->keep_fwd_rows = keep;
  int st = pipeline_filters_survivors(ctx, om, dna, &prm, &st_local, &surv, &d_pool);
  ctx->keep_fwd_rows = false;
  if (st != BATH_OK) { ctx->fwd_rows_kept = nullptr; return st; }
  clk.lap("std: cascade + survivors to the host");
  if (stats) *stats = st_local;
  for (PipelineSurvivor &o : surv) o.win_start = o.start;                  // windowsq is the ORF's own stretch of DNA (p7_pipeline.c:1755)
  st = std_domains(ctx, om, dna, surv, d_pool, DomOpts(prm, E_report), &nclust);
  ctx->fwd_rows_kept = nullptr; ctx->fwd_rows_off = nullptr;
  if (st != BATH_OK) return st;
  if (n_clustered_regions) *n_clustered_regions = nclust;
  *domains = ctx->fs_domains.data(); *n_domains = (int64_t)ctx->fs_domains.size();
  return BATH_OK;
}

// =================================================================================================================
// Full-matrix entry points over a block of amino-acid targets: what the single-target prototypes p7_Forward / p7_Backward /
// p7_Decoding / p7_OptimalAccuracy / p7_Null2_ByExpectation of impl_hip/ bind (the domain stage above runs the same kernels).
// =================================================================================================================
namespace {
struct BlockOffsets { std::vector<int64_t> x, dp; };
BlockOffsets block_offsets(const bath_hip_seqs *sq, int M) {
  BlockOffsets o;
  o.x.assign((size_t)sq->n + 1, 0); o.dp.assign((size_t)sq->n + 1, 0);
  for (int64_t i = 0; i < sq->n; i++) {
    o.x[(size_t)i + 1] = o.x[(size_t)i] + ((int64_t)sq->h_len[(size_t)i] + 1) * 6;
    o.dp[(size_t)i + 1] = o.dp[(size_t)i] + ((int64_t)sq->h_len[(size_t)i] + 1) * (M + 1) * 3;
  }
  return o;
}
}  // namespace

extern "C" int bath_hip_forward_full(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const int32_t *cfg_len, int unihit,
                                     float *sc, int32_t *status, float *dp, float *xmx) {
  if (!ctx || !om || !sq || !sc) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  const int M = om->M;
  int maxcfg = sq->maxlen;
  if (cfg_len) for (int64_t i = 0; i < n; i++) maxcfg = std::max(maxcfg, (int)cfg_len[i]);
  int st = om->ensure_len_tables(maxcfg + 1);
  if (st != BATH_OK) return st;
  const BlockOffsets o = block_offsets(sq, M);
  DevBuf &b_f = ctx->scratch[15], &b_off = ctx->scratch[20], &b_fx = ctx->scratch[18], &b_sc = ctx->scratch[21], &b_cfg = ctx->scratch[5];
  BATH_HIP_TRY(ctx, b_f.reserve((size_t)o.dp[(size_t)n] * 4 + 64)); BATH_HIP_TRY(ctx, b_fx.reserve((size_t)o.x[(size_t)n] * 4 + 64));
  BATH_HIP_TRY(ctx, b_off.reserve((size_t)(n + 1) * 16)); BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 8)); BATH_HIP_TRY(ctx, b_cfg.reserve((size_t)n * 4 + 64));
  int64_t *d_xo = b_off.as<int64_t>(), *d_dpo = d_xo + (n + 1);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_xo, o.x.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_dpo, o.dp.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (cfg_len) BATH_HIP_TRY(ctx, hipMemcpyAsync(b_cfg.p, cfg_len, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  float *d_sc = b_sc.as<float>();
  int32_t *d_st = reinterpret_cast<int32_t *>(d_sc + n);
  if ((st = launch_fwd_wave(ctx, om, sq->view(), nullptr, n, d_sc, d_st, nullptr, b_fx.as<float>(), d_xo, b_f.as<float>(), d_dpo, unihit ? 1 : 0,
                            cfg_len ? b_cfg.as<int32_t>() : nullptr)) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, d_sc, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (status) BATH_HIP_TRY(ctx, hipMemcpyAsync(status, d_st, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (dp) BATH_HIP_TRY(ctx, hipMemcpyAsync(dp, b_f.p, (size_t)o.dp[(size_t)n] * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (xmx) BATH_HIP_TRY(ctx, hipMemcpyAsync(xmx, b_fx.p, (size_t)o.x[(size_t)n] * 4, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}

extern "C" int bath_hip_std_envelopes(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, bath_std_result *res,
                                      float *pp, float *oa, float *ppx, float *oax) {
  if (!ctx || !om || !sq || !res) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  const int M = om->M;
  int st = om->ensure_len_tables(sq->maxlen + 1);
  if (st != BATH_OK) return st;
  const BlockOffsets o = block_offsets(sq, M);
  std::vector<int64_t> toff((size_t)n + 1, 0);
  for (int64_t e = 0; e < n; e++) toff[(size_t)e + 1] = toff[(size_t)e] + sq->h_len[(size_t)e] + M + 2;
  DevBuf &b_f = ctx->scratch[15], &b_b = ctx->scratch[16], &b_off = ctx->scratch[20], &b_fx = ctx->scratch[6], &b_bx = ctx->scratch[7], &b_px = ctx->scratch[18],
         &b_ox = ctx->scratch[19], &b_em = ctx->scratch[22], &b_out = ctx->scratch[21], &b_tb = ctx->scratch[10], &b_sc = ctx->scratch[2], &b_n2 = ctx->scratch[23];
  const size_t nx = (size_t)o.x[(size_t)n], ndp = (size_t)o.dp[(size_t)n];
  BATH_HIP_TRY(ctx, b_f.reserve(ndp * 4 + 64 + kStdFillSlack)); BATH_HIP_TRY(ctx, b_b.reserve(ndp * 4 + 64 + kStdFillSlack));
  BATH_HIP_TRY(ctx, b_fx.reserve(nx * 4 + 64)); BATH_HIP_TRY(ctx, b_bx.reserve(nx * 4 + 64));
  BATH_HIP_TRY(ctx, b_px.reserve(nx / 6 * 5 * 4 + 64)); BATH_HIP_TRY(ctx, b_ox.reserve(nx / 6 * 5 * 4 + 64));
  BATH_HIP_TRY(ctx, b_em.reserve((size_t)n * 2 * (M + 1) * 4 + 64)); BATH_HIP_TRY(ctx, b_out.reserve((size_t)n * sizeof(StdEnvOut) + 64));
  BATH_HIP_TRY(ctx, b_tb.reserve((size_t)toff[(size_t)n] + 64)); BATH_HIP_TRY(ctx, b_off.reserve((size_t)(n + 1) * 24));
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 16)); BATH_HIP_TRY(ctx, b_n2.reserve((size_t)n * kKp * 4 + 64));
  int64_t *d_xo = b_off.as<int64_t>(), *d_dpo = d_xo + (n + 1), *d_to = d_dpo + (n + 1);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_xo, o.x.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_dpo, o.dp.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_to, toff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  float *d_sc = b_sc.as<float>();
  int32_t *d_st = reinterpret_cast<int32_t *>(d_sc + 2 * n);
  if ((st = launch_fwd_wave(ctx, om, sq->view(), nullptr, n, d_sc, d_st, nullptr, b_fx.as<float>(), d_xo, b_f.as<float>(), d_dpo, 1)) != BATH_OK) return st;
  if ((st = launch_bwd_wave(ctx, om, sq->view(), n, b_fx.as<float>(), d_xo, d_sc + n, d_st + n, b_bx.as<float>(), b_b.as<float>(), d_dpo, 1)) != BATH_OK) return st;
  // p7_Decoding, p7_OptimalAccuracy, p7_Null2_ByExpectation (and the traceback, unused here), a lane per envelope: posteriors
  // overwrite Backward, the OA matrix overwrites Forward, as in the reference
  hipLaunchKernelGGL(std_envelope_kernel, dim3((unsigned)((n * kStdTraceSpread + 63) / 64)), dim3(64), 0, ctx->stream, sq->view(), M, om->d_tf, om->d_rf, b_f.as<float>(), b_b.as<float>(), d_dpo,
                     b_fx.as<float>(), b_bx.as<float>(), d_xo, b_px.as<float>(), b_ox.as<float>(), b_em.as<float>(), b_out.as<StdEnvOut>(),
                     (const uint8_t *)nullptr, b_tb.as<uint8_t>(), d_to, 0, (const float *)nullptr, (const float *)nullptr, (const uint8_t *)nullptr,
                     (const int64_t *)nullptr, (const int64_t *)nullptr, b_n2.as<float>());
  BATH_HIP_TRY(ctx, hipGetLastError());
  std::vector<StdEnvOut> eo((size_t)n);
  std::vector<float> h_sc((size_t)n * 2), h_n2((size_t)n * kKp);
  std::vector<int32_t> h_st((size_t)n * 2);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(eo.data(), b_out.p, (size_t)n * sizeof(StdEnvOut), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h_sc.data(), d_sc, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h_st.data(), d_st, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h_n2.data(), b_n2.p, (size_t)n * kKp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (pp) BATH_HIP_TRY(ctx, hipMemcpyAsync(pp, b_b.p, ndp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (oa) BATH_HIP_TRY(ctx, hipMemcpyAsync(oa, b_f.p, ndp * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (ppx) BATH_HIP_TRY(ctx, hipMemcpyAsync(ppx, b_px.p, nx / 6 * 5 * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (oax) BATH_HIP_TRY(ctx, hipMemcpyAsync(oax, b_ox.p, nx / 6 * 5 * 4, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t e = 0; e < n; e++) {
    bath_std_result &r = res[(size_t)e];
    r.fwdsc = h_sc[(size_t)e]; r.bcksc = h_sc[(size_t)(n + e)]; r.fwd_status = h_st[(size_t)e]; r.bck_status = h_st[(size_t)(n + e)];
    r.ok = eo[(size_t)e].ok; r.oasc = eo[(size_t)e].oasc;
    std::memcpy(r.null2, &h_n2[(size_t)e * kKp], sizeof(float) * kKp);
  }
  return BATH_OK;
}

extern "C" const char *bath_hip_domain_cigars(const bath_hip_ctx *ctx) { return ctx ? ctx->cigars.c_str() : nullptr; }

// P7_DOMAIN.tr of every domain of the last pipeline call: the trace kernels' columns expanded into the reference's arrays
extern "C" int bath_hip_domain_traces(bath_hip_ctx *ctx, const bath_domain_trace **tr, int64_t *n_traces,
                                      const int8_t **st, const int32_t **k, const int32_t **i, const int8_t **c, const float **pp) {
  if (!ctx || !tr || !n_traces) return BATH_EINVAL;
  if (ctx->tr_recs.size() != ctx->fs_domains.size()) { ctx->set_error("no traces for the domains of the last call"); return BATH_EINVAL; }
  if (!ctx->tr_valid) {
    const size_t ncols = ctx->tr_codes.size();
    ctx->tr_out.resize(ctx->tr_recs.size());
    ctx->tr_st.resize(ncols); ctx->tr_c.resize(ncols); ctx->tr_k.resize(ncols); ctx->tr_i.resize(ncols);
    for (size_t d = 0; d < ctx->tr_recs.size(); d++) {
      const bath_hip_ctx::TraceRec &t = ctx->tr_recs[d];
      ctx->tr_out[d] = bath_domain_trace{t.col_off, t.ncol, t.win_start, t.orf_start, t.frameshift};
      int kk = t.k1 - 1, ii = t.i_first - 1;
      for (int z = 0; z < t.ncol; z++) {
        const size_t q = (size_t)t.col_off + (size_t)z;
        const int s = ctx->tr_codes[q] & 0xf, cl = (ctx->tr_codes[q] >> 4) & 0xf;
        if (s == 3)      { kk++; ii += cl; ctx->tr_st[q] = BATH_T_M; ctx->tr_k[q] = kk; ctx->tr_i[q] = ii; ctx->tr_c[q] = (int8_t)cl; }
        else if (s == 5) { ii += 3;        ctx->tr_st[q] = BATH_T_I; ctx->tr_k[q] = kk; ctx->tr_i[q] = ii; ctx->tr_c[q] = 0; }
        else             { kk++;           ctx->tr_st[q] = BATH_T_D; ctx->tr_k[q] = kk; ctx->tr_i[q] = t.d_i; ctx->tr_c[q] = 0; }
      }
    }
    ctx->tr_valid = true;
  }
  *tr = ctx->tr_out.data(); *n_traces = (int64_t)ctx->tr_out.size();
  if (st) *st = ctx->tr_st.data();
  if (k) *k = ctx->tr_k.data();
  if (i) *i = ctx->tr_i.data();
  if (c) *c = ctx->tr_c.data();
  if (pp) *pp = ctx->tr_pp.data();
  return BATH_OK;
}
