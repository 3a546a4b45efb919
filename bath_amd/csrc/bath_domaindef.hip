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
// block: the 3-codon parsers (fs3_fwd_kernel, fs_bwd_kernel<.,3,.>) and the five envelope kernels of bath_frameshift.hip.
// What remains is O(L) per window (posterior sums over the special-state rows, the region heuristics) or a serial walk
// of at most L+M steps per envelope (the optimal-accuracy traceback): that is done here on the host, on the rows and
// matrices the kernels hand back.
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

enum { XE = 0, XN, XJ, XB, XC };                       // special-state columns of the generic matrices (p7G_E ... p7G_C)
enum { tMM = 0, tIM, tDM, tBM, tMD, tDD, tMI, tII };    // hmmer.h:221
enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC, sT };    // trace states

struct Trace { std::vector<int8_t> st; std::vector<int32_t> k, i, c; };

const double kLn2 = 0.69314718055994529;

double exp_logsurv(double x, double mu, double lambda) { return x < mu ? 0.0 : -lambda * (x - mu); }

// btot / etot / mocc of a window from the parsers' rows (log space), generic_decoding_frameshift.c:204-290
void domain_decoding(const float *fx, const float *bx, int L, float loopN, float loopC, float loopJ, std::vector<float> &btot, std::vector<float> &etot,
                     std::vector<float> &mocc) {
  btot.assign((size_t)L + 1, 0.f); etot.assign((size_t)L + 1, 0.f); mocc.assign((size_t)L + 1, 0.f);
  auto F = [&](int i, int s) { return fx[(size_t)i * 5 + s]; };
  auto B = [&](int i, int s) { return bx[(size_t)i * 5 + s]; };
  const float Z = flogsum_host(B(0, XN), flogsum_host(B(1, XN), B(2, XN)));
  for (int i = 3; i <= L; i++) {
    btot[(size_t)i] = btot[(size_t)i - 3] + expf(F(i - 3, XB) + B(i - 3, XB) - Z);
    etot[(size_t)i] = etot[(size_t)i - 3] + expf(F(i, XE) + B(i, XE) - Z);
  }
  // i is emitted by N, C or J in any of the three codon phases that cover it
  auto emitted = [&](int s, float loop, int a, int b) { return expf(F(a, s) + B(b, s) + loop - Z); };
  for (int i = 3; i < L - 1; i++) {
    float p = 0.0f;
    for (int s : {XN, XC, XJ}) {
      const float loop = s == XN ? loopN : (s == XC ? loopC : loopJ);
      p += emitted(s, loop, i - 3, i); p += emitted(s, loop, i - 2, i + 1); p += emitted(s, loop, i - 1, i + 2);
    }
    mocc[(size_t)i] = (float)(1. - p);
  }
  if (L >= 4) {
    float p = 0.0f;
    for (int s : {XN, XC, XJ}) { const float loop = s == XN ? loopN : (s == XC ? loopC : loopJ); p += emitted(s, loop, L - 4, L - 1); p += emitted(s, loop, L - 3, L); }
    mocc[(size_t)L - 1] = (float)(1. - p);
    p = 0.0f;
    for (int s : {XN, XC, XJ}) { const float loop = s == XN ? loopN : (s == XC ? loopC : loopJ); p += emitted(s, loop, L - 3, L); }
    mocc[(size_t)L] = (float)(1. - p);
  }
}

// more than one domain expected somewhere inside i..j? (p7_domaindef.c:684-714)
bool multidomain(const std::vector<float> &btot, const std::vector<float> &etot, int i, int j, float rt3) {
  float best = -1.0f;
  for (int ph = 0; ph < 3; ph++) {
    const int f = (j - i + 1 - ph) % 3;
    for (int z = i + 2 + ph; z <= j - f; z += 3)
      best = std::max(best, std::min(etot[(size_t)z] - etot[(size_t)(i - 1 + ph)], btot[(size_t)(j - f)] - btot[(size_t)z - 3]));
  }
  return best >= rt3;
}

// regions of a window: p7_domaindef.c:328-392 (the start / end of a frameshift-aware domain must show in all three frames)
void find_regions(const std::vector<float> &btot, const std::vector<float> &etot, const std::vector<float> &mocc, int L, std::vector<std::pair<int, int>> &single,
                  int *n_multi) {
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;      // p7_domaindef.c:80-82
  bool triggered = false;
  int d = 0;
  for (int j = 1; j < L; j++) {
    if (!triggered) { if (mocc[(size_t)j] >= rt1) triggered = true; d = j; continue; }
    bool found = false;
    while (d > 1 && !found) {
      int run = 0;
      d--;
      while (run < 3 && d > 3 && mocc[(size_t)d] - (btot[(size_t)d] - btot[(size_t)d - 3]) < rt2) { d--; run++; }
      if (run == 3) found = true;
    }
    const int i = std::max(1, d - 3);
    d = j + 1;
    found = false;
    while (d < L && !found) {
      int run = 0;
      d++;
      while (run < 3 && d < L && mocc[(size_t)d] - (etot[(size_t)d] - etot[(size_t)d - 3]) < rt2) { d++; run++; }
      if (run == 3) found = true;
    }
    j = std::min(L, d + 3);
    if (j - i + 1 >= 12) {
      if (multidomain(btot, etot, i, j, rt3)) (*n_multi)++;
      else single.emplace_back(i, j);
    }
    triggered = false;
  }
}

int first_max(const float *v, int n) { int b = 0; for (int q = 1; q < n; q++) if (v[q] > v[b]) b = q; return b; }    // esl_vec_FArgMax

// p7_GOATrace_Frameshift on one envelope.  pp: posteriors [(L+1)][(M+1)][8] {D,I,M_c0..M_c5}; ppx, oax: special rows;
// oa: OA matrix [(L+1)][(M+1)][3] {D,I,M}.  Unihit: E->J impossible.
bool oa_trace(const float *tsc, int M, int L, const float *pp, const float *ppx, const float *oa, const float *oax, Trace &tr) {
  auto delta = [&](int s, int k) { return (k >= 0 && k < M && tsc[(size_t)k * 8 + s] == -INFINITY) ? FLT_MIN : ((k < 0 || k >= M) ? FLT_MIN : 1.0f); };
  auto OM = [&](int i, int k) { return oa[((size_t)i * (M + 1) + k) * 3 + 2]; };
  auto OI = [&](int i, int k) { return oa[((size_t)i * (M + 1) + k) * 3 + 1]; };
  auto OD = [&](int i, int k) { return oa[((size_t)i * (M + 1) + k) * 3 + 0]; };
  auto OX = [&](int i, int s) { return oax[(size_t)i * 5 + s]; };
  auto PX = [&](int i, int s) { return ppx[(size_t)i * 5 + s]; };
  auto PP = [&](int i, int k, int cell) { return pp[((size_t)i * (M + 1) + k) * 8 + cell]; };
  tr.st.clear(); tr.k.clear(); tr.i.clear(); tr.c.clear();
  auto push = [&](int st, int k, int i, int c) { tr.st.push_back((int8_t)st); tr.k.push_back(k); tr.i.push_back(i); tr.c.push_back(c); };
  int i = L, k = 0, c = 0;
  push(sT, k, i, c); push(sC, k, i, c);
  int prev = sC;
  const float tCL = 1.0f, tEM = 1.0f, tJL = 1.0f, tEL = FLT_MIN, tNM = 1.0f, tJM = 1.0f;       // unihit length model: only E->J is impossible
  while (prev != sS) {
    int cur = -1;
    float path[4];
    switch (prev) {
    case sM: {
      static const int state[4] = {sM, sI, sD, sB};
      path[0] = delta(tMM, k - 1) * OM(i, k - 1); path[1] = delta(tIM, k - 1) * OI(i, k - 1);
      path[2] = delta(tDM, k - 1) * OD(i, k - 1); path[3] = delta(tBM, k - 1) * OX(i, XB);
      cur = state[first_max(path, 4)]; k--; break; }
    case sD:
      path[0] = delta(tMD, k - 1) * OM(i, k - 1); path[1] = delta(tDD, k - 1) * OD(i, k - 1);
      cur = path[0] >= path[1] ? sM : sD; k--; break;
    case sI:
      path[0] = delta(tMI, k) * OM(i - 3, k); path[1] = delta(tII, k) * OI(i - 3, k);
      cur = path[0] >= path[1] ? sM : sI; i -= 3; break;
    case sN: cur = (i == 0) ? sS : sN; break;
    case sC: {
      static const int state[4] = {sC, sC, sC, sE};
      if (i < 4) { cur = sE; break; }
      path[0] = tCL * (OX(i - 3, XC) + PX(i, XC));
      path[1] = (i < L) ? tCL * (OX(i - 2, XC) + PX(i + 1, XC)) : FLT_MIN;
      path[2] = (i < L - 1) ? tCL * (OX(i - 1, XC) + PX(i + 2, XC)) : FLT_MIN;
      path[3] = tEM * OX(i, XE);
      cur = state[first_max(path, 4)]; break; }
    case sJ:
      if (i <= 5) { cur = sE; break; }
      path[0] = tJL * (OX(i, XJ) + PX(i, XJ)); path[1] = tEL * OX(i, XE);
      cur = first_max(path, 2) == 0 ? sJ : sE; break;
    case sE: {
      float mx = -INFINITY;
      int smax = -1, kmax = -1;
      for (int q = 1; q <= M; q++) {
        if (OM(i, q) > mx) { mx = OM(i, q); smax = sM; kmax = q; }
        if (OD(i, q) > mx) { mx = OD(i, q); smax = sD; kmax = q; }
      }
      k = kmax; cur = smax; break; }
    case sB: cur = (tNM * OX(i, XN) > tJM * OX(i, XJ)) ? sN : sJ; break;
    default: return false;
    }
    if (cur < 0 || k < 0 || i < 0) return false;
    if (cur == sM) {
      float cod[5];
      for (int q = 0; q < 5; q++) cod[q] = PP(i, k, 3 + q);
      c = first_max(cod, 5) + 1;
    } else c = 0;
    push(cur, k, i, c);
    if ((cur == sN || cur == sC || cur == sJ) && cur == prev) i--;
    prev = cur;
    i -= c;
    if (tr.st.size() > (size_t)(4 * (L + M) + 64)) return false;
  }
  std::reverse(tr.st.begin(), tr.st.end()); std::reverse(tr.k.begin(), tr.k.end());
  std::reverse(tr.i.begin(), tr.i.end()); std::reverse(tr.c.begin(), tr.c.end());
  return true;
}

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
  const int M = h5.M;

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
  std::vector<int64_t> xoff((size_t)nsel + 1, 0);
  for (int i = 0; i < nsel; i++) xoff[(size_t)i + 1] = xoff[(size_t)i] + ((int64_t)regs[(size_t)i].len + 1) * 5;
  std::vector<float> fx((size_t)xoff[(size_t)nsel]), bx((size_t)xoff[(size_t)nsel]), fsc((size_t)nsel), bsc((size_t)nsel);
  {
    bath_hip_seqs view;
    if ((st = fs_gather_view(ctx, dna, regs, tt.comp, &view, nullptr)) != BATH_OK) return st;
    st = bath_hip_fs3_forward_parser(ctx, om_fs3, &view, BATH_LOGSUM_TABLE, fsc.data(), fx.data(), xoff.data());
    if (st == BATH_OK) st = bath_hip_fs3_backward_parser(ctx, om_fs3, &view, BATH_LOGSUM_TABLE, bsc.data(), bx.data(), xoff.data());
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
    if (st != BATH_OK) return st;
  }

  // ---- posterior sums over the special states -> regions -> envelopes
  const float pmove = (2.0f + 1.0f) / (100.0f + 2.0f + 1.0f);                 // p7_fs_ReconfigLength(L = 100, nj = 1), modelconfig.c:767-770
  const float loop = (float)std::log((double)(1.0f - pmove));
  struct Env { int sel, i, j; };
  std::vector<Env> envs;
  int n_multi = 0;
  for (int q = 0; q < nsel; q++) {
    if (!(bsc[(size_t)q] > -INFINITY)) continue;                              // Backward underflow: the reference skips the window (:1471)
    const int L = regs[(size_t)q].len;
    std::vector<float> btot, etot, mocc;
    domain_decoding(&fx[(size_t)xoff[(size_t)q]], &bx[(size_t)xoff[(size_t)q]], L, loop, loop, loop, btot, etot, mocc);
    std::vector<std::pair<int, int>> single;
    find_regions(btot, etot, mocc, L, single, &n_multi);
    for (auto &r : single) if (r.second - r.first + 1 >= 15) envs.push_back(Env{q, r.first, r.second});      // rescore_isolated_domain: Ld < 15 -> nothing
  }
  if (n_skipped_regions) *n_skipped_regions = n_multi;
  if (envs.empty()) return BATH_OK;

  // ---- envelopes: Forward, Backward, decoding, optimal accuracy, null2 on the GPU (unihit, length Ld/3)
  const int nenv = (int)envs.size();
  std::vector<FsWinDev> eregs((size_t)nenv);
  std::vector<int64_t> poff((size_t)nenv + 1, 0), ooff((size_t)nenv + 1, 0), exoff((size_t)nenv + 1, 0);
  for (int e = 0; e < nenv; e++) {
    const FsWinDev &wr = regs[(size_t)envs[(size_t)e].sel];
    FsWinDev d = wr;
    d.start = wr.start + envs[(size_t)e].i - 1; d.len = envs[(size_t)e].j - envs[(size_t)e].i + 1;
    eregs[(size_t)e] = d;
    const int64_t rows = (int64_t)d.len + 1;
    poff[(size_t)e + 1] = poff[(size_t)e] + rows * (M + 1) * 8; ooff[(size_t)e + 1] = ooff[(size_t)e] + rows * (M + 1) * 3; exoff[(size_t)e + 1] = exoff[(size_t)e] + rows * 5;
  }
  std::vector<bath_fs5_result> res((size_t)nenv);
  std::vector<float> pp((size_t)poff[(size_t)nenv]), oa((size_t)ooff[(size_t)nenv]), ppx((size_t)exoff[(size_t)nenv]), oax((size_t)exoff[(size_t)nenv]);
  std::vector<uint8_t> pool;
  {
    bath_hip_seqs view;
    if ((st = fs_gather_view(ctx, dna, eregs, tt.comp, &view, nullptr)) != BATH_OK) return st;
    st = fs5_envelopes_ex(ctx, om_fs5, &view, BATH_LOGSUM_TABLE, 0, res.data(), pp.data(), oa.data(), ppx.data(), oax.data());
    pool.resize((size_t)view.total_aligned + 16);
    if (st == BATH_OK && hipMemcpy(pool.data(), view.d_data, (size_t)view.total_aligned, hipMemcpyDeviceToHost) != hipSuccess) st = BATH_EFAIL;
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;
    if (st != BATH_OK) return st;
  }

  // ---- traceback, null2 along the trace, the hit's scores
  const int ml = h5.max_length;
  const float Zf = (float)st_local.nres / (float)ml;                          // pli->Z, p7_domaindef.c:1033 (here: residues of the block)
  Trace tr;
  for (int e = 0; e < nenv; e++) {
    const Env &en = envs[(size_t)e];
    const bath_fs_window &win = fw[sel[(size_t)en.sel]];
    const int Ld = eregs[(size_t)e].len;
    const uint8_t *dsq = pool.data() + eregs[(size_t)e].dst_off - 1;          // dsq[1..Ld]
    const float envsc = res[(size_t)e].fwdsc;
    if (!(envsc > -INFINITY) || !(res[(size_t)e].bcksc > -INFINITY)) continue;
    {
      const float p1 = (float)(Ld / 3) / (float)(Ld / 3 + 1);
      const float per_frame = (float)((float)(Ld / 3) * std::log((double)p1) + std::log(1. - p1));
      const float nullsc = (float)(per_frame + std::log(3.0));
      const float seqscore = (float)((envsc - nullsc) / kLn2);
      if (exp_surv(seqscore, h5.evparam[7], h5.evparam[BATH_FLAMBDA]) * (double)Zf > E_report) continue;    // :1034 (FTAUFS5)
    }
    if (!oa_trace(h5.tsc, M, Ld, &pp[(size_t)poff[(size_t)e]], &ppx[(size_t)exoff[(size_t)e]], &oa[(size_t)ooff[(size_t)e]], &oax[(size_t)exoff[(size_t)e]], tr)) continue;
    // null2 score of every nucleotide by the state (and codon) that emits it in the trace, :1086-1142
    const float *null2 = res[(size_t)e].null2;
    std::vector<float> n2((size_t)Ld + 2, 0.f);
    {
      int t = -1, u = -1, v = -1, w = -1, x = -1, pos = 1;
      size_t z = 0;
      auto amino = [&](int k, int ci, int cap) { return (int)h5.codons[(size_t)k * h5.maxcodons + (size_t)std::min(ci, cap)]; };
      while (pos <= Ld && z < tr.st.size()) {
        x = dsq[pos] < 4 ? (int)dsq[pos] : 1367;
        const int s = tr.st[z];
        if (s == sN || s == sC || s == sJ) { n2[(size_t)pos] = 0.f; if (tr.i[z] == pos && pos > 2) pos++; z++; }
        else if (s == sM) {
          if (tr.i[z] == pos) {
            int ci = 0, cap = 1364;
            switch (tr.c[z]) {
            case 1: ci = x * 341; cap = 1366; break;
            case 2: ci = x * 341 + w * 85 + 1; cap = 1365; break;
            case 3: ci = x * 341 + w * 85 + v * 21 + 2; cap = 1364; break;
            case 4: ci = x * 341 + w * 85 + v * 21 + u * 5 + 3; cap = 1365; break;
            default: ci = x * 341 + w * 85 + v * 21 + u * 5 + t + 4; cap = 1366; break;
            }
            const float sc = logf(null2[amino(tr.k[z], ci, cap)]);
            n2[(size_t)pos] = (sc == -INFINITY) ? 0.f : sc;
            z++;
          } else n2[(size_t)pos] = 0.f;
          pos++;
        } else if (s == sI) {
          if (tr.i[z] == pos) {
            const float sc = logf(null2[amino(tr.k[z], x * 341 + w * 85 + v * 21 + 2, 1364)]);
            n2[(size_t)pos] = (sc == -INFINITY) ? 0.f : sc;
            z++;
          } else n2[(size_t)pos] = 0.f;
          pos++;
        } else z++;
        t = u; u = v; v = w; w = x;
      }
    }
    float domcorrection = 0.f;
    for (int pos = 1; pos <= Ld; pos++) domcorrection += n2[(size_t)pos];
    size_t z1 = 0, z2 = tr.st.size();
    while (z1 < tr.st.size() && tr.st[z1] != sM) z1++;
    while (z2 > 0 && tr.st[z2 - 1] != sM) z2--;
    if (z1 >= tr.st.size() || z2 == 0) continue;
    z2--;
    bath_fs_domain dm{};
    dm.window = win.window; dm.strand = win.strand; dm.fs_window = sel[(size_t)en.sel];
    // window coordinates first (:1148-1163), then the sequence's (p7_pipeline.c:1035-1049); envelope i..j in the window
    const int wi = en.i, wj = en.j;
    int iali = wi - 1 + tr.i[z1] - (tr.c[z1] - 1), jali = wi - 1 + tr.i[z2], ienv = wi, jenv = wj;
    const int ali_len = jali - iali + 1, env_len = jenv - ienv + 1;
    dm.ihmm = tr.k[z1]; dm.jhmm = tr.k[z2];
    dm.envsc = envsc; dm.oasc = res[(size_t)e].oasc; dm.domcorrection = std::max(0.f, domcorrection);
    for (size_t z = 0; z < tr.st.size(); z++) if (tr.st[z] == sM && tr.c[z] != 3) dm.n_shifted_codons++;
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
