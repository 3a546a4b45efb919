// bath_ensemble.hip -- multi-domain regions: envelopes from a clustered ensemble of stochastic tracebacks (host code).
//
// Reference: region_trace_ensemble (src/p7_domaindef.c:766-850) with p7_StochasticTrace (src/impl_sse/stotrace.c:71-300),
// p7_trace_Index (src/p7_trace.c:2592), p7_Null2_ByTrace (src/impl_sse/null2.c:131-215), p7_spensemble_Add / _Cluster and
// link_spsamples (src/p7_spensemble.c:189-217, 300-440), parameters of p7_domaindef_Create_BATH (src/p7_domaindef.c:83-97).
//
// Where it runs.  The region's Forward matrix is computed on the GPU by fwd_wave_kernel (multihit, full matrix) and copied
// to the host; the 200 tracebacks are a strictly serial consumer of one random-number stream (every choice draws the next
// number), about 200 x (Lr + M) dependent steps per region, and regions of this kind are a few per thousand ORFs that pass
// the Forward filter -- so the walk, the null2 bookkeeping and the clustering stay on the host.
//
// easel is not part of the reference tree (un-pinned submodule), so its pieces are restated from their published
// algorithms: the "fast" generator of esl_randomness_CreateFast (x <- 69069 x + 1 on a Jenkins-mixed seed; u = x / 2^32;
// re-seeded with 42 before every region, p7_pipeline.c:135-143), esl_rnd_FChoose, esl_vec_FNorm, and
// esl_cluster_SingleLinkage (connected components of the link relation).  Parity of the sampled ensemble with a real
// bathsearch run is therefore unpinned; tests compare this code with the oracle's independent restatement.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "bath_common.hpp"
#include "bath_launch.hpp"

namespace {

enum { XE = 0, XN, XJ, XB, XC, XS };
enum { cM = 0, cD = 1, cI = 2 };
enum { MM = 0, IM, DM, BM, MD, DD, MI, II };
enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC, sT };

struct FastRng {
  uint32_t x;
  explicit FastRng(uint32_t seed) {
    uint32_t a = seed, b = 87654321u, c = 12345678u;
    a -= b; a -= c; a ^= (c >> 13);  b -= c; b -= a; b ^= (a << 8);   c -= a; c -= b; c ^= (b >> 13);
    a -= b; a -= c; a ^= (c >> 12);  b -= c; b -= a; b ^= (a << 16);  c -= a; c -= b; c ^= (b >> 5);
    a -= b; a -= c; a ^= (c >> 3);   b -= c; b -= a; b ^= (a << 10);  c -= a; c -= b; c ^= (b >> 15);
    x = c ? c : 42u;
  }
  double next() { x = x * 69069u + 1u; return (double)x / 4294967296.0; }
};

// pli->r and ddef->do_reseeding (p7_pipeline.c:135-143; p7_domaindef.c:781, :904).  With a seed (--seed, 42 by default) every region's
// ensemble starts from it.  Seed 0 means "an arbitrary one-time seed, no reseeding": the reference takes the time of day and lets the
// generator run on from region to region of a worker; here every region draws a stream of its own from one per-process arbitrary
// value -- results are not reproducible from run to run in either.
uint32_t region_seed(uint32_t seed) {
  if (seed != 0) return seed;
  static const uint32_t once = (uint32_t)std::chrono::steady_clock::now().time_since_epoch().count() | 1u;
  static std::atomic<uint32_t> counter{0};
  return once + 0x9e3779b9u * (counter.fetch_add(1) + 1u);
}

template <int N>
int choose(FastRng &rng, float (&p)[N]) {
  float sum = 0.f, comp = 0.f;                            // esl_vec_FNorm
  for (int q = 0; q < N; q++) { const float y = p[q] - comp, t = sum + y; comp = (t - sum) - y; sum = t; }
  for (int q = 0; q < N; q++) p[q] = (sum != 0.0f) ? p[q] / sum : 1.0f / (float)N;
  for (;;) {                                              // esl_rnd_FChoose
    const float roll = (float)rng.next();
    float acc = 0.f;
    for (int q = 0; q < N; q++) { acc += p[q]; if (roll < acc) return q; }
  }
}

struct Step { int8_t st; int32_t k, i; };
struct Seg { int idx, i, j, k, m; float prob; };

// p7_spensemble_Cluster / p7_spensemble_fs_Cluster (single linkage, significant clusters, consensus end points) and the removal
// of dominated clusters (p7_domaindef.c:815-843, 923-953)
void cluster_segments(const std::vector<Seg> &sp, int nsamples, bool fs, std::vector<std::pair<int, int>> *env) {
  const int max_diagdiff = 4;
  const float min_overlap = 0.8f, min_posterior = 0.25f, min_endpointp = 0.02f;
  // ---- p7_spensemble_Cluster
  auto linked = [&](const Seg &a, const Seg &b) {
    int nov = std::min(a.j, b.j) - std::max(a.i, b.i) + 1, n = std::min(a.j - a.i + 1, b.j - b.i + 1);
    if ((float)nov / (float)n < min_overlap) return false;
    nov = std::min(a.m, b.m) - std::max(a.k, b.k); n = std::min(a.m - a.k + 1, b.m - b.k + 1);
    if ((float)nov / (float)n < min_overlap) return false;
    if (fs) {                                             // nucleotide coordinates: diagonals in codons (link_spsamples_fs)
      if (std::abs((a.i / 3 - a.k) - (b.i / 3 - b.k)) <= max_diagdiff) return true;
      return std::abs((a.j / 3 - a.m) - (b.j / 3 - b.m)) <= max_diagdiff;
    }
    if (std::abs((a.i - a.k) - (b.i - b.k)) <= max_diagdiff) return true;
    return std::abs((a.j - a.m) - (b.j - b.m)) <= max_diagdiff;
  };
  const int nsp = (int)sp.size();
  // Single linkage = connected components of the "linked" graph; a cluster's number is the order of its first segment in <sp>.
  // 200 traces sampled from one posterior keep producing the SAME segments (a few dozen distinct (i, j, k, m) among ~400), and
  // linked() depends on those four numbers only: the components are found on the distinct segments (a few thousand link tests
  // instead of ~10^5, each two float divisions: half of an ensemble's host time) and handed back to the segments.
  std::vector<int> uniq_of((size_t)nsp), first_of;              // segment -> distinct segment; distinct segment -> its first segment
  {
    std::vector<int> order((size_t)nsp);
    for (int h = 0; h < nsp; h++) order[(size_t)h] = h;
    auto key_less = [&](int a, int b) {
      const Seg &x = sp[(size_t)a], &y = sp[(size_t)b];
      if (x.i != y.i) return x.i < y.i;
      if (x.j != y.j) return x.j < y.j;
      if (x.k != y.k) return x.k < y.k;
      if (x.m != y.m) return x.m < y.m;
      return a < b;                                             // equal segments: in <sp> order, so that the first of a run is the earliest
    };
    std::sort(order.begin(), order.end(), key_less);
    for (int z = 0; z < nsp; z++) {
      const int h = order[(size_t)z];
      const bool same = z > 0 && sp[(size_t)h].i == sp[(size_t)order[(size_t)z - 1]].i && sp[(size_t)h].j == sp[(size_t)order[(size_t)z - 1]].j &&
                        sp[(size_t)h].k == sp[(size_t)order[(size_t)z - 1]].k && sp[(size_t)h].m == sp[(size_t)order[(size_t)z - 1]].m;
      // Copies are one vertex only if the segment is linked to ITSELF.  The model overlap is nov = min(m) - max(k) WITHOUT the + 1
      // (p7_spensemble.c:207, :244), so a segment of four model nodes or fewer has (m - k) / (m - k + 1) < 0.8 and is linked to
      // nothing, its own copies included: in the reference every copy is a singleton cluster (never significant), and so it is here.
      if (!same || !linked(sp[(size_t)h], sp[(size_t)h])) first_of.push_back(h);
      uniq_of[(size_t)h] = (int)first_of.size() - 1;
    }
  }
  const int nu = (int)first_of.size();
  std::vector<int> ucomp((size_t)nu, -1), stack;
  int ncomp = 0;
  for (int u = 0; u < nu; u++) {
    if (ucomp[(size_t)u] >= 0) continue;
    ucomp[(size_t)u] = ncomp; stack.assign(1, u);
    while (!stack.empty()) {
      const int a = stack.back(); stack.pop_back();
      for (int b = 0; b < nu; b++)
        if (ucomp[(size_t)b] < 0 && linked(sp[(size_t)first_of[(size_t)a]], sp[(size_t)first_of[(size_t)b]])) { ucomp[(size_t)b] = ncomp; stack.push_back(b); }
    }
    ncomp++;
  }
  std::vector<int> assign((size_t)nsp, -1), number((size_t)ncomp, -1);
  int nc = 0;
  for (int h = 0; h < nsp; h++) {                               // clusters numbered by their first segment, as the search over <sp> numbers them
    int &id = number[(size_t)ucomp[(size_t)uniq_of[(size_t)h]]];
    if (id < 0) id = nc++;
    assign[(size_t)h] = id;
  }
  std::vector<Seg> sig;
  std::vector<int> epc;
  for (int c = 0; c < nc; c++) {
    int ninc = 0, last = -1;
    for (int h = 0; h < nsp; h++) if (assign[(size_t)h] == c) { if (sp[(size_t)h].idx != last) ninc++; last = sp[(size_t)h].idx; }
    if ((float)ninc / (float)nsamples < min_posterior) continue;
    const int thr = (int)ceilf((float)ninc * min_endpointp);
    // widest end point that at least thr segments of the cluster share, independently for i, k (leftmost) and j, m (rightmost)
    auto consensus = [&](int Seg::*f, bool leftmost) {
      int lo = 0, hi = 0; bool first = true;
      for (int h = 0; h < nsp; h++) if (assign[(size_t)h] == c) { const int v = sp[(size_t)h].*f; if (first) { lo = hi = v; first = false; } else { lo = std::min(lo, v); hi = std::max(hi, v); } }
      epc.assign((size_t)(hi - lo + 1), 0);
      for (int h = 0; h < nsp; h++) if (assign[(size_t)h] == c) epc[(size_t)(sp[(size_t)h].*f - lo)]++;
      if (leftmost) { for (int v = lo; v <= hi; v++) if (epc[(size_t)(v - lo)] >= thr) return v; }
      else          { for (int v = hi; v >= lo; v--) if (epc[(size_t)(v - lo)] >= thr) return v; }
      return lo + (int)(std::max_element(epc.begin(), epc.end()) - epc.begin());
    };
    const int bi = consensus(&Seg::i, true), bk = consensus(&Seg::k, true), bj = consensus(&Seg::j, false), bm = consensus(&Seg::m, false);
    if (bi > bj || bk > bm) continue;
    sig.push_back(Seg{c, bi, bj, bk, bm, (float)ninc / (float)nsamples});
  }
  std::stable_sort(sig.begin(), sig.end(), [](const Seg &a, const Seg &b) { return a.i < b.i; });
  // ---- clusters dominated by a more probable one that overlaps them (p7_domaindef.c:815-843)
  std::vector<char> dominated(sig.size(), 0);
  for (size_t d = 0; d < sig.size(); d++)
    for (size_t d2 = d + 1; d2 < sig.size(); d2++) {
      const int nov = std::min(sig[d].j, sig[d2].j) - std::max(sig[d].i, sig[d2].i) + 1;
      if (nov == 0) break;
      const int n = std::min(sig[d].j - sig[d].i + 1, sig[d2].j - sig[d2].i + 1);
      if ((float)nov / (float)n >= 0.8) { if (sig[d].prob > sig[d2].prob) dominated[d2] = 1; else dominated[d] = 1; }
    }
  for (size_t d = 0; d < sig.size(); d++) if (!dominated[d]) env->push_back({sig[d].i, sig[d].j});
}

}  // namespace

// One region of an ORF.  fwd: (Lr+1) x (M+1) x {M,D,I}, fx: (Lr+1) x {E,N,J,B,C,SCALE} of p7_Forward on the region with the
// model in the multihit configuration for length <cfg_L>; res[0..Lr): the region's residues.
// Out: n2sc[0..Lr) per-residue null2 log odds; env: envelopes (1-based, region-relative), ordered by start.
int bath::region_trace_ensemble(const bath_hip_oprofile *om, int cfg_L, const uint8_t *res, int Lr, const float *fwd, const float *fx,
                                std::vector<float> *n2sc_out, std::vector<std::pair<int, int>> *env, uint32_t seed) {
  const int M = om->M, Q = std::max(2, (M - 1) / 4 + 1);
  const size_t W = (size_t)(M + 1) * 3;
  const float *tf = om->tf.data();
  if (om->ensure_len_tables(cfg_L) != BATH_OK) return BATH_EFAIL;
  const float pmove = om->lt.h_pmove[(size_t)cfg_L], ploop = 1.0f - pmove;
  const float tEL = om->xf_E[0], tEM = om->xf_E[1];
  const int nsamples = 200;
  std::vector<float> &n2sc = *n2sc_out;
  n2sc.assign((size_t)Lr + 1, 0.f);                       // 1-based positions of the region
  env->clear();

  FastRng rng(region_seed(seed));
  std::vector<Step> tr;
  std::vector<Seg> sp;
  std::vector<float> cnt((size_t)M + 1);
  // match odds ratios node-major, [k][20]: the expectation sum_k cnt[k] * rf[x][k] of all 20 residues is then one pass over the
  // nodes of the domain with 20 independent accumulators (each residue's sum still runs over k in ascending order)
  static thread_local std::vector<float> rft;
  rft.resize((size_t)(M + 1) * 20);
  for (int x = 0; x < 20; x++) { const float *e = om->rf.data() + (size_t)x * (M + 1); for (int q = 0; q <= M; q++) rft[(size_t)q * 20 + x] = e[q]; }
  const int step_cap = 4 * (Lr + M) + 64;
  for (int t = 0; t < nsamples; t++) {
    tr.clear();
    int i = Lr, k = 0, s0 = sC;
    tr.push_back(Step{(int8_t)sT, 0, i}); tr.push_back(Step{(int8_t)sC, 0, i});
    while (s0 != sS) {
      int s1 = -1;
      switch (s0) {
      case sM: {
        const float *tk = tf + (size_t)k * 8, *pr = fwd + (size_t)(i - 1) * W;
        float p[4] = {fx[(size_t)(i - 1) * 6 + XB] * tk[BM], pr[(size_t)(k - 1) * 3 + cM] * tk[MM], pr[(size_t)(k - 1) * 3 + cI] * tk[IM], pr[(size_t)(k - 1) * 3 + cD] * tk[DM]};
        static const int state[4] = {sB, sM, sI, sD};
        s1 = state[choose(rng, p)]; k--; i--; break; }
      case sD: {
        const float *c = fwd + (size_t)i * W;
        float p[2] = {k - 1 >= 1 ? c[(size_t)(k - 1) * 3 + cM] * tf[(size_t)(k - 1) * 8 + MD] : 0.f, k - 1 >= 1 ? c[(size_t)(k - 1) * 3 + cD] * tf[(size_t)(k - 1) * 8 + DD] : 0.f};
        s1 = choose(rng, p) == 0 ? sM : sD; k--; break; }
      case sI: {
        const float *pr = fwd + (size_t)(i - 1) * W;
        float p[2] = {pr[(size_t)k * 3 + cM] * tf[(size_t)k * 8 + MI], pr[(size_t)k * 3 + cI] * tf[(size_t)k * 8 + II]};
        s1 = choose(rng, p) == 0 ? sM : sI; i--; break; }
      case sN: s1 = (i == 0) ? sS : sN; break;
      case sC: {
        float p[2] = {fx[(size_t)(i - 1) * 6 + XC] * ploop, fx[(size_t)i * 6 + XE] * tEM * fx[(size_t)i * 6 + XS]};
        s1 = choose(rng, p) == 0 ? sC : sE; break; }
      case sJ: {
        float p[2] = {fx[(size_t)(i - 1) * 6 + XJ] * ploop, fx[(size_t)i * 6 + XE] * tEL * fx[(size_t)i * 6 + XS]};
        s1 = choose(rng, p) == 0 ? sJ : sE; break; }
      case sE: {                                          // select_e: cumulative sum in double over the cells in striped order
        const float *c = fwd + (size_t)i * W;
        const double roll = rng.next();
        const float norm = (float)(1.0 / fx[(size_t)i * 6 + XE]);
        double sum = 0.0;
        for (int pass = 0; pass < 4 && s1 < 0; pass++)
          for (int q = 0; q < Q && s1 < 0; q++) {
            for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += kk <= M ? c[(size_t)kk * 3 + cM] * norm : 0.0f; if (roll < sum) { k = kk; s1 = sM; } }
            for (int r = 0; r < 4 && s1 < 0; r++) { const int kk = r * Q + q + 1; sum += kk <= M ? c[(size_t)kk * 3 + cD] * norm : 0.0f; if (roll < sum) { k = kk; s1 = sD; } }
          }
        break; }
      case sB: {
        float p[2] = {fx[(size_t)i * 6 + XN] * pmove, fx[(size_t)i * 6 + XJ] * pmove};
        s1 = choose(rng, p) == 0 ? sN : sJ; break; }
      default: break;
      }
      if (s1 < 0 || i < 0 || k < 0 || (int)tr.size() > step_cap) return BATH_EFAIL;
      tr.push_back(Step{(int8_t)s1, k, i});
      if ((s1 == sN || s1 == sJ || s1 == sC) && s1 == s0) i--;
      s0 = s1;
    }
    std::reverse(tr.begin(), tr.end());

    // domains of this trace (p7_trace_Index), their null2 by trace, and the per-residue bookkeeping of :790-803
    int pos = 1;
    for (size_t z = 0; z < tr.size();) {
      if (tr[z].st != sB) { z++; continue; }
      const size_t zb = z;
      int sqfrom = 0, sqto = 0, hmmfrom = 0, hmmto = 0, Ld = 0;
      std::fill(cnt.begin(), cnt.end(), 0.f);
      for (z = zb + 1; z < tr.size() && tr[z].st != sE; z++) {
        if (tr[z].st == sM) { if (!sqfrom) sqfrom = tr[z].i; if (!hmmfrom) hmmfrom = tr[z].k; sqto = tr[z].i; hmmto = tr[z].k; }
        if (tr[z].st == sM || tr[z].st == sI) { Ld++; cnt[(size_t)tr[z].k] += 1.0f; }      // inserts land in the match slot (null2.c:160-166)
      }
      sp.push_back(Seg{t, sqfrom, sqto, hmmfrom, hmmto, 0.f});
      const float norm = (float)(1.0 / (float)Ld);
      for (int q = std::max(hmmfrom, 1); q <= hmmto; q++) cnt[(size_t)q] *= norm;
      float null2[kKp];
      for (int x = 0; x < 20; x++) null2[x] = 0.f;
      for (int q = std::max(hmmfrom, 1); q <= hmmto; q++) {                 // cnt is zero outside the domain's nodes
        const float c = cnt[(size_t)q], *e = rft.data() + (size_t)q * 20;
        for (int x = 0; x < 20; x++) null2[x] += c * e[x];
      }
      static const int mem[6][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}, {-1, -1}};     // B=DN J=IL Z=EQ O=K U=C X=any
      for (int dx = 0; dx < 6; dx++) {
        float s = 0.f; int c = 0;
        if (dx == 5) { for (int y = 0; y < 20; y++) { s += null2[y]; c++; } }
        else { s += null2[mem[dx][0]]; c++; if (mem[dx][1] != mem[dx][0]) { s += null2[mem[dx][1]]; c++; } }
        null2[21 + dx] = s / (float)c;
      }
      null2[20] = 1.0f; null2[27] = 1.0f; null2[28] = 1.0f;
      for (; pos <= sqfrom; pos++) n2sc[(size_t)pos] += 1.0f;
      for (; pos <= sqto; pos++) n2sc[(size_t)pos] += null2[std::min<int>(res[pos - 1], kKp - 1)];
      z++;
    }
    for (; pos <= Lr; pos++) n2sc[(size_t)pos] += 1.0f;
  }
  for (int pos = 1; pos <= Lr; pos++) n2sc[(size_t)pos] = logf(n2sc[(size_t)pos] / (float)nsamples);

  cluster_segments(sp, nsamples, false, env);
  return BATH_OK;
}

// ---- frameshift branch: region_trace_ensemble_frameshift (p7_domaindef.c:891-958) with the generic log-space stochastic
// traceback (generic_stotrace_frameshift.c:40-215) on the region's Forward matrix from fs5_fwd_kernel (multihit).
// fwd: (Lr+1) x (M+1) x {D, I, M_C0, M_C1..M_C5}; fx: (Lr+1) x {E,N,J,B,C}; tsc: generic [M][8]; xNL/xNM: log loop / move of
// N, C, J; xE: log 1/2.  The region starts at nucleotide <ireg> of its window: segments are clustered in window coordinates
// (the link rule divides them by 3).  env: envelopes in window nucleotides, ordered by start (empty: no valid traces).
int bath::fs_region_trace_ensemble(int M, const float *tsc, float xNL, float xNM, float xE, int ireg, int Lr, const float *fwd, const float *fx,
                                   std::vector<std::pair<int, int>> *env, uint32_t seed) {
  enum { gD = 0, gI = 1, gM = 2 };
  enum { gE = 0, gN, gJ, gB, gC };
  env->clear();
  const auto t_begin = std::chrono::steady_clock::now();
  const size_t W = (size_t)(M + 1) * 8;
  auto DP = [&](int i, int k, int s) { return fwd[(size_t)i * W + (size_t)k * 8 + s]; };
  auto X = [&](int i, int s) { return fx[(size_t)i * 5 + s]; };
  auto TS = [&](int s, int k) { return tsc[(size_t)k * 8 + s]; };
  // The probability vector of a choice is a function of (state, i, k) alone, and 200 traces sampled from one posterior
  // keep coming back to the same cells: each vector is computed once (esl_vec_FLogNorm: ~2 expf per entry, 2M+1 entries in
  // the E state) and kept, so that a revisit costs one random number and a few compares.  Same arithmetic, same draws.
  struct Memo {
    std::vector<uint8_t> cflag, rflag;      // per cell: 1 = M's 4, 2 = codon's 5, 4 = D's 2, 8 = I's 2; per row: 1 = C, 2 = J, 4 = B
    std::vector<float> cell, row, epool;    // 16 floats per cell, 12 per row, 2M+1 per E row visited
    std::vector<int64_t> eoff;
  };
  static thread_local Memo mm;
  const size_t ncell = (size_t)(Lr + 1) * (size_t)(M + 1);
  const bool memo_cells = ncell * 64 <= ((size_t)192 << 20);
  if (memo_cells) { mm.cflag.assign(ncell, 0); if (mm.cell.size() < ncell * 16) mm.cell.resize(ncell * 16); }
  mm.rflag.assign((size_t)Lr + 1, 0); mm.eoff.assign((size_t)Lr + 1, -1); mm.epool.clear();
  if (mm.row.size() < ((size_t)Lr + 1) * 12) mm.row.resize(((size_t)Lr + 1) * 12);
  float tmp[8];
  auto lognorm = [&](float *v, int n) {                  // esl_vec_FLogNorm, then esl_vec_FNorm as esl_rnd_FChoose's callers do
    float mx = v[0];
    for (int q = 1; q < n; q++) mx = std::max(mx, v[q]);
    float denom;
    if (mx == INFINITY) denom = INFINITY;
    else if (mx == -INFINITY) denom = -INFINITY;
    else { float sum = 0.f; for (int q = 0; q < n; q++) if (v[q] > mx - 50.f) sum += expf(v[q] - mx); denom = logf(sum) + mx; }
    for (int q = 0; q < n; q++) v[q] = expf(v[q] - denom);
    float sum = 0.f, comp = 0.f;
    for (int q = 0; q < n; q++) { const float y = v[q] - comp, t = sum + y; comp = (t - sum) - y; sum = t; }
    for (int q = 0; q < n; q++) v[q] = (sum != 0.0f) ? v[q] / sum : 1.0f / (float)n;
  };
  auto roll = [&](FastRng &rng, const float *v, int n) {  // esl_rnd_FChoose
    for (;;) {
      const float r = (float)rng.next();
      float acc = 0.f;
      for (int q = 0; q < n; q++) { acc += v[q]; if (r < acc) return q; }
    }
  };
  // slot of a cell's vector (nullptr: not memoised, use tmp) and whether it still has to be filled
  auto cell_slot = [&](int i, int k, int bit, int at, bool *fresh) -> float * {
    if (!memo_cells) { *fresh = true; return tmp; }
    const size_t c = (size_t)i * (size_t)(M + 1) + (size_t)k;
    *fresh = !(mm.cflag[c] & bit); mm.cflag[c] |= (uint8_t)bit;
    return mm.cell.data() + c * 16 + at;
  };
  auto row_slot = [&](int i, int bit, int at, bool *fresh) -> float * {
    *fresh = !(mm.rflag[(size_t)i] & bit); mm.rflag[(size_t)i] |= (uint8_t)bit;
    return mm.row.data() + (size_t)i * 12 + at;
  };
  const int nsamples = 200, step_cap = 4 * (Lr + M) + 64;
  FastRng rng(region_seed(seed));
  std::vector<Seg> sp;
  struct FsStep { int8_t st, c; int32_t k, i; };
  std::vector<FsStep> tr;
  for (int t = 0; t < nsamples; t++) {
    tr.clear();
    int i = Lr, k = 0, c = 0, sprv = sC;
    tr.push_back(FsStep{(int8_t)sT, 0, 0, i}); tr.push_back(FsStep{(int8_t)sC, 0, 0, i});
    while (sprv != sS) {
      int scur = -1;
      switch (sprv) {
      case sC:
        if (X(i, gC) == -INFINITY) return BATH_OK;
        if (i < 4) { scur = sE; break; }
        { bool fresh; float *v = row_slot(i, 1, 0, &fresh);
          if (fresh) { v[0] = X(i - 3, gC) + xNL; v[1] = X(i - 2, gC) + xNL; v[2] = X(i - 1, gC) + xNL; v[3] = X(i, gE) + xE; lognorm(v, 4); }
          scur = roll(rng, v, 4) < 3 ? sC : sE; }
        break;
      case sE:
        if (X(i, gE) == -INFINITY) return BATH_OK;
        { if (mm.eoff[(size_t)i] < 0) {
            mm.eoff[(size_t)i] = (int64_t)mm.epool.size();
            mm.epool.resize(mm.epool.size() + (size_t)(2 * M + 1));
            float *v = mm.epool.data() + mm.eoff[(size_t)i];
            v[0] = v[(size_t)M + 1] = -INFINITY;
            for (int q = 1; q <= M; q++) v[(size_t)q] = DP(i, q, gM);
            for (int q = 2; q <= M; q++) v[(size_t)q + M] = DP(i, q, gD);
            lognorm(v, 2 * M + 1);
          }
          k = roll(rng, mm.epool.data() + mm.eoff[(size_t)i], 2 * M + 1); }
        if (k <= M) scur = sM; else { k -= M; scur = sD; }
        break;
      case sM: {
        bool fresh; float *v = cell_slot(i, k, 1, 0, &fresh);
        if (fresh) { v[0] = X(i, gB) + TS(BM, k - 1); v[1] = DP(i, k - 1, gM) + TS(MM, k - 1); v[2] = DP(i, k - 1, gI) + TS(IM, k - 1); v[3] = DP(i, k - 1, gD) + TS(DM, k - 1); lognorm(v, 4); }
        static const int state[4] = {sB, sM, sI, sD};
        scur = state[roll(rng, v, 4)]; k--; break; }
      case sD:
        if (DP(i, k, gD) == -INFINITY) return BATH_OK;
        { bool fresh; float *v = cell_slot(i, k, 4, 9, &fresh);
          if (fresh) { v[0] = DP(i, k - 1, gM) + TS(MD, k - 1); v[1] = DP(i, k - 1, gD) + TS(DD, k - 1); lognorm(v, 2); }
          scur = roll(rng, v, 2) == 0 ? sM : sD; }
        k--; break;
      case sI:
        if (DP(i, k, gI) == -INFINITY || i < 3) return BATH_OK;
        { bool fresh; float *v = cell_slot(i, k, 8, 11, &fresh);
          if (fresh) { v[0] = DP(i - 3, k, gM) + TS(MI, k); v[1] = DP(i - 3, k, gI) + TS(II, k); lognorm(v, 2); }
          scur = roll(rng, v, 2) == 0 ? sM : sI; }
        i -= 3; break;
      case sN:
        if (X(i, gN) == -INFINITY) return BATH_OK;
        scur = (i == 0) ? sS : sN; break;
      case sB:
        if (X(i, gB) == -INFINITY) return BATH_OK;
        { bool fresh; float *v = row_slot(i, 4, 8, &fresh);
          if (fresh) { v[0] = X(i, gN) + xNM; v[1] = X(i, gJ) + xNM; lognorm(v, 2); }
          scur = roll(rng, v, 2) == 0 ? sN : sJ; }
        break;
      case sJ:
        if (X(i, gJ) == -INFINITY) return BATH_OK;
        if (i < 4) { scur = sE; break; }
        { bool fresh; float *v = row_slot(i, 2, 4, &fresh);
          if (fresh) { v[0] = X(i - 3, gJ) + xNL; v[1] = X(i - 2, gJ) + xNL; v[2] = X(i - 1, gJ) + xNL; v[3] = X(i, gE) + xE; lognorm(v, 4); }
          scur = roll(rng, v, 4) < 3 ? sJ : sE; }
        break;
      default: return BATH_OK;
      }
      if (scur == sM) {                                   // codon length from the C1..C5 cells
        bool fresh; float *v = cell_slot(i, k, 2, 4, &fresh);
        if (fresh) { for (int q = 0; q < 5; q++) v[q] = DP(i, k, gM + 1 + q); lognorm(v, 5); }
        c = roll(rng, v, 5) + 1;
        if (i - c < 0) scur = sB;
      } else c = 0;
      if (scur < 0 || k < 0 || i < 0 || (int)tr.size() > step_cap) return BATH_OK;
      tr.push_back(FsStep{(int8_t)scur, (int8_t)c, k, i});
      if ((scur == sN || scur == sC || scur == sJ) && scur == sprv) i--;
      sprv = scur;
      i -= c;
      if (i < 0) return BATH_OK;
    }
    std::reverse(tr.begin(), tr.end());
    for (size_t z = 0; z < tr.size(); z++) {              // p7_trace_fs_Index
      if (tr[z].st != sB) continue;
      int sqfrom = 0, sqto = 0, hmmfrom = 0, hmmto = 0;
      for (z = z + 1; z < tr.size() && tr[z].st != sE; z++)
        if (tr[z].st == sM) { if (!sqfrom) sqfrom = tr[z].i - tr[z].c + 1; if (!hmmfrom) hmmfrom = tr[z].k; sqto = tr[z].i; hmmto = tr[z].k; }
      sp.push_back(Seg{t, sqfrom + ireg - 1, sqto + ireg - 1, hmmfrom, hmmto, 0.f});
    }
  }
  static const bool prof = [] { const char *e = std::getenv("BATH_ENS_PROF"); return e && e[0] == '1'; }();
  const auto tc0 = std::chrono::steady_clock::now();
  cluster_segments(sp, nsamples, true, env);
  if (prof) {
    const auto tc1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[ens] Lr %d: traces %.3f ms (%zu segments, %zu E rows memoised), clustering %.3f ms\n", Lr,
            std::chrono::duration<double, std::milli>(tc0 - t_begin).count(), sp.size(), mm.epool.size() / (size_t)(2 * M + 1),
            std::chrono::duration<double, std::milli>(tc1 - tc0).count());
  }
  return BATH_OK;
}


// ---- self-test hooks (include/bath_hip.h): the restated easel pieces and the frameshift ensemble, callable without a GPU
extern "C" int bath_selftest_fs_ensemble(int M, const float *tsc, float xNL, float xNM, float xE, int ireg, int Lr, const float *fwd, const float *fx,
                                         int32_t *env, int max_env, int32_t *n_env) {
  if (!tsc || !fwd || !fx || !env || !n_env || M < 1 || Lr < 1 || max_env < 0) return BATH_EINVAL;
  std::vector<std::pair<int, int>> cl;
  const int st = bath::fs_region_trace_ensemble(M, tsc, xNL, xNM, xE, ireg, Lr, fwd, fx, &cl);
  if (st != BATH_OK) return st;
  *n_env = (int32_t)cl.size();
  for (size_t e = 0; e < cl.size() && (int)e < max_env; e++) { env[2 * e] = cl[e].first; env[2 * e + 1] = cl[e].second; }
  return BATH_OK;
}

// the clustering step alone on caller-supplied segments (test hook: tests/test_ensemble_cpu.py holds it against the oracle's all-pairs search)
extern "C" int bath_selftest_cluster_segments(int n, const int32_t *idx, const int32_t *i, const int32_t *j, const int32_t *k, const int32_t *m,
                                              int nsamples, int fs, int32_t *env, int max_env, int32_t *n_env) {
  if (n < 0 || !n_env || (n > 0 && (!idx || !i || !j || !k || !m))) return BATH_EINVAL;
  std::vector<Seg> sp((size_t)n);
  for (int h = 0; h < n; h++) sp[(size_t)h] = Seg{idx[h], i[h], j[h], k[h], m[h], 0.f};
  std::vector<std::pair<int, int>> cl;
  cluster_segments(sp, nsamples, fs != 0, &cl);
  *n_env = (int32_t)cl.size();
  for (size_t e = 0; e < cl.size() && (int)e < max_env; e++) { env[2 * e] = cl[e].first; env[2 * e + 1] = cl[e].second; }
  return BATH_OK;
}

extern "C" int bath_selftest_rng_stream(uint32_t seed, int n, double *out) {
  if (!out || n < 0) return BATH_EINVAL;
  FastRng rng(seed);
  for (int i = 0; i < n; i++) out[i] = rng.next();
  return BATH_OK;
}
extern "C" int bath_selftest_fchoose(uint32_t seed, const float *p, int n, int draws, int32_t *out) {
  if (!p || !out || n < 1 || n > 8 || draws < 0) return BATH_EINVAL;
  FastRng rng(seed);
  for (int d = 0; d < draws; d++) {
    int r = 0;
    switch (n) {
#define BATH_ST_CASE(N) case N: { float v[N]; for (int q = 0; q < N; q++) v[q] = p[q]; r = choose<N>(rng, v); } break;
      BATH_ST_CASE(1) BATH_ST_CASE(2) BATH_ST_CASE(3) BATH_ST_CASE(4) BATH_ST_CASE(5) BATH_ST_CASE(6) BATH_ST_CASE(7) BATH_ST_CASE(8)
#undef BATH_ST_CASE
    }
    out[d] = r;
  }
  return BATH_OK;
}
