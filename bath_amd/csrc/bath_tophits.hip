// bath_tophits.hip -- the hit list of a search and its tabular output (host code; no kernels).
//
// Reference: what bathsearch does with the hits its workers' pipelines created (bathsearch.c:868-921):
//   p7_tophits_ComputeEvalues_BATH        src/p7_tophits.c:789-801   lnP += log(nres / (3 * max_length))
//   p7_tophits_SortBySeqidxAndAlipos      :379 with hit_sorter_by_seqidx_aliposition :285-306
//   p7_tophits_RemoveDuplicates           :816-903
//   p7_tophits_SortBySortkey              :345 with hit_sorter_by_sortkey :261-283
//   p7_tophits_Threshold                  :914-966 (pli->Z = 1, by E-value)
//   p7_tophits_TabularTargets             :1603-1729 (--tblout, with or without --cigar, with or without --fs)
// A hit is one bath_fs_domain that passed the in-pipeline E-value test (p7_pipeline.c:1080, 1246), with the names of its
// target sequence.  qsort is not stable and neither need this be: ties are broken by the comparators' own secondary keys.
#include <algorithm>
#include <cinttypes>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bath_hip.h"

namespace {

enum { IS_REPORTED = 1, IS_INCLUDED = 2, IS_DUPLICATE = 4 };

struct Hit {
  bath_fs_domain d;
  int64_t seqidx;
  int64_t target_len;
  std::string name, acc, desc, cigar;
  double lnP, sortkey;
  float score;
  int flags;
};

int strand_dir(const bath_fs_domain &d) { return d.iali < d.jali ? 1 : -1; }

}  // namespace

struct bath_tophits {
  std::vector<Hit> unsrt;
  std::vector<int> order;              // th->hit[]: indices into unsrt
  int64_t nreported = 0, nincluded = 0;
  double incE = 0.01;                  // pli->incE, p7_pipeline.c:166
  int by_E = 1, inc_by_E = 1;          // pli->by_E / pli->inc_by_E, cleared by -T / --incT (:146-176)
  double T = 0.0, incT = 0.0;
};

extern "C" bath_tophits *bath_tophits_create(void) { return new bath_tophits(); }
extern "C" void bath_tophits_destroy(bath_tophits *th) { delete th; }
extern "C" int64_t bath_tophits_count(const bath_tophits *th) { return th ? (int64_t)th->unsrt.size() : 0; }
extern "C" int64_t bath_tophits_reported(const bath_tophits *th) { return th ? th->nreported : 0; }
extern "C" void bath_tophits_set_inclusion(bath_tophits *th, double incE) { if (th) th->incE = incE; }
extern "C" void bath_tophits_set_score_thresholds(bath_tophits *th, int by_E, double T, int inc_by_E, double incT) {
  if (!th) return;
  th->by_E = by_E ? 1 : 0; th->T = T; th->inc_by_E = inc_by_E ? 1 : 0; th->incT = incT;
}
// bathsearch.c:868-881: what p7_tophits_ComputeEvalues_BATH gets as N
extern "C" int64_t bath_search_space_residues(int Z_is_set, double Z_megabases, int strands, int64_t nres_searched) {
  if (!Z_is_set) return nres_searched;
  int64_t n = (int64_t)(1000000 * Z_megabases);
  if (strands == BATH_STRAND_BOTH) n *= 2;
  return n;
}

extern "C" int bath_tophits_add(bath_tophits *th, const bath_fs_domain *dom, int64_t n, const char *cigars, int64_t seqidx0,
                                const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs, const int64_t *seq_lens) {
  if (!th || (n > 0 && (!dom || !seq_names || !seq_lens))) return BATH_EINVAL;
  for (int64_t i = 0; i < n; i++) {
    if (!dom[i].reported) continue;                       // the pipeline creates a hit only for these
    Hit h;
    h.d = dom[i];
    h.seqidx = seqidx0 + dom[i].window;
    h.target_len = seq_lens[dom[i].window];
    h.name = seq_names[dom[i].window] ? seq_names[dom[i].window] : "";
    if (seq_accs && seq_accs[dom[i].window]) h.acc = seq_accs[dom[i].window];
    if (seq_descs && seq_descs[dom[i].window]) h.desc = seq_descs[dom[i].window];
    if (cigars && dom[i].cigar_off >= 0) h.cigar = cigars + dom[i].cigar_off;
    h.lnP = dom[i].lnP;
    h.score = dom[i].bitscore;
    h.sortkey = -dom[i].lnP;                              // inc_by_E (p7_pipeline.c:1119, 1280)
    h.flags = 0;
    th->unsrt.push_back(std::move(h));
  }
  return BATH_OK;
}

extern "C" int bath_tophits_finalize(bath_tophits *th, int64_t nres, int max_length, double E) {
  if (!th || max_length <= 0) return BATH_EINVAL;
  std::vector<Hit> &H = th->unsrt;
  const int N = (int)H.size();
  // p7_tophits_ComputeEvalues_BATH(th, resCnt, om->max_length * 3)
  for (Hit &h : H) {
    h.lnP += std::log((double)((float)nres / (float)(max_length * 3)));
    h.d.lnP = h.lnP;
    h.sortkey = -1.0 * h.lnP;
  }
  th->order.resize((size_t)N);
  for (int i = 0; i < N; i++) th->order[(size_t)i] = i;
  // p7_tophits_SortBySeqidxAndAlipos
  std::stable_sort(th->order.begin(), th->order.end(), [&](int a, int b) {
    const Hit &h1 = H[(size_t)a], &h2 = H[(size_t)b];
    if (h1.seqidx != h2.seqidx) return h1.seqidx < h2.seqidx;
    int64_t s1 = h1.d.iali, e1 = h1.d.jali, s2 = h2.d.iali, e2 = h2.d.jali;
    const int dir1 = s1 < e1 ? 1 : -1, dir2 = s2 < e2 ? 1 : -1;
    if (dir1 < 0) std::swap(s1, e1);
    if (dir2 < 0) std::swap(s2, e2);
    if (dir1 != dir2) return dir2 < 0;                    // the positive strand goes first
    if (s1 != s2) return s1 < s2;
    return e1 > e2;                                       // the longer hit first
  });
  // p7_tophits_RemoveDuplicates (using_bit_cutoffs = FALSE)
  if (N >= 2) {
    int j = 0;
    for (int i = 1; i < N; i++) {
      Hit &hi = H[(size_t)th->order[(size_t)i]], &hj = H[(size_t)th->order[(size_t)j]], &hp = H[(size_t)th->order[(size_t)i - 1]];
      int64_t s_j = hj.d.iali, e_j = hj.d.jali, s_i = hi.d.iali, e_i = hi.d.jali;
      const int dir_j = s_j < e_j ? 1 : -1, dir_i = s_i < e_i ? 1 : -1;
      if (dir_j < 0) std::swap(s_j, e_j);
      if (dir_i < 0) std::swap(s_i, e_i);
      const int len_j = (int)(e_j - s_j + 1), len_i = (int)(e_i - s_i + 1);
      const int64_t is = std::max(s_i, s_j), ie = std::min(e_i, e_j);
      const int ilen = (int)(ie - is + 1);
      const int hs = std::max(hi.d.ihmm, hj.d.ihmm), he = std::min(hi.d.jhmm, hj.d.jhmm);
      const int hlen = he - hs + 1;
      if (hi.name == hp.name && hi.seqidx == hp.seqidx && dir_i == dir_j && hlen > 0 &&
          ((s_i >= s_j - 3 && s_i <= s_j + 3) || (e_i >= e_j - 3 && e_i <= e_j + 3) || (ilen >= len_i * 0.95) || (ilen >= len_j * 0.95))) {
        const bool remove_j = hi.lnP < hj.lnP;            // keep the better E-value
        (remove_j ? hj : hi).flags |= IS_DUPLICATE;
        if (remove_j) j = i;
      } else j = i;
    }
  }
  // p7_tophits_SortBySortkey
  for (int i = 0; i < N; i++) th->order[(size_t)i] = i;
  std::stable_sort(th->order.begin(), th->order.end(), [&](int a, int b) {
    const Hit &h1 = H[(size_t)a], &h2 = H[(size_t)b];
    if (h1.sortkey != h2.sortkey) return h1.sortkey > h2.sortkey;
    const int c = std::strcmp(h1.name.c_str(), h2.name.c_str());
    if (c != 0) return c < 0;
    const int dir1 = strand_dir(h1.d), dir2 = strand_dir(h2.d);
    if (dir1 != dir2) return dir2 < 0;
    return h1.d.iali < h2.d.iali;
  });
  // p7_tophits_Threshold with pli->Z = 1 (bathsearch.c:917-918), by E-value; inclusion threshold = reporting threshold here
  th->nreported = th->nincluded = 0;
  for (Hit &h : H) {
    h.flags &= ~(IS_REPORTED | IS_INCLUDED);
    const bool reportable = th->by_E ? std::exp(h.lnP) <= E : (double)h.score >= th->T;                  // p7_pli_TargetReportable, p7_pipeline.c:583
    if (!(h.flags & IS_DUPLICATE) && reportable) {
      h.flags |= IS_REPORTED; th->nreported++;
      const bool includable = th->inc_by_E ? std::exp(h.lnP) <= th->incE : (double)h.score >= th->incT;  // p7_pli_TargetIncludable, :595
      if (includable) { h.flags |= IS_INCLUDED; th->nincluded++; }
    }
  }
  return BATH_OK;
}

extern "C" int bath_tophits_get(const bath_tophits *th, int64_t rank, bath_fs_domain *dom, int64_t *seqidx, int32_t *flags) {
  if (!th || rank < 0 || rank >= (int64_t)th->unsrt.size()) return BATH_EINVAL;
  const Hit &h = th->unsrt[(size_t)(th->order.empty() ? rank : th->order[(size_t)rank])];
  if (dom) *dom = h.d;
  if (seqidx) *seqidx = h.seqidx;
  if (flags) *flags = h.flags;
  return BATH_OK;
}

namespace {
void appendf(std::string &out, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void appendf(std::string &out, const char *fmt, ...) {
  char stackbuf[512];
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(stackbuf, sizeof stackbuf, fmt, ap);
  va_end(ap);
  if (n < (int)sizeof stackbuf) { out.append(stackbuf, (size_t)std::max(n, 0)); return; }
  std::vector<char> big((size_t)n + 1);
  va_start(ap, fmt);
  vsnprintf(big.data(), big.size(), fmt, ap);
  va_end(ap);
  out.append(big.data(), (size_t)n);
}
}  // namespace

// p7_tophits_TabularTargets, p7_tophits.c:1603-1729 (pli->spliced = FALSE).  Returns the table's size in bytes; copies at
// most <cap> of them into <buf>.
extern "C" int64_t bath_tophits_tabular_targets(const bath_tophits *th, const char *qname, const char *qacc, int M, int fs_pipe, int show_cigar,
                                                int show_header, char *buf, int64_t cap) {
  if (!th || !qname) return -1;
  const std::vector<Hit> &H = th->unsrt;
  size_t maxname = 0, maxacc = 0;
  int maxpos = 0;
  for (const Hit &h : H) {
    maxname = std::max(maxname, h.name.size());
    maxacc = std::max(maxacc, h.acc.size());
    if (h.d.iali > 0) {
      char b[32];
      maxpos = std::max(maxpos, snprintf(b, sizeof b, "%" PRId64, (int64_t)h.d.iali));
      maxpos = std::max(maxpos, snprintf(b, sizeof b, "%" PRId64, (int64_t)h.d.jali));
    }
  }
  const int qnamew = (int)std::max<size_t>(20, std::strlen(qname));
  const int tnamew = (int)std::max<size_t>(20, maxname);
  const int qaccw = qacc ? (int)std::max<size_t>(10, std::strlen(qacc)) : 10;
  const int taccw = (int)std::max<size_t>(10, maxacc);
  const int posw = std::max(9, maxpos);
  std::string out;
  if (show_header) {
    appendf(out, "#%7s %-*s %-*s %-*s %-*s %9s %9s %9s %9s %9s %9s", " hit ID", tnamew - 1, " target name", taccw, " accession", qnamew, " query name", qaccw,
            " accession", "  hmm len", " hmm from", "   hmm to", "  seq len", " ali from", "   ali to");
    appendf(out, "  %9s %6s %5s %5s", "  E-value", " score", " bias", "  PID");
    if (fs_pipe) appendf(out, " %7s %6s", " shifts", " stops");
    appendf(out, " %s\n", show_cigar ? "CIGAR" : " description of target");
    appendf(out, "#%7s %-*s %-*s %-*s %-*s %9s %9s %9s %9s %9s %9s", "-------", tnamew - 1, "-------------------", taccw, "----------", qnamew, "--------------------",
            qaccw, "----------", "---------", "---------", "---------", "---------", "---------", "---------");
    appendf(out, "  %9s %6s %5s %5s", "---------", "------", "-----", "-----");
    if (fs_pipe) appendf(out, " %7s %6s", "-------", "------");
    appendf(out, " %s\n", "---------------------");
  }
  int id = 0;
  const double kLog2R = 1.44269504088896341;
  for (size_t r = 0; r < H.size(); r++) {
    const Hit &h = H[(size_t)(th->order.empty() ? (int)r : th->order[r])];
    if (!(h.flags & IS_REPORTED)) continue;
    id++;
    appendf(out, "%8d %-*s %-*s %-*s %-*s %8d  %8d  %8d  %*" PRId64 " %*" PRId64 " %*" PRId64 "", id, tnamew, h.name.c_str(), taccw, h.acc.empty() ? "-" : h.acc.c_str(),
            qnamew, qname, qaccw, (qacc && qacc[0]) ? qacc : "-", M, h.d.ihmm, h.d.jhmm, posw, h.target_len, posw, (int64_t)h.d.iali, posw, (int64_t)h.d.jali);
    appendf(out, " %9.2g %6.1f %5.1f %5.2f", std::exp(h.lnP), h.score, h.d.dombias * kLog2R, h.d.pid);
    if (fs_pipe) appendf(out, " %7d %6d", h.d.n_shifted_codons, h.d.n_stops);
    if (show_cigar) appendf(out, " %s\n", h.cigar.c_str());
    else appendf(out, " %s\n", h.desc.empty() ? "-" : h.desc.c_str());
  }
  if (buf && cap > 0) std::memcpy(buf, out.data(), (size_t)std::min<int64_t>(cap, (int64_t)out.size()));
  return (int64_t)out.size();
}

// p7_tophits_Targets, p7_tophits.c:1073-1228 (search mode, no --acc, not spliced): the "Scores for complete hits" block of
// bathsearch's main output.  textw: the --textw line width (120 by default; <= 0: unlimited).  Returns the size in bytes.
extern "C" int64_t bath_tophits_targets(const bath_tophits *th, int fs_pipe, int textw, char *buf, int64_t cap) {
  if (!th) return -1;
  const std::vector<Hit> &H = th->unsrt;
  size_t maxname = 0;
  int maxpos = 0;
  for (const Hit &h : H) {
    maxname = std::max(maxname, h.name.size());
    if (h.d.iali > 0) {
      char b[32];
      maxpos = std::max(maxpos, snprintf(b, sizeof b, "%" PRId64, (int64_t)h.d.iali));
      maxpos = std::max(maxpos, snprintf(b, sizeof b, "%" PRId64, (int64_t)h.d.jali));
    }
  }
  const int namew = (int)std::max<size_t>(8, maxname), posw = std::max(6, maxpos);
  const int descw = textw > 0 ? std::max(32, textw - namew - 2 * posw - 32) : 0;
  std::string out;
  appendf(out, "Scores for complete hits:\n");
  if (fs_pipe) {
    appendf(out, "  %9s %6s %5s  %-*s %*s %*s  %6s  %5s  %s\n", "E-value", " score", " bias", namew, "Sequence", posw, "start", posw, "end", "shifts", "stops", "Description");
    appendf(out, "  %9s %6s %5s  %-*s %*s %*s  %6s  %5s  %s\n", "-------", "------", "-----", namew, "--------", posw, "-----", posw, "-----", "------", "-----", "-----------");
  } else {
    appendf(out, "  %9s %6s %5s  %-*s %*s %*s  %s\n", "E-value", " score", " bias", namew, "Sequence", posw, "start", posw, "end", "Description");
    appendf(out, "  %9s %6s %5s  %-*s %*s %*s  %s\n", "-------", "------", "-----", namew, "--------", posw, "-----", posw, "-----", "-----------");
  }
  const double kLog2R = 1.44269504088896341;
  bool printed_incthresh = false;
  for (size_t r = 0; r < H.size(); r++) {
    const Hit &h = H[(size_t)(th->order.empty() ? (int)r : th->order[r])];
    if (!(h.flags & IS_REPORTED)) continue;
    if (!(h.flags & IS_INCLUDED) && !printed_incthresh) { appendf(out, "  ------ inclusion threshold ------\n"); printed_incthresh = true; }
    appendf(out, "%c %9.2g %6.1f %5.1f  %-*s %*" PRId64 " %*" PRId64 "  ", ' ', std::exp(h.lnP), h.score, kLog2R * h.d.dombias, namew, h.name.c_str(), posw, (int64_t)h.d.iali, posw,
            (int64_t)h.d.jali);
    if (fs_pipe) appendf(out, "%6d  %5d", h.d.n_shifted_codons, h.d.n_stops);
    if (textw > 0) appendf(out, "  %-.*s\n", descw, h.desc.c_str());
    else appendf(out, "  %s\n", h.desc.c_str());
  }
  if (th->nreported == 0) appendf(out, "\n   [No hits detected that satisfy reporting thresholds]\n");
  if (buf && cap > 0) std::memcpy(buf, out.data(), (size_t)std::min<int64_t>(cap, (int64_t)out.size()));
  return (int64_t)out.size();
}

// The head of a hit's entry in "Annotation for each hit" (p7_tophits_Domains, p7_tophits.c:1256-1378; search mode, not spliced):
// the ">> name  description" line, the two header lines and the hit's line.  The alignment block that follows in the reference's
// output (p7_alidisplay_Print_BATH) is not produced.  <rank>: position in the current sort order; returns 0 for a hit that is
// not reported, else the size in bytes.
extern "C" int64_t bath_tophits_domain_annotation(const bath_tophits *th, int64_t rank, int M, int fs_pipe, char *buf, int64_t cap) {
  if (!th || rank < 0 || rank >= (int64_t)th->unsrt.size()) return -1;
  const Hit &h = th->unsrt[(size_t)(th->order.empty() ? rank : th->order[(size_t)rank])];
  if (!(h.flags & IS_REPORTED)) return 0;
  std::string out;
  appendf(out, ">> %s  %s\n", h.name.c_str(), h.desc.c_str());
  if (fs_pipe) {
    appendf(out, "   %6s %5s %9s %10s %9s    %9s %9s    %6s  %5s %9s   %4s\n", "score", "bias", "   Evalue", "hmm-from", " hmm-to", " ali-from", "   ali-to", "shifts", "stops", "   sq-len", "acc");
    appendf(out, "   %6s %5s %9s %10s %9s    %9s %9s    %6s  %5s %9s   %4s\n", "------", "-----", "---------", "--------", "-------", "---------", "---------", "------", "-----", "---------", "----");
  } else {
    appendf(out, "   %6s %5s %9s %10s %9s    %9s %9s    %9s   %4s\n", "score", "bias", "   Evalue", "hmm-from", " hmm-to", " ali-from", "   ali-to", "   sq-len", "acc");
    appendf(out, "   %6s %5s %9s %10s %9s    %9s %9s    %9s   %4s\n", "------", "-----", "---------", "--------", "-------", "---------", "---------", "---------", "----");
  }
  const double kLog2R = 1.44269504088896341;
  const double acc = h.d.oasc / (1.0 + std::fabs((float)(h.d.jenv - h.d.ienv) / 3));
  const char inc = (h.flags & IS_INCLUDED) ? '!' : '?';
  const char h1 = h.d.ihmm == 1 ? '[' : '.', h2 = h.d.jhmm == M ? ']' : '.';
  const char s1 = h.d.iali == 1 ? '[' : '.', s2 = (int64_t)h.d.jali == h.target_len ? ']' : '.';
  if (fs_pipe)
    appendf(out, " %c %6.1f %5.1f %9.2g %10d %9d %c%c %9" PRId64 " %9" PRId64 " %c%c %6d  %5d %9" PRId64 "   %4.2f\n", inc, h.d.bitscore, h.d.dombias * kLog2R, std::exp(h.d.lnP), h.d.ihmm,
            h.d.jhmm, h1, h2, (int64_t)h.d.iali, (int64_t)h.d.jali, s1, s2, h.d.n_shifted_codons, h.d.n_stops, h.target_len, acc);
  else
    appendf(out, " %c %6.1f %5.1f %9.2g %10d %9d %c%c %9" PRId64 " %9" PRId64 " %c%c %9" PRId64 "   %4.2f\n", inc, h.d.bitscore, h.d.dombias * kLog2R, std::exp(h.d.lnP), h.d.ihmm, h.d.jhmm, h1, h2,
            (int64_t)h.d.iali, (int64_t)h.d.jali, s1, s2, h.target_len, acc);
  if (buf && cap > 0) std::memcpy(buf, out.data(), (size_t)std::min<int64_t>(cap, (int64_t)out.size()));
  return (int64_t)out.size();
}

// p7_pli_Statistics (p7_pipeline.c:1836-1873) for a finished search: the "Internal pipeline statistics summary" block of the main
// output without its two timing lines.  n_output / pos_output are taken over the reported hits as bathsearch.c:950-957 does.
extern "C" int64_t bath_tophits_pipeline_statistics(const bath_tophits *th, const bath_pipeline_stats *st, const bath_pipeline_params *prm, int64_t nmodels,
                                                    int64_t nnodes, int64_t nseqs, char *buf, int64_t cap) {
  if (!th || !st || !prm) return -1;
  int64_t n_output = 0, pos_output = 0;
  for (const Hit &h : th->unsrt)
    if ((h.flags & IS_REPORTED) && !(h.flags & IS_DUPLICATE)) { n_output++; pos_output += 1 + (h.d.jali > h.d.iali ? h.d.jali - h.d.iali : h.d.iali - h.d.jali); }
  const double denom = (double)(st->nres * nmodels);
  std::string out;
  appendf(out, "Internal pipeline statistics summary:\n");
  appendf(out, "-------------------------------------\n");
  appendf(out, "Query model(s):              %15" PRId64 "  (%" PRId64 " nodes)\n", nmodels, nnodes);
  appendf(out, "Target %-12s          %15" PRId64 "  (%" PRId64 " residues searched)\n", "sequence(s):", nseqs, (int64_t)st->nres);
  appendf(out, "Residues passing SSV filter: %15" PRId64 "  (%.3g); expected (%.3g)\n", (int64_t)st->pos_past_msv, (double)st->pos_past_msv / denom, prm->F1);
  appendf(out, "Residues passing bias filter:%15" PRId64 "  (%.3g); expected (%.3g)\n", (int64_t)st->pos_past_bias, (double)st->pos_past_bias / denom, prm->F1);
  appendf(out, "Residues passing Vit filter: %15" PRId64 "  (%.3g); expected (%.3g)\n", (int64_t)st->pos_past_vit, (double)st->pos_past_vit / denom, prm->F2);
  appendf(out, "Residues passing Fwd filter: %15" PRId64 "  (%.3g); expected (%.3g)\n", (int64_t)st->pos_past_fwd, (double)st->pos_past_fwd / denom, prm->F3);
  appendf(out, "Total number of hits:        %15d  (%.3g)\n", (int)n_output, (double)pos_output / denom);
  if (buf && cap > 0) std::memcpy(buf, out.data(), (size_t)std::min<int64_t>(cap, (int64_t)out.size()));
  return (int64_t)out.size();
}
