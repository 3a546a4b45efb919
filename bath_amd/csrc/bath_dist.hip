// bath_dist.hip -- what a C host needs to run one search on several GPUs: the hit list as a byte stream, and the division of the
// work (host code; nothing here touches the GPU or a network: the host's own transport -- MPI, RCCL, sockets -- moves the bytes).
//
// Reference.  bathsearch's workers each fill a P7_TOPHITS and a P7_PIPELINE; the master merges them (p7_tophits_Merge,
// p7_pipeline_Merge; bathsearch.c:884-905).  Between processes a hit travels as p7_hit_Serialize / p7_hit_Deserialize write and
// read it (src/p7_hit.c:174-411, :411-640): a self-delimiting record -- its own size first, then every fixed-width field in NETWORK
// byte order, a byte of presence flags, the optional strings NUL-terminated -- followed by the hit's domain with its alignment
// (p7_domain_Serialize, p7_alidisplay_Serialize).  bath_hits_serialize follows that scheme for what a hit is on this path
// (bath_fs_domain + its --cigar string + optionally its trace, from which the alignment display is made); the layout is
// documented in include/bath_hip.h.
// Work division: the weighted cut of a multi-query job into (query, window group) items and their deal to the ranks, longest
// processing time first -- what bath_amd/dist.py did in Python until round 5 and now calls here, so that a C host and the Python
// harness split a job identically.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "bath_hip.h"

struct bath_hits {
  std::vector<bath_fs_domain> dom;
  std::string cigars;
  std::vector<bath_domain_trace> tr;
  std::vector<int8_t> st, c;
  std::vector<int32_t> k, i;
  std::vector<float> pp;
  bool has_traces = false;
};

namespace {

const uint32_t kMagic = 0x42484954u;        // "BHIT"
const uint32_t kVersion = 1;
enum { CIGAR_PRESENT = 1, TRACE_PRESENT = 2 };
// the fixed part of a hit: ser_size, window (i64), strand + fs_window + 6 coordinates (8 i32), 6 x f32, lnP (f64), reported /
// shifted / stops (3 i32), pid (f32), ali_columns (i32), 1 byte of presence flags
const uint32_t kHitBase = 4 + 8 + 8 * 4 + 6 * 4 + 8 + 3 * 4 + 4 + 4 + 1;

struct Writer {
  uint8_t *buf; int64_t cap, n = 0;
  void raw(const void *p, size_t len) { if (buf && n + (int64_t)len <= cap) std::memcpy(buf + n, p, len); n += (int64_t)len; }
  void u8(uint8_t v) { raw(&v, 1); }
  void u32(uint32_t v) { const uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; raw(b, 4); }   // esl_hton32
  void u64(uint64_t v) { u32((uint32_t)(v >> 32)); u32((uint32_t)v); }                                                                  // esl_hton64
  void i32(int32_t v) { u32((uint32_t)v); }
  void i64(int64_t v) { u64((uint64_t)v); }
  void f32(float v) { uint32_t x; std::memcpy(&x, &v, 4); u32(x); }
  void f64(double v) { uint64_t x; std::memcpy(&x, &v, 8); u64(x); }
};
struct Reader {
  const uint8_t *buf; int64_t len, n = 0; bool ok = true;
  bool need(int64_t k) { if (n + k > len) ok = false; return ok; }
  uint8_t u8() { if (!need(1)) return 0; return buf[n++]; }
  uint32_t u32() { if (!need(4)) return 0; const uint8_t *b = buf + n; n += 4; return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
  uint64_t u64() { const uint64_t hi = u32(); return (hi << 32) | u32(); }
  int32_t i32() { return (int32_t)u32(); }
  int64_t i64() { return (int64_t)u64(); }
  float f32() { const uint32_t x = u32(); float v; std::memcpy(&v, &x, 4); return v; }
  double f64() { const uint64_t x = u64(); double v; std::memcpy(&v, &x, 8); return v; }
};

}  // namespace

extern "C" int64_t bath_hits_serialize(const bath_fs_domain *dom, int64_t n, const char *cigars, const bath_domain_trace *tr,
                                       const int8_t *st, const int32_t *k, const int32_t *i, const int8_t *c, const float *pp,
                                       uint8_t *buf, int64_t cap) {
  if (n < 0 || (n > 0 && !dom) || (tr && (!st || !k || !i || !c || !pp))) return -1;
  Writer w{buf, buf ? cap : 0};
  w.u32(kMagic); w.u32(kVersion); w.u64((uint64_t)n);
  for (int64_t h = 0; h < n; h++) {
    const bath_fs_domain &d = dom[h];
    const char *cig = (cigars && d.cigar_off >= 0) ? cigars + d.cigar_off : nullptr;
    const uint32_t cig_size = cig ? (uint32_t)std::strlen(cig) + 1 : 0;
    const uint32_t tr_size = tr ? 16 + (uint32_t)tr[h].N * 14 : 0;
    const uint8_t flags = (uint8_t)((cig ? CIGAR_PRESENT : 0) | (tr ? TRACE_PRESENT : 0));
    w.u32(kHitBase + cig_size + tr_size);                                  // field 1: the record's own size, as p7_hit_Serialize writes it
    w.i64(d.window); w.i32(d.strand); w.i32(d.fs_window);
    w.i32(d.ienv); w.i32(d.jenv); w.i32(d.iali); w.i32(d.jali); w.i32(d.ihmm); w.i32(d.jhmm);
    w.f32(d.envsc); w.f32(d.oasc); w.f32(d.domcorrection); w.f32(d.dombias); w.f32(d.bitscore); w.f32(d.pre_score);
    w.f64(d.lnP);
    w.i32(d.reported); w.i32(d.n_shifted_codons); w.i32(d.n_stops);
    w.f32(d.pid); w.i32(d.ali_columns);
    w.u8(flags);
    if (cig) w.raw(cig, cig_size);
    if (tr) {
      const bath_domain_trace &t = tr[h];
      w.i32(t.N); w.i32(t.win_start); w.i32(t.orf_start); w.i32(t.frameshift);
      for (int32_t z = 0; z < t.N; z++) {
        const size_t q = (size_t)t.off + (size_t)z;
        w.u8((uint8_t)st[q]); w.u8((uint8_t)c[q]); w.i32(k[q]); w.i32(i[q]); w.f32(pp[q]);
      }
    }
  }
  return (buf && w.n > cap) ? -1 : w.n;
}

extern "C" int bath_hits_deserialize(const uint8_t *buf, int64_t nbytes, bath_hits **ret) {
  if (!buf || !ret || nbytes < 16) return BATH_EINVAL;
  *ret = nullptr;
  Reader r{buf, nbytes};
  if (r.u32() != kMagic || r.u32() != kVersion) return BATH_EFORMAT;
  const uint64_t n = r.u64();
  if (n > (uint64_t)(nbytes - 16) / kHitBase) return BATH_EFORMAT;          // every record has a fixed part: the stream cannot hold more
  bath_hits *H = new bath_hits();
  H->dom.reserve((size_t)n);
  for (uint64_t h = 0; h < n && r.ok; h++) {
    const int64_t start = r.n;
    const uint32_t size = r.u32();
    bath_fs_domain d{};
    d.window = r.i64(); d.strand = r.i32(); d.fs_window = r.i32();
    d.ienv = r.i32(); d.jenv = r.i32(); d.iali = r.i32(); d.jali = r.i32(); d.ihmm = r.i32(); d.jhmm = r.i32();
    d.envsc = r.f32(); d.oasc = r.f32(); d.domcorrection = r.f32(); d.dombias = r.f32(); d.bitscore = r.f32(); d.pre_score = r.f32();
    d.lnP = r.f64();
    d.reported = r.i32(); d.n_shifted_codons = r.i32(); d.n_stops = r.i32();
    d.pid = r.f32(); d.ali_columns = r.i32();
    const uint8_t flags = r.u8();
    d.cigar_off = -1;
    if (flags & CIGAR_PRESENT) {
      const char *s = reinterpret_cast<const char *>(buf + r.n);
      const void *z = r.ok ? std::memchr(s, 0, (size_t)(nbytes - r.n)) : nullptr;
      if (!z) { r.ok = false; break; }
      const size_t len = (size_t)((const char *)z - s);
      d.cigar_off = (int64_t)H->cigars.size();
      H->cigars.append(s, len); H->cigars.push_back('\0');
      r.n += (int64_t)len + 1;
    }
    bath_domain_trace t{(int64_t)H->st.size(), 0, 0, 0, 0};
    if (flags & TRACE_PRESENT) {
      H->has_traces = true;
      t.N = r.i32(); t.win_start = r.i32(); t.orf_start = r.i32(); t.frameshift = r.i32();
      if (t.N < 0 || !r.need((int64_t)t.N * 14)) { r.ok = false; break; }
      for (int32_t z = 0; z < t.N; z++) {
        H->st.push_back((int8_t)r.u8()); H->c.push_back((int8_t)r.u8()); H->k.push_back(r.i32()); H->i.push_back(r.i32()); H->pp.push_back(r.f32());
      }
    }
    H->tr.push_back(t);
    H->dom.push_back(d);
    if (r.n - start != (int64_t)size) r.ok = false;                        // consistency check, as p7_hit_Deserialize's
  }
  if (!r.ok || H->dom.size() != (size_t)n || r.n != nbytes) { delete H; return BATH_EFORMAT; }
  *ret = H;
  return BATH_OK;
}

// the size of the stream that starts at <buf> (several streams may lie back to back in one message): walks the records' size fields
extern "C" int64_t bath_hits_stream_size(const uint8_t *buf, int64_t nbytes) {
  if (!buf || nbytes < 16) return -1;
  Reader r{buf, nbytes};
  if (r.u32() != kMagic || r.u32() != kVersion) return -1;
  const uint64_t n = r.u64();
  for (uint64_t h = 0; h < n; h++) {
    const int64_t start = r.n;
    const uint32_t size = r.u32();
    if (!r.ok || size < kHitBase || start + (int64_t)size > nbytes) return -1;
    r.n = start + (int64_t)size;
  }
  return r.n;
}

extern "C" void bath_hits_destroy(bath_hits *h) { delete h; }
extern "C" int64_t bath_hits_count(const bath_hits *h) { return h ? (int64_t)h->dom.size() : 0; }
extern "C" bath_fs_domain *bath_hits_domains(bath_hits *h) { return h ? h->dom.data() : nullptr; }
extern "C" const char *bath_hits_cigars(const bath_hits *h, int64_t *nbytes) { if (nbytes) *nbytes = h ? (int64_t)h->cigars.size() : 0; return h ? h->cigars.c_str() : nullptr; }
extern "C" int bath_hits_traces(const bath_hits *h, const bath_domain_trace **tr, const int8_t **st, const int32_t **k, const int32_t **i, const int8_t **c, const float **pp) {
  if (!h || !h->has_traces) return BATH_EINVAL;
  if (tr) *tr = h->tr.data();
  if (st) *st = h->st.data();
  if (k) *k = h->k.data();
  if (i) *i = h->i.data();
  if (c) *c = h->c.data();
  if (pp) *pp = h->pp.data();
  return BATH_OK;
}

// p7_tophits_Merge from a byte stream: the hits of another rank join this list (their window indices must already be the
// search's own: the sender adds its shard's first window to bath_fs_domain.window before serializing, or passes window_shift here).
// The stream comes from another process: a hit whose shifted window is not one of the <n_seqs> sequences the name / length arrays
// describe is refused (BATH_EFORMAT) before anything indexes them, and nothing is added.
extern "C" int bath_tophits_add_serialized(bath_tophits *th, const uint8_t *buf, int64_t nbytes, int64_t window_shift, int64_t n_seqs, int64_t seqidx0,
                                           const char *const *seq_names, const char *const *seq_accs, const char *const *seq_descs, const int64_t *seq_lens) {
  if (!th || n_seqs < 0) return BATH_EINVAL;
  bath_hits *H = nullptr;
  const int st = bath_hits_deserialize(buf, nbytes, &H);
  if (st != BATH_OK) return st;
  for (bath_fs_domain &d : H->dom) {
    if (d.window < 0 || d.window > INT64_MAX - std::max<int64_t>(window_shift, 0)) { delete H; return BATH_EFORMAT; }
    d.window += window_shift;
    if (d.window < 0 || d.window >= n_seqs) { delete H; return BATH_EFORMAT; }
  }
  const int rc = bath_tophits_add(th, H->dom.data(), (int64_t)H->dom.size(), H->cigars.c_str(), seqidx0, seq_names, seq_accs, seq_descs, seq_lens);
  delete H;
  return rc;
}

// ---- work division ---------------------------------------------------------------------------------------------------------
// Contiguous shares of n units: rank r of <world> gets [lo, hi), the first n % world ranks one unit more (the block queue of
// bathsearch's threads hands out consecutive blocks; a static cut of the same list)
extern "C" void bath_dist_shard_range(int64_t n, int rank, int world, int64_t *lo, int64_t *hi) {
  if (world < 1 || rank < 0 || rank >= world || n < 0) {                    // no such share: an empty range, never a division by zero
    if (lo) *lo = 0;
    if (hi) *hi = 0;
    return;
  }
  const int64_t base = n / world, extra = n % world;
  const int64_t a = (int64_t)rank * base + std::min<int64_t>(rank, extra);
  if (lo) *lo = a;
  if (hi) *hi = a + base + (rank < extra ? 1 : 0);
}

// A multi-query job (bathsearch's loop over the queries of an HMM database, bathsearch.c:737-844) as (query, window group) items:
// every query's windows [0, n_q) cut into g_q consecutive groups, g_q = round(share_q x T) in [1, n_q], share_q = cost_q / sum of
// costs, T = max(queries, items_per_rank x world).  costs NULL: every query in G groups, G the smallest count that gives every rank
// about <items_per_rank> items (ceil(items_per_rank x world / queries)).  Returns the number of items (also when <cap> is smaller;
// only the first <cap> are written).  Deterministic: every rank computes the same list, no communication.
extern "C" int64_t bath_dist_items(const int64_t *n_windows_by_query, const double *costs_by_query, int n_queries, int world, int items_per_rank,
                                   bath_dist_item *items, int64_t cap) {
  if (!n_windows_by_query || n_queries < 0 || world < 1 || items_per_rank < 1) return -1;
  double total = 0.0;
  for (int q = 0; q < n_queries; q++) total += costs_by_query ? costs_by_query[q] : 1.0;
  if (!(total > 0.0)) total = 1.0;
  const int64_t T = std::max<int64_t>(n_queries, (int64_t)items_per_rank * world);
  int64_t n_items = 0;
  for (int q = 0; q < n_queries; q++) {
    const int64_t n = n_windows_by_query[q];
    const double share = (costs_by_query ? costs_by_query[q] : 1.0) / total;
    const int64_t G = std::max<int64_t>(1, ((int64_t)items_per_rank * world + std::max(n_queries, 1) - 1) / std::max(n_queries, 1));
    const int64_t g = std::max<int64_t>(1, std::min<int64_t>(costs_by_query ? (int64_t)(share * (double)T + 0.5) : G, n));
    for (int64_t kk = 0; kk < g; kk++) {
      int64_t lo, hi;
      bath_dist_shard_range(n, (int)kk, (int)g, &lo, &hi);
      if (hi <= lo) continue;
      if (items && n_items < cap) items[n_items] = bath_dist_item{q, lo, hi};
      n_items++;
    }
  }
  return n_items;
}

// Owner rank of every item: longest processing time first onto the least loaded rank (ties: the earlier item, the lowest rank)
extern "C" int bath_dist_deal(const double *costs, int64_t n_items, int world, int32_t *owner) {
  if (n_items < 0 || world < 1 || (n_items > 0 && (!costs || !owner))) return BATH_EINVAL;
  std::vector<int64_t> order((size_t)n_items);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return costs[a] > costs[b]; });
  std::vector<double> load((size_t)world, 0.0);
  for (int64_t x : order) {
    int best = 0;
    for (int r = 1; r < world; r++) if (load[(size_t)r] < load[(size_t)best]) best = r;
    owner[x] = best;
    load[(size_t)best] += costs[x];
  }
  return BATH_OK;
}
