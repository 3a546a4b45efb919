// bath_common.hpp -- internal types shared by the HIP translation units of libbathhip.
#pragma once
#include <set>
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "bath_hip.h"

namespace bath {

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised first.  The attribute
// belongs to the KERNEL, not to a launch: worker contexts on several host threads launch the same instantiation with different sizes,
// and a thread that lowered it between another thread's "set" and "launch" would make that launch exceed it.  So it is raised ONCE per
// (device, kernel), under a lock, to all the LDS a workgroup can have (160 KB on gfx950, less the kernel's static share), and never touched again.
inline hipError_t allow_max_lds(const void *fn) {
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  if (done.count({dev, fn})) return hipSuccess;
  hipFuncAttributes at{};
  size_t fixed = 0;
  if (hipFuncGetAttributes(&at, fn) == hipSuccess) fixed = at.sharedSizeBytes; else (void)hipGetLastError();
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - (int)std::min<size_t>(fixed, 96 * 1024)));
  if (e == hipSuccess) done.insert({dev, fn});
  return e;
}

constexpr int kKp = BATH_KP_AMINO;       // 29 amino symbols
constexpr int kRowReset = 29;            // extra "row" of the SSV cost table: every cost +127 (resets every diagonal)
constexpr int kSsvRows = 30;
constexpr int kStop = 27;                // '*'
constexpr int kXaa = 26;                 // 'X'
constexpr int kOrfBins = 2048;           // ORF length histogram of the work-list sort (longer ORFs share the last bin)

#define BATH_HIP_TRY(ctx, call)                                                            \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->set_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
      return BATH_EFAIL;                                                                   \
    }                                                                                      \
  } while (0)

// A growable device buffer.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// A growable pinned host buffer (D2H copies of matrices the host walks: page-locked memory copies at PCIe speed).
struct HostBuf {
  void *p = nullptr;
  size_t cap = 0;
  // coherent: fine-grained memory, for buffers a KERNEL writes and the host reads while the kernel is still running (device
  // stores go straight out instead of sitting in L2 until the end of the kernel; __threadfence_system() orders them)
  bool coherent = false;
  hipError_t reserve(size_t bytes, bool want_coherent = false) {
    if (bytes <= cap && (coherent || !want_coherent)) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    coherent = coherent || want_coherent;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipHostMalloc(&p, want, coherent ? (hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) : hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Host threads that stay alive across pipeline calls: one per concurrent part (lane) of a block.  run(k, f) has workers 0..k-1
// call f(worker) and returns when all have finished.
struct LanePool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> th;
  std::function<void(int)> fn;
  int gen = 0, pending = 0, K = 0;
  bool stop = false;
  void loop(int id) {
    int seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> l(m);
        cv_work.wait(l, [&] { return stop || (gen != seen && id < K); });
        if (stop) return;
        seen = gen; f = fn;
      }
      f(id);
      { std::lock_guard<std::mutex> l(m); if (--pending == 0) cv_done.notify_all(); }
    }
  }
  void run(int k, std::function<void(int)> f) {
    while ((int)th.size() < k) { const int id = (int)th.size(); th.emplace_back([this, id] { loop(id); }); }
    { std::lock_guard<std::mutex> l(m); fn = std::move(f); K = k; pending = k; gen++; }
    cv_work.notify_all();
    std::unique_lock<std::mutex> l(m);
    cv_done.wait(l, [&] { return pending == 0; });
  }
  ~LanePool() {
    { std::lock_guard<std::mutex> l(m); stop = true; }
    cv_work.notify_all();
    for (std::thread &t : th) t.join();
  }
};

// Worker contexts that run whole --fs passes on one GPU (the reference's worker threads, bathsearch.c:1119-1290) meet at the strict
// 3-codon parsers: a chain block (bath_fs_chain.hip) takes a CU's whole register file and most of its LDS, the Forward parser's
// launch takes nearly every CU, and whatever holds CUs when it starts pushes part of its blocks into a second round (a launch
// lasts as long as its longest window: a second round doubles it).  BATH_HIP_FS_GATE makes the turn-taking explicit, per device:
//   0 (default)  no gate: the hardware queues interleave the workers' kernels
//   1            one chain stage (Forward or Backward parser) at a time
//   2            one Forward parser at a time; the Backward parser (half the CUs or fewer) runs beside anything
//   3            the Forward parser alone on the chip: it waits for the other workers' cascades and envelope stages, and they for it
// Measured with the bench's --fs block (tools/fs_workers_probe.py), ms per block: DESIGN.md 4.6c.
inline std::shared_mutex &stage_gate_mutex(int device) { static std::shared_mutex g[16]; return g[device & 15]; }
inline int stage_gate_mode() { static const int m = [] { const char *e = std::getenv("BATH_HIP_FS_GATE"); return e ? std::atoi(e) : 0; }(); return m; }
struct StageGate {
  enum Kind { kFwdChain, kBwdChain, kCascade, kEnvelopes, kNone };
  std::unique_lock<std::shared_mutex> ex;
  std::shared_lock<std::shared_mutex> sh;
  StageGate(int device, Kind kind) {
    const int m = stage_gate_mode();
    if (device < 0 || m <= 0 || kind == kNone) return;
    const bool exclusive = (kind == kFwdChain) || (kind == kBwdChain && m == 1);
    const bool shared = m == 3 && (kind == kCascade || kind == kEnvelopes);
    if (exclusive) ex = std::unique_lock<std::shared_mutex>(stage_gate_mutex(device));
    else if (shared) sh = std::shared_lock<std::shared_mutex>(stage_gate_mutex(device));
  }
  void release() { if (ex.owns_lock()) ex.unlock(); if (sh.owns_lock()) sh.unlock(); }
};

struct StageTiming { const char *name; float ms; int64_t launches; };
// one kernel launch of the frameshift / domain stages, timed with HIP events on the stream it was launched on
struct KernelSpan { const char *name; hipEvent_t a, b; double cells, bytes; };

// An ORF that passed the Forward filter, as the domain-definition stage needs it (bath_domaindef.hip)
struct PipelineSurvivor {
  int64_t window;          // sequence index in the block
  int64_t aa_off;          // its residues in the amino-acid stream pool
  int32_t strand, start;   // start: first nucleotide, 1-based on the strand being read
  int32_t n;               // residues
  int32_t win_start = 0;   // windowsq->start on that strand: the ORF's own start, or the DNA window's in the --fs pipeline
  int32_t fs_window = -1;  // index of that DNA window in the frameshift stage's output
  int64_t fx_off = -1;     // where the cascade's Forward parser left this ORF's special-state rows (floats into ctx->fwd_rows_kept), -1: not kept
};

}  // namespace bath

struct bath_hip_ctx;
namespace bath {
// contexts alive in the process, and how many of them other contexts created for their own stages: the difference is what the host holds
extern std::atomic<int> g_ctx_total, g_ctx_internal;
inline int host_contexts() { return g_ctx_total.load() - g_ctx_internal.load(); }
inline void mark_internal(bath_hip_ctx *c);
}
struct bath_hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t tail_stream = nullptr;    // BATH_HIP_TAIL_PRIO=1 (experiment): the cascade's kernels after SSV, highest priority
  hipStream_t side_stream = nullptr;    // created on first use: kernels that may overlap the main stream's (pipeline)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t copy_stream = nullptr;    // created on first use: host -> device uploads of packed blocks (bath_hip_seqs_upload_packed)
  hipDeviceProp_t prop{};
  int fs_strict = 1;                    // frameshift log-sums along the model in the reference's serial order (bit-identical); bath_hip_set_fs_strict(ctx, 0): wavefront scans
  int fs_serial = -1;                   // envelopes' Backward after Forward on one stream instead of beside it (timing probes); -1: BATH_HIP_FS_SERIAL decides
  uint64_t tabs_uid = 0;                // whose SSV score table sits in scratch[8] (bath_pipeline.hip: uploaded once per profile, not per call)
  const void *tabs_ptr = nullptr;
  uint64_t fsw_pad_uid = 0;             // whose window padding fractions sit in scratch[52] (bath_fs_windows.hip)
  int orf_tables_id = -1;               // the NCBI table whose codon tables sit in scratch[28] (bath_orfs.hip: built and uploaded once per context and table)
  std::string err;
  void set_error(const std::string &m) { err = m; }
  // scratch owned by the context (reused across calls)
  bath::DevBuf scratch[60];             // [50]-[52]: the device-side DNA-window builder (bath_fs_windows.hip)
  // page-locked staging for small tables a launch uploads (job order, batch starts, offsets): an asynchronous copy from here needs
  // no synchronize before the local it was built in goes away.  One slot per call site; a site is reused by its context only
  // after the stage that used it has synchronized its stream.  [0] fs_schedule, [1]/[2] chain_batches (Forward / Backward), [3] wavefront Backward
  bath::HostBuf stage[12];                // ... [4] / [5]: the standard branch's domain stage, its small uploads / downloads (bath_domaindef.hip: std_domains)
  template <class T> int stage_upload(int slot, void *dst, const T *src, size_t n, hipStream_t s) {
    if (stage[slot].reserve(n * sizeof(T) + 64) != hipSuccess) { set_error("cannot allocate page-locked staging memory"); return BATH_EFAIL; }
    std::memcpy(stage[slot].p, src, n * sizeof(T));
    if (hipMemcpyAsync(dst, stage[slot].p, n * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) { set_error("staging upload failed"); return BATH_EFAIL; }
    return BATH_OK;
  }
  bath::HostBuf pinned[4];                // [0,1]: standard-branch region matrices; [2,3]: frameshift-branch ones (read by ensemble threads)
  bath::HostBuf results_pinned;           // bath_hip_pipeline_filters output: the ORF records, page-locked
  bath_orf_result *d_records = nullptr;   // ... as the last cascade pass left them on the device (bath_records.hip)
  int64_t n_records = 0;
  std::vector<bath_orf> orfs;             // bath_hip_translate_orfs output
  std::vector<bath_fs_window> fs_windows; // bath_hip_pipeline_frameshift output
  // The domain stage of the plain pipeline reads the Forward parser's special-state rows of the ORFs that passed F3 -- the matrix the
  // reference keeps in pli->oxf between the filter and p7_domaindef (p7_pipeline.c:1741-1771).  keep_fwd_rows: the cascade's Forward
  // launch writes them (one-lane blocks: the domain stage then copies the survivors' rows instead of running the parser again);
  // fwd_rows_kept / fwd_rows_off: where, by candidate, until the next cascade call of this context.
  bool keep_fwd_rows = false;
  const float *fwd_rows_kept = nullptr;
  const int64_t *fwd_rows_off = nullptr;
  bool fs_want_regions = false;           // set by the domain stage: the decision stage also runs the Backward parser and the region heuristics
  // Speculative Backward (bath_frameshift.hip: fs3_backward_spec): the 3-codon Backward parser of the LONGEST DNA windows runs beside the
  // Forward parser of all of them, before the branch is known -- the two parsers of a window do not read each other, and the pass lasts
  // as long as its longest window's Forward, then Backward, then regions' Forward.  fs_spec_rows[w] >= 0: window w's special-state rows
  // lie at that offset (floats) of scratch[53]; the domain stage runs the parser for the other frameshift-branch windows only.
  hipStream_t spec_stream = nullptr;
  hipEvent_t ev_spec = nullptr;
  bool fs_spec_valid = false;
  std::vector<int64_t> fs_spec_rows;
  std::vector<int64_t> fs_keep_xoff;      // the decision stage's Forward parser rows stay on the device (scratch[45]): offsets per DNA window, in floats
  std::vector<int32_t> fs_regions_all;    // ... for every DNA window: 1 + 3*fs_max_regions() ints each (bath_frameshift.hip: fs3_regions)
  std::vector<bath_fs_domain> fs_domains; // bath_hip_pipeline_frameshift_domains output
  std::string cigars;                     // NUL-terminated CIGAR strings of fs_domains (bath_fs_domain.cigar_off)
  // the domains' traces (P7_DOMAIN.tr, first to last match state) as the trace kernels left them, one record per entry of fs_domains;
  // bath_hip_domain_traces expands them into the reference's arrays on demand
  struct TraceRec { int64_t col_off; int32_t ncol, k1, i_first, win_start, orf_start, frameshift, d_i; };
  std::vector<TraceRec> tr_recs;
  std::vector<uint16_t> tr_codes;         // per column: state (3 M, 4 D, 5 I) | codon length << 4 | indel label << 8
  std::vector<float> tr_pp;               // per column: the state's posterior probability
  void traces_clear() { tr_recs.clear(); tr_codes.clear(); tr_pp.clear(); tr_valid = false; }
  void trace_push(const uint16_t *codes, const float *pp, int ncol, int k1, int i_first, int win_start, int orf_start, int frameshift, int d_i) {
    tr_recs.push_back(TraceRec{(int64_t)tr_codes.size(), ncol, k1, i_first, win_start, orf_start, frameshift, d_i});
    tr_codes.insert(tr_codes.end(), codes, codes + ncol);
    if (pp) tr_pp.insert(tr_pp.end(), pp, pp + ncol); else tr_pp.insert(tr_pp.end(), (size_t)ncol, 0.0f);
    tr_valid = false;
  }
  bool tr_valid = false;                  // tr_out .. tr_pp_out hold the expansion of tr_recs
  std::vector<bath_domain_trace> tr_out;
  std::vector<int8_t> tr_st, tr_c;
  std::vector<int32_t> tr_k, tr_i;
  std::vector<bath::PipelineSurvivor> fs_std_orfs;   // ORFs of the windows that take the standard branch (p7_pipeline.c:1479-1510)
  const uint8_t *fs_std_pool = nullptr;              // their residues: the amino-acid streams of the last cascade
  std::vector<uint8_t> orf_aa;
  std::vector<bath_hmm_window> hmm_windows;   // bath_hip_vitfilter_bath / bath_hip_ssvfilter_bath output
  std::vector<bath::StageTiming> timings;
  std::vector<hipEvent_t> ev_pool;
  // per-kernel device times of the stages after the cascade (bath_hip_kernel_times): spans recorded since the last reset
  std::vector<bath::KernelSpan> spans;
  std::vector<hipEvent_t> span_events;
  size_t span_events_used = 0;
  void spans_reset() { spans.clear(); span_events_used = 0; }
  int span_begin(const char *name, hipStream_t s, double cells, double bytes) {
    auto get = [&]() -> hipEvent_t {
      if (span_events_used == span_events.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; span_events.push_back(e); }
      return span_events[span_events_used++];
    };
    bath::KernelSpan k{name, get(), get(), cells, bytes};
    if (!k.a || !k.b) return -1;
    (void)hipEventRecord(k.a, s);
    spans.push_back(k);
    return (int)spans.size() - 1;
  }
  void span_end(int idx, hipStream_t s) { if (idx >= 0) (void)hipEventRecord(spans[(size_t)idx].b, s); }
  // worker lanes: contexts with their own stream and scratch, used by the pipeline to run parts of a block concurrently
  std::vector<bath_hip_ctx *> lanes;
  bath::LanePool *lane_pool = nullptr;   // the lanes' host threads, kept across calls
  hipEvent_t ev_lanes = nullptr;         // orders the lanes' streams after what the context's stream holds (uploads, expansion kernels)
  bath_hip_ctx *aux = nullptr;   // a context of its own (stream, scratch) for the standard-branch domains that run beside the frameshift branch
  bath_hip_ctx *aux2 = nullptr;  // ... and one for the multi-domain regions' Forward, which runs beside the first batch of envelopes (strict mode)
  bath_hip_ctx *aux3 = nullptr;  // ... and one for the clusters' envelopes, which run beside the tail of the single-domain batch (strict mode)
  bool internal = false;          // created by another context (a lane, the standard branch, the regions' Forward, the clusters' envelopes): not one of the host's own
};

namespace bath {
inline void mark_internal(bath_hip_ctx *c) { if (c && !c->internal) { c->internal = true; g_ctx_internal.fetch_add(1); } }
}

// Device view of a sequence block.
struct SeqView {
  const uint8_t *data;     // residues, each sequence 16-byte aligned
  const int64_t *off;      // [n]
  const int32_t *len;      // [n]
  int64_t n;
  const int32_t *context;  // [n] or null: leading nucleotides already searched with the previous window (ESL_SQ.C)
};

struct bath_hip_seqs {
  bath_hip_ctx *ctx = nullptr;
  int64_t n = 0;
  int64_t total = 0;       // total residues
  int64_t total_aligned = 0;   // bytes of d_data in use (every sequence padded to 16)
  mutable int64_t cache_minlen = -1, cache_nres = 0, cache_max_orfs = 0;   // pipeline sizing, per min_orf_len
  mutable int64_t ntiles = -1;             // translation tiles (bath_orfs.hip), built on first use
  mutable void *d_tile_desc = nullptr;     // int4 per tile: {window offset / 16, window length, window, tile index in the window}
  mutable int32_t *d_tile_first = nullptr;
  // parts of this block (consecutive windows, about equal in residues) for the pipeline's concurrent lanes: views into
  // d_data / d_len with their own rebased offsets; built on first use
  bool is_part = false;
  int64_t first_window = 0;
  mutable std::vector<bath_hip_seqs *> parts;
  int32_t maxlen = 0;
  uint8_t *d_data = nullptr;
  int64_t *d_off = nullptr;
  int32_t *d_len = nullptr;
  int32_t *d_context = nullptr; // [n] ESL_SQ.C of every window (bath_hip_seqs_set_context); null: all zero
  // streamed blocks (bath_hip_seqs_upload_packed): the 2-bit form as it arrives, its per-sequence byte offsets, the exceptions
  uint8_t *d_packed = nullptr;
  int64_t *d_poff = nullptr;
  int64_t packed_bytes = 0;
  void *d_exc = nullptr;        // {int64 sequence, int32 position, int32 code} records
  int64_t exc_cap = 0, n_exc = 0;
  hipEvent_t ev_upload = nullptr;
  bool upload_pending = false;
  std::vector<int64_t> h_off;   // device offsets (aligned)
  std::vector<int32_t> h_len;
  std::vector<int32_t> h_context;
  SeqView view() const { return SeqView{d_data, d_off, d_len, n, d_context}; }
};

// Per-length scalars of the limited-precision score systems (p7_oprofile_ReconfigLength, p7_oprofile.c:1261):
// computed on the host with the reference's exact libm calls, looked up by the kernels.
struct LenTables {
  int32_t maxL = -1;
  uint8_t *d_tjb = nullptr;     // [maxL+1] tjb_b(L)
  int16_t *d_xwmove = nullptr;  // [maxL+1] xw[N|C|J][MOVE](L)
  float   *d_pmove = nullptr;   // [maxL+1] xf[..][MOVE](L); LOOP = 1 - pmove
  float   *d_nullsc = nullptr;  // [maxL+1] p7_bg_NullOne for length L
  float   *d_lt1 = nullptr;     // [maxL+1] (float)L*logf(p1)      } the two length terms p7_bg_FilterScore adds, in order
  float   *d_lt2 = nullptr;     // [maxL+1] logf(1-p1)             } (p7_bg.c:501)
  float   *d_p1 = nullptr;      // [maxL+1] bg->p1 = L/(L+1)
  std::vector<uint8_t> h_tjb;
  std::vector<int16_t> h_xwmove;
  std::vector<float> h_pmove, h_nullsc;
};

struct bath_hip_oprofile {
  bath_hip_ctx *ctx = nullptr;
  uint64_t uid = 0;             // distinct for every profile ever created in the process (a context remembers whose per-call tables it holds)
  int M = 0, max_length = 0, L0 = 0;
  float nj = 1.f;
  float evparam[BATH_NEVPARAM];
  float compo[BATH_K_AMINO];
  // host copies, unstriped
  std::vector<uint8_t> rb;      // [Kp][M+1]
  std::vector<int16_t> rw;      // [Kp][M+1]
  std::vector<int16_t> tw;      // [M+1][8]
  std::vector<float> rf, tf;    // [Kp][M+1], [M+1][8]
  uint8_t tbm_b = 0, tec_b = 0, base_b = 0, bias_b = 0;
  float scale_b = 0;
  int16_t xw_E[2] = {0, 0};
  float scale_w = 0;
  int16_t base_w = 0, ddbound_w = 0;
  float xf_E[2] = {0, 0};
  std::string consensus;              // [M+2], ' ' where unset (bath_hip_oprofile_set_consensus)
  std::vector<uint8_t> cons_digital;  // [M+1] residue code of consensus[k], 255 = none
  uint8_t *d_cons = nullptr;
  std::vector<float> prefix_lengths, suffix_lengths;   // [M+1] P7_SCOREDATA window padding fractions (p7_scoredata.c:357-380)
  // device tables
  int NR = 0, G = 1;            // SSV kernel tile: NR registers (two cells each) per lane, G lanes per target (2*NR*G >= M)
  int ssv_row_bytes = 0;
  int16_t *d_ssv = nullptr;     // [kSsvRows][ssv_row_bytes/2] signed SSV costs (sf_conversion), +127 padding
  int16_t *d_msv = nullptr;     // the same layout with MSV's increments (bias - rb) * 2^-11, unclipped (bath_msv_lane.hip); models of one lane tile <= 76 registers
  uint8_t *d_rb = nullptr;      // [Kp][rb_stride]
  int rb_stride = 0;
  int16_t *d_rw = nullptr;      // [Kp][M+1]
  int16_t *d_tw = nullptr;      // [M+1][8]
  float *d_rf = nullptr, *d_tf = nullptr;
  float *d_rfb = nullptr, *d_tfb = nullptr;   // d_rf / d_tf padded to M+2 nodes (zeros), for the Backward kernel when the tables stay in global memory
  float *d_msc = nullptr, *d_tsc = nullptr;   // log-odds match scores [Kp][M+1] and log transitions [M][8] (p7_pli_computeAliScores_BATH)
  float *d_bias_eo = nullptr;   // [Kp][2] emission odds of the 2-state bias filter HMM for om->compo
  // lane-per-target Viterbi kernel tables (bath_viterbi.hip); vit_NR == 0 when the model is too long for it
  int vit_NR = 0, vit_rw_pitch = 0;
  int16_t *d_vit_rw = nullptr; uint32_t *d_vit_tw2 = nullptr; int16_t *d_vit_rank = nullptr;
  mutable LenTables lt;
  // The per-length tables and the emission thresholds grow on demand and the profile is shared by the clones of several host
  // threads (p7_oprofile_Clone): growth is serialised, and a replaced table is kept until the profile dies -- a kernel of
  // another thread may still be reading it.
  mutable std::mutex grow_mu;
  mutable std::vector<void *> retired;
  // SSV emission thresholds per ORF length for the pipeline (bath_pipeline.hip: build_emit_table), cached per F1
  mutable int16_t *d_emit = nullptr;
  mutable double emit_F1 = -1.0;
  mutable int emit_maxlen = -1;
  int ensure_len_tables(int maxL) const;
};

// implemented in bath_profile.hip
namespace bath {
int build_len_tables(const bath_hip_oprofile *om, int maxL);
void bias_filter_eo(const float *compo, float eo[kKp][2]);
double gumbel_surv(double x, double mu, double lambda);
double gumbel_invsurv(double p, double mu, double lambda);
double exp_surv(double x, double mu, double lambda);
struct StageClock {                         // BATH_HIP_TIMING=1: wall time of the host-visible stages, to stderr
  bool on; std::chrono::steady_clock::time_point t;
  StageClock() { const char *e = std::getenv("BATH_HIP_TIMING"); on = e && e[0] == '1'; t = std::chrono::steady_clock::now(); }
  void lap(const char *what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bath timing] %-34s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

}  // namespace bath
