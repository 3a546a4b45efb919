// bath_fs_windows.hip -- p7_pli_BuildDNAWindows and the per-window ORF summary of p7_pli_Frameshift ON THE DEVICE, and the
// branch decision that follows the 3-codon Forward parser (p7_pipeline.c:462-572, :1364-1415, :1425-1465).
//
// Rounds 1-5 built the DNA windows on the host: the ORFs that passed F4 and their hit windows came back over PCIe, were sorted
// and merged by host threads, the window descriptors went up again, and after the Forward parser the scores came back for a host
// loop of three exp() per window -- 2.4 + 1.0 ms of a 60 ms pass in which nothing ran on the GPU (profiles/r05_fs_pass_timeline.txt).
// Here the same steps are kernels on the context's stream, fed by what the cascade's lanes left in device memory:
//   merge     the lanes' F4 survivors and hit windows, ids made the block's own, sort keys formed
//   rank      the ORFs ordered by counting (keys are unique: rank = number of smaller keys; a few thousand records, tiled through
//             LDS, the comparisons spread over the chip) by (sequence, strand, the order esl_gencode emits a strand's ORFs)
//   hits      per hit window: its ORF's best window (highest score, then longest, then first) and k_min / k_max by atomics on the
//             ORF's slot -- the hit windows need no order of their own
//   orf       per ORF: its best hit window -> the DNA window it asks for (:486-527)
//   group     per (sequence, strand): windows ordered by start (p7_hmmwindow_SortByStart) ...
//   fuse      ... and overlapping ones fused, serially within the group as the reference does (:541-566)
//   summary   per fused window: the ORFs inside it -- count, k_min / k_max, the table log-sum of their Forward scores in the
//             reference's order, P_min, P_tot (:1376-1415, :1457)
//   layout    the pool offsets of the windows' copies and the sequence-block view the parsers read
//   decide    (after the parsers) null / bias / Forward scores -> P-values -> which branch each window takes (:1425-1465)
// The host reads the result ONCE (the window records with their branch, and the ordered ORF list it needs for the standard
// branch's ORF lists) through page-locked memory.  Same records as the host path (BATH_HIP_FS_WINDOWS_HOST=1 keeps that path for
// A/B; it is also the fallback for inputs this path does not take: more than 32768 surviving ORFs or 65536 hit windows in a block,
// or more than 1024 surviving ORFs on one strand of one sequence).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"
#include "bath_fs_device.hpp"

namespace bath {

namespace {

constexpr int kMaxOrfs = 32768, kMaxHitWins = 65536, kMaxGroup = 1024;
// BATH_HIP_FSW_MAX_GROUP=g (tests): groups of more than g ORFs send the block to the host path, so that the fallback runs on small inputs
static int max_group() { static const int g = [] { const char *e = std::getenv("BATH_HIP_FSW_MAX_GROUP"); return e ? std::max(1, std::min(std::atoi(e), kMaxGroup)) : kMaxGroup; }(); return g; }
struct Key { uint64_t hi, lo; };
__device__ __forceinline__ bool key_less(const Key &a, const Key &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

struct LaneArgs { FsLaneSurv l[16]; int n; int c_begin[17], w_begin[17]; };

// the lanes' survivors as one list with the block's own ids, and the key that orders the ORFs
__global__ void fsw_merge_kernel(LaneArgs L, const int32_t *__restrict__ seq_len, FsOrfDev *__restrict__ orfs, Key *__restrict__ okey,
                                 WindowRec *__restrict__ wins) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_c = L.c_begin[L.n], n_w = L.w_begin[L.n];
  if (t < n_c) {
    int k = 0;
    while (t >= L.c_begin[k + 1]) k++;
    const FsCandRec q = L.l[k].d_c[t - L.c_begin[k]];
    FsOrfDev o;
    o.w = q.window + L.l[k].first_window; o.aa_off = q.aa_off + L.l[k].dpool; o.P = q.P; o.cand = q.cand + L.l[k].cand_base;
    o.strand = q.sf / 3; o.n = q.len; o.start = q.sf % 3 + 3 * q.startj + 1; o.end = o.start + 3 * o.n - 1;
    o.fwd_null = q.fwdsc - q.nullsc;                                   // pli_tmp->fwdsc (p7_pipeline.c:1782)
    o.wb = -1; o.we = 0; o.dw_n = 0; o.dw_len = 0; o.dw_k = 0; o.kmin = 0x7fffffff; o.kmax = 0; o.g0 = o.g1 = 0; o.pad_ = 0;
    orfs[t] = o;
    // esl_gencode emits a strand's ORFs when their closing stop codon is read; ORFs still open at the end follow, frame by frame
    const int n_seq = seq_len[o.w];
    const uint64_t ea = (o.end + 3 > n_seq) ? 1 : 0;
    const uint64_t val = ea ? (uint64_t)((o.start - 1) % 3) : (uint64_t)o.end;
    okey[t] = Key{((uint64_t)o.w << 1) | (uint64_t)o.strand, (ea << 62) | (val << 31) | (uint64_t)(uint32_t)o.cand};
  }
  if (t < n_w) {
    int k = 0;
    while (t >= L.w_begin[k + 1]) k++;
    WindowRec w = L.l[k].d_w[t - L.w_begin[k]];
    w.cand += L.l[k].cand_base;
    wins[t] = w;
  }
}

// rank[i] += number of keys of this block's column range that are smaller than key i (the keys are pairwise different; rank zeroed
// before): the list's order without a sort network.  grid (rows of 256 keys, column ranges): the n^2 comparisons fill the chip.
__global__ __launch_bounds__(256) void fsw_rank_kernel(const Key *__restrict__ key, int n, int32_t *__restrict__ rank) {
  __shared__ Key tile[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const Key mine = i < n ? key[i] : Key{~0ull, ~0ull};
  const int ntiles = (n + 255) / 256;
  const int t0 = (int)((long long)ntiles * blockIdx.y / gridDim.y), t1 = (int)((long long)ntiles * (blockIdx.y + 1) / gridDim.y);
  int r = 0;
  for (int t = t0; t < t1; t++) {
    const int j0 = t * 256;
    __syncthreads();
    if (j0 + (int)threadIdx.x < n) tile[threadIdx.x] = key[j0 + threadIdx.x];
    __syncthreads();
    const int m = min(256, n - j0);
    for (int j = 0; j < m; j++) r += key_less(tile[j], mine) ? 1 : 0;
  }
  if (i < n && r) atomicAdd(&rank[i], r);
}

// after the ORFs are in order: slot_of[cand] = the ORF's place (slot_of is -1 elsewhere)
__global__ void fsw_slot_kernel(const FsOrfDev *__restrict__ orfs, int n, int32_t *__restrict__ slot_of) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n) slot_of[orfs[s].cand] = s;
}

// a hit window's rank among its ORF's windows as one integer: highest score first, then the longest, then the leftmost (:486-495)
__device__ __forceinline__ unsigned long long hit_key(const WindowRec &x) {
  uint32_t u = __float_as_uint(x.score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                     // floats in their order as unsigned integers
  return ((unsigned long long)u << 32) | ((unsigned long long)((uint32_t)x.length & 0xfffu) << 20) | (unsigned long long)(0xfffffu - ((uint32_t)x.n & 0xfffffu));
}
// per hit window, pass 1: the best key and k_min / k_max of its ORF; pass 2: among the windows that hold the best key, the first
__global__ void fsw_hits1_kernel(const WindowRec *__restrict__ wins, int nhw, const int32_t *__restrict__ slot_of, unsigned long long *__restrict__ best_key, FsOrfDev *__restrict__ orfs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nhw) return;
  const WindowRec x = wins[t];
  const int s = slot_of[x.cand];
  if (s < 0) return;
  atomicMax(&best_key[s], hit_key(x));
  atomicMin(&orfs[s].kmin, x.k - x.length + 1);
  atomicMax(&orfs[s].kmax, x.k);
}
__global__ void fsw_hits2_kernel(const WindowRec *__restrict__ wins, int nhw, const int32_t *__restrict__ slot_of, const unsigned long long *__restrict__ best_key, FsOrfDev *__restrict__ orfs) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nhw) return;
  const WindowRec x = wins[t];
  const int s = slot_of[x.cand];
  if (s < 0 || hit_key(x) != best_key[s]) return;
  atomicMin(reinterpret_cast<unsigned int *>(&orfs[s].wb), (unsigned int)t);          // wb starts at -1 = 0xffffffff: "no hit window"
}

template <class T>
__global__ void fsw_scatter_kernel(const T *__restrict__ in, const int32_t *__restrict__ rank, int n, T *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[rank[i]] = in[i];
}

struct BuildParams {
  int M, max_length;
  const float *prefix, *suffix;       // [M+1] P7_SCOREDATA window padding fractions
  double F3, ftau, flambda;
  int std_pipe;
  const float *tbl;                   // p7_FLogsum's table
};

// per ORF (in order): its best hit window says which DNA window the ORF asks for (p7_pipeline.c:486-527)
__global__ void fsw_orf_kernel(FsOrfDev *__restrict__ orfs, int n, const WindowRec *__restrict__ wins, const int32_t *__restrict__ seq_len, BuildParams p) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  FsOrfDev o = orfs[s];
  const int best = o.wb;
  int32_t cn, ck, cl;
  if (best >= 0) { cn = wins[best].n; ck = wins[best].k; cl = wins[best].length; o.we = best + 1; }
  else if (o.n >= p.M) { cn = (o.n - p.M) / 2 + 1; ck = p.M; cl = p.M; }            // :500-510: no window, centre of the model
  else { cn = 1; ck = p.M - ((p.M - o.n) / 2); cl = o.n; }
  if (best < 0) { o.wb = 0; o.we = 0; o.kmin = p.M; o.kmax = 0; }
  const int n_seq = seq_len[o.w];
  int64_t ws = (int64_t)((double)(uint32_t)cn - ((double)p.max_length * (0.1 + (double)p.prefix[ck - cl + 1])) + 1);     // :513
  int64_t wend = (int64_t)((double)((uint32_t)cn + (uint32_t)cl) + ((double)p.max_length * (0.1 + (double)p.suffix[ck])) - 2);   // :514
  ws = min((int64_t)0, ws);                                            // :516-517 (sic)
  wend = max((int64_t)o.n, wend);
  ws = max((int64_t)1, (int64_t)o.start + ws * 3);                     // :520-527, o.start already on the strand being read
  wend = min((int64_t)n_seq, (int64_t)o.start + wend * 3);
  o.dw_n = (int32_t)ws; o.dw_k = ck; o.dw_len = (int32_t)(wend - ws + 1);
  orfs[s] = o;
}

struct DnaWinDev { int32_t n, k, length; };

// per ORF: the bounds of its (sequence, strand) group, and its window's place in the group's list ordered by start
// (p7_hmmwindow_SortByStart; equal starts keep the ORFs' order)
__global__ void fsw_group_kernel(FsOrfDev *__restrict__ orfs, int n, DnaWinDev *__restrict__ sorted, int *__restrict__ flags, int group_cap) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int64_t w = orfs[s].w;
  const int strand = orfs[s].strand;
  int g0 = s, g1 = s + 1;
  while (g0 > 0 && orfs[g0 - 1].w == w && orfs[g0 - 1].strand == strand && s - g0 <= kMaxGroup) g0--;
  while (g1 < n && orfs[g1].w == w && orfs[g1].strand == strand && g1 - s <= kMaxGroup) g1++;
  if (g1 - g0 > group_cap) { atomicOr(flags, 1); return; }             // the host path takes such a block
  const int32_t my_n = orfs[s].dw_n;
  int r = 0;
  for (int b = g0; b < g1; b++) { const int32_t bn = orfs[b].dw_n; r += (bn < my_n || (bn == my_n && b < s)) ? 1 : 0; }
  sorted[g0 + r] = DnaWinDev{my_n, orfs[s].dw_k, orfs[s].dw_len};
  orfs[s].g0 = g0; orfs[s].g1 = g1;
}

// per group (its first ORF's thread): overlapping windows fused, left to right (:541-566, pct_overlap = 0); cnt[g0] = windows left
__global__ void fsw_fuse_kernel(const FsOrfDev *__restrict__ orfs, int n, DnaWinDev *__restrict__ wl, int32_t *__restrict__ cnt, int max_length) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int g0 = orfs[s].g0, g1 = orfs[s].g1;
  if (s != g0) { cnt[s] = 0; return; }
  int keep = g0;
  for (int i = g0 + 1; i < g1; i++) {
    DnaWinDev prev = wl[keep];
    const DnaWinDev cur = wl[i];
    const int64_t pe = (int64_t)prev.n + prev.length - 1, ce = (int64_t)cur.n + cur.length - 1;
    const int32_t ov = (int32_t)(min(pe, ce) - max((int64_t)prev.n, (int64_t)cur.n) + 1);
    const int64_t ms = min((int64_t)prev.n, (int64_t)cur.n), me = max(pe, ce);
    const int32_t ml = (int32_t)(me - ms + 1);
    if (((float)ov / (float)min(prev.length, cur.length) > 0.f) && ml < (2 * (max_length * 3))) { prev.n = (int32_t)ms; prev.length = ml; wl[keep] = prev; }
    else wl[++keep] = cur;
  }
  cnt[s] = keep - g0 + 1;
}

// exclusive prefix sums of cnt[0..n) by one block; total -> hdr[0]
__global__ __launch_bounds__(1024) void fsw_scan_kernel(const int32_t *__restrict__ cnt, int n, int32_t *__restrict__ base, int32_t *__restrict__ hdr) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(n, b + per);
  int s = 0;
  for (int i = b; i < e; i++) s += cnt[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = b; i < e; i++) { base[i] = run; run += cnt[i]; }
  if (threadIdx.x == 1023) hdr[0] = part[1023];
}

// per fused window: the summary of the ORFs that lie inside it (p7_pipeline.c:1376-1415), the record and the gather descriptor
__global__ void fsw_summary_kernel(const FsOrfDev *__restrict__ orfs, int n, const DnaWinDev *__restrict__ wl, const int32_t *__restrict__ cnt,
                                   const int32_t *__restrict__ base, const int64_t *__restrict__ seq_off, const int32_t *__restrict__ seq_len, BuildParams p,
                                   bath_fs_window *__restrict__ out, FsWinDev *__restrict__ desc, int32_t *__restrict__ grp) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int g0 = orfs[s].g0, g1 = orfs[s].g1;
  if (s - g0 >= cnt[g0]) return;
  const DnaWinDev dw = wl[s];
  const int64_t w = orfs[g0].w;
  const int strand = orfs[g0].strand;
  const int n_seq = seq_len[w];
  const int64_t dstart = strand ? n_seq : 1;                           // dnasq->start of a whole sequence
  const int64_t wstart = strand ? dstart - ((int64_t)dw.n + dw.length) : dstart + dw.n - 1;       // :1373-1374
  const int64_t wend = strand ? dstart - dw.n + 1 : wstart + dw.length - 1;
  int orf_cnt = 0, k_min = p.M, k_max = 0;
  float tot = -INFINITY;
  double P_min = INFINITY;
  for (int b = g0; b < g1; b++) {
    const FsOrfDev &o = orfs[b];
    int64_t os, oe;
    if (strand) { const int64_t rs = (int64_t)n_seq - o.start + 1, re = (int64_t)n_seq - o.end + 1; os = dstart - (n_seq - re + 1) + 1; oe = dstart - (n_seq - rs + 1) + 1; }
    else { os = dstart + o.start - 1; oe = dstart + o.end - 1; }
    if (!(os >= wstart && oe <= wend)) continue;                       // :1405
    P_min = fmin(P_min, o.P);
    tot = flogsum_g(tot, o.fwd_null, p.tbl);
    orf_cnt++;
    if (o.we > o.wb) { k_min = min(k_min, o.kmin); k_max = max(k_max, o.kmax); }
  }
  const int idx = base[g0] + (s - g0);
  bath_fs_window r;
  memset(&r, 0, sizeof r);
  r.window = w; r.strand = strand; r.n = dw.n; r.length = dw.length;
  r.orf_cnt = orf_cnt; r.k_min = k_min; r.k_max = k_max; r.tot_orfsc = tot; r.P_min = P_min;
  const double x = (double)tot / 0.69314718055994529;
  r.P_tot = p.std_pipe ? ((x < p.ftau) ? 1.0 : exp(-p.flambda * (x - p.ftau))) : 1.0;              // :1457; --fsonly: 1
  out[idx] = r;
  FsWinDev d;
  d.src_off = seq_off[w]; d.dst_off = 0; d.seq_n = n_seq; d.start = dw.n; d.len = dw.length; d.strand = strand; d.kmin = k_min; d.kmax = k_max;
  desc[idx] = d;
  grp[2 * idx] = g0; grp[2 * idx + 1] = g1;
}

// the pool offsets of the windows' copies (each padded to 16 bytes + 16), the view's off[] / len[], and the header:
// hdr = {windows, flags, longest window, -, pool bytes (2 ints), total nucleotides (2 ints)}
__global__ __launch_bounds__(1024) void fsw_layout_kernel(FsWinDev *__restrict__ desc, int32_t *__restrict__ hdr, int64_t *__restrict__ voff, int32_t *__restrict__ vlen) {
  __shared__ long long part[1024];
  __shared__ int pmax[1024];
  const int nw = hdr[0];
  const int per = (nw + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(nw, b + per);
  long long s = 0, tot = 0;
  int mx = 0;
  for (int i = b; i < e; i++) { const int len = desc[i].len; s += ((long long)len + 15) / 16 * 16 + 16; tot += len; mx = max(mx, len); }
  part[threadIdx.x] = s; pmax[threadIdx.x] = mx;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const long long v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    const int m = threadIdx.x >= d ? pmax[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v; pmax[threadIdx.x] = max(pmax[threadIdx.x], m);
    __syncthreads();
  }
  long long run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = b; i < e; i++) {
    const int len = desc[i].len;
    desc[i].dst_off = run; voff[i] = run; vlen[i] = len;
    run += ((long long)len + 15) / 16 * 16 + 16;
  }
  // total nucleotides: a second, tiny reduction through the same array
  __syncthreads();
  const long long pool = part[1023];
  const int longest = pmax[1023];
  __syncthreads();
  part[threadIdx.x] = tot;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) { if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) {
    hdr[2] = longest;
    memcpy(&hdr[4], &pool, 8);
    const long long total = part[0];
    memcpy(&hdr[6], &total, 8);
  }
}

// p7_pipeline.c:1425-1465 per window: null score (p7_bg_fs_NullOne), bias filter score (p7_bg_fs_FilterScore: the three frames'
// Forward scores of the 2-state filter HMM from fs_bias_kernel, summed with the table, plus the length term), Forward score ->
// P-values -> branch
__global__ void fsw_decide_kernel(bath_fs_window *__restrict__ out, int nw, const float *__restrict__ bias /* [nw][2][3] */, const float *__restrict__ fsc,
                                  const float *__restrict__ tbl, int do_biasfilter, int std_pipe, double F3, double tau3, double lambda) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nw) return;
  bath_fs_window r = out[i];
  const double kLn2 = 0.69314718055994529;
  const int L3 = r.length / 3;
  const float p1 = (float)L3 / (float)(L3 + 1);
  const float per_frame = (float)((float)L3 * log((double)p1) + log(1. - p1));                        // p7_bg_fs_NullOne, p7_bg.c:380
  r.nullsc = (float)(per_frame + log(3.0));
  if (do_biasfilter) {
    float f2[2];
    for (int pass = 0; pass < 2; pass++) {
      const float *b = &bias[((size_t)i * 2 + pass) * 3];
      float sum = -INFINITY;
      for (int f = 0; f < 3; f++) sum = flogsum_g(sum, b[f], tbl);
      f2[pass] = (float)((double)sum + ((double)((float)L3 * logf(p1) + logf((float)(1. - p1))) + log(3.0)));   // p7_bg.c:561
    }
    r.filtersc = f2[0];
    if (r.k_min <= r.k_max && f2[1] > r.filtersc) r.filtersc = f2[1];                                  // :1432-1440
  } else r.filtersc = r.nullsc;
  r.fwdsc = fsc[i];
  const float seqscore = (float)((r.fwdsc - r.filtersc) / kLn2);
  const double x1 = (double)seqscore, x2 = (r.fwdsc - r.nullsc) / kLn2;
  r.P_fs = (x1 < tau3) ? 1.0 : exp(-lambda * (x1 - tau3));
  r.P_null = (x2 < tau3) ? 1.0 : exp(-lambda * (x2 - tau3));
  if (r.P_fs <= F3 && (r.P_null < r.P_tot || (r.P_null == r.P_tot && r.orf_cnt > 1) || r.P_min > F3)) r.branch = 1;
  else r.branch = std_pipe ? 2 : 0;                                                                    // :1480: --fsonly has no standard branch
  out[i] = r;
}

}  // namespace

bool fs_windows_on_device() {
  static const bool host = [] { const char *e = std::getenv("BATH_HIP_FS_WINDOWS_HOST"); return e && e[0] == '1'; }();
  return !host;
}

int fs_build_windows_device(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3, const bath_hip_seqs *dna,
                            const bath_pipeline_params *prm, const FsLaneSurv *lanes, int nlanes, int nc_total, FsWinBuild *B) {
  *B = FsWinBuild{};
  if (nlanes > 16) return BATH_ENORESULT;
  LaneArgs L{};
  L.n = nlanes;
  for (int k = 0; k < nlanes; k++) { L.l[k] = lanes[k]; L.c_begin[k + 1] = L.c_begin[k] + lanes[k].n_c; L.w_begin[k + 1] = L.w_begin[k] + lanes[k].n_w; }
  const int n = L.c_begin[nlanes], nhw = L.w_begin[nlanes];
  if (n == 0) return BATH_OK;
  if (n > kMaxOrfs || nhw > kMaxHitWins) return BATH_ENORESULT;
  const int M = om->M;
  // the window padding fractions of P7_SCOREDATA, once per profile and context
  DevBuf &b_pad = ctx->scratch[52];
  static_assert(sizeof(FsOrfDev) % 8 == 0 && sizeof(bath_fs_window) % 8 == 0 && sizeof(FsWinDev) % 8 == 0, "records are laid out back to back");
  if (ctx->fsw_pad_uid != om->uid || !b_pad.p) {
    BATH_HIP_TRY(ctx, b_pad.reserve((size_t)(M + 1) * 2 * sizeof(float) + 64));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(b_pad.p, om->prefix_lengths.data(), (size_t)(M + 1) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(b_pad.as<float>() + (M + 1), om->suffix_lengths.data(), (size_t)(M + 1) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->fsw_pad_uid = om->uid;
  }
  // workspace: the unordered ORFs, their keys and ranks, the hit windows, candidate -> slot, best keys, the groups' window lists,
  // counts.  Results, two regions: A = what the host needs BEFORE it can launch the parsers (header, the windows' pool offsets and
  // lengths), B = what it reads after the branch decision (ordered ORFs, group bounds; the records travel then, completed).
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_orf0 = 0, o_okey = o_orf0 + al((size_t)n * sizeof(FsOrfDev)), o_orank = o_okey + al((size_t)n * sizeof(Key)),
               o_win = o_orank + al((size_t)n * 4), o_slot = o_win + al((size_t)std::max(nhw, 1) * sizeof(WindowRec)), o_best = o_slot + al((size_t)std::max(nc_total, 1) * 4),
               o_wl = o_best + al((size_t)n * 8), o_cnt = o_wl + al((size_t)n * sizeof(DnaWinDev)), o_base = o_cnt + al((size_t)n * 4), ws_bytes = o_base + al((size_t)n * 4);
  const size_t r_hdr = 0, r_voff = 256, r_vlen = r_voff + al((size_t)n * 8), a_bytes = r_vlen + al((size_t)n * 4),
               r_orf = a_bytes, r_grp = r_orf + al((size_t)n * sizeof(FsOrfDev)), b_end = r_grp + al((size_t)n * 8),
               r_out = b_end, r_desc = r_out + al((size_t)n * sizeof(bath_fs_window)), res_bytes = r_desc + al((size_t)n * sizeof(FsWinDev));
  DevBuf &b_ws = ctx->scratch[50], &b_res = ctx->scratch[51];
  BATH_HIP_TRY(ctx, b_ws.reserve(ws_bytes + 256));
  BATH_HIP_TRY(ctx, b_res.reserve(res_bytes + 256));
  if (ctx->stage[6].reserve(b_end + 256) != hipSuccess) { ctx->set_error("cannot allocate page-locked staging memory"); return BATH_EFAIL; }
  char *ws = b_ws.as<char>(), *rs = b_res.as<char>();
  FsOrfDev *d_orf0 = reinterpret_cast<FsOrfDev *>(ws + o_orf0), *d_orf = reinterpret_cast<FsOrfDev *>(rs + r_orf);
  Key *d_okey = reinterpret_cast<Key *>(ws + o_okey);
  int32_t *d_orank = reinterpret_cast<int32_t *>(ws + o_orank), *d_slot = reinterpret_cast<int32_t *>(ws + o_slot);
  WindowRec *d_win = reinterpret_cast<WindowRec *>(ws + o_win);
  unsigned long long *d_best = reinterpret_cast<unsigned long long *>(ws + o_best);
  DnaWinDev *d_wl = reinterpret_cast<DnaWinDev *>(ws + o_wl);
  int32_t *d_cnt = reinterpret_cast<int32_t *>(ws + o_cnt), *d_base = reinterpret_cast<int32_t *>(ws + o_base);
  int32_t *d_hdr = reinterpret_cast<int32_t *>(rs + r_hdr), *d_grp = reinterpret_cast<int32_t *>(rs + r_grp);
  bath_fs_window *d_out = reinterpret_cast<bath_fs_window *>(rs + r_out);
  FsWinDev *d_desc = reinterpret_cast<FsWinDev *>(rs + r_desc);
  int64_t *d_voff = reinterpret_cast<int64_t *>(rs + r_voff);
  int32_t *d_vlen = reinterpret_cast<int32_t *>(rs + r_vlen);
  hipStream_t s = ctx->stream;
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_hdr, 0, 256, s));
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_orank, 0, (size_t)n * 4, s));
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_slot, 0xff, (size_t)std::max(nc_total, 1) * 4, s));
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_best, 0, (size_t)n * 8, s));
  const BuildParams p{M, om->max_length, b_pad.as<float>(), b_pad.as<float>() + (M + 1), prm->F3, (double)om->evparam[BATH_FTAU], (double)om->evparam[BATH_FLAMBDA],
                      prm->std_pipe, om_fs3->d_logsum};
  const int T = 256, gn = (n + T - 1) / T, gm = (std::max(n, nhw) + T - 1) / T, gh = (nhw + T - 1) / T;
  const int cols = std::max(1, std::min(gn, (ctx->prop.multiProcessorCount * 4) / std::max(gn, 1)));      // column ranges: about four blocks per CU in all
  hipLaunchKernelGGL(fsw_merge_kernel, dim3(gm), dim3(T), 0, s, L, dna->d_len, d_orf0, d_okey, d_win);
  hipLaunchKernelGGL(fsw_rank_kernel, dim3(gn, cols), dim3(256), 0, s, d_okey, n, d_orank);
  hipLaunchKernelGGL(fsw_scatter_kernel<FsOrfDev>, dim3(gn), dim3(T), 0, s, d_orf0, d_orank, n, d_orf);
  if (nhw > 0) {
    hipLaunchKernelGGL(fsw_slot_kernel, dim3(gn), dim3(T), 0, s, d_orf, n, d_slot);
    hipLaunchKernelGGL(fsw_hits1_kernel, dim3(gh), dim3(T), 0, s, d_win, nhw, d_slot, d_best, d_orf);
    hipLaunchKernelGGL(fsw_hits2_kernel, dim3(gh), dim3(T), 0, s, d_win, nhw, d_slot, d_best, d_orf);
  }
  hipLaunchKernelGGL(fsw_orf_kernel, dim3(gn), dim3(T), 0, s, d_orf, n, d_win, dna->d_len, p);
  hipLaunchKernelGGL(fsw_group_kernel, dim3(gn), dim3(T), 0, s, d_orf, n, d_wl, d_hdr + 1, max_group());
  hipLaunchKernelGGL(fsw_fuse_kernel, dim3(gn), dim3(T), 0, s, d_orf, n, d_wl, d_cnt, om->max_length);
  hipLaunchKernelGGL(fsw_scan_kernel, dim3(1), dim3(1024), 0, s, d_cnt, n, d_base, d_hdr);
  hipLaunchKernelGGL(fsw_summary_kernel, dim3(gn), dim3(T), 0, s, d_orf, n, d_wl, d_cnt, d_base, dna->d_off, dna->d_len, p, d_out, d_desc, d_grp);
  hipLaunchKernelGGL(fsw_layout_kernel, dim3(1), dim3(1024), 0, s, d_desc, d_hdr, d_voff, d_vlen);
  BATH_HIP_TRY(ctx, hipGetLastError());
  // region A now (the one synchronize of this stage), region B behind it on the same stream: it is complete long before the
  // decision's synchronize, which is the next time the host looks
  char *h = static_cast<char *>(ctx->stage[6].p);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h, rs, a_bytes, hipMemcpyDeviceToHost, s));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(s));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h + a_bytes, rs + a_bytes, b_end - a_bytes, hipMemcpyDeviceToHost, s));
  const int32_t *hdr = reinterpret_cast<const int32_t *>(h + r_hdr);
  if (hdr[1] != 0) { BATH_HIP_TRY(ctx, hipStreamSynchronize(s)); return BATH_ENORESULT; }      // a group beyond kMaxGroup: the host path
  B->n_orfs = n; B->nw = hdr[0]; B->maxlen = hdr[2];
  std::memcpy(&B->pool_bytes, &hdr[4], 8); std::memcpy(&B->total, &hdr[6], 8);
  B->h_orfs = reinterpret_cast<const FsOrfDev *>(h + r_orf); B->h_grp = reinterpret_cast<const int32_t *>(h + r_grp);
  B->h_voff = reinterpret_cast<const int64_t *>(h + r_voff); B->h_vlen = reinterpret_cast<const int32_t *>(h + r_vlen);
  B->d_out = d_out; B->d_desc = d_desc; B->d_voff = d_voff; B->d_vlen = d_vlen;
  return BATH_OK;
}

int fs_decide_device(bath_hip_ctx *ctx, const bath_hip_fsprofile *om_fs3, const bath_pipeline_params *prm, const FsWinBuild &B, const float *d_bias,
                     const float *d_fsc, bath_fs_window *h_out) {
  if (B.nw == 0) return BATH_OK;
  const float *ev3 = om_fs3->evparam;
  hipLaunchKernelGGL(fsw_decide_kernel, dim3((unsigned)((B.nw + 255) / 256)), dim3(256), 0, ctx->stream, B.d_out, B.nw, d_bias, d_fsc, om_fs3->d_logsum,
                     prm->do_biasfilter, prm->std_pipe, prm->F3, (double)ev3[BATH_FTAUFS3], (double)ev3[BATH_FLAMBDA]);
  BATH_HIP_TRY(ctx, hipGetLastError());
  if (ctx->stage[7].reserve((size_t)B.nw * sizeof(bath_fs_window) + 64) != hipSuccess) { ctx->set_error("cannot allocate page-locked staging memory"); return BATH_EFAIL; }
  BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[7].p, B.d_out, (size_t)B.nw * sizeof(bath_fs_window), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(h_out, ctx->stage[7].p, (size_t)B.nw * sizeof(bath_fs_window));
  return BATH_OK;
}

}  // namespace bath
