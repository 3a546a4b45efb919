// bath_fs_device.hpp -- device helpers shared by the frameshift kernels (bath_frameshift.hip, bath_fs_wavefront.hip):
// p7_FLogsum with its table in LDS (logsum.c:105), DPP lane moves, the longest-first job queue, the device profile.
#pragma once
#include <algorithm>
#include <vector>
#include <cmath>
#include <mutex>
#include <vector>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

struct bath_hip_fsprofile {
  bath_hip_ctx *ctx = nullptr;
  int M = 0, codon_lengths = 0, maxcodons = 0, max_length = 0;
  int pitch = 0;                 // floats per emission row (M+1 rounded up to 4)
  float fsprob = 0.f;
  float evparam[BATH_NEVPARAM];
  float *d_rsc = nullptr;        // [(maxcodons+Kp)][pitch]
  float *d_tf = nullptr;         // [(M+2)][8] forward-ordered transitions per node
  float *d_tb = nullptr;         // [(M+2)][8] backward-ordered transitions per node
  float *d_logsum = nullptr;     // [16000]
  std::vector<float> h_tsc;      // [M*8] generic log transitions (OA traceback deltas on the host)
  std::vector<uint8_t> h_codons; // [(M+1)*maxcodons] best amino acid per (node, quasi-codon) (null2 along a trace)
  uint8_t *d_codons = nullptr;   // the same on the device (5-codon profiles)
  uint8_t *d_indel = nullptr;    // [(M+1)*maxcodons] indel-type label of that choice (hmmer.h:259-276), for the alignment display
  // length model: xsc[N|C|J][LOOP|MOVE] for L_amino, multihit (nj=1) and unihit (nj=0); host libm log()
  // (the tables grow under <grow_mu>; the old ones are retired, not freed: the profile is shared by contexts running side by side --
  // the regions' Forward beside the envelopes, worker contexts -- and another stream's kernel may still be reading them)
  mutable int maxL = -1;
  mutable float *d_loop[2] = {nullptr, nullptr}, *d_move[2] = {nullptr, nullptr};
  mutable std::mutex grow_mu;
  mutable std::vector<void *> retired;
  int ensure_len(int maxL_amino) const;
};

namespace bath {

constexpr int kLogsumTbl = 16000;
// threads per block of the DP kernels that hold p7_FLogsum's 64 KB table in LDS: 8 waves share one copy, so two blocks = 16 waves
// fit a CU (4 per SIMD); with 4 waves per block the table limited a CU to 8 waves, and these kernels live on latency hiding
#ifndef BATH_FS_BLOCK
#define BATH_FS_BLOCK 512
#endif
#ifndef BATH_FS_WAVES           /* waves per SIMD the compiler must leave room for when a lane holds <= 3 nodes (0: no constraint) */
#define BATH_FS_WAVES 4
#endif
constexpr int kFsBlock = BATH_FS_BLOCK;
// Two 512-thread blocks (one 64 KB table each) fit a CU's LDS: 4 waves per SIMD if a wave keeps to 128 VGPRs.  Left alone the
// compiler takes 134-161 for the straight-line rows (one block per CU, 2 waves per SIMD); told to stay within 128 it spills
// 7-39 registers.  Measured on the bench's --fs pass (tools/fs_variants.sh): the parsers and Backward gain from the
// cap (fs3_fwd 9.5 -> 7.6 ms, fs_bwd<3> 7.1 -> 5.4, fs_bwd<5> 6.0 -> 5.4), the 5-codon Forward loses (3.7 -> 5.4: 39 spills
// in its row chain) and is left uncapped.  Models with more than 3 nodes per lane need the registers.
constexpr int fs_min_waves(int C) { return (C <= 3 && BATH_FS_WAVES > 0) ? BATH_FS_WAVES : 1; }

struct FsDev {
  int M, pitch, maxcodons;
  const float *rsc, *tf, *tb, *logsum;
};

// p7_FLogsum (logsum.c:105-111): truncating table lookup, or the exact form (logsum.c:109)
// Straight-line code: the table is read unconditionally at a clamped index and the early-out cases are a select.  With the
// obvious `if (...) return mx;` every log-sum became its own exec-masked basic block (165 branches in the 3-codon Forward
// kernel) and the compiler could not overlap the independent log-sums of a lane's nodes; the values are identical.
// v_max_f32 as the instruction: fmaxf() first quiets each operand the compiler cannot prove canonical (a loaded value, a DPP
// move) with a v_max_f32 x, x of its own -- two of the ten instructions of a table log-sum, in kernels that are chains of them.
// The operands here are never NaN (scores are finite or -inf).
__device__ __forceinline__ float vmax_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool EXACT>
__device__ __forceinline__ float flogsum(float a, float b, const float *tbl) {
  if (EXACT) {
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    if (mn == -INFINITY || (mx - mn) >= 15.7f) return mx;
    return mx + log1pf(expf(mn - mx));
  }
  // <tbl> is the kernels' LDS copy of the table, ZERO from entry 15700 on (fs_load_logsum_table): the reference's early outs
  // "mn == -inf or mx - mn >= 15.7 -> mx" are then the look-up itself (mx + 0), and a log-sum is max, |a - b|, min, mul, cvt,
  // shift, ds_read, add.  (int)(d * 1000.f) >= 15700 exactly when d >= 15.7f: 15.7f * 1000.f rounds to 15700.0f and the float
  // below 15.7f to 15699.999.  a = b = -inf: |NaN| -> v_min returns 15.999 -> -inf + 0.
  const float dc = fminf(fabsf(a - b), 15.999f);
  return vmax_raw(a, b) + tbl[(int)(dc * 1000.f)];
}

// the same on the unpadded table in global memory (kernels that take a few log-sums per target)
__device__ __forceinline__ float flogsum_g(float a, float b, const float *tbl) {
  const float mx = fmaxf(a, b), mn = fminf(a, b);
  const float d = mx - mn;                                    // +inf when mn = -inf, NaN when both are
  const float dc = fminf(d, 15.999f);                         // (v_min_f32 returns the number when one operand is NaN)
  const float t = tbl[(int)(dc * 1000.f)];
  return (mn == -INFINITY || d >= 15.7f) ? mx : mx + t;
}

// p7_FLogsum's table into LDS, its entries for differences >= 15.7 (which the reference never reads) zeroed
__device__ __forceinline__ void fs_load_logsum_table(float *s_tbl, const float *g_tbl) {
  for (int i = threadIdx.x; i < 16000; i += blockDim.x) s_tbl[i] = (i < 15700) ? g_tbl[i] : 0.f;
}

// Cross-lane moves by DPP instead of ds_bpermute (__shfl_*): these kernels are chains of dependent operations, and a shuffle through
// the LDS crossbar costs ~100+ cycles of that chain where a DPP operand costs a VALU instruction.  <old> is what a lane without a
// source keeps: the identity of the combining operation, so that such lanes need no select.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_shr1(float v, float fill) { return dpp_f<0x138>(v, fill); }          // lane l <- lane l-1, lane 0 <- fill
__device__ __forceinline__ float wave_bcast_last(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); }

// log-sum of a value per lane, result in every lane: an inclusive scan in lane order (row_shr 1/2/4/8, row_bcast 15/31), lane 63
// broadcast.  BATH_FS_BPERMUTE: the xor butterfly through ds_bpermute this replaced.
template <bool EXACT>
__device__ __forceinline__ float wave_logsum(float v, const float *tbl) {
#ifdef BATH_FS_BPERMUTE
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = flogsum<EXACT>(v, __shfl_xor(v, d, 64), tbl);
  return v;
#else
  v = flogsum<EXACT>(v, dpp_f<0x111>(v, -INFINITY), tbl);
  v = flogsum<EXACT>(v, dpp_f<0x112>(v, -INFINITY), tbl);
  v = flogsum<EXACT>(v, dpp_f<0x114>(v, -INFINITY), tbl);
  v = flogsum<EXACT>(v, dpp_f<0x118>(v, -INFINITY), tbl);
  v = flogsum<EXACT>(v, dpp_f<0x142, 0xa>(v, -INFINITY), tbl);
  v = flogsum<EXACT>(v, dpp_f<0x143, 0xc>(v, -INFINITY), tbl);
  return wave_bcast_last(v);
#endif
}

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// Work distribution of the wave-per-window kernels.  Their duration is the longest chain of rows any one wave walks, so the
// windows are handed out longest first from a shared counter (<order> lists them by decreasing length): a wave that drew a
// long window early draws fewer later, instead of every wave taking windows wid, wid + nwaves, ... whatever their lengths.
// Sequences 0..n-1 by decreasing length, equal lengths in index order (what std::stable_sort gives): a counting sort -- the launchers
// sort every batch of DNA windows or envelopes once or twice per stage while the GPU waits (7.6 k windows: 0.2-0.3 ms per std::sort,
// ~20 us this way).  <sorted_len>, if not null, receives the lengths in that order.
inline void fs_order_by_length_desc(const int32_t *len, int64_t n, std::vector<int32_t> *order, std::vector<int32_t> *sorted_len = nullptr) {
  int32_t mx = 0;
  for (int64_t i = 0; i < n; i++) mx = std::max(mx, len[i]);
  if (order) order->resize((size_t)n);
  if (sorted_len) sorted_len->resize((size_t)n);
  if (n == 0) return;
  if ((int64_t)mx > 8 * n + 4096) {                              // a few long sequences: the comparison sort is the cheaper one
    std::vector<int32_t> ord((size_t)n);
    for (int64_t i = 0; i < n; i++) ord[(size_t)i] = (int32_t)i;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return len[a] > len[b]; });
    if (sorted_len) for (int64_t i = 0; i < n; i++) (*sorted_len)[(size_t)i] = len[ord[(size_t)i]];
    if (order) order->swap(ord);
    return;
  }
  std::vector<int32_t> start((size_t)mx + 2, 0);
  for (int64_t i = 0; i < n; i++) start[(size_t)(mx - std::max(len[i], 0)) + 1]++;       // bucket 0 = the longest
  for (int32_t b = 0; b <= mx; b++) start[(size_t)b + 1] += start[(size_t)b];
  for (int64_t i = 0; i < n; i++) {
    const int32_t at = start[(size_t)(mx - std::max(len[i], 0))]++;
    if (order) (*order)[(size_t)at] = (int32_t)i;
    if (sorted_len) (*sorted_len)[(size_t)at] = len[i];
  }
}

struct FsJobs { const int32_t *order; unsigned *counter; };
__device__ __forceinline__ int64_t fs_next_job(const FsJobs &q, int64_t n, int lane) {
  unsigned j = 0;
  if (lane == 0) j = atomicAdd(q.counter, 1u);
  j = (unsigned)__shfl((int)j, 0, 64);
  return (int64_t)j < n ? (int64_t)q.order[j] : (int64_t)-1;
}


// ---- row-per-lane wavefront kernels of the envelopes (bath_fs_wavefront.hip)
int launch_fs5_fwd_wf(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int exact, int c5_compat,
                      float *d_sc, float *d_fwd, const int64_t *d_foff, float *d_xmx, const int64_t *d_xoff, DevBuf &ring_scratch, FsJobs jobs);
int launch_fs5_bwd_wf(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int exact,
                      float *d_sc, float *d_bck, const int64_t *d_boff, float *d_xmx, const int64_t *d_xoff, DevBuf &ring_scratch, DevBuf &terms_scratch, DevBuf &toff_scratch,
                      FsJobs jobs_sweep, FsJobs jobs_x);

// ---- decoding + optimal-accuracy fill with several waves per envelope, long models (bath_fs_decode.hip)
int fs5_decode_oa_mw_shape(int M, int *nodes_per_lane);     // waves per envelope (0: the one-wave kernel of bath_frameshift.hip)
int launch_fs5_decode_oa_mw(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, const float *d_bsc,
                            float *d_fwd, const int64_t *d_foff, float *d_fx, const int64_t *d_xoff, const float *d_bck, const int64_t *d_boff, const float *d_bx,
                            float *d_colsum, float *d_oa, float *d_osc, float *d_ox, FsJobs jobs, int store_pp = 1, float *d_rowden = nullptr);

// ---- multihit recursions in the reference's serial order, chains batched per block (bath_fs_chain.hip)
int launch_fs3_fwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM,
                         float *d_sc, float *d_xmx, const int64_t *d_xoff, FsJobs jobs, int cu_share = 1);
int launch_fs3_bwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM,
                         float *d_sc, float *d_xmx, const int64_t *d_xoff, FsJobs jobs, int cu_share = 1 /* 2: the Forward parser runs beside it */,
                         int bst_slot = 48, int stage_slot = 2 /* scratch / staging slots of the launch's batch starts */);
int launch_fs5_fwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM, int c5_compat,
                         float *d_sc, float *d_fwd, const int64_t *d_foff, float *d_xmx, const int64_t *d_xoff, int cfg_len, FsJobs jobs, int *d_done);
inline bool fs_chain_enabled() { static const bool off = [] { const char *e = std::getenv("BATH_HIP_FS_HANDOFF"); return e && e[0] == '1'; }(); return !off; }   // BATH_HIP_FS_HANDOFF=1: the 64-step lane hand-off kernels, for A/B runs

}  // namespace bath
