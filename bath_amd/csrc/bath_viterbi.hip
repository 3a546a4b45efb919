// bath_viterbi.hip -- p7_ViterbiFilter[_BATH] with one LANE per target (models up to 224 nodes).
//
// Reference: src/impl_sse/vitfilter.c:83-248 (score), :286-465 (score + diagonal windows).
//
// Only ~1% of the ORFs reach the Viterbi filter, but there are still several hundred thousand of them
// per block of 10^6 windows and they are short (mean ~45 aa), so the wave-per-target kernel
// (vit_wave_kernel, bath_filters.hip) spends its time in cross-lane shuffles.  Here the three DP rows
// (M, I, D) of a target live in one lane's VGPRs as packed int16, 64 targets per wavefront advance in
// lock step (targets are bucketed by length first): no shuffles, no reductions.
// Register r holds nodes r+1 (low half) and NR+r+1 (high half), so "the node to the left" is the previous register for both
// halves and the D->D chain D(k+1) = max(M(k)+tMD(k), D(k)+tDD(k)) runs down the registers as TWO chains in one packed
// add + max.  The high chain starts at node NR+1, whose D is the low chain's last value: it is run from -inf first and the
// paths that enter through node NR+1 are added where the row is read again (the next row's pass), max(D, D(NR+1) + running sum of
// tDD) -- equal to the serial chain because every tDD is <= 0 (a saturating add of a sum of non-positive terms is the chain of
// saturating adds).
// Round 2 started with adjacent nodes per register (three v_alignbit per pair for the left neighbours and the D chain as
// ~10 unpacked instructions per pair): 28 instructions per pair, now 17.  All arithmetic is the reference's: saturating 16-bit adds
// (v_pk_add_i16 clamp == adds_epi16), int16 wrap-around for the special states.
// The D row is evaluated exactly on every row; the reference's lazy-F test only skips D->D work that
// cannot reach any M cell (vitfilter.c:183-196), so M rows, xE and the score are identical.
#include <algorithm>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

namespace bath {

__device__ __forceinline__ s16x2 pk_adds(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s16x2 as_s2(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ unsigned as_u(s16x2 v) { return __builtin_bit_cast(unsigned, v); }

struct VitLaneTables {
  const int16_t *rw;      // [30][pitch] emission words, node k at index k-1, padding -32768; row 29 = all -32768
  int rw_pitch_bytes;
  const uint32_t *tw2;    // [NR][8] packed pairs (node r+1 | node NR+r+1 << 16), order MM IM DM BM MD DD MI II
  const int16_t *rank;    // [2*NR] visiting rank (vitfilter.c:390-396) of node r+1 at 2r, of node NR+r+1 at 2r+1; 32767 for padding
};

struct VitLaneConsts {
  int M, base_w, xwE_loop, xwE_move, Q8;
  float scale_w;
  double invP_vit, invP_msv;
  float scale_b;
  int base_b, tec_b, bias_b;
};

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

template <int NR>
#ifndef BATH_VIT_WAVES
#define BATH_VIT_WAVES 1
#endif
__global__ __launch_bounds__(256, (NR <= 76 ? BATH_VIT_WAVES : 1)) void vit_lane_kernel(SeqView sq, VitLaneTables tb, const uint32_t *__restrict__ tw2g, VitLaneConsts c, const int16_t *__restrict__ xwmove_tab,
                                                          const uint8_t *__restrict__ tjb_tab, const int32_t *__restrict__ todo, int64_t ntodo,
                                                          const int *__restrict__ ntodo_dev, const int *__restrict__ skip_dev, float *__restrict__ sc, int32_t *__restrict__ status,
                                                          const float *__restrict__ filtersc, const uint8_t *__restrict__ ssv_scores,
                                                          WindowRec *__restrict__ wins, int *__restrict__ win_count, int win_cap, int32_t *__restrict__ kminmax) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *s_rw = lds;                                              // 30 rows
  int16_t *s_rank = reinterpret_cast<int16_t *>(lds + 30 * tb.rw_pitch_bytes);
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tb.rw);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_rw);
    for (int i = threadIdx.x; i < 30 * tb.rw_pitch_bytes / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < 2 * NR; i += blockDim.x) s_rank[i] = tb.rank[i];
  }
  __syncthreads();
  if (ntodo_dev) ntodo = *ntodo_dev;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + (skip_dev ? *skip_dev : 0);   // the first *skip_dev entries are someone else's
  if (t - (threadIdx.x & 63) >= ntodo) return;                   // whole wave idle
  const bool live = t < ntodo;
  const int64_t sid = live ? (todo ? (int64_t)todo[t] : t) : (todo ? (int64_t)todo[0] : 0);
  const int L = live ? sq.len[sid] : 0;
  const uint8_t *s = sq.data + sq.off[sid];
  const int Lw = wave_max_i32(L);
  const bool do_win = (wins != nullptr);
  const int M = c.M;

  const int xw_move = xwmove_tab[L];
  int sc_thresh = 0, sc_ext_thresh = 0, skip_until = 0, kmin = 1 << 30, kmax = 0;
  if (do_win && live) {
    const double fsc = (double)filtersc[sid];
    sc_thresh = (int)(int16_t)(int)ceil(((fsc + 0.69314718055994529 * c.invP_vit + 3.0) * (double)c.scale_w) -
                                        (double)(float)c.xwE_move - (double)(float)xw_move + (double)(float)c.base_w);
    sc_ext_thresh = (int)ceil(((fsc + 0.69314718055994529 * c.invP_msv + 3.0) * (double)c.scale_b) + c.base_b + c.tec_b + (int)tjb_tab[L]);
  }
  const s16x2 NEG = {-32768, -32768};
  s16x2 Mr[NR], Ir[NR], Dr[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) Mr[r] = Ir[r] = Dr[r] = NEG;
  int xN = c.base_w;
  int xB = (int16_t)(xN + xw_move);
  int xJ = -32768, xC = -32768;
  bool overflow = false;
  short ePrev = -32768;                                                          // D(NR+1) of the previous row

  uint32_t wnext = (0 < L) ? *reinterpret_cast<const uint32_t *>(s) : 0x1d1d1d1du;
  for (int i0 = 0; i0 < Lw; i0 += 4) {
    const uint32_t w = wnext;
    wnext = (i0 + 4 < L) ? *reinterpret_cast<const uint32_t *>(s + i0 + 4) : 0x1d1d1d1du;
#pragma unroll 1
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j + 1;
      const bool on = (i <= L) && !overflow;
      int x = (w >> (8 * j)) & 0xff;
      x = (i <= L) ? min(x, kKp - 1) : 29;
      const char *rowbase = s_rw + x * tb.rw_pitch_bytes;
      const s16x2 xBv = {(short)xB, (short)xB};
      s16x2 xEv = NEG;
      // one ascending pass, in place.  The previous row's values of the pair to the left are carried in three
      // registers (they are overwritten before the next pair needs them); the D->D chain rides along in <d>.
      // Transition pairs are wave-uniform: read straight from global memory so that they arrive by scalar loads.
      // left neighbours of register 0: node 0 (nothing) for the low half, node NR = the low half of register NR-1 for the high half
      unsigned pM = (as_u(Mr[NR - 1]) << 16) | 0x8000u, pI = (as_u(Ir[NR - 1]) << 16) | 0x8000u, pD = (as_u(Dr[NR - 1]) << 16) | 0x8000u;   // (low halves: never corrected)
      s16x2 dpk = NEG;                                                           // D(1) | D(NR+1) without the paths through node NR
      // The previous row's D is stored WITHOUT the paths that enter the high chain through node NR+1; they are added here, where
      // that row is read: max(D, D(NR+1) + tDD(NR+1) + ... ), the right-hand side carried along as one saturating running sum (the
      // low halves hold -32768 throughout: no effect).  No second pass over the registers, no table.
      s16x2 cpk = {(short)-32768, ePrev};                                        // D(NR+1) of the previous row, less the tDD passed so far (low half: stays -32768)
      int twz = 0;
      asm volatile("" : "+s"(twz));                                              // opaque zero: keeps the (row-invariant) loads inside the row loop
      const uint32_t *tw = tw2g + twz;
      // Software pipeline: transitions one unit of two pairs ahead (s_load_dwordx16 issued by hand so that the wait
      // for it can be placed by hand: the scalar cache returns out of order and shares its counter with LDS), the
      // emissions one group of four pairs ahead (LDS).  Every unit starts with a full wait, then issues the next
      // unit's loads, then computes on registers only.
      u32x16 t_c, t_n;
      int4 e_c = *reinterpret_cast<const int4 *>(rowbase), e_n = e_c;
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(t_c) : "s"(tw));
#pragma unroll
      for (int r = 0; r < NR; r += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if ((r & 3) == 0 && r + 4 < NR) e_n = *reinterpret_cast<const int4 *>(rowbase + 4 * (r + 4));
        if (r + 2 < NR) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(t_n) : "s"(tw), "i"((r + 2) * 32));
        __builtin_amdgcn_sched_barrier(0);
        const unsigned em[2] = {(r & 3) ? (unsigned)e_c.z : (unsigned)e_c.x, (r & 3) ? (unsigned)e_c.w : (unsigned)e_c.y};
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int rr = r + q;
          // MM IM DM BM | MD DD MI II
          const unsigned tMM = t_c[8 * q + 0], tIM = t_c[8 * q + 1], tDM = t_c[8 * q + 2], tBM = t_c[8 * q + 3];
          const unsigned tMD = t_c[8 * q + 4], tDD = t_c[8 * q + 5], tMI = t_c[8 * q + 6], tII = t_c[8 * q + 7];
          const unsigned oM = as_u(Mr[rr]), oI = as_u(Ir[rr]);
          const unsigned oD = as_u(pk_max(Dr[rr], cpk));
          cpk = pk_adds(cpk, as_s2(tDD));
          const s16x2 ms = as_s2(pM), is = as_s2(pI), ds = as_s2(pD);
          s16x2 sv = pk_adds(xBv, as_s2(tBM));
          sv = pk_max(sv, pk_adds(ms, as_s2(tMM)));
          sv = pk_max(sv, pk_adds(is, as_s2(tIM)));
          sv = pk_max(sv, pk_adds(ds, as_s2(tDM)));
          sv = pk_adds(sv, as_s2(em[q]));
          Ir[rr] = pk_max(pk_adds(as_s2(oM), as_s2(tMI)), pk_adds(as_s2(oI), as_s2(tII)));
          const s16x2 dcv = pk_adds(sv, as_s2(tMD));                             // M(node)+tMD(node): the M->D part of D(node+1)
          xEv = pk_max(xEv, sv);
          Mr[rr] = sv;
          // both D chains: D(node+1) = max(M(node)+tMD(node), D(node)+tDD(node))
          Dr[rr] = dpk;
          dpk = pk_max(dcv, pk_adds(dpk, as_s2(tDD)));
          pM = oM; pI = oI; pD = oD;
        }
        if (r + 2 < NR) t_c = t_n;
        if ((r & 3) == 2 && r + 2 < NR) e_c = e_n;
        // pin this unit's results here: pure arithmetic otherwise floats past the barrier and drags its scalars along
        asm volatile("" : "+v"(Mr[r]), "+v"(Mr[r + 1]), "+v"(Ir[r]), "+v"(Ir[r + 1]), "+v"(Dr[r]), "+v"(Dr[r + 1]), "+v"(dpk), "+v"(cpk), "+v"(xEv));
        __builtin_amdgcn_sched_barrier(0);
      }
      ePrev = dpk.x;                                                             // D(NR+1) of this row: the next row adds the paths through it
      if (on) {
        const int xE = max((int)xEv.x, (int)xEv.y);
        if (xE >= 32767) overflow = true;
        else {
          xN = (int16_t)(xN + 0);
          xC = (int16_t)max(xC + 0, xE + c.xwE_move);
          xJ = (int16_t)max(xJ + 0, xE + c.xwE_loop);
          xB = (int16_t)max(xJ + xw_move, xN + xw_move);
          if (do_win && i > skip_until && xE >= sc_thresh) {          // vitfilter.c:386-424 (rare)
            int rank = 32767;
#pragma unroll
            for (int r = 0; r < NR; r++) {
              if ((int)Mr[r].x == xE) rank = min(rank, (int)s_rank[2 * r]);
              if ((int)Mr[r].y == xE) rank = min(rank, (int)s_rank[2 * r + 1]);
            }
            const int k_start = (rank / 8) + c.Q8 * (rank % 8) + 1;
            int max_k_end = k_start, max_i_end = i, sc_ext = sc_ext_thresh, max_sc_ext = sc_ext, since = 0;
            int kk = k_start + 1, nn = i + 1;
            while (kk <= M && nn <= L) {
              sc_ext += c.bias_b - (int)ssv_scores[(size_t)kk * kKp + min((int)s[nn - 1], kKp - 1)];
              if (sc_ext >= max_sc_ext) { max_sc_ext = sc_ext; max_k_end = kk; max_i_end = nn; since = 0; }
              else if (++since == 5) break;
              kk++; nn++;
            }
            const int slot = atomicAdd(win_count, 1);
            if (slot < win_cap) wins[slot] = WindowRec{(int32_t)sid, i, max_k_end, max_k_end - k_start + 1, 0.0f};
            kmax = max(kmax, max_k_end);
            kmin = min(kmin, k_start);
            skip_until = max_i_end;
          }
        }
      }
    }
  }
  if (live) {
    if (overflow) { sc[sid] = INFINITY; status[sid] = BATH_ERANGE; }
    else if (xC > -32768) {
      float r = (float)xC + (float)xw_move - (float)c.base_w;
      r /= c.scale_w;
      r = (float)((double)r - 3.0);
      sc[sid] = r; status[sid] = BATH_OK;
    } else { sc[sid] = -INFINITY; status[sid] = BATH_OK; }
    if (do_win && kminmax) { kminmax[2 * sid] = kmin; kminmax[2 * sid + 1] = kmax; }
  }
}

// ---- counting sort of a work list by target length, so that the 64 lanes of a wave finish together ----
// Global atomics on a few hot addresses serialise (~3 ns each on MI355X), so both passes count in LDS first and
// touch each global bin once per block.
constexpr int kLenBins = 2048;
__device__ __forceinline__ void len_block_range(int n, int &lo, int &hi) {
  const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  lo = min(n, (int)blockIdx.x * per); hi = min(n, lo + per);
}
__global__ __launch_bounds__(256) void len_hist_kernel(const int32_t *__restrict__ todo, const int *__restrict__ ntodo_dev, const int32_t *__restrict__ len,
                                                       int *__restrict__ hist) {
  __shared__ int s_cnt[kLenBins];
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) s_cnt[i] = 0;
  __syncthreads();
  int lo, hi;
  len_block_range(*ntodo_dev, lo, hi);
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) atomicAdd(&s_cnt[min(len[todo[j]], kLenBins - 1)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) if (s_cnt[i]) atomicAdd(&hist[i], s_cnt[i]);
}
__global__ void len_scan_kernel(int *__restrict__ hist /* in: counts, out: start offsets; longest first */) {
  __shared__ int tmp[kLenBins];
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) tmp[i] = hist[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = kLenBins - 1; b >= 0; b--) { const int cnt = tmp[b]; tmp[b] = run; run += cnt; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) hist[i] = tmp[i];
}
__global__ __launch_bounds__(256) void len_scatter_kernel(const int32_t *__restrict__ todo, const int *__restrict__ ntodo_dev, const int32_t *__restrict__ len,
                                                          int *__restrict__ cursor, int32_t *__restrict__ sorted) {
  __shared__ int s_cnt[kLenBins];
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) s_cnt[i] = 0;
  __syncthreads();
  int lo, hi;
  len_block_range(*ntodo_dev, lo, hi);
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) atomicAdd(&s_cnt[min(len[todo[j]], kLenBins - 1)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kLenBins; i += blockDim.x) { const int c = s_cnt[i]; if (c) s_cnt[i] = atomicAdd(&cursor[i], c); }
  __syncthreads();
  for (int j = lo + threadIdx.x; j < hi; j += blockDim.x) {
    const int cnd = todo[j];
    sorted[atomicAdd(&s_cnt[min(len[cnd], kLenBins - 1)], 1)] = cnd;
  }
}

int vit_lane_supported(const bath_hip_oprofile *om) { return om->vit_NR > 0; }

// todo (device list, count on device) -> sorted by length into <d_sorted>; <d_bins> is kLenBins ints of scratch
const int *len_sort_count_longer(const int *d_bins, int T) { return d_bins + std::min(T + 1, kLenBins - 1); }   // after the sort: #targets longer than T

int launch_len_sort(bath_hip_ctx *ctx, const int32_t *d_todo, const int *d_ntodo, const int32_t *d_len, int *d_bins, int32_t *d_sorted) {
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_bins, 0, kLenBins * sizeof(int), ctx->stream));
  hipLaunchKernelGGL(len_hist_kernel, dim3(256), dim3(256), 0, ctx->stream, d_todo, d_ntodo, d_len, d_bins);
  hipLaunchKernelGGL(len_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_bins);
  hipLaunchKernelGGL(len_scatter_kernel, dim3(256), dim3(256), 0, ctx->stream, d_todo, d_ntodo, d_len, d_bins, d_sorted);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_vit_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, const int *ntodo_dev,
                    float *d_sc, int32_t *d_status, const VitWindowArgs *wa, const int *skip_dev) {
  if (ntodo == 0) return BATH_OK;
  VitLaneTables tb{om->d_vit_rw, om->vit_rw_pitch, om->d_vit_tw2, om->d_vit_rank};
  VitLaneConsts c{};
  c.M = om->M; c.base_w = om->base_w; c.xwE_loop = om->xw_E[0]; c.xwE_move = om->xw_E[1]; c.Q8 = std::max(2, ((om->M - 1) / 8) + 1);
  c.scale_w = om->scale_w; c.scale_b = om->scale_b; c.base_b = om->base_b; c.tec_b = om->tec_b; c.bias_b = om->bias_b;
  const float *fsc = nullptr; const uint8_t *ssv = nullptr; WindowRec *wins = nullptr; int *wc = nullptr; int cap = 0; int32_t *kmm = nullptr;
  if (wa) { c.invP_vit = wa->invP_vit; c.invP_msv = wa->invP_msv; fsc = wa->d_filtersc; ssv = wa->d_ssv_scores; wins = (WindowRec *)wa->d_wins; wc = wa->d_win_count; cap = wa->win_cap; kmm = wa->d_kminmax; }
  const int NRv = om->vit_NR;
  const size_t shmem = (size_t)30 * om->vit_rw_pitch + (size_t)2 * NRv * 2 + 16;
  const int blocks = (int)((ntodo + 255) / 256);
  bool launched = false;
#define BATH_VITL_CASE(N)                                                                                                              \
  if (!launched && NRv == N) {                                                                                                         \
    hipLaunchKernelGGL(vit_lane_kernel<N>, dim3(blocks), dim3(256), shmem, ctx->stream, v, tb, tb.tw2, c, om->lt.d_xwmove, om->lt.d_tjb, d_todo, ntodo, \
                       ntodo_dev, skip_dev, d_sc, d_status, fsc, ssv, wins, wc, cap, kmm);                                                       \
    launched = true;                                                                                                                   \
  }
  BATH_VITL_CASE(16) BATH_VITL_CASE(32) BATH_VITL_CASE(48) BATH_VITL_CASE(64) BATH_VITL_CASE(68) BATH_VITL_CASE(72) BATH_VITL_CASE(76) BATH_VITL_CASE(80)
  BATH_VITL_CASE(96) BATH_VITL_CASE(112)
#undef BATH_VITL_CASE
  if (!launched) { ctx->set_error("lane-per-target Viterbi kernel: unsupported model length"); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
