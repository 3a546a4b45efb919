// bath_filters.hip -- HMMER3 filter-cascade kernels for gfx950 and their batched C entry points.
//
//   ssv_lane_kernel     <- p7_SSVFilter / get_xE / calc_band_N    src/impl_sse/ssvfilter.c:668-925
//   msv_wave_kernel     <- p7_MSVFilter (J-state path)            src/impl_sse/msvfilter.c:106-207
//   vit_wave_kernel     <- p7_ViterbiFilter[_BATH]                src/impl_sse/vitfilter.c:83-465
//   fwd_wave_kernel     <- p7_ForwardParser / forward_engine      src/impl_sse/fwdback.c:256-463
//   bias_lane_kernel    <- p7_bg_FilterScore / esl_hmm_Forward    src/p7_bg.c:491-505
//
// Work decomposition (MI355X-first, not the reference's 16-lane stripes):
//  * SSV has no dependency along a DP row (only along diagonals), so ONE LANE owns one target and
//    keeps the whole DP row in VGPRs, two cells (binary16, see bath_kernels.hpp) per register; 64 targets per
//    wavefront advance in lock step, emission costs are gathered from an LDS table by residue.
//    No cross-lane traffic at all.  1.5 VALU ops per 2 cells.
//  * MSV(J)/Viterbi/Forward have a serial dependency along the row (xE->xB, D->D), and only the
//    ~2% of targets that survive SSV reach them: ONE WAVEFRONT owns one target, lanes own
//    contiguous blocks of model nodes, the row recurrence is a wavefront scan (shuffle), and the
//    along-sequence recurrence stays in registers.
#include <cmath>
#include <cstring>
#include <algorithm>
#include <numeric>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

namespace bath {

// ============================================================================================
// SSV, lane per target.
// ============================================================================================

// Amino-acid targets: G adjacent lanes score sequence order[t] (or t); writes the raw maximum v in the
// kernel's signed domain (begin score = -128), i.e. get_xE()'s byte minus 256.
template <int NR, int G>
__global__ __launch_bounds__(256, NR <= 76 ? 4 : 1) void ssv_lane_kernel(SeqView sq, const int32_t *__restrict__ order,
                                                       const int16_t *__restrict__ cost_tab, int row_bytes,
                                                       int16_t *__restrict__ out_v) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  {
    const int n32 = kSsvRows * row_bytes / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(cost_tab);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < n32; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane_ = threadIdx.x & 63;
  const int64_t t = (gt >> 6) * (64 / G) + SsvGroups<G>::slot(lane_);       // G >= 4: a group's lanes are 64/G apart (bath_kernels.hpp)
  const int grank = SsvGroups<G>::rank(lane_);
  const bool live = t < sq.n;
  const int64_t sid = live ? (order ? (int64_t)order[t] : t) : 0;
  const int L = live ? sq.len[sid] : 0;
  const uint8_t *s = sq.data + sq.off[sid];
  const int Lw = wave_max_i32(L);
  const char *tile = lds + grank * (4 * NR);
  const unsigned tile_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)tile;   // LDS byte address

  s16x2 reg[NR];
  const s16x2 fl = {0, 0};                                      // the begin score
#pragma unroll
  for (int r = 0; r < NR; r++) reg[r] = fl;
  s16x2 xE = fl, xE2 = fl;

  uint32_t wnext = (0 < L) ? *reinterpret_cast<const uint32_t *>(s) : 0x1d1d1d1du;
  for (int i0 = 0; i0 < Lw; i0 += 4) {
    const uint32_t w = wnext;
    wnext = (i0 + 4 < L) ? *reinterpret_cast<const uint32_t *>(s + i0 + 4) : 0x1d1d1d1du;
#pragma unroll 1                   // one row body: the 128-VGPR budget of 4 waves per SIMD (as in ssv_orf_kernel)
    for (int j = 0; j < 4; j++) {
      int x = (w >> (8 * j)) & 0xff;
      x = (i0 + j < L) ? min(x, kKp - 1) : kRowReset;
      const unsigned carry = ssv_carry<NR, G>(reg, grank);
      ssv_row_pipe<NR, 3>(reg, xE, xE2, tile_addr + (unsigned)(x * row_bytes), carry);
    }
  }
  xE = ssv_max3(xE, xE2, xE2);
  const int v = ssv_group_max<G>(xE);
  if (live && grank == 0) out_v[sid] = (int16_t)min(v, 32767);
}

__global__ void ssv_classify_kernel(int64_t n, const int32_t *__restrict__ len, const int16_t *__restrict__ v,
                                    const uint8_t *__restrict__ tjb_tab, MsvConsts c, float *__restrict__ sc,
                                    int32_t *__restrict__ status) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float s = 0.f;
  const int st = ssv_classify(v[t], tjb_tab[len[t]], c, &s);
  sc[t] = s;
  status[t] = st;
}

// ============================================================================================
// Full MSV with the J state, wave per target (only for targets SSV could not decide).
// Lane l owns nodes l*C+1 .. l*C+C.
// ============================================================================================
template <int C>
__global__ __launch_bounds__(256) void msv_wave_kernel(SeqView sq, int M, const uint8_t *__restrict__ rb, int rb_stride,
                                                       const uint8_t *__restrict__ tjb_tab, MsvConsts c,
                                                       const int32_t *__restrict__ todo, int64_t ntodo, const int *__restrict__ ntodo_dev,
                                                       float *__restrict__ sc, int32_t *__restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (ntodo_dev) ntodo = *ntodo_dev;
  for (int64_t job = wid; job < ntodo; job += nw) {
    const int64_t sid = todo ? (int64_t)todo[job] : job;
    const int L = sq.len[sid];
    const uint8_t *s = sq.data + sq.off[sid];
    const int tjb = tjb_tab[L];
    const int tjbm = (uint8_t)((int8_t)tjb + (int8_t)c.tbm);
    int dp[C];
#pragma unroll
    for (int k = 0; k < C; k++) dp[k] = 0;
    int xJ = 0;
    int xB = satu8(c.base - tjbm);
    bool overflow = false;
    const int ph = threadIdx.x & 63;
    int rbuf = (ph < L) ? (int)s[ph] : 0, rnext = 0;          // residues 64 rows at a time, a lane each, the next 64 in flight (see fwd_wave_kernel)
    for (int i = 0; i < L; i++) {
      const int j = i & 63;
      if (j == 0) { const int q = i + 64 + ph; rnext = (q < L) ? (int)s[q] : 0; }
      const int x = min(__builtin_amdgcn_readlane(rbuf, j), kKp - 1);
      if (j == 63) rbuf = rnext;
      const uint8_t *row = rb + (size_t)x * rb_stride;
      int prev = wave_shr1_i32(dp[C - 1], 0);
      int xE = 0;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int node = lane * C + k + 1;
        const int cost = (node <= M) ? (int)row[node] : 255;
        int sv = max(prev, xB);
        sv = satu8(sv + c.bias);
        sv = satu8(sv - cost);
        prev = dp[k];
        dp[k] = sv;
        xE = max(xE, sv);
      }
      xE = wave_max_i32(xE);
      if (satu8(xE + c.bias) == 255) { overflow = true; break; }
      xE = satu8(xE - c.tec);
      xJ = max(xJ, xE);
      xB = satu8(max(c.base, xJ) - tjbm);
    }
    if (lane == 0) {
      if (overflow) { sc[sid] = INFINITY; status[sid] = BATH_ERANGE; }
      else {
        float r = ((float)(xJ - tjb) - (float)c.base);
        r /= c.scale_b;
        r = (float)((double)r - 3.0);
        sc[sid] = r; status[sid] = BATH_OK;
      }
    }
  }
}

__host__ __device__ constexpr int vit_em_stride(int M) { return (M + 7) & ~7; }
// ============================================================================================
// Viterbi filter, wave per target.  int16 semantics of the reference kept exactly: every add
// saturates to [-32768, 32767] (adds_epi16), special states wrap like int16_t assignments.
// The D->D chain is evaluated exactly every row with a wavefront scan over (max,+) maps; the
// reference's "lazy F" shortcut only skips work that provably cannot change any M cell.
// ============================================================================================
struct VitConsts {
  int base_w, xwE_loop, xwE_move;
  float scale_w;
  // window finding (p7_ViterbiFilter_BATH), all optional
  double invP_vit, invP_msv;     // esl_gumbel_invsurv(F2, VMU,VLAMBDA) / (F2, MMU,MLAMBDA), as float->double
  float scale_b;
  int base_b, tec_b, bias_b;
  int Q8;
};

template <int C>
__global__ __launch_bounds__(256) void vit_wave_kernel(SeqView sq, int M, const int16_t *__restrict__ g_rw,
                                                       const int16_t *__restrict__ g_tw,
                                                       const int16_t *__restrict__ xwmove_tab, const uint8_t *__restrict__ tjb_tab,
                                                       VitConsts c, const int32_t *__restrict__ todo, int64_t ntodo, const int *__restrict__ ntodo_dev,
                                                       float *__restrict__ sc, int32_t *__restrict__ status,
                                                       // BATH window outputs (null => plain p7_ViterbiFilter)
                                                       const float *__restrict__ filtersc, const uint8_t *__restrict__ ssv_scores,
                                                       WindowRec *__restrict__ wins, int *__restrict__ win_count, int win_cap,
                                                       int32_t *__restrict__ kminmax) {
  // Emission rows in LDS as [Kp][S] int16, entry node - 1, S a multiple of 8 (a lane's C nodes: aligned vector reads that follow the
  // previous lane's, no bank conflicts; 64 C entries of padding behind the last row); the lane's transitions in registers, read
  // once (as 16 bytes per node in LDS they were 16 C bytes apart from lane to lane: a 32-way conflict at C = 8).  See fwd_wave_kernel.
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int16_t *s_rw = reinterpret_cast<int16_t *>(lds);
  const int S = vit_em_stride(M);
  for (int i = threadIdx.x; i < kKp * S + 64 * C; i += blockDim.x) { const int x = i / S, k = i - x * S; s_rw[i] = (x < kKp && k < M) ? g_rw[(size_t)x * (M + 1) + k + 1] : (int16_t)-32768; }
  __syncthreads();
  enum { MM, IM, DM, BM, MD, DD, MI, II };
  const int lane = threadIdx.x & 63;
  int4 twq[C];
#pragma unroll
  for (int k = 0; k < C; k++) twq[k] = *reinterpret_cast<const int4 *>(g_tw + (size_t)min(lane * C + k + 1, M) * 8);
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const bool do_win = (wins != nullptr);
  if (ntodo_dev) ntodo = *ntodo_dev;

  for (int64_t job = wid; job < ntodo; job += nw) {
    const int64_t sid = todo ? (int64_t)todo[job] : job;
    const int L = sq.len[sid];
    const uint8_t *s = sq.data + sq.off[sid];
    const int xw_move = xwmove_tab[L];
    int sc_thresh = 0, sc_ext_thresh = 0, skip_until = 0, kmin = 1 << 30, kmax = 0;
    if (do_win) {
      const double fsc = (double)filtersc[sid];
      sc_thresh = (int)(int16_t)(int)ceil(((fsc + 0.69314718055994529 * c.invP_vit + 3.0) * (double)c.scale_w) -
                                          (double)(float)c.xwE_move - (double)(float)xw_move + (double)(float)c.base_w);
      sc_ext_thresh = (int)ceil(((fsc + 0.69314718055994529 * c.invP_msv + 3.0) * (double)c.scale_b) + c.base_b + c.tec_b + (int)tjb_tab[L]);
    }
    int Mp[C], Ip[C], Dp[C];
#pragma unroll
    for (int k = 0; k < C; k++) Mp[k] = Ip[k] = Dp[k] = -32768;
    int xN = c.base_w;
    int xB = (int16_t)(xN + xw_move);
    int xJ = -32768, xC = -32768, xE = -32768;
    bool overflow = false;

    const int ph = threadIdx.x & 63;
    int rbuf = (ph < L) ? (int)s[ph] : 0, rnext = 0;          // residues 64 rows at a time, a lane each, the next 64 in flight (see fwd_wave_kernel)
    for (int i = 1; i <= L; i++) {
      const int j = (i - 1) & 63;
      if (j == 0) { const int q = i - 1 + 64 + ph; rnext = (q < L) ? (int)s[q] : 0; }
      const int x = min(__builtin_amdgcn_readlane(rbuf, j), kKp - 1);
      if (j == 63) rbuf = rnext;
      int em[C];
      {
        const int16_t *rw = s_rw + (size_t)x * S + lane * C;
        if constexpr (C % 8 == 0) {
#pragma unroll
          for (int q = 0; q < C / 8; q++) {
            const int4 v = *reinterpret_cast<const int4 *>(rw + 8 * q);
            const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int h = 0; h < 4; h++) { em[8 * q + 2 * h] = (int16_t)(w[h] & 0xffff); em[8 * q + 2 * h + 1] = w[h] >> 16; }
          }
        } else if constexpr (C % 4 == 0) {
#pragma unroll
          for (int q = 0; q < C / 4; q++) {
            const int2 v = *reinterpret_cast<const int2 *>(rw + 4 * q);
            em[4 * q] = (int16_t)(v.x & 0xffff); em[4 * q + 1] = v.x >> 16; em[4 * q + 2] = (int16_t)(v.y & 0xffff); em[4 * q + 3] = v.y >> 16;
          }
        } else if constexpr (C % 2 == 0) {
#pragma unroll
          for (int q = 0; q < C / 2; q++) { const int v = *reinterpret_cast<const int *>(rw + 2 * q); em[2 * q] = (int16_t)(v & 0xffff); em[2 * q + 1] = v >> 16; }
        } else {
#pragma unroll
          for (int q = 0; q < C; q++) em[q] = (int)rw[q];
        }
      }
      const int mIn = wave_shr1_i32(Mp[C - 1], -32768), iIn = wave_shr1_i32(Ip[C - 1], -32768), dIn = wave_shr1_i32(Dp[C - 1], -32768);
      int Mc[C], Ic[C], dcv[C], tdd[C];
      int xEl = -32768;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int node = lane * C + k + 1;
        if (node <= M) {
          const int4 tq = twq[k];
          const int tMM = (int16_t)(tq.x & 0xffff), tIM = (int16_t)(tq.x >> 16);
          const int tDM = (int16_t)(tq.y & 0xffff), tBM = (int16_t)(tq.y >> 16);
          const int tMD = (int16_t)(tq.z & 0xffff), tDD = (int16_t)(tq.z >> 16);
          const int tMI = (int16_t)(tq.w & 0xffff), tII = (int16_t)(tq.w >> 16);
          const int m1 = (k == 0) ? mIn : Mp[k - 1];
          const int i1 = (k == 0) ? iIn : Ip[k - 1];
          const int d1 = (k == 0) ? dIn : Dp[k - 1];
          int sv = sat16(xB + tBM);
          sv = max(sv, sat16(m1 + tMM));
          sv = max(sv, sat16(i1 + tIM));
          sv = max(sv, sat16(d1 + tDM));
          sv = sat16(sv + em[k]);
          Mc[k] = sv;
          xEl = max(xEl, sv);
          dcv[k] = sat16(sv + tMD);
          tdd[k] = tDD;
          Ic[k] = max(sat16(Mp[k] + tMI), sat16(Ip[k] + tII));
        } else {
          Mc[k] = -32768; Ic[k] = -32768; dcv[k] = -32768; tdd[k] = -32768;
        }
      }
      xE = wave_max_i32(xEl);
      if (xE >= 32767) { overflow = true; break; }
      xN = (int16_t)(xN + 0);
      xC = (int16_t)max(xC + 0, xE + c.xwE_move);
      xJ = (int16_t)max(xJ + 0, xE + c.xwE_loop);
      xB = (int16_t)max(xJ + xw_move, xN + xw_move);

      if (do_win && i > skip_until && xE >= sc_thresh) {           // vitfilter.c:386-424
        int rank = 1 << 30;
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int node = lane * C + k + 1;
          if (node <= M && Mc[k] == xE) rank = min(rank, ((node - 1) % c.Q8) * 8 + (node - 1) / c.Q8);
        }
        rank = wave_min_i32(rank);
        const int k_start = (rank / 8) + c.Q8 * (rank % 8) + 1;
        int max_k_end = k_start, max_i_end = i, sc_ext = sc_ext_thresh, max_sc_ext = sc_ext, since = 0;
        int kk = k_start + 1, nn = i + 1;
        while (kk <= M && nn <= L) {
          sc_ext += c.bias_b - (int)ssv_scores[(size_t)kk * kKp + min((int)s[nn - 1], kKp - 1)];
          if (sc_ext >= max_sc_ext) { max_sc_ext = sc_ext; max_k_end = kk; max_i_end = nn; since = 0; }
          else if (++since == 5) break;
          kk++; nn++;
        }
        if (lane == 0) {
          int slot = atomicAdd(win_count, 1);
          if (slot < win_cap) wins[slot] = WindowRec{(int32_t)sid, i, max_k_end, max_k_end - k_start + 1, 0.0f};
        }
        kmax = max(kmax, max_k_end);
        kmin = min(kmin, k_start);
        skip_until = max_i_end;
      }

      // exact D row: D(node+1) = max(dcv(node), D(node)+tDD(node)); chain across lanes by scan
      int A = -(1 << 28), B = 0;
#pragma unroll
      for (int k = 0; k < C; k++) { A = max(dcv[k], A + tdd[k]); B += tdd[k]; A = max(A, -(1 << 28)); }
      // inclusive scan of f_l(x) = max(A_l, x + B_l)
      // by DPP; lanes without a source see the identity map (A = -2^28, B = 0).  Integer (max,+): any scan order gives the same D row
#define BATH_VIT_STEP(CTRL, MASK) { const int Ap = dpp_i<CTRL, MASK>(A, -(1 << 28)), Bp = dpp_i<CTRL, MASK>(B, 0); A = max(A, max(Ap, -(1 << 28)) + B); B = max(B + Bp, -(1 << 28)); }
      BATH_VIT_STEP(0x111, 0xf) BATH_VIT_STEP(0x112, 0xf) BATH_VIT_STEP(0x114, 0xf) BATH_VIT_STEP(0x118, 0xf) BATH_VIT_STEP(0x142, 0xa) BATH_VIT_STEP(0x143, 0xc)
#undef BATH_VIT_STEP
      int din = wave_shr1_i32(A, -32768);
      din = max(din, -32768);
      int Dc[C];
      Dc[0] = din;
#pragma unroll
      for (int k = 1; k < C; k++) Dc[k] = max(dcv[k - 1], sat16(Dc[k - 1] + tdd[k - 1]));
#pragma unroll
      for (int k = 0; k < C; k++) { Mp[k] = Mc[k]; Ip[k] = Ic[k]; Dp[k] = Dc[k]; }
    }
    if (lane == 0) {
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
}

// ============================================================================================
// Forward parser, wave per target (odds-ratio space, sparse rescaling as fwdback.c:418-434).
// ============================================================================================
struct FwdConsts { float xfE_loop, xfE_move; };

// Emission rows of the wave-per-target Forward / Backward kernels in LDS: [Kp][S] floats, entry node - 1 (S a multiple of 4, 64 C
// zeros behind the last row), so that a lane's C consecutive nodes are one or two 16-byte reads at 16-byte-aligned addresses that
// follow the previous lane's: no bank conflicts.  (Indexed by node with a read per node, the lanes of a wave were C x 4 bytes apart:
// an 8-way conflict at C = 8 -- and the transitions, 32 bytes per node read as two float4, 64-way: they are in registers now.)
__host__ __device__ constexpr int wave_em_stride(int M) { return (M + 4) & ~3; }          // >= M + 1: entry M is the zero column of node M + 1
template <int C>
__device__ __forceinline__ void wave_em_load(const float *row, int lane, float (&e)[C]) {
  const float *p = row + lane * C;
  if constexpr (C % 4 == 0) {
#pragma unroll
    for (int q = 0; q < C / 4; q++) { const float4 v = *reinterpret_cast<const float4 *>(p + 4 * q); e[4 * q] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w; }
  } else if constexpr (C % 2 == 0) {
#pragma unroll
    for (int q = 0; q < C / 2; q++) { const float2 v = *reinterpret_cast<const float2 *>(p + 2 * q); e[2 * q] = v.x; e[2 * q + 1] = v.y; }
  } else {
#pragma unroll
    for (int q = 0; q < C; q++) e[q] = p[q];
  }
}
template <int C>
__device__ __forceinline__ void wave_em_fill(float *w_rf, const float *g_rf, int M, int src_stride) {
  const int S = wave_em_stride(M), n = kKp * S + 64 * C;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { const int x = i / S, k = i - x * S; w_rf[i] = (x < kKp && k < M) ? g_rf[(size_t)x * src_stride + k + 1] : 0.f; }
}

// GT: the tables of a model too long for the LDS (above ~1100 nodes) are read from global memory (L2-resident) instead
template <int C, bool GT = false>
__global__ __launch_bounds__(256) void fwd_wave_kernel(SeqView sq, int M, const float *__restrict__ g_rf, const float *__restrict__ g_tf,
                                                       const float *__restrict__ pmove_tab, FwdConsts c,
                                                       const int32_t *__restrict__ todo, int64_t ntodo, const int *__restrict__ ntodo_dev,
                                                       float *__restrict__ sc, int32_t *__restrict__ status,
                                                       float *__restrict__ xmx, const int64_t *__restrict__ xmx_off,
                                                       float *__restrict__ dp, const int64_t *__restrict__ dp_off, int unihit,
                                                       const int32_t *__restrict__ cfg_len) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const float *s_rf = reinterpret_cast<const float *>(lds);                      // !GT: [Kp][wave_em_stride(M)], entry node - 1
  const int S = wave_em_stride(M);
  if (ntodo_dev) ntodo = *ntodo_dev;
  if (!GT) {
    wave_em_fill<C>(reinterpret_cast<float *>(lds), g_rf, M, M + 1);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // the lane's transitions, once (!GT): MM IM DM BM / MD DD MI II of its C nodes
  float4 tra[GT ? 1 : C], trb[GT ? 1 : C];
  if constexpr (!GT) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int node = min(lane * C + k + 1, M);
      tra[k] = *reinterpret_cast<const float4 *>(g_tf + (size_t)node * 8); trb[k] = *reinterpret_cast<const float4 *>(g_tf + (size_t)node * 8 + 4);
    }
  }

  for (int64_t job = wid; job < ntodo; job += nw) {
    const int64_t sid = todo ? (int64_t)todo[job] : job;
    const int L = sq.len[sid];
    const uint8_t *s = sq.data + sq.off[sid];
    // unihit (envelope rescoring, p7_oprofile_ReconfigUnihit): nj = 0, so pmove = 2/(L+2); a plain float division, as on the host
    // cfg_len: the length the model was configured for when that is not this target's (a region of an ORF, p7_domaindef.c:553)
    const float pmove = unihit ? (2.0f / ((float)L + 2.0f)) : pmove_tab[cfg_len ? cfg_len[sid] : L], ploop = 1.0f - pmove;
    float *dprow = dp ? dp + dp_off[sid] : nullptr;        // full matrix (p7_Forward): (L+1) x (M+1) x {M, D, I}
    if (dprow) for (int k = lane; k <= M; k += 64) { dprow[(size_t)k * 3] = dprow[(size_t)k * 3 + 1] = dprow[(size_t)k * 3 + 2] = 0.f; }
    float Mp[C], Ip[C], Dp[C];
#pragma unroll
    for (int k = 0; k < C; k++) Mp[k] = Ip[k] = Dp[k] = 0.f;
    float xN = 1.f, xE = 0.f, xJ = 0.f, xC = 0.f, xB = pmove;
    float totscale = 0.f;
    float *xrow = (xmx && xmx_off[sid] >= 0) ? xmx + xmx_off[sid] : nullptr;      // (L+1) x {E,N,J,B,C,SCALE}, P7_OMX xmx (impl_sse.h:253-262); a negative offset: not wanted for this target
    if (xrow && lane == 0) { xrow[0] = xE; xrow[1] = xN; xrow[2] = xJ; xrow[3] = xB; xrow[4] = xC; xrow[5] = 1.0f; }

    // The residues arrive 64 rows at a time, a lane each, the next 64 in flight: with a load per row every row of a launch of a
    // few dozen targets (one query's envelopes) waited for its own trip to memory
    int rbuf = (lane < L) ? (int)s[lane] : 0, rnext = 0;
    for (int i = 1; i <= L; i++) {
      const int j = (i - 1) & 63;
      if (j == 0) { const int q = i - 1 + 64 + lane; rnext = (q < L) ? (int)s[q] : 0; }
      const int x = min(__builtin_amdgcn_readlane(rbuf, j), kKp - 1);
      if (j == 63) rbuf = rnext;
      float em[C];
      if constexpr (!GT) wave_em_load<C>(s_rf + (size_t)x * S, lane, em);
      else {
#pragma unroll
        for (int k = 0; k < C; k++) em[k] = g_rf[(size_t)x * (M + 1) + min(lane * C + k + 1, M)];
      }
      const float mIn = wave_shr1_f32(Mp[C - 1], 0.f), iIn = wave_shr1_f32(Ip[C - 1], 0.f), dIn = wave_shr1_f32(Dp[C - 1], 0.f);
      float Mc[C], Ic[C], md[C], tdd[C];
      float sumE = 0.f;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int node = lane * C + k + 1;
        if (node <= M) {
          float4 ta, tb;                                                                      // MM IM DM BM / MD DD MI II
          if constexpr (GT) { ta = *reinterpret_cast<const float4 *>(g_tf + (size_t)node * 8); tb = *reinterpret_cast<const float4 *>(g_tf + (size_t)node * 8 + 4); }
          else { ta = tra[k]; tb = trb[k]; }
          const float m1 = (k == 0) ? mIn : Mp[k - 1];
          const float i1 = (k == 0) ? iIn : Ip[k - 1];
          const float d1 = (k == 0) ? dIn : Dp[k - 1];
          float sv = xB * ta.w;
          sv = sv + m1 * ta.x;
          sv = sv + i1 * ta.y;
          sv = sv + d1 * ta.z;
          sv = sv * em[k];
          Mc[k] = sv;
          sumE += sv;
          md[k] = sv * tb.x;
          tdd[k] = tb.y;
          Ic[k] = Mp[k] * tb.z + Ip[k] * tb.w;
        } else { Mc[k] = 0.f; Ic[k] = 0.f; md[k] = 0.f; tdd[k] = 0.f; }
      }
      // D(node+1) = md(node) + D(node)*tDD(node): affine maps composed by wavefront scan
      float A = 0.f, B = 1.f;
#pragma unroll
      for (int k = 0; k < C; k++) { A = md[k] + A * tdd[k]; B *= tdd[k]; }
      // by DPP; lanes without a source see the identity map (A = 0, B = 1): A + 0 * B = A, B * 1 = B
#define BATH_FWD_STEP(CTRL, MASK) { const float Ap = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, A), CTRL, MASK, 0xf, false)), \
                                                 Bp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, B), CTRL, MASK, 0xf, false)); \
                                    A = A + Ap * B; B = B * Bp; }
      BATH_FWD_STEP(0x111, 0xf) BATH_FWD_STEP(0x112, 0xf) BATH_FWD_STEP(0x114, 0xf) BATH_FWD_STEP(0x118, 0xf) BATH_FWD_STEP(0x142, 0xa) BATH_FWD_STEP(0x143, 0xc)
#undef BATH_FWD_STEP
      const float din = wave_shr1_f32(A, 0.f);
      float Dc[C];
      Dc[0] = din;
#pragma unroll
      for (int k = 1; k < C; k++) Dc[k] = md[k - 1] + Dc[k - 1] * tdd[k - 1];
#pragma unroll
      for (int k = 0; k < C; k++) sumE += Dc[k];
      xE = wave_sum_f32(sumE);
      xN = xN * ploop;
      xC = (xC * ploop) + (xE * c.xfE_move);
      xJ = (xJ * ploop) + (xE * c.xfE_loop);
      xB = (xJ * pmove) + (xN * pmove);
      float scale = 1.0f;
      if (xE > 1.0e4f) {
        xN = xN / xE; xC = xC / xE; xJ = xJ / xE; xB = xB / xE;
        const float inv = (float)(1.0 / (double)xE);
#pragma unroll
        for (int k = 0; k < C; k++) { Mc[k] *= inv; Dc[k] *= inv; Ic[k] *= inv; }
        totscale += (float)log((double)xE);
        scale = xE;
        xE = 1.0f;
      }
      if (xrow && lane == 0) { float *r = xrow + (size_t)i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = scale; }
      if (dprow) {
        float *r = dprow + (size_t)i * (M + 1) * 3;
        if (lane == 0) r[0] = r[1] = r[2] = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) { const int node = lane * C + k + 1; if (node <= M) { r[(size_t)node * 3] = Mc[k]; r[(size_t)node * 3 + 1] = Dc[k]; r[(size_t)node * 3 + 2] = Ic[k]; } }
      }
#pragma unroll
      for (int k = 0; k < C; k++) { Mp[k] = Mc[k]; Ip[k] = Ic[k]; Dp[k] = Dc[k]; }
    }
    if (lane == 0) {
      if (isnan(xC) || (L > 0 && xC == 0.0f) || isinf(xC)) { sc[sid] = -INFINITY; status[sid] = BATH_ERANGE; }
      else { sc[sid] = (float)((double)totscale + log((double)(xC * pmove))); status[sid] = BATH_OK; }
    }
  }
}

// ============================================================================================
// Backward parser, wave per target: p7_BackwardParser / backward_engine (src/impl_sse/fwdback.c:468-740), odds-ratio
// fp32, rescaled row by row with the Forward pass's scale factors (fwd xmx SCALE) unless xB outgrows 1e16 (:668-675).
// Lane l owns nodes l*C+1 .. l*C+C; D(i,k) = ... + D(i,k+1)*tDD(k) is a right-to-left scan over affine maps.
// ============================================================================================
// GT: g_rf / g_tf are then the padded copies ([Kp][M+2], [(M+2)*8]) the oprofile keeps for this purpose
template <int C, bool GT = false>
__global__ __launch_bounds__(256) void bwd_wave_kernel(SeqView sq, int M, const float *__restrict__ g_rf, const float *__restrict__ g_tf,
                                                       const float *__restrict__ pmove_tab, FwdConsts c, int64_t ntodo,
                                                       const float *__restrict__ fwd_xmx, const int64_t *__restrict__ xmx_off,
                                                       float *__restrict__ sc, int32_t *__restrict__ status, float *__restrict__ bck_xmx,
                                                       float *__restrict__ dp, const int64_t *__restrict__ dp_off, int unihit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const float *s_tf = g_tf;                                                      // GT: [(M+2)*8], node M+1 all zero (!GT: read once, below)
  const float *s_rf = GT ? g_rf : reinterpret_cast<const float *>(lds);          // GT: [Kp][M+2], column M+1 zero; !GT: [Kp][wave_em_stride(M)], entry node - 1
  const int S = wave_em_stride(M);
  if (!GT) {
    wave_em_fill<C>(reinterpret_cast<float *>(lds), g_rf, M, M + 1);
    __syncthreads();
  }
  // lanes own their nodes in DESCENDING order (logical lane = 63 - physical): the chains towards node M are then upward DPP scans
  const int lane = 63 - (threadIdx.x & 63);
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // the lane's transitions, once (!GT): MD DD MI II and BM of its C nodes (zero beyond node M), MM IM DM of the nodes to their right
  // (zero from node M on)
  float4 trb[GT ? 1 : C], trn[GT ? 1 : C]; float trbm[GT ? 1 : C];
  if constexpr (!GT) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int node = lane * C + k + 1;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      trb[k] = node <= M ? *reinterpret_cast<const float4 *>(g_tf + (size_t)node * 8 + 4) : z;
      trbm[k] = node <= M ? g_tf[(size_t)node * 8 + 3] : 0.f;
      trn[k] = node < M ? *reinterpret_cast<const float4 *>(g_tf + (size_t)(node + 1) * 8) : z;
    }
  }
  for (int64_t sid = wid; sid < ntodo; sid += nw) {
    const int L = sq.len[sid];
    const uint8_t *s = sq.data + sq.off[sid];
    const float pmove = unihit ? (2.0f / ((float)L + 2.0f)) : pmove_tab[L], ploop = 1.0f - pmove;
    float *dprow = dp ? dp + dp_off[sid] : nullptr;        // full matrix (p7_Backward), same layout as Forward's
    const float *fx = fwd_xmx + xmx_off[sid];
    float *bx = bck_xmx ? bck_xmx + xmx_off[sid] : nullptr;
    float xJ = 0.f, xB = 0.f, xN = 0.f, xC = pmove, xE = xC * c.xfE_move;
    float Mn[C], In[C], Dn[C];
    // ---- row L: M = D = xE plus the D->D->..->E and M->D->..->E paths (:499-532)
    {
      float A = 0.f, B = 1.f;                              // D(k) = xE + D(k+1)*tDD(k)
#pragma unroll
      for (int k = C - 1; k >= 0; k--) { const int node = lane * C + k + 1; float tdd; if constexpr (GT) tdd = (node <= M) ? s_tf[(size_t)node * 8 + 5] : 0.f; else tdd = trb[k].y; A = ((node <= M) ? xE : 0.f) + A * tdd; B *= tdd; }
      // lanes without a source see the identity map (A = 0, B = 1)
#define BATH_BWD_STEP(CTRL, MASK) { const float An = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, A), CTRL, MASK, 0xf, false)), \
                                                 Bn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, B), CTRL, MASK, 0xf, false)); \
                                    A = A + An * B; B = B * Bn; }
      BATH_BWD_STEP(0x111, 0xf) BATH_BWD_STEP(0x112, 0xf) BATH_BWD_STEP(0x114, 0xf) BATH_BWD_STEP(0x118, 0xf) BATH_BWD_STEP(0x142, 0xa) BATH_BWD_STEP(0x143, 0xc)
#undef BATH_BWD_STEP
      float dnext = wave_shr1_f32(A, 0.f);                 // D of the first node of the logical lane above (the physical lane below)
#pragma unroll
      for (int k = C - 1; k >= 0; k--) {
        const int node = lane * C + k + 1;
        if (node <= M) {
          float tdd, tmd;
          if constexpr (GT) { tdd = s_tf[(size_t)node * 8 + 5]; tmd = s_tf[(size_t)node * 8 + 4]; } else { tdd = trb[k].y; tmd = trb[k].x; }
          Dn[k] = xE + dnext * tdd;
          Mn[k] = xE + dnext * tmd;
          dnext = Dn[k];
        } else { Dn[k] = 0.f; Mn[k] = 0.f; dnext = 0.f; }
        In[k] = 0.f;
      }
    }
    float totscale;
    {
      const float scl = (L >= 1) ? fx[(size_t)L * 6 + 5] : 1.0f;
      if (scl > 1.0f) {
        xE /= scl; xN /= scl; xC /= scl; xJ /= scl; xB /= scl;
        const float inv = (float)(1.0 / (double)scl);
#pragma unroll
        for (int k = 0; k < C; k++) { Mn[k] *= inv; Dn[k] *= inv; In[k] *= inv; }
      }
      totscale = (float)log((double)scl);
      if (bx && lane == 0) { float *r = bx + (size_t)L * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = scl; }
      if (dprow) {
        float *r = dprow + (size_t)L * (M + 1) * 3;
        if (lane == 0) r[0] = r[1] = r[2] = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) { const int node = lane * C + k + 1; if (node <= M) { r[(size_t)node * 3] = Mn[k]; r[(size_t)node * 3 + 1] = Dn[k]; r[(size_t)node * 3 + 2] = In[k]; } }
        for (int k = lane; k <= M; k += 64) { dprow[(size_t)k * 3] = dprow[(size_t)k * 3 + 1] = dprow[(size_t)k * 3 + 2] = 0.f; }      // row 0
      }
    }
    bool own_scales = false;
    // residue i+1 and Forward's scale factor of row i, 64 rows at a time (a physical lane each, the next 64 in flight): see fwd_wave_kernel
    const int ph = threadIdx.x & 63;
    int rbuf = (L - 1 - ph >= 0) ? (int)s[L - 1 - ph] : 0, rnext = 0;
    float fbuf = (L - 1 - ph >= 0) ? fx[(size_t)(L - 1 - ph) * 6 + 5] : 1.0f, fnext = 1.0f;
    for (int i = L - 1; i >= 0; i--) {
      const int j = (L - 1 - i) & 63;
      if (j == 0) { const int q = i - 64 - ph; rnext = (q >= 0) ? (int)s[q] : 0; fnext = (q >= 0) ? fx[(size_t)q * 6 + 5] : 1.0f; }
      const int x = min(__builtin_amdgcn_readlane(rbuf, j), kKp - 1);               // residue i+1
      const float fs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fbuf), j));
      if (j == 63) { rbuf = rnext; fbuf = fnext; }
      // B(i) = sum_k M(i+1,k) * e(k, x_{i+1}) * tBM(k)
      float me[C];                                         // M(i+1,k) * e(k, x_{i+1})
      float b = 0.f;
      if constexpr (GT) {
        const float *rf = s_rf + (size_t)x * (M + 2);
#pragma unroll
        for (int k = 0; k < C; k++) { const int node = lane * C + k + 1; me[k] = Mn[k] * rf[min(node, M + 1)]; b += me[k] * s_tf[(size_t)min(node, M + 1) * 8 + 3]; }
      } else {
        float em[C];
        wave_em_load<C>(s_rf + (size_t)x * S, lane, em);  // (beyond node M + 1: the next row's or the pad's finite numbers, times Mn = 0)
#pragma unroll
        for (int k = 0; k < C; k++) { me[k] = Mn[k] * em[k]; b += me[k] * trbm[k]; }
      }
      xB = wave_sum_f32(b);
      if (i == 0) { xN = (xB * pmove) + (xN * ploop); break; }                                       // :695-740: only N and B are reachable
      xC = xC * ploop;
      xJ = (xB * pmove) + (xJ * ploop);
      xN = (xB * pmove) + (xN * ploop);
      xE = (xC * c.xfE_move) + (xJ * c.xfE_loop);
      // the node to the right of each lane's block
      const float meR = wave_shr1_f32(me[0], 0.f);
      float mnext[C];                                      // M(i+1,k+1) * e(k+1)
#pragma unroll
      for (int k = 0; k < C; k++) mnext[k] = (k + 1 < C) ? me[k + 1] : meR;
      // D(k) = mnext(k)*tDM(k+1) + xE + D(k+1)*tDD(k): right-to-left affine scan
      float A = 0.f, B = 1.f, dconst[C], tddv[C];
#pragma unroll
      for (int k = C - 1; k >= 0; k--) {
        const int node = lane * C + k + 1;
        const bool in = node <= M;
        float tdm;
        if constexpr (GT) { tdm = (node < M) ? s_tf[(size_t)(node + 1) * 8 + 2] : 0.f; tddv[k] = in ? s_tf[(size_t)node * 8 + 5] : 0.f; }
        else { tdm = trn[k].z; tddv[k] = trb[k].y; }
        dconst[k] = in ? (mnext[k] * tdm) : 0.f;
        // reference order (:608-612 restated): Dc = mnext*tdm + Dc[k+1]*tdd + xE
        A = in ? ((dconst[k] + A * tddv[k]) + xE) : 0.f; B = in ? B * tddv[k] : 0.f;
      }
#define BATH_BWD_STEP(CTRL, MASK) { const float An = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, A), CTRL, MASK, 0xf, false)), \
                                                 Bn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3f800000, __builtin_bit_cast(int, B), CTRL, MASK, 0xf, false)); \
                                    A = A + An * B; B = B * Bn; }
      BATH_BWD_STEP(0x111, 0xf) BATH_BWD_STEP(0x112, 0xf) BATH_BWD_STEP(0x114, 0xf) BATH_BWD_STEP(0x118, 0xf) BATH_BWD_STEP(0x142, 0xa) BATH_BWD_STEP(0x143, 0xc)
#undef BATH_BWD_STEP
      float dnext = wave_shr1_f32(A, 0.f);
      float Mc[C], Ic[C], Dc[C];
#pragma unroll
      for (int k = C - 1; k >= 0; k--) {
        const int node = lane * C + k + 1;
        if (node <= M) {
          float tmm, tim, t4, t6, t7;                              // MM, IM enter node k+1 (zero at M+1); MD, MI, II leave node k
          if constexpr (GT) {
            const float *t = s_tf + (size_t)node * 8, *t1 = s_tf + (size_t)(node + 1) * 8;
            tmm = (node < M) ? t1[0] : 0.f; tim = (node < M) ? t1[1] : 0.f; t4 = t[4]; t6 = t[6]; t7 = t[7];
          } else { tmm = trn[k].x; tim = trn[k].y; t4 = trb[k].x; t6 = trb[k].z; t7 = trb[k].w; }
          Ic[k] = In[k] * t7 + mnext[k] * tim;
          Dc[k] = (dconst[k] + dnext * tddv[k]) + xE;
          Mc[k] = ((In[k] * t6 + mnext[k] * tmm) + xE) + dnext * t4;
          dnext = Dc[k];
        } else { Mc[k] = Ic[k] = Dc[k] = 0.f; dnext = 0.f; }
      }
      if (xB > 1.0e16f) own_scales = true;
      const float scl = own_scales ? ((xB > 1.0e4f) ? xB : 1.0f) : fs;
      if (scl > 1.0f) {
        xE /= scl; xN /= scl; xJ /= scl; xB /= scl; xC /= scl;
        const float inv = (float)(1.0 / (double)scl);
#pragma unroll
        for (int k = 0; k < C; k++) { Mc[k] *= inv; Dc[k] *= inv; Ic[k] *= inv; }
        totscale += (float)log((double)scl);
      }
      if (bx && lane == 0) { float *r = bx + (size_t)i * 6; r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; r[5] = scl; }
      if (dprow) {
        float *r = dprow + (size_t)i * (M + 1) * 3;
        if (lane == 0) r[0] = r[1] = r[2] = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) { const int node = lane * C + k + 1; if (node <= M) { r[(size_t)node * 3] = Mc[k]; r[(size_t)node * 3 + 1] = Dc[k]; r[(size_t)node * 3 + 2] = Ic[k]; } }
      }
#pragma unroll
      for (int k = 0; k < C; k++) { Mn[k] = Mc[k]; In[k] = Ic[k]; Dn[k] = Dc[k]; }
    }
    if (lane == 0) {
      if (bx) { bx[0] = 0.f; bx[1] = xN; bx[2] = 0.f; bx[3] = xB; bx[4] = 0.f; bx[5] = 1.0f; }
      if (isnan(xN) || (L > 0 && xN == 0.0f) || isinf(xN)) { sc[sid] = -INFINITY; status[sid] = BATH_ERANGE; }
      else { sc[sid] = (float)((double)totscale + log((double)xN)); status[sid] = BATH_OK; }
    }
  }
}

// ============================================================================================
// Bias filter: Forward of the 2-state HMM (p7_bg_FilterScore), lane per target.
// eo is [Kp][2] emission odds; per-target eo (local composition) if eo_stride != 0.
// ============================================================================================
__global__ __launch_bounds__(256) void bias_lane_kernel(SeqView sq, int M, const float *__restrict__ eo, int eo_stride,
                                                        const float *__restrict__ p1_tab, const float *__restrict__ lt1_tab, const float *__restrict__ lt2_tab,
                                                        const float *__restrict__ nullsc_tab,
                                                        const int32_t *__restrict__ todo, int64_t ntodo,
                                                        float *__restrict__ nullsc, float *__restrict__ filtersc) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntodo) return;
  const int64_t sid = todo ? (int64_t)todo[t] : t;
  const int L = sq.len[sid];
  const uint8_t *s = sq.data + sq.off[sid];
  const float *e = eo + (size_t)(eo_stride ? t * eo_stride : 0);
  const float logsc = bias_forward(s, L, M, e, p1_tab[L]);
  if (nullsc) nullsc[sid] = nullsc_tab[L];
  filtersc[sid] = (logsc + lt1_tab[L]) + lt2_tab[L];
}

}  // namespace bath

// ============================================================================================
// Host side: batched entry points.
// ============================================================================================
namespace bath {

MsvConsts msv_consts(const bath_hip_oprofile *om) { return MsvConsts{om->tbm_b, om->tec_b, om->base_b, om->bias_b, om->scale_b}; }

VitConsts vit_consts(const bath_hip_oprofile *om) {
  VitConsts c{};
  c.base_w = om->base_w; c.xwE_loop = om->xw_E[0]; c.xwE_move = om->xw_E[1]; c.scale_w = om->scale_w;
  c.scale_b = om->scale_b; c.base_b = om->base_b; c.tec_b = om->tec_b; c.bias_b = om->bias_b;
  c.Q8 = std::max(2, ((om->M - 1) / 8) + 1);
  return c;
}

// sequences sorted by length so the 64 lanes of a wavefront finish together
static int length_order(bath_hip_ctx *ctx, const bath_hip_seqs *sq, DevBuf &buf) {
  std::vector<int32_t> order((size_t)sq->n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return sq->h_len[a] > sq->h_len[b]; });
  BATH_HIP_TRY(ctx, buf.reserve(order.size() * sizeof(int32_t) + 16));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(buf.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  return BATH_OK;
}

int launch_ssv_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_order, int16_t *d_v) {
  if (v.n == 0) return BATH_OK;
  const int G = om->G;
  const size_t shmem = (size_t)kSsvRows * om->ssv_row_bytes;
  const int threads = 256;
  const int blocks = (int)((v.n * G + threads - 1) / threads);
  const int rb = om->ssv_row_bytes;
  bool launched = false;
#define BATH_SSV_CASE(N, GG)                                                                                             \
  if (!launched && om->NR == N && G == GG) {                                                                             \
    if (shmem > 64 * 1024) (void)bath::allow_max_lds((const void *)ssv_lane_kernel<N, GG>); \
    hipLaunchKernelGGL((ssv_lane_kernel<N, GG>), dim3(blocks), dim3(threads), shmem, ctx->stream, v, d_order, om->d_ssv, rb, d_v); \
    launched = true;                                                                                                     \
  }
  BATH_SSV_SHAPES(BATH_SSV_CASE)
#undef BATH_SSV_CASE
  if (!launched) { ctx->set_error("SSV kernel: no tile shape for model length " + std::to_string(om->M)); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

constexpr size_t kWaveTableLdsMax = 150 * 1024;      // Forward / Backward tables above this stay in global memory (160 KB of LDS per CU)

static int columns_per_lane(int M) {
  int c = (M + 63) / 64;
  for (int opt : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32}) if (c <= opt) return opt;
  return -1;
}

#define BATH_C_SWITCH(C, BODY)                                   \
  switch (C) {                                                   \
    case 1: { constexpr int CC = 1; BODY } break;                \
    case 2: { constexpr int CC = 2; BODY } break;                \
    case 3: { constexpr int CC = 3; BODY } break;                \
    case 4: { constexpr int CC = 4; BODY } break;                \
    case 6: { constexpr int CC = 6; BODY } break;                \
    case 8: { constexpr int CC = 8; BODY } break;                \
    case 12: { constexpr int CC = 12; BODY } break;              \
    case 16: { constexpr int CC = 16; BODY } break;              \
    case 24: { constexpr int CC = 24; BODY } break;              \
    case 32: { constexpr int CC = 32; BODY } break;              \
    default: ctx->set_error("model too long (> 2048 nodes)"); return BATH_EINVAL; \
  }

static int wave_grid(bath_hip_ctx *ctx, int64_t njobs) {
  int64_t blocks = (njobs + 3) / 4;
  int64_t cap = (int64_t)ctx->prop.multiProcessorCount * 8;
  return (int)std::max<int64_t>(1, std::min(blocks, cap));
}

int launch_msv_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev, bool lane_ok) {
  if (ntodo == 0) return BATH_OK;
  if (lane_ok) {                                                 // models of up to 152 nodes: a lane per target (bath_msv_lane.hip); not for a few thousand targets
    const int st = launch_msv_lane(ctx, om, v, d_todo, ntodo, d_sc, d_status, ntodo_dev);
    if (st != BATH_ENORESULT) return st;
  }
  const int C = columns_per_lane(om->M);
  const int grid = wave_grid(ctx, ntodo);
  BATH_C_SWITCH(C, hipLaunchKernelGGL(msv_wave_kernel<CC>, dim3(grid), dim3(256), 0, ctx->stream, v, om->M, om->d_rb, om->rb_stride,
                                      om->lt.d_tjb, msv_consts(om), d_todo, ntodo, ntodo_dev, d_sc, d_status);)
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_vit_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status,
                    const VitWindowArgs *wa, const int *ntodo_dev) {
  if (ntodo == 0) return BATH_OK;
  const int C = columns_per_lane(om->M);
  const int grid = wave_grid(ctx, ntodo);
  const size_t shmem = ((size_t)kKp * vit_em_stride(om->M) + 64 * (size_t)C) * sizeof(int16_t);
  VitConsts c = vit_consts(om);
  const float *fsc = nullptr; const uint8_t *ssv = nullptr; WindowRec *wins = nullptr; int *wc = nullptr; int cap = 0; int32_t *kmm = nullptr;
  if (wa) { c.invP_vit = wa->invP_vit; c.invP_msv = wa->invP_msv; fsc = wa->d_filtersc; ssv = wa->d_ssv_scores; wins = (WindowRec *)wa->d_wins; wc = wa->d_win_count; cap = wa->win_cap; kmm = wa->d_kminmax; }
  BATH_C_SWITCH(C, {
    if (shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)vit_wave_kernel<CC>));
    hipLaunchKernelGGL(vit_wave_kernel<CC>, dim3(grid), dim3(256), shmem, ctx->stream, v, om->M, om->d_rw, om->d_tw, om->lt.d_xwmove, om->lt.d_tjb, c,
                       d_todo, ntodo, ntodo_dev, d_sc, d_status, fsc, ssv, wins, wc, cap, kmm);
  })
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_fwd_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev,
                    float *d_xmx, const int64_t *d_xmx_off, float *d_dp, const int64_t *d_dp_off, int unihit, const int32_t *d_cfg_len) {
  if (ntodo == 0) return BATH_OK;
  const int C = columns_per_lane(om->M);
  const int grid = wave_grid(ctx, ntodo);
  const size_t shmem = ((size_t)kKp * wave_em_stride(om->M) + 64 * (size_t)C) * sizeof(float);
  FwdConsts c{om->xf_E[0], om->xf_E[1]};
  if (unihit) { c.xfE_loop = 0.0f; c.xfE_move = 1.0f; }         // p7_oprofile_ReconfigUnihit, p7_oprofile.c:1421-1422
  if (shmem > kWaveTableLdsMax || C > 16) {                     // tables stay in global memory (beyond 16 nodes per lane the transitions do not fit the registers either)
    BATH_C_SWITCH(C, {
      hipLaunchKernelGGL((fwd_wave_kernel<CC, true>), dim3(grid), dim3(256), 0, ctx->stream, v, om->M, om->d_rf, om->d_tf, om->lt.d_pmove, c, d_todo, ntodo, ntodo_dev, d_sc, d_status, d_xmx, d_xmx_off, d_dp, d_dp_off, unihit, d_cfg_len);
    })
  } else {
    BATH_C_SWITCH(C, {
      if (shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fwd_wave_kernel<CC>));
      hipLaunchKernelGGL(fwd_wave_kernel<CC>, dim3(grid), dim3(256), shmem, ctx->stream, v, om->M, om->d_rf, om->d_tf, om->lt.d_pmove, c, d_todo, ntodo, ntodo_dev, d_sc, d_status, d_xmx, d_xmx_off, d_dp, d_dp_off, unihit, d_cfg_len);
    })
  }
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_bwd_wave(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, int64_t n, const float *d_fwd_xmx, const int64_t *d_xmx_off,
                    float *d_sc, int32_t *d_status, float *d_bck_xmx, float *d_dp, const int64_t *d_dp_off, int unihit) {
  if (n == 0) return BATH_OK;
  const int C = columns_per_lane(om->M);
  const int grid = wave_grid(ctx, n);
  const size_t shmem = ((size_t)kKp * wave_em_stride(om->M) + 64 * (size_t)C) * sizeof(float);
  FwdConsts c{om->xf_E[0], om->xf_E[1]};
  if (unihit) { c.xfE_loop = 0.0f; c.xfE_move = 1.0f; }
  if (shmem > kWaveTableLdsMax || C > 16) {
    BATH_C_SWITCH(C, {
      hipLaunchKernelGGL((bwd_wave_kernel<CC, true>), dim3(grid), dim3(256), 0, ctx->stream, v, om->M, om->d_rfb, om->d_tfb, om->lt.d_pmove, c, n, d_fwd_xmx, d_xmx_off, d_sc, d_status, d_bck_xmx, d_dp, d_dp_off, unihit);
    })
  } else {
    BATH_C_SWITCH(C, {
      if (shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)bwd_wave_kernel<CC>));
      hipLaunchKernelGGL(bwd_wave_kernel<CC>, dim3(grid), dim3(256), shmem, ctx->stream, v, om->M, om->d_rf, om->d_tf, om->lt.d_pmove, c, n, d_fwd_xmx, d_xmx_off, d_sc, d_status, d_bck_xmx, d_dp, d_dp_off, unihit);
    })
  }
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_bias_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const float *d_eo, int eo_stride, const int32_t *d_todo, int64_t ntodo,
                     float *d_nullsc, float *d_filtersc) {
  if (ntodo == 0) return BATH_OK;
  const int blocks = (int)((ntodo + 255) / 256);
  hipLaunchKernelGGL(bias_lane_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v, om->M, d_eo ? d_eo : om->d_bias_eo, eo_stride, om->lt.d_p1,
                     om->lt.d_lt1, om->lt.d_lt2, om->lt.d_nullsc, d_todo, ntodo, d_nullsc, d_filtersc);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_ssv_classify(bath_hip_ctx *ctx, const bath_hip_oprofile *om, int64_t n, const int32_t *d_len, const int16_t *d_v, float *d_sc, int32_t *d_status) {
  if (n == 0) return BATH_OK;
  hipLaunchKernelGGL(ssv_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n, d_len, d_v, om->lt.d_tjb, msv_consts(om), d_sc, d_status);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath

static int check_batch(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq) {
  if (!ctx || !om || !sq) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return om->ensure_len_tables(sq->maxlen);
}

static int ssv_or_msv(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status, bool do_msv) {
  int st = check_batch(ctx, om, sq);
  if (st != BATH_OK) return st;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  DevBuf &b_order = ctx->scratch[0], &b_v = ctx->scratch[1], &b_sc = ctx->scratch[2], &b_st = ctx->scratch[3], &b_todo = ctx->scratch[4];
  if ((st = length_order(ctx, sq, b_order)) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, b_v.reserve((size_t)n * sizeof(int16_t)));
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * sizeof(float)));
  BATH_HIP_TRY(ctx, b_st.reserve((size_t)n * sizeof(int32_t)));
  if ((st = launch_ssv_lane(ctx, om, sq->view(), b_order.as<int32_t>(), b_v.as<int16_t>())) != BATH_OK) return st;
  if ((st = launch_ssv_classify(ctx, om, n, sq->d_len, b_v.as<int16_t>(), b_sc.as<float>(), b_st.as<int32_t>())) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, b_sc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(status, b_st.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (!do_msv) return BATH_OK;
  std::vector<int32_t> todo;
  for (int64_t i = 0; i < n; i++) if (status[i] == BATH_ENORESULT) todo.push_back((int32_t)i);
  if (todo.empty()) return BATH_OK;
  BATH_HIP_TRY(ctx, b_todo.reserve(todo.size() * sizeof(int32_t)));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(b_todo.p, todo.data(), todo.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  if ((st = launch_msv_wave(ctx, om, sq->view(), b_todo.as<int32_t>(), (int64_t)todo.size(), b_sc.as<float>(), b_st.as<int32_t>(), nullptr)) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, b_sc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(status, b_st.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}

extern "C" int bath_hip_ssvfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status) {
  return ssv_or_msv(ctx, om, sq, sc, status, false);
}
extern "C" int bath_hip_msvfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status) {
  return ssv_or_msv(ctx, om, sq, sc, status, true);
}

template <class F>
static int score_batch(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status, F launch) {
  int st = check_batch(ctx, om, sq);
  if (st != BATH_OK) return st;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  DevBuf &b_sc = ctx->scratch[2], &b_st = ctx->scratch[3];
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * sizeof(float)));
  BATH_HIP_TRY(ctx, b_st.reserve((size_t)n * sizeof(int32_t)));
  if ((st = launch(b_sc.as<float>(), b_st.as<int32_t>())) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, b_sc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (status) BATH_HIP_TRY(ctx, hipMemcpyAsync(status, b_st.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}

extern "C" int bath_hip_vitfilter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status) {
  if (vit_lane_supported(om) && sq && sq->n > 0) {       // lane-per-target kernel: targets sorted by length
    int st0 = check_batch(ctx, om, sq);
    if (st0 != BATH_OK) return st0;
    DevBuf &b_order = ctx->scratch[0];
    if ((st0 = length_order(ctx, sq, b_order)) != BATH_OK) return st0;
    return score_batch(ctx, om, sq, sc, status, [&](float *d_sc, int32_t *d_st) {
      return launch_vit_lane(ctx, om, sq->view(), b_order.as<int32_t>(), sq->n, nullptr, d_sc, d_st, nullptr); });
  }
  return score_batch(ctx, om, sq, sc, status, [&](float *d_sc, int32_t *d_st) { return launch_vit_wave(ctx, om, sq->view(), nullptr, sq->n, d_sc, d_st, nullptr, nullptr); });
}
extern "C" int bath_hip_forward_parser(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *sc, int32_t *status) {
  return score_batch(ctx, om, sq, sc, status, [&](float *d_sc, int32_t *d_st) { return launch_fwd_wave(ctx, om, sq->view(), nullptr, sq->n, d_sc, d_st, nullptr, nullptr, nullptr); });
}

// p7_ForwardParser + p7_BackwardParser with their special-state rows (what p7_domaindef reads): xmx layout (L_i+1) x
// {E,N,J,B,C,SCALE} at xmx_offsets[i] floats, for both passes.
extern "C" int bath_hip_fwdback_parser(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const int64_t *xmx_offsets,
                                       float *fwd_sc, float *bck_sc, int32_t *fwd_status, int32_t *bck_status, float *fwd_xmx, float *bck_xmx) {
  int st = check_batch(ctx, om, sq);
  if (st != BATH_OK) return st;
  if (!xmx_offsets || !fwd_sc || !bck_sc) return BATH_EINVAL;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  const size_t nx = (size_t)xmx_offsets[n];
  DevBuf &b_sc = ctx->scratch[2], &b_st = ctx->scratch[3], &b_off = ctx->scratch[4], &b_fx = ctx->scratch[6], &b_bx = ctx->scratch[7];
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 2 * sizeof(float)));
  BATH_HIP_TRY(ctx, b_st.reserve((size_t)n * 2 * sizeof(int32_t)));
  BATH_HIP_TRY(ctx, b_off.reserve((size_t)(n + 1) * sizeof(int64_t)));
  BATH_HIP_TRY(ctx, b_fx.reserve(nx * sizeof(float) + 64));
  BATH_HIP_TRY(ctx, b_bx.reserve(nx * sizeof(float) + 64));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(b_off.p, xmx_offsets, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  float *d_fsc = b_sc.as<float>(), *d_bsc = d_fsc + n;
  int32_t *d_fst = b_st.as<int32_t>(), *d_bst = d_fst + n;
  if ((st = launch_fwd_wave(ctx, om, sq->view(), nullptr, n, d_fsc, d_fst, nullptr, b_fx.as<float>(), b_off.as<int64_t>())) != BATH_OK) return st;
  if ((st = launch_bwd_wave(ctx, om, sq->view(), n, b_fx.as<float>(), b_off.as<int64_t>(), d_bsc, d_bst, b_bx.as<float>())) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(fwd_sc, d_fsc, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(bck_sc, d_bsc, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (fwd_status) BATH_HIP_TRY(ctx, hipMemcpyAsync(fwd_status, d_fst, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (bck_status) BATH_HIP_TRY(ctx, hipMemcpyAsync(bck_status, d_bst, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (fwd_xmx) BATH_HIP_TRY(ctx, hipMemcpyAsync(fwd_xmx, b_fx.p, nx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (bck_xmx) BATH_HIP_TRY(ctx, hipMemcpyAsync(bck_xmx, b_bx.p, nx * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}
extern "C" int bath_hip_bias_filter(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, float *nullsc, float *filtersc) {
  int st = check_batch(ctx, om, sq);
  if (st != BATH_OK) return st;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  DevBuf &b_a = ctx->scratch[2], &b_b = ctx->scratch[5];
  BATH_HIP_TRY(ctx, b_a.reserve((size_t)n * sizeof(float)));
  BATH_HIP_TRY(ctx, b_b.reserve((size_t)n * sizeof(float)));
  if ((st = launch_bias_lane(ctx, om, sq->view(), nullptr, 0, nullptr, n, b_a.as<float>(), b_b.as<float>())) != BATH_OK) return st;
  if (nullsc) BATH_HIP_TRY(ctx, hipMemcpyAsync(nullsc, b_a.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(filtersc, b_b.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}
