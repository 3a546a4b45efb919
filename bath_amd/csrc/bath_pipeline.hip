// bath_pipeline.hip -- the filter cascade of p7_Pipeline_BATH() as a batched GPU pipeline.
//
// Reference control flow (src/p7_pipeline.c:1632-1791), per ORF of a DNA window and strand:
//   MSV (F1) -> bias filter (F1) -> ViterbiFilter_BATH (F2) or SSVFilter_BATH windows
//   -> local-composition re-filter -> ForwardParser (F3, or F4 when fs_pipe)
// with the ORFs produced by easel's six-frame translation (src/bathsearch.c:384-392).
//
// GPU formulation:
//   1. bath_orfs.hip: six-frame translation, ORF finding and a length-sorted ORF work list (lane per stream).
//   2. ssv_orf_kernel: one LANE per ORF (G lanes for long models) runs the SSV recurrence with the whole DP row in
//      registers, two binary16 cells each; lanes of a wave hold ORFs of equal length.  The lane compares its maximum with a
//      per-length threshold table computed on the host with the reference's double-precision P-value maths (so the
//      F1 decision is bit-identical) and appends the ~2% survivors to a candidate list.
//   3. survivors flow through decision kernels (lane per candidate) and DP kernels (lane or wave per candidate)
//      with device-side work lists: no host round trip until the final copy-out.  Their residues are read in place
//      from the amino-acid streams written by step 1.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <map>
#include <thread>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"
#include "host_model.hpp"

using namespace bath;

namespace bath {

static const double kLog2 = 0.69314718055994529;
constexpr int kVitLongOrf = 128;         // ORFs longer than this take the wave-per-ORF Viterbi kernel (see the pipeline)

// ---------------------------------------------------------------------------------------------
// candidate storage (structure of arrays, indexed by candidate id)
// ---------------------------------------------------------------------------------------------
enum { FLAG_VIT_RUN = 1, FLAG_HAS_WIN = 2 };

struct Counters {                     // device-side counters, one struct per pipeline call
  int cand_count, todo_msv, todo_vit, todo_ssvb, todo_vit2, todo_fwd, win_count, overflow;
  int todo_lb, pad_;                                        // candidates that need the local-composition bias filter
  unsigned long long n_orfs, orf_res;                       // ORFs >= minlen, their total aa
  unsigned long long n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
  unsigned long long pos_past_msv, pos_past_bias, pos_past_vit, pos_past_fwd;
  unsigned long long res_vit, res_fwd;                      // residues entering Viterbi / Forward
  unsigned long long res_seen;                              // residues of ORFs lying inside a window's context: scored by SSV but not the reference's (p7_pipeline.c:1635)
};

struct Params {
  double F1, F2, F3, F4;
  int do_bias, fs_pipe, minlen;
  float evparam[BATH_NEVPARAM];
};

// ---------------------------------------------------------------------------------------------
// 1. SSV over the length-sorted ORF list, lane per ORF (persistent waves striding over the list)
// ---------------------------------------------------------------------------------------------
// (tiles of at most 76 registers: four 256-thread blocks per CU, 128 VGPRs each; a group of G lanes with such tiles has rows of at
// most 4 * 76 * G bytes, so four copies of its cost table always fit a CU's LDS -- a 1024-thread variant existed and could never be chosen)
template <int NR, int G>
__global__ __launch_bounds__(256, NR <= 76 ? 4 : 1) void ssv_orf_kernel(const uint8_t *__restrict__ aa, const OrfRec *__restrict__ orfs, const int *__restrict__ n_orfs_dev,
                                                      SeqView dna, const int16_t *__restrict__ cost_tab, int row_bytes,
                                                      const int16_t *__restrict__ emit_thresh, int thresh_max,
                                                      Cand cand, int cand_cap, Counters *__restrict__ ctr, int chunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // chunk > 0: not persistent -- a wave scores <chunk> consecutive groups of the list and exits, so that wave slots come free
  // all the time and kernels of higher-priority streams get onto the chip while this one is running
  if (chunk > 0 && (int64_t)blockIdx.x * (blockDim.x >> 6) * chunk * (64 / G) >= (int64_t)*n_orfs_dev) return;
  {
    const int n32 = kSsvRows * row_bytes / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(cost_tab);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < n32; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  constexpr int TPW = 64 / G;                                   // ORFs per wave
  const int64_t n_orfs = *n_orfs_dev;
  const int lane = threadIdx.x & 63;
  const int grank = SsvGroups<G>::rank(lane);
  const char *tile = lds + grank * (4 * NR);
  const unsigned tile_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)tile;   // LDS byte address
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const s16x2 fl = {0, 0};                                      // the begin score
  const int64_t tb_first = chunk > 0 ? wave0 * chunk * TPW : wave0 * TPW, tb_step = chunk > 0 ? TPW : nwaves * TPW;
  const int64_t tb_end = chunk > 0 ? (tb_first + (int64_t)chunk * TPW < n_orfs ? tb_first + (int64_t)chunk * TPW : n_orfs) : n_orfs;
  for (int64_t tb = tb_first; tb < tb_end; tb += tb_step) {
    const int64_t t = tb + SsvGroups<G>::slot(lane);
    const bool live = t < n_orfs;
    OrfRec rec{0, 0, 0};
    if (live) rec = orfs[t];
    const int L = rec.len_sf & 0x0fffffff;
#ifdef BATH_SSV_POOL_PROBE
    // Timing probe (WRONG results; tools/ssv_pool_probe.sh): the residue reads as they would be from a pool written in work-list
    // order and interleaved per wave -- the 8 bytes of a wave's 64 lanes contiguous (512 B per load instruction), a wave's groups
    // one after the other -- i.e. with NO over-fetch: what the SSV kernel itself could gain from such a pool, before the pool's cost
    const uint8_t *s = aa + ((size_t)(tb / TPW) * 8192) % ((size_t)1 << 27) + (size_t)lane * 8;
#define BATH_SSV_RES(i8) (s + (size_t)(i8) * 64)
#else
    const uint8_t *s = aa + rec.aa_off;
#define BATH_SSV_RES(i8) (s + (i8))
#endif
    const int Lw = wave_max_i32(L);
    s16x2 reg[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) reg[r] = fl;
    s16x2 xE = fl, xE2 = fl;
    // residues 8 at a time, the next 8 in flight: an ORF's 64-byte sectors are touched by half as many loads as with a dword
    // per 4 rows (HBM traffic of the launch -32%, kernel -4%; the loop also stops at the wave's longest ORF instead of the next
    // multiple of 4 rows).  Staging the residues through LDS instead (three global_load_lds_dwordx4 per wave and 32 rows, double
    // buffered) was measured at -2% time and -20% traffic and not kept: most of the excess traffic is 128-byte lines shared by
    // ORFs of different lengths, which no read pattern of this kernel can merge.
    uint64_t qnext = (0 < L) ? *reinterpret_cast<const uint64_t *>(BATH_SSV_RES(0)) : 0x1d1d1d1d1d1d1d1dull;
    for (int i0 = 0; i0 < Lw; i0 += 8) {
      const uint64_t q = qnext;
      qnext = (i0 + 8 < L) ? *reinterpret_cast<const uint64_t *>(BATH_SSV_RES(i0 + 8)) : 0x1d1d1d1d1d1d1d1dull;
      const int nrow = min(8, Lw - i0);                              // wave-uniform
      // the row's overhead is kept to full-rate 32-bit ops: a bit-field extract on the dword holding the residue (no 64-bit
      // shift), a 24-bit multiply for the row offset (v_mul_lo_u32 is quarter rate); bytes inside an ORF are residue codes < 29
      for (int h = 0; h < 2; h++) {
        const uint32_t qh = h ? (uint32_t)(q >> 32) : (uint32_t)q;
        const int nh = min(4, nrow - 4 * h);                         // wave-uniform
        for (int j = 0; j < nh; j++) {
          int x = (int)__builtin_amdgcn_ubfe(qh, 8 * j, 8);
          x = (i0 + 4 * h + j < L) ? x : kRowReset;
          const unsigned carry = ssv_carry<NR, G>(reg, grank);
          ssv_row_pipe<NR, 3>(reg, xE, xE2, tile_addr + __umul24((unsigned)x, (unsigned)row_bytes), carry);
        }
      }
    }
    xE = ssv_max3(xE, xE2, xE2);
    const int v = ssv_group_max<G>(xE);
    if (dna.context) {                  // windows with context only: ORFs the previous window already searched do not count as MSV work
      unsigned long long r = 0;
      if (live && grank == 0) {
        const int sf = (int)((unsigned)rec.len_sf >> 28);
        const int64_t w = rec.w;
        const int startj = (int32_t)(rec.aa_off - (2 * dna.off[w] + 96 * w + (int64_t)sf * orf_stream_pitch(dna.len[w])));
        const int C = dna.context[w], start_s = sf % 3 + 3 * startj + 1;
        if (sf < 3 ? (start_s + 3 * L - 1 < C) : (dna.len[w] - start_s + 1 < C)) r = (unsigned long long)L;
      }
      if (__ballot(r != 0)) { for (int d = 32; d >= 1; d >>= 1) r += __shfl_xor(r, d, 64); if (lane == 0) atomicAdd(&ctr->res_seen, r); }
    }
    if (live && grank == 0 && v >= (int)emit_thresh[min(L, thresh_max)]) {
      const int sf = (int)((unsigned)rec.len_sf >> 28);
      const int64_t w = rec.w;
      const int64_t stream = 2 * dna.off[w] + 96 * w + (int64_t)sf * orf_stream_pitch(dna.len[w]);
      const int startj = (int32_t)(rec.aa_off - stream);
      bool seen = false;                // an ORF that lies inside the previous window's share of this one (p7_pipeline.c:1635-1637)
      if (dna.context) {
        const int C = dna.context[w], start_s = sf % 3 + 3 * startj + 1;
        seen = sf < 3 ? (start_s + 3 * L - 1 < C) : (dna.len[w] - start_s + 1 < C);
      }
      const int slot = seen ? cand_cap + 1 : atomicAdd(&ctr->cand_count, 1);
      if (seen) {
      } else if (slot < cand_cap) {
        cand.window[slot] = w; cand.sf[slot] = sf; cand.startj[slot] = startj; cand.len[slot] = L;
        cand.v[slot] = (int16_t)min(v, 32767); cand.off[slot] = rec.aa_off;
      } else atomicOr(&ctr->overflow, 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// decision kernels (lane per candidate)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double d_gumbel_surv(double x, double mu, double lambda) {
  const double ey = -exp(-lambda * (x - mu));
  if (fabs(ey) < 5e-9) return -ey;
  return 1.0 - exp(ey);
}
__device__ __forceinline__ double d_exp_surv(double x, double mu, double lambda) { return (x < mu) ? 1.0 : exp(-lambda * (x - mu)); }

// classify the SSV maxima (ssvfilter.c:876-925); undecided targets go to the full-MSV list
__global__ void classify_kernel(Cand cand, int cand_cap, Counters *__restrict__ ctr, const uint8_t *__restrict__ tjb_tab, MsvConsts mc,
                                int32_t *__restrict__ todo_msv) {
  const int ncand = min(ctr->cand_count, cand_cap);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x) {
    float sc = 0.f;
    const int st = ssv_classify(cand.v[c], tjb_tab[cand.len[c]], mc, &sc);
    cand.usc[c] = sc; cand.msv_status[c] = st; cand.stage[c] = 0; cand.flags[c] = 0;
    cand.vfsc[c] = -INFINITY; cand.fwdsc[c] = -INFINITY; cand.vit_status[c] = 0; cand.filtersc[c] = 0.f; cand.P[c] = 1.0;
    cand.kminmax[2 * c] = 1 << 30; cand.kminmax[2 * c + 1] = 0;
    if (st == BATH_ENORESULT) todo_msv[atomicAdd(&ctr->todo_msv, 1)] = c;
  }
}

// MSV P-value (F1), bias filter (F1), then route to Viterbi (P > F2) or SSV windows (P <= F2): p7_pipeline.c:1649-1677
__global__ void f1_bias_kernel(Cand cand, int cand_cap, Counters *__restrict__ ctr, Params p, const uint8_t *__restrict__ pool, int M,
                               const float *__restrict__ eo, const float *__restrict__ nullsc_tab, const float *__restrict__ p1_tab,
                               const float *__restrict__ lt1_tab, const float *__restrict__ lt2_tab,
                               int32_t *__restrict__ todo_vit, int32_t *__restrict__ todo_ssvb) {
  __shared__ float s_eo[kKp * 2];                          // the filter HMM's emission odds: read once per residue, from LDS
  for (int i = threadIdx.x; i < kKp * 2; i += blockDim.x) s_eo[i] = eo[i];
  __syncthreads();
  const int ncand = min(ctr->cand_count, cand_cap);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x) {
    const int L = cand.len[c];
    const float usc = cand.usc[c];
    const float nullsc = nullsc_tab[L];
    cand.nullsc[c] = nullsc;
    float seqsc = (float)((double)(usc - nullsc) / kLog2);
    double P = d_gumbel_surv(seqsc, p.evparam[0], p.evparam[1]);
    cand.P[c] = P;
    if (P > p.F1) continue;
    atomicAdd(&ctr->n_past_msv, 1ull); atomicAdd(&ctr->pos_past_msv, (unsigned long long)L * 3ull);
    cand.stage[c] = 1;
    float filtersc = nullsc;
    if (p.do_bias) {
      filtersc = bias_forward(pool + cand.off[c], L, M, s_eo, p1_tab[L]);
      filtersc = (filtersc + lt1_tab[L]) + lt2_tab[L];
      seqsc = (float)((double)(usc - filtersc) / kLog2);
      P = d_gumbel_surv(seqsc, p.evparam[0], p.evparam[1]);
      cand.P[c] = P; cand.filtersc[c] = filtersc;
      if (P > p.F1) continue;
    }
    cand.filtersc[c] = filtersc;
    atomicAdd(&ctr->n_past_bias, 1ull); atomicAdd(&ctr->pos_past_bias, (unsigned long long)L * 3ull);
    cand.stage[c] = 2;
    if (P > p.F2) { cand.flags[c] |= FLAG_VIT_RUN; todo_vit[atomicAdd(&ctr->todo_vit, 1)] = c; atomicAdd(&ctr->res_vit, (unsigned long long)L); }
    else todo_ssvb[atomicAdd(&ctr->todo_ssvb, 1)] = c;
  }
}

// p7_SSVFilter_BATH (msvfilter.c:250-427): diagonal windows for strong MSV hits, wave per candidate.
// Lane l owns nodes l*C+1 .. l*C+C (byte arithmetic of the reference kept exactly).  When the row maximum
// reaches the threshold, the reference scans its striped vectors q=0..Q-1, lanes 0..15 and keeps the strictly
// greatest byte (msvfilter.c:358-366): i.e. the first maximal cell in that order, found here by a wave min
// over the striped rank.
template <int C>
__global__ __launch_bounds__(256) void ssv_bath_kernel(Cand cand, const Counters *__restrict__ ctr, const int32_t *__restrict__ todo,
                                                       const uint8_t *__restrict__ pool, int M, const uint8_t *__restrict__ rb, int rb_stride,
                                                       const uint8_t *__restrict__ ssv_scores, const uint8_t *__restrict__ tjb_tab,
                                                       const float *__restrict__ nullsc_tab, MsvConsts mc, double invP_f1,
                                                       WindowRec *__restrict__ wins, int win_cap, Counters *__restrict__ ctrw) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int ntodo = ctr->todo_ssvb;
  const int Q = max(2, ((M - 1) / 16) + 1);
  for (int64_t job = wid; job < ntodo; job += nw) {
    const int c = todo[job];
    const int L = cand.len[c];
    const uint8_t *s = pool + cand.off[c];
    const int tjb = tjb_tab[L];
    const float nullsc = nullsc_tab[L];
    const int sc_thresh = (uint8_t)(int)ceil((((double)nullsc + ((double)(float)invP_f1 * kLog2) + 3.0) * (double)mc.scale_b) + mc.base + mc.tec + tjb);
    const int tjbm = (uint8_t)((int8_t)tjb + (int8_t)mc.tbm);
    const int xB = satu8(mc.base - tjbm);
    int kmin = 1 << 30, kmax = 0;
    int dp[C];
#pragma unroll
    for (int k = 0; k < C; k++) dp[k] = 0;
    const int ph = threadIdx.x & 63;
    int rbuf = (ph < L) ? (int)s[ph] : 0, rnext = 0;          // residues 64 rows at a time, a lane each, the next 64 in flight (see fwd_wave_kernel)
    for (int i = 1; i <= L; i++) {
      const int j = (i - 1) & 63;
      if (j == 0) { const int q = i - 1 + 64 + ph; rnext = (q < L) ? (int)s[q] : 0; }
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
        sv = satu8(sv + mc.bias);
        sv = satu8(sv - cost);
        prev = dp[k];
        dp[k] = sv;
        xE = max(xE, sv);
      }
      xE = wave_max_i32(xE);
      if (xE >= sc_thresh) {
        int rank = 1 << 30;
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int node = lane * C + k + 1;
          if (node <= M && dp[k] == xE) rank = min(rank, ((node - 1) % Q) * 16 + (node - 1) / Q);
        }
        rank = wave_min_i32(rank);
        int end = (rank / 16) + Q * (rank % 16) + 1;
        int rem_sc = xE;
#pragma unroll
        for (int k = 0; k < C; k++) dp[k] = 0;
        int start = end, target_end = i, target_start = i;
        int sc = rem_sc;
        while (rem_sc > mc.base - tjb - mc.tbm && start >= 1 && target_start >= 1) {
          rem_sc -= mc.bias - (int)ssv_scores[(size_t)start * kKp + min((int)s[target_start - 1], kKp - 1)];
          --start; --target_start;
        }
        start++; target_start++;
        int k = end + 1, n = target_end + 1, max_end = target_end, max_sc = sc, since = 0;
        while (k < M && n <= L) {
          sc += mc.bias - (int)ssv_scores[(size_t)k * kKp + min((int)s[n - 1], kKp - 1)];
          if (sc >= max_sc) { max_sc = sc; max_end = n; since = 0; }
          else { since++; if (since == 5) break; }
          k++; n++;
        }
        end += (max_end - target_end);
        target_end = max_end;
        float ret = ((float)(max_sc - tjb) - (float)mc.base);
        ret /= mc.scale_b;
        ret = (float)((double)ret - 3.0);
        if (lane == 0) {
          const int slot = atomicAdd(&ctrw->win_count, 1);
          if (slot < win_cap) wins[slot] = WindowRec{c, target_start, end, end - start + 1, ret};
        }
        kmin = min(kmin, start); kmax = max(kmax, end);
        i = target_end;
        { const int base = i & ~63; rbuf = (base + ph < L) ? (int)s[base + ph] : 0; rnext = (base + 64 + ph < L) ? (int)s[base + 64 + ph] : 0; }   // the row loop jumped: refill
      }
    }
    if (lane == 0) { cand.kminmax[2 * c] = kmin; cand.kminmax[2 * c + 1] = kmax; }
  }
}

// p7_pli_ComputeLocalCompo (p7_pipeline.c:427-458) + p7_bg_SetFilter + esl_hmm_Configure for one candidate
// p7_pli_ComputeLocalCompo's sum for ONE residue x over the (widened) node range: the 20 sums are independent of each other
__device__ __forceinline__ float local_compo_x(const uint8_t *ssv_scores, int M, int base_b, float scale_b, const float *bgf, int k_start, int k_end, int x) {
  const int k_len = k_end - k_start + 1;
  if (k_len < 20) { k_start -= (20 - k_len) / 2; k_end += (20 - k_len) / 2; }
  k_start = max(1, k_start); k_end = min(M, k_end);
  float acc = 0.0f;
  for (int k = k_start; k <= k_end; k++) {
    const float lo = ((float)base_b - (float)ssv_scores[(size_t)k * kKp + x]) / scale_b;
    acc += bgf[x] * expf(lo);
  }
  return acc;
}
__device__ void local_compo_finish(float *compo /* [20], the sums */, const float *bgf, float *eo /* [Kp][2] */);
__device__ void local_compo_eo(const uint8_t *ssv_scores, int M, int base_b, float scale_b, const float *bgf, int k_start, int k_end, float *eo /* [Kp][2] */) {
  float compo[20];
  for (int x = 0; x < 20; x++) compo[x] = local_compo_x(ssv_scores, M, base_b, scale_b, bgf, k_start, k_end, x);
  local_compo_finish(compo, bgf, eo);
}
__device__ void local_compo_finish(float *compo, const float *bgf, float *eo) {
  float sum = 0.f, cc = 0.f;
  for (int x = 0; x < 20; x++) { const float y = compo[x] - cc; const float t = sum + y; cc = (t - sum) - y; sum = t; }
  if (sum != 0.0f) for (int x = 0; x < 20; x++) compo[x] /= sum;
  else for (int x = 0; x < 20; x++) compo[x] = 1.0f / 20.0f;
  for (int x = 0; x < 20; x++) { eo[2 * x] = bgf[x] / bgf[x]; eo[2 * x + 1] = compo[x] / bgf[x]; }
  eo[2 * 20] = eo[2 * 20 + 1] = 1.0f; eo[2 * 27] = eo[2 * 27 + 1] = 1.0f; eo[2 * 28] = eo[2 * 28 + 1] = 1.0f;
  // degenerate residues 21..26: B=DN J=IL Z=EQ O=K U=C X=all
  const int mem[6][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}, {-1, -1}};
  for (int dx = 0; dx < 6; dx++) {
    float n0 = 0.f, n1 = 0.f, den = 0.f;
    if (dx == 5) { for (int y = 0; y < 20; y++) { n0 += bgf[y]; n1 += compo[y]; den += bgf[y]; } }
    else {
      const int a = min(mem[dx][0], mem[dx][1]), b = max(mem[dx][0], mem[dx][1]);
      n0 += bgf[a]; n1 += compo[a]; den += bgf[a];
      if (b != a) { n0 += bgf[b]; n1 += compo[b]; den += bgf[b]; }
    }
    eo[2 * (21 + dx)] = den > 0.f ? n0 / den : 0.f;
    eo[2 * (21 + dx) + 1] = den > 0.f ? n1 / den : 0.f;
  }
}

// after Viterbi / SSV windows: F2 test, local-composition re-filter (p7_pipeline.c:1672-1718)
__global__ void post_vit_kernel(Cand cand, int cand_cap, Counters *__restrict__ ctr, Params p, int32_t *__restrict__ todo_lb, int32_t *__restrict__ todo_fwd) {
  const int ncand = min(ctr->cand_count, cand_cap);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x) {
    if (cand.stage[c] != 2) continue;
    const int L = cand.len[c];
    if (cand.flags[c] & FLAG_VIT_RUN) {
      const float seqsc = (float)((double)(cand.vfsc[c] - cand.filtersc[c]) / kLog2);
      const double P = d_gumbel_surv(seqsc, p.evparam[2], p.evparam[3]);
      cand.P[c] = P;
      if (P > p.F2) { cand.kminmax[2 * c] = 1 << 30; cand.kminmax[2 * c + 1] = 0; continue; }
    }
    atomicAdd(&ctr->n_past_vit, 1ull); atomicAdd(&ctr->pos_past_vit, (unsigned long long)L * 3ull);
    if (p.do_bias && cand.kminmax[2 * c] <= cand.kminmax[2 * c + 1]) {
      // the local-composition bias filter (p7_pipeline.c:1688-1712) is ~10^3 times the work of everything else here and only the
      // few per cent of candidates that passed Viterbi with a hit window need it: they go to a list and a dense kernel
      // (post_vit_local_kernel); done in place, every wave of this kernel would walk the long path for one or two of its lanes
      cand.flags[c] |= FLAG_HAS_WIN;
      todo_lb[atomicAdd(&ctr->todo_lb, 1)] = c;
      continue;
    }
    cand.stage[c] = 3;
    todo_fwd[atomicAdd(&ctr->todo_fwd, 1)] = c; atomicAdd(&ctr->res_fwd, (unsigned long long)L);
  }
}

// The local-composition bias filter in three dense steps.  (1) compo_terms_kernel, once per call: the 20 x 256 possible terms
// bg_f[x] * exp((base_b - s) / scale_b) of p7_pli_ComputeLocalCompo (s: a byte score).  (2) local_compo_kernel: three candidates per
// wave, lanes 20g..20g+19 each add the terms of one residue over the window's nodes (the 20 sums are independent; each still adds in
// ascending k).  (3) post_vit_local_kernel: a lane per candidate does the serial rest -- normalisation, the 2-state filter HMM
// over the ORF, the decisions -- with its emission odds in LDS.
__global__ void compo_terms_kernel(const float *__restrict__ bgf, int base_b, float scale_b, float *__restrict__ terms /* [20][256] */) {
  const int x = blockIdx.x, sc = threadIdx.x;
  const float lo = ((float)base_b - (float)sc) / scale_b;
  terms[x * 256 + sc] = bgf[x] * expf(lo);
}

__global__ __launch_bounds__(64) void local_compo_kernel(Cand cand, const Counters *__restrict__ ctr, int M, const uint8_t *__restrict__ ssv_scores,
                                                         const float *__restrict__ terms, const int32_t *__restrict__ todo_lb, float *__restrict__ compo_out /* [todo_lb][20] */) {
  const int lane = threadIdx.x, g = lane / 20, x = lane % 20;
  const int ntodo = ctr->todo_lb;
  for (int base = blockIdx.x * 3; base < ntodo; base += gridDim.x * 3) {
    const int job = base + g;
    if (g >= 3 || job >= ntodo) continue;
    const int c = todo_lb[job];
    int k_start = cand.kminmax[2 * c], k_end = cand.kminmax[2 * c + 1];
    const int k_len = k_end - k_start + 1;
    if (k_len < 20) { k_start -= (20 - k_len) / 2; k_end += (20 - k_len) / 2; }
    k_start = max(1, k_start); k_end = min(M, k_end);
    const float *tx = terms + x * 256;
    float acc = 0.0f;
    for (int k = k_start; k <= k_end; k++) acc += tx[ssv_scores[(size_t)k * kKp + x]];
    compo_out[(size_t)job * 20 + x] = acc;
  }
}

__global__ __launch_bounds__(64) void post_vit_local_kernel(Cand cand, Counters *__restrict__ ctr, Params p, const uint8_t *__restrict__ pool, int M,
                                                            const float *__restrict__ bgf, const float *__restrict__ compo_in,
                                                            const float *__restrict__ p1_tab, const float *__restrict__ lt1_tab, const float *__restrict__ lt2_tab,
                                                            const int32_t *__restrict__ todo_lb, int32_t *__restrict__ todo_vit2, int32_t *__restrict__ todo_fwd) {
  __shared__ float s_eo[64][kKp * 2 + 1];
  const int ntodo = ctr->todo_lb;
  for (int job = blockIdx.x * blockDim.x + threadIdx.x; job < ntodo; job += gridDim.x * blockDim.x) {
    const int c = todo_lb[job];
    const int L = cand.len[c];
    const float usc = cand.usc[c];
    float filtersc = cand.filtersc[c];
    const float vfsc = (cand.flags[c] & FLAG_VIT_RUN) ? cand.vfsc[c] : -INFINITY;
    float compo[20];
    for (int y = 0; y < 20; y++) compo[y] = compo_in[(size_t)job * 20 + y];
    float *eo = s_eo[threadIdx.x];
    local_compo_finish(compo, bgf, eo);
    float lf = bias_forward(pool + cand.off[c], L, M, eo, p1_tab[L]);
    lf = (lf + lt1_tab[L]) + lt2_tab[L];
    bool to_fwd = true;
    if (lf > filtersc) {
      filtersc = lf;
      cand.filtersc[c] = filtersc;
      if (vfsc == -INFINITY) {
        const float seqsc = (float)((double)(usc - filtersc) / kLog2);
        const double P = d_gumbel_surv(seqsc, p.evparam[0], p.evparam[1]);
        cand.P[c] = P;
        if (P > p.F2) { cand.stage[c] = 5; todo_vit2[atomicAdd(&ctr->todo_vit2, 1)] = c; atomicAdd(&ctr->res_vit, (unsigned long long)L); to_fwd = false; }
      } else {
        const float seqsc = (float)((double)(vfsc - filtersc) / kLog2);
        const double P = d_gumbel_surv(seqsc, p.evparam[2], p.evparam[3]);
        cand.P[c] = P;
        if (P > p.F2) to_fwd = false;                      // rejected; stays at stage 2 (the reference has already counted it past Vit)
      }
    }
    if (to_fwd) { cand.stage[c] = 3; todo_fwd[atomicAdd(&ctr->todo_fwd, 1)] = c; atomicAdd(&ctr->res_fwd, (unsigned long long)L); }
  }
}

// plain p7_ViterbiFilter re-run for candidates whose local filter score demanded it (p7_pipeline.c:1703-1708)
__global__ void post_vit2_kernel(Cand cand, Counters *__restrict__ ctr, Params p, const int32_t *__restrict__ todo_vit2, int32_t *__restrict__ todo_fwd) {
  const int ntodo = ctr->todo_vit2;
  for (int job = blockIdx.x * blockDim.x + threadIdx.x; job < ntodo; job += gridDim.x * blockDim.x) {
    const int c = todo_vit2[job];
    const float seqsc = (float)((double)(cand.vfsc[c] - cand.filtersc[c]) / kLog2);
    const double P = d_gumbel_surv(seqsc, p.evparam[2], p.evparam[3]);
    cand.P[c] = P;
    if (P > p.F2) { cand.stage[c] = 2; continue; }
    cand.stage[c] = 3;
    todo_fwd[atomicAdd(&ctr->todo_fwd, 1)] = c; atomicAdd(&ctr->res_fwd, (unsigned long long)cand.len[c]);
  }
}

// Forward P-value: F3, or F4 in the frameshift pipeline (p7_pipeline.c:1735-1738, 1779-1785)
__global__ void final_kernel(Cand cand, Counters *__restrict__ ctr, Params p, const int32_t *__restrict__ todo_fwd) {
  const int ntodo = ctr->todo_fwd;
  for (int job = blockIdx.x * blockDim.x + threadIdx.x; job < ntodo; job += gridDim.x * blockDim.x) {
    const int c = todo_fwd[job];
    const float seqsc = (float)((double)(cand.fwdsc[c] - cand.filtersc[c]) / kLog2);
    const double P = d_exp_surv(seqsc, p.evparam[4], p.evparam[5]);
    cand.P[c] = P;
    if (P > (p.fs_pipe ? p.F4 : p.F3)) continue;
    cand.stage[c] = 4;
    atomicAdd(&ctr->n_past_fwd, 1ull);
    if (!p.fs_pipe) atomicAdd(&ctr->pos_past_fwd, (unsigned long long)cand.len[c] * 3ull);
  }
}

// wave-per-candidate kernels take their work list length from device memory
__global__ void copy_count_kernel(const int *src, int64_t *dst) { *dst = *src; }

}  // namespace bath


// =================================================================================================
// host orchestration
// =================================================================================================
namespace bath {

// host twin of ssv_classify (ssvfilter.c:876-925), used only to build the emission threshold table
static int ssv_classify_host(int v, int tjb, const MsvConsts &c, float *sc) {
  if (tjb + c.tbm + c.tec + c.bias >= 127) return BATH_ENORESULT;
  unsigned xE = (v >= -1 - c.bias) ? 255u : (unsigned)(v + 256);
  if ((int)xE >= 255 - c.bias) { *sc = INFINITY; return (c.base - tjb - c.tbm < 128) ? BATH_ENORESULT : BATH_ERANGE; }
  xE = (xE + (unsigned)(c.base - tjb - c.tbm) - 128u) & 0xffffu;
  if ((int)xE >= 255 - c.bias) { *sc = INFINITY; return BATH_ERANGE; }
  unsigned xJ = (xE - (unsigned)c.tec) & 0xffffu;
  if ((int)xJ > c.base) return BATH_ENORESULT;
  float r = ((float)((int)xJ - tjb) - (float)c.base);
  r /= c.scale_b;
  r = (float)((double)r - 3.0);
  *sc = r;
  return BATH_OK;
}

// E[len] = smallest raw SSV maximum for which an ORF of <len> residues must be kept: either p7_SSVFilter
// would not return eslOK (overflow / J state possible -> needs the full MSV) or its score passes F1
// (p7_pipeline.c:1650-1652, same float/double operation order).
static void build_emit_table(const bath_hip_oprofile *om, double F1, int maxlen, std::vector<int16_t> &tab) {
  const MsvConsts mc = msv_consts(om);
  tab.assign((size_t)maxlen + 1, (int16_t)32767);
  for (int len = 0; len <= maxlen; len++) {
    const int tjb = om->lt.h_tjb[len];
    const float nullsc = om->lt.h_nullsc[len];
    auto keep = [&](int v) {
      float usc = 0.f;
      if (ssv_classify_host(v, tjb, mc, &usc) != BATH_OK) return true;
      const float seqsc = (float)((double)(usc - nullsc) / kLog2);
      return !(gumbel_surv(seqsc, om->evparam[0], om->evparam[1]) > F1);
    };
    // keep(v) is monotone in v (the score grows with v, and beyond the representable range the filter reports
    // overflow / "J state possible", which is kept too): smallest v with keep(v) by bisection over -128..127
    if (!keep(127)) continue;
    int lo = -128, hi = 127;
    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (keep(mid)) hi = mid; else lo = mid + 1; }
    tab[(size_t)len] = (int16_t)lo;
  }
}

// The table depends on the model, F1 and the longest ORF only: kept with the profile (host and device) between calls.
static int ensure_emit_table(bath_hip_ctx *ctx, const bath_hip_oprofile *om, double F1, int maxlen) {
  std::lock_guard<std::mutex> lock(om->grow_mu);
  if (om->emit_F1 == F1 && om->emit_maxlen >= maxlen && om->d_emit) return BATH_OK;
  std::vector<int16_t> tab;
  build_emit_table(om, F1, maxlen, tab);
  if (om->d_emit) om->retired.push_back(om->d_emit);            // freed with the profile (bath_hip_oprofile_destroy)
  om->d_emit = nullptr;
  BATH_HIP_TRY(ctx, hipMalloc((void **)&om->d_emit, tab.size() * sizeof(int16_t) + 64));
  BATH_HIP_TRY(ctx, hipMemcpy(om->d_emit, tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  om->emit_F1 = F1; om->emit_maxlen = maxlen;
  return BATH_OK;
}

static int wave_grid_blocks(bath_hip_ctx *ctx) { return ctx->prop.multiProcessorCount * 8; }

struct PipelineWork {
  // device allocations live in ctx->scratch[8..]; this struct only carves them up
  Cand cand;
  int cand_cap = 0;
  uint8_t *pool = nullptr;             // the amino-acid streams of bath_orfs.hip (ctx->scratch[24])
  int32_t *todo_msv = nullptr, *todo_vit = nullptr, *todo_ssvb = nullptr, *todo_vit2 = nullptr, *todo_fwd = nullptr, *todo_sorted = nullptr;
  int *len_bins = nullptr;
  WindowRec *wins = nullptr;
  int win_cap = 0;
  Counters *ctr = nullptr;
};

template <class T>
static T *carve(char *&p, size_t n) {
  T *r = reinterpret_cast<T *>(p);
  p += (n * sizeof(T) + 255) / 256 * 256;
  return r;
}

static size_t layout(PipelineWork &w, char *base, int cap, int win_cap) {
  char *p = base;
  const size_t n = (size_t)cap;
  w.cand_cap = cap; w.win_cap = std::max(2 * cap, win_cap);
  if (win_cap == 0) {                                     // tests: start with a window buffer that is too small (BATH_HIP_TEST_WINCAP)
    static const int forced = [] { const char *e = std::getenv("BATH_HIP_TEST_WINCAP"); return e ? std::atoi(e) : 0; }();
    if (forced > 0) w.win_cap = forced;
  }
  w.ctr = carve<Counters>(p, 1);
  w.cand.window = carve<int64_t>(p, n); w.cand.off = carve<int64_t>(p, n); w.cand.P = carve<double>(p, n);
  w.cand.sf = carve<int32_t>(p, n); w.cand.startj = carve<int32_t>(p, n); w.cand.len = carve<int32_t>(p, n);
  w.cand.msv_status = carve<int32_t>(p, n); w.cand.vit_status = carve<int32_t>(p, n); w.cand.fwd_status = carve<int32_t>(p, n); w.cand.stage = carve<int32_t>(p, n);
  w.cand.flags = carve<int32_t>(p, n); w.cand.kminmax = carve<int32_t>(p, 2 * n);
  w.cand.usc = carve<float>(p, n); w.cand.nullsc = carve<float>(p, n); w.cand.filtersc = carve<float>(p, n);
  w.cand.vfsc = carve<float>(p, n); w.cand.fwdsc = carve<float>(p, n);
  w.cand.v = carve<int16_t>(p, n);
  w.todo_msv = carve<int32_t>(p, n); w.todo_vit = carve<int32_t>(p, n); w.todo_ssvb = carve<int32_t>(p, n);
  w.todo_vit2 = carve<int32_t>(p, n); w.todo_fwd = carve<int32_t>(p, n); w.todo_sorted = carve<int32_t>(p, n);
  w.len_bins = carve<int>(p, 2048);
  w.wins = carve<WindowRec>(p, (size_t)w.win_cap);
  return (size_t)(p - base);
}

}  // namespace bath

extern "C" void bath_pipeline_params_default(bath_pipeline_params *p, int fs_pipe) {   // p7_pipeline.c:219-222, bathsearch.c:104
  p->F1 = 0.02; p->F2 = 1e-3; p->F3 = 1e-5; p->F4 = 5e-4;
  p->do_biasfilter = 1; p->fs_pipe = fs_pipe; p->min_orf_len = 20; p->ncbi_table = 1;
  p->nres_before = 0;
  p->do_null2 = 1; p->std_pipe = 1; p->strands = BATH_STRAND_BOTH; p->initiator = BATH_INIT_ANY;   // p7_pipeline.c:107, :199; bathsearch.c:97, :718-719
  p->inc_by_E = 1; p->seed = 42; p->T = 0.0;                                                        // p7_pipeline.c:98, :148, :165
}

extern "C" int bath_hip_pipeline_timings(const bath_hip_ctx *ctx, int max, const char **names, float *ms, int64_t *launches) {
  int n = std::min<int>(max, (int)ctx->timings.size());
  for (int i = 0; i < n; i++) { names[i] = ctx->timings[i].name; ms[i] = ctx->timings[i].ms; launches[i] = ctx->timings[i].launches; }
  return n;
}

namespace bath {
struct FilterState {               // what the frameshift stage reads after the cascade (device memory stays in ctx->scratch)
  PipelineWork W;
  Counters hc{};                     // zero when the block is empty and the cascade returns before running
  const uint8_t *d_ssvsc = nullptr;
  const float *d_bgf = nullptr;
  OrfTablesDev tt{};
};
}  // namespace bath

// Where the Forward parser of the cascade leaves the special-state rows of its candidates (ctx->keep_fwd_rows): offsets in work-list
// order, (len + 1) x 6 floats each, by one wave (a chunk of 64 candidates per step, prefix sums by shuffle); a candidate whose rows
// would not fit <cap> floats gets -1 and no rows.
__global__ __launch_bounds__(64) void fwd_keep_offsets_kernel(const int32_t *__restrict__ todo, const int *__restrict__ ntodo_dev, const int32_t *__restrict__ len,
                                                              int64_t *__restrict__ off, int64_t cap) {
  const int lane = threadIdx.x, n = *ntodo_dev;
  int64_t base = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const int sid = j < n ? todo[j] : -1;
    const int64_t v = sid >= 0 ? ((int64_t)len[sid] + 1) * 6 : 0;
    int64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const int64_t u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
    if (sid >= 0) off[sid] = (base + incl <= cap) ? base + incl - v : (int64_t)-1;
    base += __shfl(incl, 63, 64);
  }
}

static int run_filters(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                       const bath_pipeline_params *prm, bath_pipeline_stats *stats,
                       const bath_orf_result **results, int64_t *n_results, FilterState *state) {
  if (!ctx || !om || !dna || !prm) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_results) *n_results = 0;
  if (results) *results = nullptr;
  if (stats) std::memset(stats, 0, sizeof *stats);
  const int64_t nwin = dna->n;
  if (nwin == 0) return BATH_OK;
  const int M = om->M;
  const int max_orf = dna->maxlen / 3 + 1;
  int st = om->ensure_len_tables(max_orf);
  if (st != BATH_OK) return st;

  // ---- per-call tables
  OrfTablesDev tt{};
  if (prm->strands < 0 || prm->strands > 2 || prm->initiator < 0 || prm->initiator > 2) { ctx->set_error("bath_pipeline_params: strands / initiator out of range"); return BATH_EINVAL; }
  if ((st = orf_tables_upload(ctx, prm->ncbi_table, &tt, prm->initiator)) != BATH_OK) return st;
  if ((st = ensure_emit_table(ctx, om, prm->F1, max_orf)) != BATH_OK) return st;
  DevBuf &b_tabs = ctx->scratch[8], &b_work = ctx->scratch[9];
  const size_t ssv_bytes = (size_t)(M + 1) * kKp;
  const size_t tabs_bytes = 8192 + ssv_bytes + 256 + 20 * 4 + 256;
  if (tabs_bytes > b_tabs.cap) ctx->tabs_uid = 0;                 // the buffer is about to be replaced: whatever it held is gone
  BATH_HIP_TRY(ctx, b_tabs.reserve(tabs_bytes));
  char *tp = b_tabs.as<char>();
  const int16_t *d_emit = om->d_emit;
  uint8_t *d_ssvsc = reinterpret_cast<uint8_t *>(tp); tp += (ssv_bytes + 255) / 256 * 256;
  float *d_bgf = reinterpret_cast<float *>(tp);
  if (ctx->tabs_uid != om->uid || ctx->tabs_ptr != b_tabs.p) {   // once per (context, profile): a database pass of small queries repeats this call per query
    std::vector<uint8_t> ssv_scores(ssv_bytes, 0);
    bath_hip_oprofile_get_ssv_scores(om, ssv_scores.data());
    BATH_HIP_TRY(ctx, hipMemcpyAsync(d_ssvsc, ssv_scores.data(), ssv_scores.size(), hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(d_bgf, kAminoBg, 20 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host vector goes out of scope here
    ctx->tabs_uid = om->uid; ctx->tabs_ptr = b_tabs.p;
  }

  Params P{};
  P.F1 = prm->F1; P.F2 = prm->F2; P.F3 = prm->F3; P.F4 = prm->F4;
  P.do_bias = prm->do_biasfilter; P.fs_pipe = prm->fs_pipe; P.minlen = prm->min_orf_len;
  std::memcpy(P.evparam, om->evparam, sizeof P.evparam);
  const MsvConsts mc = msv_consts(om);
  const double invP_f1 = gumbel_invsurv(prm->F1, om->evparam[0], om->evparam[1]);   // p7_SSVFilter_BATH's threshold, msvfilter.c:302
  VitWindowArgs wa{};
  wa.invP_vit = (double)(float)gumbel_invsurv(prm->F2, om->evparam[2], om->evparam[3]);   // vitfilter.c:314 (float invP)
  wa.invP_msv = (double)(float)gumbel_invsurv(prm->F2, om->evparam[0], om->evparam[1]);   // vitfilter.c:319

  // ---- capacity guess: all ORFs that could exist, scaled by the expected MSV pass rate (+ slack); retried on overflow
  if (dna->cache_minlen != prm->min_orf_len) {
    int64_t nres_c = 0, max_orfs_c = 0;
    for (int64_t i = 0; i < nwin; i++) {
      const int n = dna->h_len[i];
      if (n < 15) continue;
      nres_c += 2 * (int64_t)(n - (dna->h_context.empty() ? 0 : dna->h_context[(size_t)i]));    // dnaSeq->W per strand
      max_orfs_c += 6 * (int64_t)((n / 3 + 1) / (prm->min_orf_len + 1) + 1);
    }
    dna->cache_minlen = prm->min_orf_len; dna->cache_nres = nres_c; dna->cache_max_orfs = max_orfs_c;
  }
  const int64_t nres = prm->strands == BATH_STRAND_BOTH ? dna->cache_nres : dna->cache_nres / 2;   // W once per strand searched (bathsearch.c:1071, :1084)
  const int64_t max_orfs = dna->cache_max_orfs;
  if (max_orfs >= (int64_t)INT32_MAX) { ctx->set_error("DNA block too large for one pipeline call (ORF list indices are 32-bit): split it"); return BATH_EINVAL; }
  int64_t cap = std::max<int64_t>(4096, (int64_t)((double)max_orfs * std::min(1.0, prm->F1 * 4.0 + 0.01)));
  cap = std::min<int64_t>(cap, std::max<int64_t>(max_orfs, 4096));

  // ---- translation / ORF work-list buffers (bath_orfs.hip)
  if ((st = orf_tiles_ensure(ctx, dna)) != BATH_OK) return st;
  const size_t nent = (size_t)dna->ntiles * 6;
  DevBuf &b_aa = ctx->scratch[24], &b_slots = ctx->scratch[25], &b_orfs = ctx->scratch[26], &b_misc = ctx->scratch[27];
  BATH_HIP_TRY(ctx, b_aa.reserve(orf_aa_bytes(dna)));
  BATH_HIP_TRY(ctx, b_slots.reserve(((size_t)dna->ntiles * (size_t)orf_slot_cap(prm->min_orf_len) + 64) * 8));
  BATH_HIP_TRY(ctx, b_orfs.reserve((size_t)(max_orfs + 64) * sizeof(OrfRec)));
  BATH_HIP_TRY(ctx, b_misc.reserve((5 * nent + 2 * kOrfBins + 64) * sizeof(int32_t)));
  OrfBuffers ob{};
  orf_buffers_carve(&ob, b_aa.p, b_slots.p, b_orfs.p, b_misc.p, nent);

  const int NRk = om->NR;
  const size_t ssv_shmem = (size_t)kSsvRows * om->ssv_row_bytes;
  if (ssv_shmem > 160 * 1024) { ctx->set_error("model too long for the LDS-resident SSV cost table"); return BATH_EINVAL; }
  const int dec_blocks = ctx->prop.multiProcessorCount * 4;

  std::vector<hipEvent_t> &ev = ctx->ev_pool;
  while (ev.size() < 12) { hipEvent_t e; BATH_HIP_TRY(ctx, hipEventCreate(&e)); ev.push_back(e); }

  PipelineWork W;
  Counters hc{};
  int win_need = 0;                                         // hit windows seen by a pass that ran out of room for them
  for (int attempt = 0;; attempt++) {
    size_t need = layout(W, nullptr, (int)cap, win_need);
    BATH_HIP_TRY(ctx, b_work.reserve(need + 4096));
    layout(W, b_work.as<char>(), (int)cap, win_need);
    W.pool = b_aa.as<uint8_t>();
    BATH_HIP_TRY(ctx, hipMemsetAsync(W.ctr, 0, sizeof(Counters), ctx->stream));
    SeqView cv{W.pool, W.cand.off, W.cand.len, cap};

    int e = 0;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 1. six-frame translation, ORFs, length-sorted work list
    if ((st = launch_orf_scan(ctx, dna, tt, prm->min_orf_len, ob, &W.ctr->n_orfs, &W.ctr->orf_res, prm->strands)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 2. SSV + F1 threshold, lane per ORF
    {
      static const int ssv_chunk = [] { const char *e = std::getenv("BATH_HIP_SSV_CHUNK"); return e ? std::atoi(e) : 0; }();
      int blocks = ctx->prop.multiProcessorCount * 4;
      const int ssv_threads = 256;
      const int wpb = ssv_threads / 64;
      if (ssv_chunk > 0) blocks = (int)std::min<int64_t>((max_orfs + (int64_t)wpb * (64 / om->G) * ssv_chunk - 1) / ((int64_t)wpb * (64 / om->G) * ssv_chunk), 1 << 30);
      bool launched = false;
#define BATH_ORF_CASE(N, GG)                                                                                                     \
  if (!launched && NRk == N && om->G == GG) {                                                                                    \
    if (ssv_shmem > 64 * 1024) (void)bath::allow_max_lds((const void *)ssv_orf_kernel<N, GG>); \
    hipLaunchKernelGGL((ssv_orf_kernel<N, GG>), dim3(blocks), dim3(ssv_threads), ssv_shmem, ctx->stream, W.pool, ob.sorted, ob.ntotal,                  \
                       dna->view(), om->d_ssv, om->ssv_row_bytes, d_emit, max_orf, W.cand, W.cand_cap, W.ctr, ssv_chunk);        \
    launched = true;                                                                                                             \
  }
      BATH_SSV_SHAPES(BATH_ORF_CASE)
#undef BATH_ORF_CASE
      if (!launched) { ctx->set_error("SSV kernel: no tile shape for this model length"); return BATH_EINVAL; }
      BATH_HIP_TRY(ctx, hipGetLastError());
    }
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // everything after SSV works on a few 10^5 survivors in chains of latency-bound kernels: on a stream of its own with the
    // highest priority, so that (with a non-persistent SSV) it runs while another part's translation + SSV fill the chip
    hipStream_t bulk_stream = ctx->stream;
    struct StreamRestore { bath_hip_ctx *c; hipStream_t s; ~StreamRestore() { c->stream = s; } } stream_restore{ctx, bulk_stream};
    if (ctx->tail_stream) {
      BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->tail_stream, ev[e - 1], 0));
      ctx->stream = ctx->tail_stream;
    }
    // The lane-per-ORF kernels pay off when the candidates fill the chip's lanes (65 k of them): a block of 10^9 nt leaves 4 x 10^5
    // Viterbi candidates, a 25 Mb query of configs[3] 5 x 10^3 -- 80 waves that each last as long as their longest ORF (0.7 ms whatever
    // their number) where the wave-per-ORF kernel takes 0.2 ms.  The candidate counts live on the device; the block's size is the
    // host's proxy.  tools/lane_crossover.py, M = 145, windows of 1 kb: Viterbi 0.72 (lane) / 0.19 (wave) ms at 12.5 k windows,
    // 0.71 / 0.63 at 100 k, 0.75 / 1.11 at 200 k; MSV 0.19 / 0.08, 0.22 / 0.10, 0.23 / 0.13, and 1.08 / 1.41 at 400 k.
    static const int64_t lane_min_nt = [] { const char *e = std::getenv("BATH_HIP_LANE_MIN_NT"); return e ? std::atoll(e) : (int64_t)150'000'000; }();
    const bool few_cands = dna->total < lane_min_nt, few_msv = dna->total < 2 * lane_min_nt;
    // 3. SSV status; full MSV for the undecided
    hipLaunchKernelGGL(classify_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, om->lt.d_tjb, mc, W.todo_msv);
    BATH_HIP_TRY(ctx, hipGetLastError());
    // (the lane-per-ORF kernels pay off when the candidates fill the chip's lanes: see few_cands at the Viterbi stage below)
    if ((st = launch_msv_wave(ctx, om, cv, W.todo_msv, cap, W.cand.usc, W.cand.msv_status, &W.ctr->todo_msv, !few_msv)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 3. F1 on the MSV score, bias filter
    hipLaunchKernelGGL(f1_bias_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, P, W.pool, M, om->d_bias_eo,
                       om->lt.d_nullsc, om->lt.d_p1, om->lt.d_lt1, om->lt.d_lt2, W.todo_vit, W.todo_ssvb);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 4. Viterbi filter with windows (P > F2) / SSV windows (P <= F2)
    wa.d_filtersc = W.cand.filtersc; wa.d_ssv_scores = d_ssvsc; wa.d_wins = W.wins; wa.d_win_count = &W.ctr->win_count; wa.win_cap = W.win_cap;
    wa.d_kminmax = W.cand.kminmax;
    // p7_SSVFilter_BATH's windows for the candidates with P <= F2: other candidates than the Viterbi kernels', so it runs beside them
    auto launch_ssvb = [&](hipStream_t stream) -> int {
      const int Cc = (M + 63) / 64;
      int Cs = -1;
      for (int opt : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 52}) if (Cc <= opt) { Cs = opt; break; }
#define BATH_SSVB_CASE(N)                                                                                                          \
  case N:                                                                                                                          \
    hipLaunchKernelGGL(ssv_bath_kernel<N>, dim3(wave_grid_blocks(ctx) / 4), dim3(256), 0, stream, W.cand, W.ctr, W.todo_ssvb, W.pool, M,      \
                       om->d_rb, om->rb_stride, d_ssvsc, om->lt.d_tjb, om->lt.d_nullsc, mc, invP_f1, W.wins, W.win_cap, W.ctr);    \
    break;
      switch (Cs) {
        BATH_SSVB_CASE(1) BATH_SSVB_CASE(2) BATH_SSVB_CASE(3) BATH_SSVB_CASE(4) BATH_SSVB_CASE(6) BATH_SSVB_CASE(8)
        BATH_SSVB_CASE(12) BATH_SSVB_CASE(16) BATH_SSVB_CASE(24) BATH_SSVB_CASE(32) BATH_SSVB_CASE(52)
        default: ctx->set_error("model too long for the SSV window kernel"); return BATH_EINVAL;
      }
#undef BATH_SSVB_CASE
      return BATH_OK;
    };
    bool ssvb_done = false;
    if (vit_lane_supported(om) && !few_cands) {       // lane per ORF, ORFs bucketed by length (bath_viterbi.hip)
      if ((st = launch_len_sort(ctx, W.todo_vit, &W.ctr->todo_vit, W.cand.len, W.len_bins, W.todo_sorted)) != BATH_OK) return st;
      // The lane kernel runs one wave per SIMD and a wave takes as long as its longest ORF (3.6 us per residue): the few
      // long ORFs at the head of the sorted list would set the duration of the whole stage.  They go to the
      // wave-per-ORF kernel on a side stream instead, concurrently with the lane kernel on the rest.
      if (!ctx->side_stream) {
        BATH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
      }
      static const int long_orf = [] { const char *e = std::getenv("BATH_HIP_VIT_LONG"); return e ? std::atoi(e) : kVitLongOrf; }();
      const int *d_nlong = len_sort_count_longer(W.len_bins, long_orf);
      BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
      {
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->side_stream;
        st = launch_vit_wave(ctx, om, cv, W.todo_sorted, cap, W.cand.vfsc, W.cand.vit_status, &wa, d_nlong);
        ctx->stream = main_stream;
        if (st != BATH_OK) return st;
        if ((st = launch_ssvb(ctx->side_stream)) != BATH_OK) return st;      // ... and the SSV windows, all under the lane kernel
        ssvb_done = true;
      }
      BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
      if ((st = launch_vit_lane(ctx, om, cv, W.todo_sorted, cap, &W.ctr->todo_vit, W.cand.vfsc, W.cand.vit_status, &wa, d_nlong)) != BATH_OK) return st;
      BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    } else if ((st = launch_vit_wave(ctx, om, cv, W.todo_vit, cap, W.cand.vfsc, W.cand.vit_status, &wa, &W.ctr->todo_vit)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    if (!ssvb_done && (st = launch_ssvb(ctx->stream)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 5. F2, local composition re-filter, optional plain Viterbi re-run
    hipLaunchKernelGGL(post_vit_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, P, W.todo_msv /* free by now */, W.todo_fwd);
    if (P.do_bias) {
      DevBuf &b_compo = ctx->scratch[36];                                       // 20 x 256 terms, then 20 sums per candidate
      BATH_HIP_TRY(ctx, b_compo.reserve((size_t)20 * 256 * 4 + (size_t)W.cand_cap * 20 * 4 + 256));
      float *d_terms = b_compo.as<float>(), *d_compo = d_terms + 20 * 256;
      hipLaunchKernelGGL(compo_terms_kernel, dim3(20), dim3(256), 0, ctx->stream, d_bgf, (int)om->base_b, om->scale_b, d_terms);
      hipLaunchKernelGGL(local_compo_kernel, dim3(16384), dim3(64), 0, ctx->stream, W.cand, W.ctr, M, d_ssvsc, d_terms, W.todo_msv, d_compo);
      hipLaunchKernelGGL(post_vit_local_kernel, dim3(std::max(1, dec_blocks)), dim3(64), 0, ctx->stream, W.cand, W.ctr, P, W.pool, M, d_bgf, d_compo,
                         om->lt.d_p1, om->lt.d_lt1, om->lt.d_lt2, W.todo_msv, W.todo_vit2, W.todo_fwd);
    }
    BATH_HIP_TRY(ctx, hipGetLastError());
    if ((st = launch_vit_wave(ctx, om, cv, W.todo_vit2, cap, W.cand.vfsc, W.cand.vit_status, nullptr, &W.ctr->todo_vit2)) != BATH_OK) return st;
    hipLaunchKernelGGL(post_vit2_kernel, dim3(64), dim3(256), 0, ctx->stream, W.cand, W.ctr, P, W.todo_vit2, W.todo_fwd);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 6. Forward parser, F3/F4 (for the domain stage of a one-lane block it also leaves its special-state rows: ctx->keep_fwd_rows)
    ctx->fwd_rows_kept = nullptr; ctx->fwd_rows_off = nullptr;
    float *d_keep = nullptr; int64_t *d_keep_off = nullptr;
    if (ctx->keep_fwd_rows) {
      constexpr int64_t kKeepFloats = (int64_t)8 << 20;                        // 32 MB: ~1.4 M residue rows; what does not fit is computed again by the domain stage
      DevBuf &b_rows = ctx->scratch[38], &b_roff = ctx->scratch[39];
      BATH_HIP_TRY(ctx, b_rows.reserve((size_t)kKeepFloats * sizeof(float) + 64)); BATH_HIP_TRY(ctx, b_roff.reserve((size_t)cap * sizeof(int64_t) + 64));
      d_keep = b_rows.as<float>(); d_keep_off = b_roff.as<int64_t>();
      hipLaunchKernelGGL(fwd_keep_offsets_kernel, dim3(1), dim3(64), 0, ctx->stream, W.todo_fwd, &W.ctr->todo_fwd, W.cand.len, d_keep_off, kKeepFloats);
      BATH_HIP_TRY(ctx, hipGetLastError());
    }
    if ((st = launch_fwd_wave(ctx, om, cv, W.todo_fwd, cap, W.cand.fwdsc, W.cand.fwd_status, &W.ctr->todo_fwd, d_keep, d_keep_off)) != BATH_OK) return st;
    if (d_keep) { ctx->fwd_rows_kept = d_keep; ctx->fwd_rows_off = d_keep_off; }
    hipLaunchKernelGGL(final_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.ctr, P, W.todo_fwd);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));

    BATH_HIP_TRY(ctx, hipMemcpyAsync(&hc, W.ctr, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // the reference's window list grows without limit (p7_hmmwindow.c:83); here the kernels count every window they find and
    // drop those beyond the buffer, so a pass that found more windows than fit is repeated with room for all of them
    if (hc.overflow || hc.cand_count > cap || hc.win_count > W.win_cap) {
      if (attempt >= 4) { ctx->set_error("candidate / window buffers overflowed repeatedly"); return BATH_EMEM; }
      if (hc.overflow || hc.cand_count > cap) cap = std::max<int64_t>(cap * 2, (int64_t)hc.cand_count + 1024);
      if (hc.win_count > W.win_cap) win_need = hc.win_count + hc.win_count / 4 + 1024;
      continue;
    }
    static const char *names[] = {"translate_orfs", "ssv_f1", "classify_msv", "f1_bias", "viterbi_windows", "ssv_windows", "post_vit", "forward_final"};
    static const int64_t launches[] = {4, 1, 2, 1, 4, 1, 3, 2};
    ctx->timings.clear();
    for (int i = 0; i + 1 < e; i++) {
      float ms = 0.f;
      BATH_HIP_TRY(ctx, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      ctx->timings.push_back(StageTiming{names[i], ms, launches[i]});
    }
    break;
  }

  if (stats) {
    stats->nres = nres; stats->n_orfs = (int64_t)hc.n_orfs;
    stats->n_past_msv = (int64_t)hc.n_past_msv; stats->n_past_bias = (int64_t)hc.n_past_bias;
    stats->n_past_vit = (int64_t)hc.n_past_vit; stats->n_past_fwd = (int64_t)hc.n_past_fwd;
    stats->pos_past_msv = (int64_t)hc.pos_past_msv; stats->pos_past_bias = (int64_t)hc.pos_past_bias;
    stats->pos_past_vit = (int64_t)hc.pos_past_vit; stats->pos_past_fwd = (int64_t)hc.pos_past_fwd;
    stats->cells_msv = (int64_t)(hc.orf_res - hc.res_seen) * M; stats->cells_vit = (int64_t)hc.res_vit * M; stats->cells_fwd = (int64_t)hc.res_fwd * M;
  }

  if (results) {
    // one record per ORF past MSV, assembled and ordered on the device (bath_records.hip).  A part of a larger block leaves them
    // there: bath_hip_pipeline_filters copies every part's records straight into their place in one page-locked array.
    bath_orf_result *d_rec = nullptr;
    int64_t nrec = 0;
    // the records' sort key holds the window in 32 bits and the ORF's first codon in 28 (bath_records.hip)
    if (dna->n > (int64_t)UINT32_MAX || dna->maxlen / 3 >= (1 << 28)) { ctx->set_error("block too large for the ORF records' sort key (2^32 windows, 805 Mnt per window)"); return BATH_ERANGE; }
    if ((st = build_orf_records(ctx, W.cand, hc.cand_count, dna->is_part ? dna->first_window : 0, &d_rec, &nrec)) != BATH_OK) return st;
    ctx->d_records = d_rec; ctx->n_records = nrec;
    if (!dna->is_part) {
      BATH_HIP_TRY(ctx, ctx->results_pinned.reserve((size_t)std::max<int64_t>(nrec, 1) * sizeof(bath_orf_result)));
      if (nrec > 0) BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->results_pinned.p, d_rec, (size_t)nrec * sizeof(bath_orf_result), hipMemcpyDeviceToHost, ctx->stream));
      BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      *results = ctx->results_pinned.as<bath_orf_result>();
    } else *results = nullptr;
    if (n_results) *n_results = nrec;
  } else if (n_results) *n_results = (int64_t)hc.n_past_msv;
  if (state) { state->W = W; state->hc = hc; state->d_ssvsc = d_ssvsc; state->d_bgf = d_bgf; state->tt = tt; }
  return BATH_OK;
}

// ---- concurrent lanes --------------------------------------------------------------------------------------------
// The cascade's tail (decision kernels, wave-per-candidate DP over a few 10^5 survivors) is latency bound and leaves
// most of the chip idle, while SSV saturates the VALUs.  A large block is therefore cut into K parts of consecutive
// windows that run the whole cascade concurrently, each on its own HIP stream with its own scratch memory (a "lane"
// context) driven by its own host thread: one part's tail overlaps another part's translation and SSV.
static int pipeline_lane_count(const bath_hip_seqs *dna) {
  const char *e = std::getenv("BATH_HIP_LANES");                  // override for tests and tuning
  const int forced = e ? std::atoi(e) : 0;
  if (forced > 0) return (int)std::min<int64_t>(forced, std::max<int64_t>(dna->n, 1));
  if (dna->is_part || dna->n < 8) return 1;
  // measured on MI355X, 10^6 x 1 kb: 1 lane 17.2 ms, 2 lanes 15.5 ms per pass; 3 lanes 15.4-19.3 ms depending on how the
  // persistent grids of the three parts happen to interleave, 4 lanes 19.7 ms: two lanes is the robust choice
  int K = (dna->total >= ((int64_t)1 << 28)) ? 2 : 1;           // blocks under 256 MB: one launch sequence is short enough
  // ... and as many parts as it takes to keep a part's ORF list within 32-bit indices (run_filters: ~1 ORF per 27 nt on both
  // strands of iid DNA; a part of 16 GB holds ~6e8): a larger block is cut into more parts instead of being refused
  const int64_t by_size = (dna->total + ((int64_t)1 << 34) - 1) >> 34;
  if (by_size > K) K = (int)std::min<int64_t>(by_size, std::max<int64_t>(dna->n, 1));
  return K;
}

static int ensure_parts(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int K) {
  if ((int)dna->parts.size() == K) return BATH_OK;
  for (bath_hip_seqs *p : dna->parts) bath_hip_seqs_destroy(p);
  dna->parts.clear();
  int64_t w0 = 0;
  for (int k = 0; k < K; k++) {
    // cut where the running residue count passes the part's share of the total.  With two parts the shares are unequal: the
    // second part's tail (decision kernels, Viterbi, Forward on its survivors) is the only work nothing else overlaps, so the
    // second part is the smaller one (BATH_HIP_LANE_SPLIT = share of the first part, tools/ab_probe.py)
    static const double split2 = [] { const char *e = std::getenv("BATH_HIP_LANE_SPLIT"); const double v = e ? std::atof(e) : 0.0; return (v > 0.05 && v < 0.95) ? v : 0.5; }();
    int64_t w1 = w0;
    if (k == K - 1) w1 = dna->n;
    else {
      const int64_t target = (K == 2) ? (int64_t)((double)dna->total_aligned * split2) : dna->total_aligned / K * (k + 1);
      w1 = std::lower_bound(dna->h_off.begin() + w0, dna->h_off.end(), target) - dna->h_off.begin();
      w1 = std::max(w1, std::min(w0 + 1, dna->n));
    }
    bath_hip_seqs *p = new bath_hip_seqs();
    p->ctx = ctx; p->is_part = true; p->first_window = w0; p->n = w1 - w0;
    const int64_t base = w0 < dna->n ? dna->h_off[(size_t)w0] : dna->total_aligned;
    p->h_off.resize((size_t)p->n); p->h_len.assign(dna->h_len.begin() + w0, dna->h_len.begin() + w1);
    for (int64_t i = 0; i < p->n; i++) {
      p->h_off[(size_t)i] = dna->h_off[(size_t)(w0 + i)] - base;
      p->maxlen = std::max(p->maxlen, p->h_len[(size_t)i]);
      p->total += p->h_len[(size_t)i];
    }
    p->total_aligned = (w1 < dna->n ? dna->h_off[(size_t)w1] : dna->total_aligned) - base;
    p->d_data = dna->d_data + base;
    p->d_len = dna->d_len + w0;
    if (dna->d_context) { p->d_context = dna->d_context + w0; p->h_context.assign(dna->h_context.begin() + w0, dna->h_context.begin() + w1); }
    dna->parts.push_back(p);
    if (hipMalloc((void **)&p->d_off, (size_t)std::max<int64_t>(p->n, 1) * sizeof(int64_t)) != hipSuccess) { ctx->set_error("hipMalloc (block parts)"); return BATH_EMEM; }
    if (p->n > 0 && hipMemcpy(p->d_off, p->h_off.data(), (size_t)p->n * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) { ctx->set_error("hipMemcpy (block parts)"); return BATH_EFAIL; }
    w0 = w1;
  }
  return BATH_OK;
}

// The cascade of a block, as K concurrent parts when the block is large.  <after>(k, lane, part, S) runs on the lane's own host
// thread right after that part's cascade, with the lane's device state still in place (S: its candidate arrays, windows, pool).
template <class After>
static int run_filters_lanes(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                             const bath_pipeline_params *prm, bath_pipeline_stats *stats,
                             const bath_orf_result **results, int64_t *n_results, std::vector<FilterState> *states, After after) {
  if (!ctx || !om || !dna || !prm) return BATH_EINVAL;
  const int K = pipeline_lane_count(dna);
  if (K <= 1) {
    states->assign(1, FilterState{});
    int st = run_filters(ctx, om, dna, prm, stats, results, n_results, &(*states)[0]);
    if (st != BATH_OK) return st;
    return after(0, ctx, dna, (*states)[0]);
  }
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  int st = om->ensure_len_tables(dna->maxlen / 3 + 1);            // the mutable state of the profile: fill it before the threads start
  if (st != BATH_OK) return st;
  if ((st = ensure_emit_table(ctx, om, prm->F1, dna->maxlen / 3 + 1)) != BATH_OK) return st;
  while ((int)ctx->lanes.size() < K) {
    bath_hip_ctx *lane = nullptr;
    if ((st = bath_hip_init(ctx->device, &lane)) != BATH_OK) { ctx->set_error("cannot create a pipeline lane"); return st; }
    mark_internal(lane);
    // Descending stream priorities: when two parts have work ready, the earlier part's workgroups are dispatched first.  Left to
    // the hardware queues the interleaving of the parts is a race and about every third process lands on a schedule that is
    // 20% slower (13.4 vs 16 ms per step on the bench block); with priorities it is 13.2-13.9 ms every time (tools/ab_probe.py).
    // BATH_HIP_LANE_PRIO=0 turns it off.
    const char *pe = std::getenv("BATH_HIP_LANE_PRIO");
    if (!(pe && pe[0] == '0')) {
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
        (void)hipStreamDestroy(lane->stream);
        const char *te = std::getenv("BATH_HIP_TAIL_PRIO");
        const bool tails = te && te[0] == '1';
        // hi is numerically the smallest value = highest priority.  With tail streams: tails highest, the first part's bulk work in the middle
        const int prio = tails ? (ctx->lanes.empty() ? std::min(lo, hi + 1) : lo) : (ctx->lanes.empty() ? hi : lo);
        if (hipStreamCreateWithPriority(&lane->stream, hipStreamNonBlocking, prio) != hipSuccess) { ctx->set_error("hipStreamCreateWithPriority"); return BATH_EFAIL; }
        if (tails && hipStreamCreateWithPriority(&lane->tail_stream, hipStreamNonBlocking, hi) != hipSuccess) { ctx->set_error("hipStreamCreateWithPriority"); return BATH_EFAIL; }
      }
    }
    ctx->lanes.push_back(lane);
  }
  for (bath_hip_ctx *lane : ctx->lanes) lane->fs_strict = ctx->fs_strict;
  if ((st = ensure_parts(ctx, dna, K)) != BATH_OK) return st;
  std::vector<bath_pipeline_stats> pst((size_t)K);
  std::vector<const bath_orf_result *> pres((size_t)K, nullptr);
  std::vector<int64_t> pn((size_t)K, 0);
  std::vector<int> rc((size_t)K, BATH_OK);
  states->assign((size_t)K, FilterState{});
  // The lanes' streams are non-blocking streams of their own: nothing orders them after the context's stream, which may still
  // hold the block's upload and expansion kernels (bath_hip_seqs_upload_packed / _upload_wait) or the part offsets copied above.
  if (!ctx->ev_lanes) BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_lanes, hipEventDisableTiming));
  BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_lanes, ctx->stream));
  for (int k = 0; k < K; k++) BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->lanes[(size_t)k]->stream, ctx->ev_lanes, 0));
  if (!ctx->lane_pool) ctx->lane_pool = new LanePool();            // the lanes' host threads live as long as the context
  ctx->lane_pool->run(K, [&](int k) {
    bath_hip_ctx *lane = ctx->lanes[(size_t)k];
    rc[(size_t)k] = run_filters(lane, om, dna->parts[(size_t)k], prm, &pst[(size_t)k], results ? &pres[(size_t)k] : nullptr, &pn[(size_t)k], &(*states)[(size_t)k]);
    if (rc[(size_t)k] == BATH_OK) rc[(size_t)k] = after(k, lane, dna->parts[(size_t)k], (*states)[(size_t)k]);
  });
  for (int k = 0; k < K; k++)
    if (rc[(size_t)k] != BATH_OK) { ctx->set_error(ctx->lanes[(size_t)k]->err); return rc[(size_t)k]; }

  bath_pipeline_stats tot{};
  int64_t ntot = 0;
  for (int k = 0; k < K; k++) {
    const bath_pipeline_stats &a = pst[(size_t)k];
    tot.nres += a.nres; tot.n_orfs += a.n_orfs; tot.n_past_msv += a.n_past_msv; tot.n_past_bias += a.n_past_bias; tot.n_past_vit += a.n_past_vit;
    tot.n_past_fwd += a.n_past_fwd; tot.pos_past_msv += a.pos_past_msv; tot.pos_past_bias += a.pos_past_bias; tot.pos_past_vit += a.pos_past_vit;
    tot.pos_past_fwd += a.pos_past_fwd; tot.cells_msv += a.cells_msv; tot.cells_vit += a.cells_vit; tot.cells_fwd += a.cells_fwd;
    ntot += pn[(size_t)k];
  }
  if (stats) *stats = tot;
  if (results) {                                                   // parts are consecutive windows; each part's records are ordered, on its device
    BATH_HIP_TRY(ctx, ctx->results_pinned.reserve((size_t)std::max<int64_t>(ntot, 1) * sizeof(bath_orf_result)));
    bath_orf_result *dst = ctx->results_pinned.as<bath_orf_result>();
    int64_t at = 0;
    for (int k = 0; k < K; k++) {
      bath_hip_ctx *lane = ctx->lanes[(size_t)k];
      const int64_t nk = pn[(size_t)k];
      if (nk > 0) BATH_HIP_TRY(ctx, hipMemcpyAsync(dst + at, lane->d_records, (size_t)nk * sizeof(bath_orf_result), hipMemcpyDeviceToHost, lane->stream));
      at += nk;
    }
    for (int k = 0; k < K; k++) BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->lanes[(size_t)k]->stream));
    *results = dst;
  }
  if (n_results) *n_results = ntot;
  // stage timings: device time summed over the lanes (they overlap in wall-clock time)
  ctx->timings.clear();
  for (int k = 0; k < K; k++) {
    const std::vector<StageTiming> &t = ctx->lanes[(size_t)k]->timings;
    for (size_t i = 0; i < t.size(); i++) {
      if (k == 0) ctx->timings.push_back(t[i]);
      else if (i < ctx->timings.size()) { ctx->timings[i].ms += t[i].ms; ctx->timings[i].launches += t[i].launches; }
    }
  }
  return BATH_OK;
}

// experiment: every lane runs its part <reps> times back to back (what a pipelined block loop would look like in steady state)
extern "C" int bath_hip_pipeline_filters_repeat(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                                                const bath_pipeline_params *prm, int reps, bath_pipeline_stats *stats) {
  std::vector<FilterState> states;
  int st = run_filters_lanes(ctx, om, dna, prm, stats, nullptr, nullptr, &states,
                             [&](int, bath_hip_ctx *lane, const bath_hip_seqs *part, const FilterState &) {
                               for (int r = 1; r < reps; r++) {
                                 bath_pipeline_stats s2{};
                                 int64_t n2 = 0;
                                 const int rc = run_filters(lane, om, part, prm, &s2, nullptr, &n2, nullptr);
                                 if (rc != BATH_OK) return rc;
                               }
                               return (int)BATH_OK;
                             });
  return st;
}

extern "C" int bath_hip_pipeline_filters(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                                         const bath_pipeline_params *prm, bath_pipeline_stats *stats,
                                         const bath_orf_result **results, int64_t *n_results) {
  std::vector<FilterState> states;
  return run_filters_lanes(ctx, om, dna, prm, stats, results, n_results, &states,
                           [](int, bath_hip_ctx *, const bath_hip_seqs *, const FilterState &) { return (int)BATH_OK; });
}

// =================================================================================================
// Frameshift stage: p7_pli_BuildDNAWindows + p7_pli_Frameshift up to the branch decision
// (p7_pipeline.c:462-572, 1339-1470).  The ORFs that passed F4 are few (10^-4 of all ORFs): window building is
// the reference's own serial logic on the host; the DNA windows' bias filter and 3-codon frameshift Forward run
// on the GPU, batched over all windows of the block.
// =================================================================================================
namespace bath {


// copy each window out of its sequence, reverse-complemented for the bottom strand
__global__ void fs_window_gather_kernel(const uint8_t *__restrict__ dna, const FsWinDev *__restrict__ wins, int nw, const uint8_t *__restrict__ comp,
                                        uint8_t *__restrict__ pool) {
  for (int w = blockIdx.x; w < nw; w += gridDim.x) {
    const FsWinDev d = wins[w];
    const uint8_t *src = dna + d.src_off;
    uint8_t *dst = pool + d.dst_off;
    for (int i = threadIdx.x; i < d.len; i += blockDim.x) {
      const int r = d.start - 1 + i;                     // 0-based position on the strand being read
      dst[i] = d.strand ? comp[min((int)src[d.seq_n - 1 - r], 17)] : src[r];
    }
  }
}

// p7_bg_fs_FilterScore (p7_bg.c:522-561) without its length term: esl_hmm_Forward of the 2-state filter HMM over the
// canonical residues of each of the three frames.  Lane per (window, pass): pass 0 uses the model's composition,
// pass 1 the local composition of nodes kmin..kmax (p7_pli_ComputeLocalCompo).  out[(w*2+pass)*3 + frame].
__global__ void fs_bias_kernel(const uint8_t *__restrict__ pool, const FsWinDev *__restrict__ wins, int nw, const uint8_t *__restrict__ aa_full, int M,
                               const float *__restrict__ eo_global, const uint8_t *__restrict__ ssv_scores, int base_b, float scale_b,
                               const float *__restrict__ bgf, float *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * nw) return;
  const int w = t >> 1, pass = t & 1;
  const FsWinDev d = wins[w];
  float eo_local[2 * kKp];
  const float *e = eo_global;
  if (pass == 1) {
    if (d.kmin > d.kmax) { out[t * 3] = out[t * 3 + 1] = out[t * 3 + 2] = -INFINITY; return; }
    local_compo_eo(ssv_scores, M, base_b, scale_b, bgf, d.kmin, d.kmax, eo_local);
    e = eo_local;
  }
  const uint8_t *s = pool + d.dst_off;
  const int L = d.len, L3 = L / 3;
  const float p1 = (float)L3 / (float)(L3 + 1);                             // p7_bg_SetLength(bg, length/3), p7_bg.c:193
  const float L1 = (float)M / 8.0f;
  const float t00 = p1, t01 = 1.0f - p1, t10 = 1.0f / (L1 + 1.0f), t11 = L1 / (L1 + 1.0f);
  for (int f = 0; f < 3; f++) {
    float logsc = 0.0f, d0 = 0.f, d1 = 0.f;
    int cnt = 0;
    for (int i = f; i + 2 < L; i += 3) {
      const int a = min((int)s[i], 17), b = min((int)s[i + 1], 17), c = min((int)s[i + 2], 17);
      const int x = aa_full[(a * 18 + b) * 18 + c];
      if (x >= 20) continue;                                                // esl_abc_XIsCanonical
      float n0, n1;
      if (cnt == 0) { n0 = e[2 * x] * 0.999f; n1 = e[2 * x + 1] * 0.001f; }
      else {
        n0 = 0.0f + d0 * t00; n0 = n0 + d1 * t10; n0 *= e[2 * x];
        n1 = 0.0f + d0 * t01; n1 = n1 + d1 * t11; n1 *= e[2 * x + 1];
      }
      const float mx = fmaxf(fmaxf(n0, 0.0f), n1);
      d0 = n0 / mx; d1 = n1 / mx;
      logsc += (float)log((double)mx);
      cnt++;
    }
    if (cnt == 0) logsc = -INFINITY;                                         // esl_hmm_Forward, L = 0: log(pi[M]) = log 0
    else { float end = 0.0f + d0 * 1.0f; end = end + d1 * 1.0f; logsc += (float)log((double)end); }
    out[t * 3 + f] = logsc;
  }
}

// Copies the regions described by <regs> (dst_off is filled in here) into one pool on the device, reverse-complemented
// for the bottom strand, and returns a sequence-block view of the pool (borrowed pointers: clear them before <view> dies).
int fs_gather_view(bath_hip_ctx *ctx, const bath_hip_seqs *dna, std::vector<FsWinDev> &regs, const uint8_t *d_comp, bath_hip_seqs *view, const FsWinDev **d_desc_out) {
  const int nw = (int)regs.size();
  int64_t pool_bytes = 0;
  for (FsWinDev &d : regs) { d.dst_off = pool_bytes; pool_bytes += ((int64_t)d.len + 15) / 16 * 16 + 16; }
  DevBuf &b_pool = ctx->scratch[29], &b_desc = ctx->scratch[30];
  BATH_HIP_TRY(ctx, b_pool.reserve((size_t)pool_bytes + 256));
  const size_t desc_bytes = ((size_t)nw * sizeof(FsWinDev) + 255) / 256 * 256;
  BATH_HIP_TRY(ctx, b_desc.reserve(desc_bytes + (size_t)nw * 12 + 64));          // descriptors, then the view's off[] and len[]
  BATH_HIP_TRY(ctx, hipMemsetAsync(b_pool.p, 0x1d, (size_t)pool_bytes + 256, ctx->stream));
  FsWinDev *d_desc = b_desc.as<FsWinDev>();
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_desc, regs.data(), (size_t)nw * sizeof(FsWinDev), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(fs_window_gather_kernel, dim3((unsigned)std::max(1, std::min(nw, 65535))), dim3(256), 0, ctx->stream, dna->d_data, d_desc, nw, d_comp, b_pool.as<uint8_t>());
  BATH_HIP_TRY(ctx, hipGetLastError());
  view->ctx = ctx; view->n = nw; view->d_data = b_pool.as<uint8_t>(); view->is_part = true;   // is_part: the destructor path must not free borrowed memory
  view->h_off.resize((size_t)nw); view->h_len.resize((size_t)nw);
  view->maxlen = 0; view->total = 0;
  for (int i = 0; i < nw; i++) {
    view->h_off[(size_t)i] = regs[(size_t)i].dst_off; view->h_len[(size_t)i] = regs[(size_t)i].len;
    view->maxlen = std::max(view->maxlen, regs[(size_t)i].len); view->total += regs[(size_t)i].len;
  }
  view->total_aligned = pool_bytes;
  int64_t *d_voff = reinterpret_cast<int64_t *>(reinterpret_cast<char *>(d_desc) + desc_bytes);
  int32_t *d_vlen = reinterpret_cast<int32_t *>(d_voff + nw);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_voff, view->h_off.data(), (size_t)nw * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_vlen, view->h_len.data(), (size_t)nw * 4, hipMemcpyHostToDevice, ctx->stream));
  view->d_off = d_voff; view->d_len = d_vlen;
  if (d_desc_out) *d_desc_out = d_desc;
  return BATH_OK;
}

// the same for windows built on the device (bath_fs_windows.hip): descriptors with their pool offsets, the view's off[] and
// len[] are already there; only the pool is reserved and the copy kernel launched
int fs_gather_view_built(bath_hip_ctx *ctx, const bath_hip_seqs *dna, const FsWinBuild &B, const uint8_t *d_comp, bath_hip_seqs *view, const FsWinDev **d_desc_out) {
  const int nw = B.nw;
  DevBuf &b_pool = ctx->scratch[56];                                          // its own pool: the speculative Backward reads it while the domain stage gathers into scratch[29]
  BATH_HIP_TRY(ctx, b_pool.reserve((size_t)B.pool_bytes + 256));
  BATH_HIP_TRY(ctx, hipMemsetAsync(b_pool.p, 0x1d, (size_t)B.pool_bytes + 256, ctx->stream));
  hipLaunchKernelGGL(fs_window_gather_kernel, dim3((unsigned)std::max(1, std::min(nw, 65535))), dim3(256), 0, ctx->stream, dna->d_data, B.d_desc, nw, d_comp, b_pool.as<uint8_t>());
  BATH_HIP_TRY(ctx, hipGetLastError());
  view->ctx = ctx; view->n = nw; view->d_data = b_pool.as<uint8_t>(); view->is_part = true;
  view->h_off.resize((size_t)nw); view->h_len.resize((size_t)nw);
  for (int i = 0; i < nw; i++) { view->h_off[(size_t)i] = B.h_voff[i]; view->h_len[(size_t)i] = B.h_vlen[i]; }
  view->maxlen = B.maxlen; view->total = B.total; view->total_aligned = B.pool_bytes;
  view->d_off = const_cast<int64_t *>(B.d_voff); view->d_len = const_cast<int32_t *>(B.d_vlen);
  if (d_desc_out) *d_desc_out = B.d_desc;
  return BATH_OK;
}

float flogsum_host(float a, float b) {               // p7_FLogsum, logsum.c:105-111 (table of logsum.c:89)
  // (a function-local static with an initialiser: built once, thread-safe -- worker contexts and the window threads call this concurrently)
  static const std::vector<float> tbl = [] { std::vector<float> t(16000); for (int i = 0; i < 16000; i++) t[i] = (float)std::log(1. + std::exp((double)-i / 1000.f)); return t; }();
  const float mx = std::max(a, b), mn = std::min(a, b);
  return (mn == -INFINITY || (mx - mn) >= 15.7f) ? mx : mx + tbl[(int)((mx - mn) * 1000.f)];
}

// The frameshift stage needs only the ORFs that passed F4 (a few thousand of the block's 10^5-10^6 MSV survivors) and their
// hit windows: select them on the device, so that what crosses PCIe is a few hundred KB instead of every candidate array.
// (FsCandRec: bath_launch.hpp)
__global__ void fs_select_cands_kernel(Cand cand, int nc, FsCandRec *__restrict__ out, int *__restrict__ count, const int64_t *__restrict__ fxoff /* kept Forward rows by candidate, or null */) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    if (cand.stage[c] != 4) continue;
    const int slot = atomicAdd(count, 1);
    out[slot] = FsCandRec{cand.window[c], cand.off[c], cand.P[c], c, cand.sf[c], cand.startj[c], cand.len[c], cand.fwdsc[c], cand.nullsc[c], fxoff ? fxoff[c] : (int64_t)-1};
  }
}
__global__ void fs_select_wins_kernel(const WindowRec *__restrict__ wins, int nwins, const int32_t *__restrict__ stage, WindowRec *__restrict__ out, int *__restrict__ count) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwins; i += gridDim.x * blockDim.x) {
    const WindowRec w = wins[i];
    if (stage[w.cand] != 4) continue;
    out[atomicAdd(count, 1)] = w;
  }
}

struct FsOrf {                      // an ORF that passed F4, host side
  int cand;
  int64_t w = 0;                    // its sequence in the block
  int strand = 0;
  int64_t aa_off = 0;               // its residues in the amino-acid stream pool
  int32_t start, end, n;            // nt coordinates on the strand being read, residues
  double P;
  float fwd_null;                   // fwdsc - nullsc (pli_tmp->fwdsc, p7_pipeline.c:1782)
  int32_t wb = 0, we = 0;           // its hit windows, a range of the window list (ordered by start)
};
struct DnaWin { int64_t n; int32_t k, length; };

// The ORFs that passed F4 and (frameshift pipeline) their hit windows, from every lane of the cascade: candidate ids are made
// unique over the lanes, windows are the block's, and aa_off addresses the residues relative to <pool> (the first lane's pool;
// another lane's pool is another allocation in the same flat address space, its ORFs carry the distance between the two).
struct SurvivorSet {
  std::vector<FsCandRec> sel;        // by candidate id
  std::vector<WindowRec> wins;       // by (candidate id, n)
  int nc_total = 0;
  const uint8_t *pool = nullptr;
  FilterState S0;                    // the first lane's state: the model-dependent tables any lane holds alike
  // device_only: the lists stayed in the lanes' device memory (counts known, nothing copied): what bath_fs_windows.hip reads;
  // survivors_to_host() completes sel / wins from them when the host path must run after all
  bool on_device = false;
  std::vector<FsLaneSurv> lanes;
  std::vector<bath_hip_ctx *> lane_ctx;
  int n_states = 0;
};

static int fetch_survivors(bath_hip_ctx *lane, const FsLaneSurv &ls, std::vector<FsCandRec> *sel, std::vector<WindowRec> *wins) {
  sel->resize((size_t)ls.n_c); wins->resize((size_t)ls.n_w);
  if (ls.n_c) BATH_HIP_TRY(lane, hipMemcpyAsync(sel->data(), ls.d_c, sel->size() * sizeof(FsCandRec), hipMemcpyDeviceToHost, lane->stream));
  if (ls.n_w) BATH_HIP_TRY(lane, hipMemcpyAsync(wins->data(), ls.d_w, wins->size() * sizeof(WindowRec), hipMemcpyDeviceToHost, lane->stream));
  BATH_HIP_TRY(lane, hipStreamSynchronize(lane->stream));
  // the kernels append in completion order: candidate order makes what follows deterministic
  std::sort(sel->begin(), sel->end(), [](const FsCandRec &a, const FsCandRec &b) { return a.cand < b.cand; });
  std::stable_sort(wins->begin(), wins->end(), [](const WindowRec &a, const WindowRec &b) { return a.cand != b.cand ? a.cand < b.cand : a.n < b.n; });
  return BATH_OK;
}

static int select_survivors(bath_hip_ctx *lane, const FilterState &S, bool want_wins, std::vector<FsCandRec> *sel, std::vector<WindowRec> *wins,
                            FsLaneSurv *dev = nullptr /* non-null: leave the lists on the device, report where */) {
  sel->clear(); wins->clear();
  if (dev) *dev = FsLaneSurv{};
  const int nc = S.hc.cand_count;
  if (nc <= 0) return BATH_OK;
  const int nwins = want_wins ? std::min(S.hc.win_count, S.W.win_cap) : 0;
  DevBuf &b_sel = lane->scratch[32];
  const size_t o_c = 256, o_w = o_c + ((size_t)nc * sizeof(FsCandRec) + 255) / 256 * 256;
  BATH_HIP_TRY(lane, b_sel.reserve(o_w + (size_t)std::max(nwins, 1) * sizeof(WindowRec) + 256));
  int *d_cnt = b_sel.as<int>();
  FsCandRec *d_c = reinterpret_cast<FsCandRec *>(b_sel.as<char>() + o_c);
  WindowRec *d_w = reinterpret_cast<WindowRec *>(b_sel.as<char>() + o_w);
  BATH_HIP_TRY(lane, hipMemsetAsync(d_cnt, 0, 256, lane->stream));
  const int blocks = lane->prop.multiProcessorCount * 4;
  hipLaunchKernelGGL(fs_select_cands_kernel, dim3(blocks), dim3(256), 0, lane->stream, S.W.cand, nc, d_c, d_cnt, lane->fwd_rows_kept ? lane->fwd_rows_off : nullptr);
  if (nwins > 0) hipLaunchKernelGGL(fs_select_wins_kernel, dim3(blocks), dim3(256), 0, lane->stream, S.W.wins, nwins, S.W.cand.stage, d_w, d_cnt + 1);
  BATH_HIP_TRY(lane, hipGetLastError());
  int h_cnt[2] = {0, 0};
  BATH_HIP_TRY(lane, hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, lane->stream));
  BATH_HIP_TRY(lane, hipStreamSynchronize(lane->stream));
  FsLaneSurv ls{};
  ls.d_c = d_c; ls.d_w = d_w; ls.n_c = h_cnt[0]; ls.n_w = h_cnt[1];
  if (dev) { *dev = ls; return BATH_OK; }
  return fetch_survivors(lane, ls, sel, wins);
}

static void merge_lane_lists(SurvivorSet *out, const std::vector<std::vector<FsCandRec>> &lsel, const std::vector<std::vector<WindowRec>> &lwin) {
  out->sel.clear(); out->wins.clear();
  for (size_t k = 0; k < (size_t)out->n_states; k++) {
    const FsLaneSurv &ls = out->lanes[k];
    for (FsCandRec q : lsel[k]) { q.cand += ls.cand_base; q.window += ls.first_window; q.aa_off += ls.dpool; if (out->n_states > 1) q.fxoff = -1; out->sel.push_back(q); }
    for (WindowRec w : lwin[k]) { w.cand += ls.cand_base; out->wins.push_back(w); }
  }
}

// the host path after a device-only cascade (an input bath_fs_windows.hip does not take): the lanes' lists come over after all
static int survivors_to_host(SurvivorSet *sv) {
  if (!sv->on_device) return BATH_OK;
  std::vector<std::vector<FsCandRec>> lsel((size_t)sv->n_states);
  std::vector<std::vector<WindowRec>> lwin((size_t)sv->n_states);
  for (size_t k = 0; k < (size_t)sv->n_states; k++) {
    const int st = fetch_survivors(sv->lane_ctx[k], sv->lanes[k], &lsel[k], &lwin[k]);
    if (st != BATH_OK) return st;
  }
  merge_lane_lists(sv, lsel, lwin);
  sv->on_device = false;
  return BATH_OK;
}

static int filters_with_survivors(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                                  bath_pipeline_stats *stats, const bath_orf_result **results, int64_t *n_results, bool want_wins, SurvivorSet *out,
                                  bool device_only = false) {
  std::vector<FilterState> states;
  const size_t nl = (size_t)std::max(1, pipeline_lane_count(dna));
  std::vector<std::vector<FsCandRec>> lsel(nl);
  std::vector<std::vector<WindowRec>> lwin(nl);
  std::vector<int64_t> first(nl, 0);
  std::vector<FsLaneSurv> ldev(nl);
  std::vector<bath_hip_ctx *> lctx(nl, nullptr);
  StageGate gate(prm->fs_pipe ? ctx->device : -1, StageGate::kCascade);      // (BATH_HIP_FS_GATE=3: not while another worker's Forward parser has the chip)
  int st = run_filters_lanes(ctx, om, dna, prm, stats, results, n_results, &states,
                             [&](int k, bath_hip_ctx *lane, const bath_hip_seqs *part, const FilterState &S) {
                               if ((size_t)k >= nl) { lane->set_error("more pipeline lanes than the survivor merge was sized for"); return (int)BATH_EFAIL; }
                               first[(size_t)k] = part->is_part ? part->first_window : 0;
                               lctx[(size_t)k] = lane;
                               return select_survivors(lane, S, want_wins, &lsel[(size_t)k], &lwin[(size_t)k], device_only ? &ldev[(size_t)k] : nullptr);
                             });
  gate.release();
  if (st != BATH_OK) return st;
  out->sel.clear(); out->wins.clear(); out->nc_total = 0;
  out->S0 = states.empty() ? FilterState{} : states[0];
  out->pool = out->S0.W.pool;
  out->n_states = (int)states.size();
  out->lanes.assign(ldev.begin(), ldev.begin() + (ptrdiff_t)states.size());
  out->lane_ctx.assign(lctx.begin(), lctx.begin() + (ptrdiff_t)states.size());
  for (size_t k = 0; k < states.size(); k++) {
    FsLaneSurv &ls = out->lanes[k];
    ls.cand_base = out->nc_total;
    ls.first_window = first[k];
    ls.dpool = (int64_t)(reinterpret_cast<intptr_t>(states[k].W.pool) - reinterpret_cast<intptr_t>(out->pool));
    out->nc_total += std::max(states[k].hc.cand_count, 0);
  }
  out->on_device = device_only;
  if (!device_only) merge_lane_lists(out, lsel, lwin);                     // (the lanes fetched and ordered their lists as they finished)
  return BATH_OK;
}

}  // namespace bath

int bath::pipeline_filters_survivors(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna, const bath_pipeline_params *prm,
                                     bath_pipeline_stats *stats, std::vector<PipelineSurvivor> *out, const uint8_t **d_pool) {
  out->clear();
  SurvivorSet sv;
  int st = filters_with_survivors(ctx, om, dna, prm, stats, nullptr, nullptr, false, &sv);
  if (st != BATH_OK) return st;
  *d_pool = sv.pool;
  out->reserve(sv.sel.size());
  for (const FsCandRec &q : sv.sel) {
    PipelineSurvivor o;
    o.window = q.window; o.aa_off = q.aa_off; o.strand = q.sf / 3; o.start = q.sf % 3 + 3 * q.startj + 1; o.n = q.len;
    o.fx_off = ctx->fwd_rows_kept ? q.fxoff : (int64_t)-1;
    out->push_back(o);
  }
  std::sort(out->begin(), out->end(), [](const PipelineSurvivor &a, const PipelineSurvivor &b) {
    if (a.window != b.window) return a.window < b.window;
    if (a.strand != b.strand) return a.strand < b.strand;
    return a.start < b.start;
  });
  return BATH_OK;
}

extern "C" int bath_hip_pipeline_frameshift(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_fsprofile *om_fs3, const bath_hip_seqs *dna,
                                            const bath_pipeline_params *prm_in, bath_pipeline_stats *stats,
                                            const bath_orf_result **results, int64_t *n_results,
                                            const bath_fs_window **fs_windows, int64_t *n_fs_windows) {
  if (!ctx || !om || !om_fs3 || !dna || !prm_in || !fs_windows || !n_fs_windows) return BATH_EINVAL;
  if (fsprofile_codon_lengths(om_fs3) != 3) { ctx->set_error("the frameshift stage needs the 3-codon frameshift profile"); return BATH_EINVAL; }
  *fs_windows = nullptr; *n_fs_windows = 0;
  ctx->fs_windows.clear();
  if (ctx->spec_stream) BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->spec_stream));   // (a speculative Backward of the previous call that nothing waited for)
  ctx->fs_spec_valid = false;
  bath_pipeline_params prm = *prm_in;
  prm.fs_pipe = 1;
  bath_pipeline_stats st_local{};
  StageClock clk;
  SurvivorSet sv;
  const bool try_device = fs_windows_on_device();
  int st = filters_with_survivors(ctx, om, dna, &prm, &st_local, results, n_results, true, &sv, try_device);
  if (st != BATH_OK) return st;
  clk.lap("fs:   cascade + F4 survivors selected");
  const FilterState &S = sv.S0;
  const int nc = sv.nc_total;
  const int M = om->M;
  const double kLn2 = 0.69314718055994529;

  // ---- p7_pli_BuildDNAWindows + the per-window ORF summary of p7_pli_Frameshift: on the device (bath_fs_windows.hip), from the
  // lists the cascade's lanes left there; the host path below is the A/B twin (BATH_HIP_FS_WINDOWS_HOST=1) and the fallback
  std::vector<bath_fs_window> out;
  std::vector<FsWinDev> dev;
  std::vector<PipelineSurvivor> std_all;                                      // the ORFs the standard branch would take (:1479-1487) ...
  std::vector<int32_t> std_begin;                                            // ... of window i: [std_begin[i], std_begin[i+1])
  FsWinBuild B;
  bool built = false;
  if (try_device) {
    st = fs_build_windows_device(ctx, om, om_fs3, dna, &prm, sv.lanes.data(), sv.n_states, nc, &B);
    if (st == BATH_OK) {
      built = true;
      out.assign((size_t)B.nw, bath_fs_window{});                           // (the records arrive completed, after the branch decision)
      clk.lap("fs:   DNA windows (device)");
    } else if (st != BATH_ENORESULT) return st;
    else if ((st = survivors_to_host(&sv)) != BATH_OK) return st;
  }
  if (!built) {
  // ---- the ORFs that passed F4, grouped by (sequence, strand) in the order esl_gencode emits a strand's ORFs: when the closing
  // stop codon is read; ORFs still open at the end of the sequence follow, frame by frame.  Each ORF's hit windows are a range
  // of <h_wins> (sorted by candidate, then start).
  const std::vector<WindowRec> &h_wins = sv.wins;
  std::vector<FsOrf> orfs_all;
  orfs_all.reserve(sv.sel.size());
  {
    size_t wi = 0;
    for (const FsCandRec &q : sv.sel) {                                      // ascending candidate id, like h_wins
      const int strand = q.sf / 3, frame = q.sf % 3;
      FsOrf o;
      o.cand = q.cand; o.w = q.window; o.strand = strand; o.aa_off = q.aa_off; o.n = q.len; o.start = frame + 3 * q.startj + 1; o.end = o.start + 3 * o.n - 1; o.P = q.P; o.fwd_null = q.fwdsc - q.nullsc;
      while (wi < h_wins.size() && h_wins[wi].cand < q.cand) wi++;
      o.wb = (int32_t)wi;
      while (wi < h_wins.size() && h_wins[wi].cand == q.cand) wi++;
      o.we = (int32_t)wi;
      orfs_all.push_back(o);
    }
  }
  std::stable_sort(orfs_all.begin(), orfs_all.end(), [dna](const FsOrf &a, const FsOrf &b) {
    if (a.w != b.w) return a.w < b.w;
    if (a.strand != b.strand) return a.strand < b.strand;
    const int n_seq = dna->h_len[(size_t)a.w];
    const bool ea = a.end + 3 > n_seq, eb = b.end + 3 > n_seq;
    if (ea != eb) return !ea;
    if (!ea) return a.end < b.end;
    return (a.start - 1) % 3 < (b.start - 1) % 3;
  });

  // ---- p7_pli_BuildDNAWindows + the per-window ORF summary of p7_pli_Frameshift (host path)
  // The groups (sequence, strand) are independent: host threads take contiguous runs of groups and their outputs are joined in
  // order (the GPU waits for this step: 1.6 -> 1.2 ms for the bench block's 8 k ORFs, the sort of the ORFs included).
  std::vector<size_t> gstart;                                                // first ORF of every group, and the end
  for (size_t g0 = 0; g0 < orfs_all.size();) {
    gstart.push_back(g0);
    size_t g1 = g0;
    while (g1 < orfs_all.size() && orfs_all[g1].w == orfs_all[g0].w && orfs_all[g1].strand == orfs_all[g0].strand) g1++;
    g0 = g1;
  }
  gstart.push_back(orfs_all.size());
  const size_t ngroups = gstart.size() - 1;
  struct WinPart { std::vector<bath_fs_window> out; std::vector<FsWinDev> dev; std::vector<PipelineSurvivor> std_all; std::vector<int32_t> std_begin; };
  static const int wthreads = [] { const char *e = std::getenv("BATH_HIP_WINDOW_THREADS"); return e ? std::max(1, std::atoi(e)) : 4; }();
  const int nparts = (int)std::max<size_t>(1, std::min<size_t>((size_t)wthreads, ngroups / 256));
  std::vector<WinPart> parts((size_t)nparts);
  auto build = [&](int part) {
    WinPart &P = parts[(size_t)part];
    std::vector<bath_fs_window> &out = P.out;
    std::vector<FsWinDev> &dev = P.dev;
    std::vector<PipelineSurvivor> &std_all = P.std_all;
    std::vector<int32_t> &std_begin = P.std_begin;
    const size_t gb = ngroups * (size_t)part / (size_t)nparts, ge = ngroups * (size_t)(part + 1) / (size_t)nparts;
    std::vector<DnaWin> wl;
    for (size_t gi = gb; gi < ge; gi++) {
    const size_t g0 = gstart[gi], g1 = gstart[gi + 1];
    const int64_t w = orfs_all[g0].w;
    const int strand = orfs_all[g0].strand;
    const int n_seq = dna->h_len[(size_t)w];
    const FsOrf *orfs = orfs_all.data() + g0, *orfs_end = orfs_all.data() + g1;
    wl.clear();
    for (const FsOrf *po = orfs; po != orfs_end; po++) {
      const FsOrf &o = *po;
      int best = -1;
      float best_score = -INFINITY;
      for (int i = o.wb; i < o.we; i++) {                                    // :486-495
        const WindowRec &x = h_wins[(size_t)i];
        if (x.score > best_score || (x.score == best_score && x.length > (best >= 0 ? h_wins[(size_t)best].length : 0))) { best_score = x.score; best = i; }
      }
      int32_t cn, ck, cl;
      if (best >= 0) { cn = h_wins[(size_t)best].n; ck = h_wins[(size_t)best].k; cl = h_wins[(size_t)best].length; }
      else if (o.n >= M) { cn = (o.n - M) / 2 + 1; ck = M; cl = M; }        // :500-510: no window, centre of the model
      else { cn = 1; ck = M - ((M - o.n) / 2); cl = o.n; }
      int64_t ws = (int64_t)((double)(uint32_t)cn - (om->max_length * (0.1 + om->prefix_lengths[(size_t)(ck - cl + 1)])) + 1);      // :513
      int64_t we = (int64_t)((double)((uint32_t)cn + (uint32_t)cl) + (om->max_length * (0.1 + om->suffix_lengths[(size_t)ck])) - 2);   // :514
      ws = std::min<int64_t>(0, ws);                                        // :516-517 (sic)
      we = std::max<int64_t>(o.n, we);
      ws = std::max<int64_t>(1, (int64_t)o.start + ws * 3);                 // :520-527, o.start already on the strand being read
      we = std::min<int64_t>(n_seq, (int64_t)o.start + we * 3);
      wl.push_back(DnaWin{ws, ck, (int32_t)(we - ws + 1)});
    }
    std::stable_sort(wl.begin(), wl.end(), [](const DnaWin &a, const DnaWin &b) { return a.n < b.n; });      // p7_hmmwindow_SortByStart
    size_t keep = 0;
    for (size_t i = 1; i < wl.size(); i++) {                                // :541-566, pct_overlap = 0
      DnaWin &prev = wl[keep];
      const DnaWin &cur = wl[i];
      const int64_t pe = prev.n + prev.length - 1, ce = cur.n + cur.length - 1;
      const int32_t ov = (int32_t)(std::min(pe, ce) - std::max(prev.n, cur.n) + 1);
      const int64_t ms = std::min(prev.n, cur.n), me = std::max(pe, ce);
      const int32_t ml = (int32_t)(me - ms + 1);
      if (((float)ov / std::min(prev.length, cur.length) > 0.f) && ml < (2 * (om->max_length * 3))) { prev.n = ms; prev.length = ml; }
      else wl[++keep] = wl[i];
    }
    wl.resize(wl.empty() ? 0 : keep + 1);

    const int64_t dstart = strand ? n_seq : 1;                             // dnasq->start of a whole sequence
    for (const DnaWin &dw : wl) {
      bath_fs_window r{};
      r.window = w; r.strand = strand; r.n = (int32_t)dw.n; r.length = dw.length;
      const int64_t wstart = strand ? dstart - (dw.n + dw.length) : dstart + dw.n - 1;       // :1373-1374
      const int64_t wend = strand ? dstart - dw.n + 1 : wstart + dw.length - 1;
      int orf_cnt = 0, k_min = M, k_max = 0;
      float tot = -INFINITY;
      double P_min = INFINITY;
      std_begin.push_back((int32_t)std_all.size());
      for (const FsOrf *po = orfs; po != orfs_end; po++) {
        const FsOrf &o = *po;
        int64_t os, oe;
        if (strand) { const int64_t rs = (int64_t)n_seq - o.start + 1, re = (int64_t)n_seq - o.end + 1; os = dstart - (n_seq - re + 1) + 1; oe = dstart - (n_seq - rs + 1) + 1; }
        else { os = dstart + o.start - 1; oe = dstart + o.end - 1; }
        if (!(os >= wstart && oe <= wend)) continue;                        // :1405
        P_min = std::min(P_min, o.P);
        tot = flogsum_host(tot, o.fwd_null);
        orf_cnt++;
        for (int i = o.wb; i < o.we; i++) { const WindowRec &x = h_wins[(size_t)i]; k_min = std::min(k_min, x.k - x.length + 1); k_max = std::max(k_max, x.k); }
        if (!(o.P > prm.F3)) {                                              // :1483-1487
          PipelineSurvivor ps;
          ps.window = w; ps.aa_off = o.aa_off; ps.strand = strand; ps.start = o.start; ps.n = o.n; ps.win_start = (int32_t)dw.n; ps.fs_window = o.cand;
          std_all.push_back(ps);                                            // fs_window carries the candidate id until the branch is known
        }
      }
      r.orf_cnt = orf_cnt; r.k_min = k_min; r.k_max = k_max; r.tot_orfsc = tot; r.P_min = P_min;
      r.P_tot = prm.std_pipe ? exp_surv((double)tot / kLn2, om->evparam[BATH_FTAU], om->evparam[BATH_FLAMBDA]) : 1.0;    // :1457: --fsonly
      out.push_back(r);
      FsWinDev d{};
      d.src_off = dna->h_off[w]; d.dst_off = 0; d.seq_n = n_seq; d.start = (int32_t)dw.n; d.len = dw.length; d.strand = strand;
      d.kmin = k_min; d.kmax = k_max;
      dev.push_back(d);
    }
  }
  };
  if (nparts == 1) build(0);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < nparts; t++) th.emplace_back(build, t);
    build(0);
    for (std::thread &t : th) t.join();
  }
  for (WinPart &P : parts) {
    const int32_t shift = (int32_t)std_all.size();
    out.insert(out.end(), P.out.begin(), P.out.end());
    dev.insert(dev.end(), P.dev.begin(), P.dev.end());
    for (int32_t b : P.std_begin) std_begin.push_back(b + shift);
    std_all.insert(std_all.end(), P.std_all.begin(), P.std_all.end());
  }
  std_begin.push_back((int32_t)std_all.size());
  clk.lap("fs:   DNA windows (host)");
  }   // !built
  const int nw = (int)out.size();
  int64_t pos_fwd = 0;
  ctx->fs_std_orfs.clear();
  ctx->fs_std_pool = sv.pool;
  std::vector<char> aligned((size_t)std::max(nc, 1), 0);                   // oxf_holder[i] == NULL: an overlapping window already took the ORF (:1485)
  if (nw > 0) {
    // ---- windows -> device, bias filter and 3-codon frameshift Forward for all of them
    DevBuf &b_out = ctx->scratch[31];
    bath_hip_seqs view;
    const FsWinDev *d_desc = nullptr;
    if (built) st = fs_gather_view_built(ctx, dna, B, S.tt.comp, &view, &d_desc);      // descriptors, offsets and lengths are already on the device
    else st = fs_gather_view(ctx, dna, dev, S.tt.comp, &view, &d_desc);
    if (st != BATH_OK) return st;
    BATH_HIP_TRY(ctx, b_out.reserve((size_t)nw * 6 * sizeof(float) + 64));
    // the bias filter (a lane per window and pass, serial over the window) and the Forward parser are independent: the bias
    // kernel goes to the side stream, behind the gather
    if ((st = fs_fork(ctx)) != BATH_OK) return st;
    hipLaunchKernelGGL(fs_bias_kernel, dim3((unsigned)((2 * nw + 63) / 64)), dim3(64), 0, ctx->side_stream, view.d_data, d_desc, nw, S.tt.full, M, om->d_bias_eo,
                       S.d_ssvsc, (int)om->base_b, om->scale_b, S.d_bgf, b_out.as<float>());
    BATH_HIP_TRY(ctx, hipGetLastError());
    std::vector<float> h_bias((size_t)nw * 6);
    std::vector<float> h_fsc((size_t)nw);
    ctx->fs_regions_all.clear();
    ctx->fs_keep_xoff.clear();
    static const bool spec_all = [] { const char *e = std::getenv("BATH_HIP_FS_SPEC_ALL"); return e && e[0] == '1'; }();   // (probe: both parsers side by side whatever the count)
    static const bool spec_force = [] { const char *e = std::getenv("BATH_HIP_FS_SPEC_FORCE"); return e && e[0] == '1'; }();   // tests: the speculative path for a handful of windows
    if (ctx->fs_want_regions && !spec_force && (spec_all || nw <= (int64_t)ctx->prop.multiProcessorCount * 16)) {
      // The domain stage follows: its Backward parser, domain decoding and region heuristics run here, for every window,
      // next to the Forward parser whose score decides the branch.  These kernels are bound by the row chain of the longest
      // window as long as every window has a wave of its own (16 waves per CU), so up to that many windows the ones that will
      // take the standard branch cost nothing extra and the frameshift-branch windows have their regions before the decision
      // is made.  Beyond it the kernels take a second round of windows and the Backward parser of the windows that turn out
      // not to need it costs more than it saves (bench block, 7.6 k windows: 49.6 ms instead of 35 + 11.6 ms).
      ctx->fs_regions_all.assign((size_t)nw * (size_t)(1 + 3 * fs_max_regions()), 0);
      const float pmove = (2.0f + 1.0f) / (100.0f + 2.0f + 1.0f);              // p7_fs_ReconfigLength(L = 100, nj = 1): the saved length (p7_domaindef.c:318)
      st = fs3_regions(ctx, om_fs3, &view, (float)std::log((double)(1.0f - pmove)), ctx->fs_regions_all.data(), h_fsc.data());
    } else {
      // more windows than both parsers can run side by side for: Forward for all of them, and -- when the domain stage follows, the
      // windows were built on the device (their pool is not the one that stage gathers into) and this context is the host's only one
      // at work -- the Backward parser of the LONGEST windows speculatively beside it (fs3_backward_spec).
      // BATH_HIP_FS_SPEC_K: how many.  DEFAULT 0 = OFF: measured on the bench block (profiles/r06_fs_spec_probe.txt), the speculation
      // does not pay -- the Forward launch is batched so that all of its 238 blocks end together (chain_batches), so it holds 238 of the
      // 256 CUs for its whole 9.4 ms and the speculative blocks beyond the 18 free CUs start when Forward ENDS, later than the domain
      // stage would have started them; launched first instead, they push Forward's blocks into a second round (21.4 -> 25.6-30.8 ms for
      // 12.9-9.3 ms of Backward saved).  Kept, tested (BATH_HIP_FS_SPEC_FORCE=1 with a handful of windows), off.
      static const int spec_k = [] { const char *e = std::getenv("BATH_HIP_FS_SPEC_K"); return e ? std::atoi(e) : 0; }();
      const int k = spec_k;
      // launched AFTER Forward's kernel (the hook): Forward's blocks take their CUs first, the speculation gets the ones that are left
      // and the ones Forward's short batches give back -- launched first it delays Forward by more than it saves (measured)
      const std::function<int()> spec = [&]() -> int { return fs3_backward_spec(ctx, om_fs3, &view, k); };
      const bool speculate = ctx->fs_want_regions && built && k > 0 && host_contexts() <= 1;
      st = fs3_forward_scores(ctx, om_fs3, &view, h_fsc.data(), speculate ? &spec : nullptr);
    }
    view.d_data = nullptr; view.d_off = nullptr; view.d_len = nullptr;     // borrowed pointers: nothing for a destructor to free
    if (st != BATH_OK) return st;
    if (built) {
      // ---- scores -> P-values -> branch on the device (fsw_decide_kernel); the host reads the completed records once and keeps the
      // bookkeeping: pos_past_fwd, and the ORFs of the windows that take the standard branch (:1479-1487), from the ordered ORF list
      if ((st = fs_join(ctx)) != BATH_OK) return st;                          // the bias kernel's side stream
      if ((st = fs_decide_device(ctx, om_fs3, &prm, B, b_out.as<float>(), ctx->scratch[12].as<float>(), out.data())) != BATH_OK) return st;
      clk.lap("fs:   gather + bias + 3-codon Forward + branch decision (device)");
      for (int i = 0; i < nw; i++) {
        const bath_fs_window &r = out[(size_t)i];
        if (r.branch == 1) { pos_fwd += r.length; continue; }
        if (r.branch != 2) continue;
        const int n_seq = dna->h_len[(size_t)r.window];
        const int64_t dstart = r.strand ? n_seq : 1;
        const int64_t wstart = r.strand ? dstart - ((int64_t)r.n + r.length) : dstart + r.n - 1;
        const int64_t wend = r.strand ? dstart - r.n + 1 : wstart + r.length - 1;
        for (int32_t z = B.h_grp[2 * i]; z < B.h_grp[2 * i + 1]; z++) {
          const FsOrfDev &o = B.h_orfs[z];
          int64_t os, oe;
          if (r.strand) { const int64_t rs = (int64_t)n_seq - o.start + 1, re = (int64_t)n_seq - o.end + 1; os = dstart - (n_seq - re + 1) + 1; oe = dstart - (n_seq - rs + 1) + 1; }
          else { os = dstart + o.start - 1; oe = dstart + o.end - 1; }
          if (!(os >= wstart && oe <= wend) || o.P > prm.F3) continue;        // :1405, :1483
          if (aligned[(size_t)o.cand]) continue;                               // oxf_holder[i] == NULL: an overlapping window already took the ORF (:1485)
          aligned[(size_t)o.cand] = 1;
          pos_fwd += (int64_t)o.n * 3;
          PipelineSurvivor ps;
          ps.window = r.window; ps.aa_off = o.aa_off; ps.strand = r.strand; ps.start = o.start; ps.n = o.n; ps.win_start = r.n; ps.fs_window = i;
          ctx->fs_std_orfs.push_back(ps);
        }
      }
    } else {
    BATH_HIP_TRY(ctx, hipMemcpyAsync(h_bias.data(), b_out.p, h_bias.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->side_stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->side_stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    clk.lap("fs:   gather + bias + 3-codon Forward");

    // ---- scores -> P-values -> which branch each window takes (:1425-1464)
    const float *ev3 = fsprofile_evparam(om_fs3);
    for (int i = 0; i < nw; i++) {
      bath_fs_window &r = out[(size_t)i];
      const int L = r.length, L3 = L / 3;
      const float p1 = (float)L3 / (float)(L3 + 1);
      const float per_frame = (float)((float)L3 * std::log((double)p1) + std::log(1. - p1));           // p7_bg_fs_NullOne, p7_bg.c:380
      r.nullsc = (float)(per_frame + std::log(3.0));
      if (prm.do_biasfilter) {
        float fsc[2];
        for (int pass = 0; pass < 2; pass++) {
          const float *b = &h_bias[((size_t)i * 2 + pass) * 3];
          float sum = -INFINITY;
          for (int f = 0; f < 3; f++) sum = flogsum_host(sum, b[f]);
          fsc[pass] = (float)((double)sum + ((double)((float)L3 * logf(p1) + logf((float)(1. - p1))) + std::log(3.0)));   // p7_bg.c:561
        }
        r.filtersc = fsc[0];
        if (r.k_min <= r.k_max && fsc[1] > r.filtersc) r.filtersc = fsc[1];                             // :1432-1440
      } else r.filtersc = r.nullsc;
      r.fwdsc = h_fsc[(size_t)i];
      const float seqscore = (float)((r.fwdsc - r.filtersc) / kLn2);
      r.P_fs = exp_surv(seqscore, ev3[BATH_FTAUFS3], ev3[BATH_FLAMBDA]);
      r.P_null = exp_surv((r.fwdsc - r.nullsc) / kLn2, ev3[BATH_FTAUFS3], ev3[BATH_FLAMBDA]);
      if (r.P_fs <= prm.F3 && (r.P_null < r.P_tot || (r.P_null == r.P_tot && r.orf_cnt > 1) || r.P_min > prm.F3)) { r.branch = 1; pos_fwd += L; }
      else if (!prm.std_pipe) r.branch = 0;                                                             // :1480: --fsonly has no standard branch
      else {
        r.branch = 2;
        for (int32_t z = std_begin[(size_t)i]; z < std_begin[(size_t)i + 1]; z++) {
          PipelineSurvivor ps = std_all[(size_t)z];
          if (aligned[(size_t)ps.fs_window]) continue;
          aligned[(size_t)ps.fs_window] = 1;
          pos_fwd += (int64_t)ps.n * 3;
          ps.fs_window = i;
          ctx->fs_std_orfs.push_back(ps);
        }
      }
    }
    }   // host decision
  }
  clk.lap("fs:   branch decision");
  st_local.pos_past_fwd = pos_fwd;                                         // in the fs pipeline only this stage counts it (:1468, :1490)
  if (stats) *stats = st_local;
  ctx->fs_windows = out;
  *fs_windows = ctx->fs_windows.data(); *n_fs_windows = nw;
  return BATH_OK;
}

// =================================================================================================
// p7_ViterbiFilter_BATH / p7_SSVFilter_BATH over a block of amino-acid targets, with their hit windows: the batched forms
// the single-target prototypes of impl_hip/ bind (the pipeline above runs the same kernels on its own candidate lists).
// =================================================================================================
namespace bath {
static int windows_out(bath_hip_ctx *ctx, const WindowRec *d_wins, const int *d_count, int cap, const bath_hmm_window **wins, int64_t *nwins) {
  int n = 0;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(&n, d_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n > cap) { ctx->set_error("hit-window buffer too small"); return BATH_EMEM; }
  std::vector<WindowRec> h((size_t)n);
  if (n) BATH_HIP_TRY(ctx, hipMemcpy(h.data(), d_wins, (size_t)n * sizeof(WindowRec), hipMemcpyDeviceToHost));
  // the kernels append in completion order: restore the reference's order (by target, then by position in the target)
  std::sort(h.begin(), h.end(), [](const WindowRec &a, const WindowRec &b) { return a.cand != b.cand ? a.cand < b.cand : a.n < b.n; });
  ctx->hmm_windows.resize((size_t)n);
  for (int i = 0; i < n; i++) ctx->hmm_windows[(size_t)i] = bath_hmm_window{h[(size_t)i].cand, h[(size_t)i].n, h[(size_t)i].k, h[(size_t)i].length, h[(size_t)i].score};
  *wins = ctx->hmm_windows.data(); *nwins = n;
  return BATH_OK;
}
}  // namespace bath

extern "C" int bath_hip_vitfilter_bath(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, const float *filtersc, double P,
                                       float *sc, int32_t *status, const bath_hmm_window **wins, int64_t *nwins) {
  if (!ctx || !om || !sq || !filtersc || !sc || !wins || !nwins) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  *wins = nullptr; *nwins = 0;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  const int M = om->M;
  int st = om->ensure_len_tables(sq->maxlen + 1);
  if (st != BATH_OK) return st;
  std::vector<uint8_t> ssv_scores((size_t)(M + 1) * kKp, 0);
  bath_hip_oprofile_get_ssv_scores(om, ssv_scores.data());
  const int cap = (int)std::min<int64_t>((int64_t)1 << 26, sq->total / 2 + 16 * n + 1024);      // a window ends at least one residue after the last
  DevBuf &b = ctx->scratch[9];
  const size_t o_sc = 0, o_st = o_sc + (size_t)n * 4, o_f = o_st + (size_t)n * 4, o_km = o_f + (size_t)n * 4, o_cnt = o_km + (size_t)n * 8, o_ssv = o_cnt + 256,
               o_w = (o_ssv + ssv_scores.size() + 255) / 256 * 256, total = o_w + (size_t)cap * sizeof(WindowRec);
  BATH_HIP_TRY(ctx, b.reserve(total + 256));
  char *p = b.as<char>();
  BATH_HIP_TRY(ctx, hipMemcpyAsync(p + o_f, filtersc, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(p + o_ssv, ssv_scores.data(), ssv_scores.size(), hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemsetAsync(p + o_cnt, 0, 256, ctx->stream));
  VitWindowArgs wa{};
  wa.invP_vit = (double)(float)gumbel_invsurv(P, om->evparam[2], om->evparam[3]);          // vitfilter.c:314 (float invP)
  wa.invP_msv = (double)(float)gumbel_invsurv(P, om->evparam[0], om->evparam[1]);          // vitfilter.c:319
  wa.d_filtersc = reinterpret_cast<const float *>(p + o_f); wa.d_ssv_scores = reinterpret_cast<const uint8_t *>(p + o_ssv);
  wa.d_wins = p + o_w; wa.d_win_count = reinterpret_cast<int *>(p + o_cnt); wa.win_cap = cap; wa.d_kminmax = reinterpret_cast<int32_t *>(p + o_km);
  if ((st = launch_vit_wave(ctx, om, sq->view(), nullptr, n, reinterpret_cast<float *>(p + o_sc), reinterpret_cast<int32_t *>(p + o_st), &wa, nullptr)) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, p + o_sc, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (status) BATH_HIP_TRY(ctx, hipMemcpyAsync(status, p + o_st, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  return windows_out(ctx, reinterpret_cast<const WindowRec *>(p + o_w), wa.d_win_count, cap, wins, nwins);
}

extern "C" int bath_hip_ssvfilter_bath(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *sq, double P,
                                       const bath_hmm_window **wins, int64_t *nwins) {
  if (!ctx || !om || !sq || !wins || !nwins) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  *wins = nullptr; *nwins = 0;
  const int64_t n = sq->n;
  if (n == 0) return BATH_OK;
  if (n >= (int64_t)INT32_MAX) return BATH_EINVAL;
  const int M = om->M;
  int st = om->ensure_len_tables(sq->maxlen + 1);
  if (st != BATH_OK) return st;
  std::vector<uint8_t> ssv_scores((size_t)(M + 1) * kKp, 0);
  bath_hip_oprofile_get_ssv_scores(om, ssv_scores.data());
  const int cap = (int)std::min<int64_t>((int64_t)1 << 26, sq->total / 2 + 16 * n + 1024);
  DevBuf &b = ctx->scratch[9];
  const size_t o_todo = 0, o_km = o_todo + (size_t)n * 4, o_ctr = (o_km + (size_t)n * 8 + 255) / 256 * 256, o_ssv = o_ctr + 256 + sizeof(Counters),
               o_w = (o_ssv + ssv_scores.size() + 255) / 256 * 256, total = o_w + (size_t)cap * sizeof(WindowRec);
  BATH_HIP_TRY(ctx, b.reserve(total + 256));
  char *p = b.as<char>();
  std::vector<int32_t> todo((size_t)n);
  for (int64_t i = 0; i < n; i++) todo[(size_t)i] = (int32_t)i;
  Counters hc{};
  hc.todo_ssvb = (int)n;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(p + o_todo, todo.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(p + o_ctr, &hc, sizeof hc, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(p + o_ssv, ssv_scores.data(), ssv_scores.size(), hipMemcpyHostToDevice, ctx->stream));
  Cand cand{};                                        // the kernel reads a target's offset and length and writes its model range
  cand.off = sq->d_off; cand.len = sq->d_len; cand.kminmax = reinterpret_cast<int32_t *>(p + o_km);
  Counters *d_ctr = reinterpret_cast<Counters *>(p + o_ctr);
  const MsvConsts mc = msv_consts(om);
  const double invP = gumbel_invsurv(P, om->evparam[0], om->evparam[1]);                   // msvfilter.c:302
  const int Cc = (M + 63) / 64;
  int Cs = -1;
  for (int opt : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 52}) if (Cc <= opt) { Cs = opt; break; }
#define BATH_SSVB_CASE(N)                                                                                                          \
  case N:                                                                                                                          \
    hipLaunchKernelGGL(ssv_bath_kernel<N>, dim3(wave_grid_blocks(ctx) / 4), dim3(256), 0, ctx->stream, cand, d_ctr, reinterpret_cast<const int32_t *>(p + o_todo), \
                       sq->d_data, M, om->d_rb, om->rb_stride, reinterpret_cast<const uint8_t *>(p + o_ssv), om->lt.d_tjb, om->lt.d_nullsc, mc, invP,       \
                       reinterpret_cast<WindowRec *>(p + o_w), cap, d_ctr);                                                        \
    break;
  switch (Cs) {
    BATH_SSVB_CASE(1) BATH_SSVB_CASE(2) BATH_SSVB_CASE(3) BATH_SSVB_CASE(4) BATH_SSVB_CASE(6) BATH_SSVB_CASE(8)
    BATH_SSVB_CASE(12) BATH_SSVB_CASE(16) BATH_SSVB_CASE(24) BATH_SSVB_CASE(32) BATH_SSVB_CASE(52)
    default: ctx->set_error("model too long for the SSV window kernel"); return BATH_EINVAL;
  }
#undef BATH_SSVB_CASE
  BATH_HIP_TRY(ctx, hipGetLastError());
  return windows_out(ctx, reinterpret_cast<const WindowRec *>(p + o_w), &d_ctr->win_count, cap, wins, nwins);
}
