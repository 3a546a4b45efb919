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
//      packed int16 registers; lanes of a wave hold ORFs of equal length.  The lane compares its maximum with a
//      per-length threshold table computed on the host with the reference's double-precision P-value maths (so the
//      F1 decision is bit-identical) and appends the ~2% survivors to a candidate list.
//   3. survivors flow through decision kernels (lane per candidate) and DP kernels (lane or wave per candidate)
//      with device-side work lists: no host round trip until the final copy-out.  Their residues are read in place
//      from the amino-acid streams written by step 1.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"
#include "host_model.hpp"

using namespace bath;

namespace bath {

static const double kLog2 = 0.69314718055994529;

// ---------------------------------------------------------------------------------------------
// candidate storage (structure of arrays, indexed by candidate id)
// ---------------------------------------------------------------------------------------------
struct Cand {
  int64_t *window;  int32_t *sf;      // strand*3+frame
  int32_t *startj;  int32_t *len;     // first codon index within the stream, length in aa
  int16_t *v;                         // raw SSV maximum
  int64_t *off;                       // offset of the amino-acid sequence in the pool
  int32_t *msv_status, *vit_status, *fwd_status, *stage, *flags;
  float *usc, *nullsc, *filtersc, *vfsc, *fwdsc;
  double *P;
  int32_t *kminmax;                   // [2*cap]
};

enum { FLAG_VIT_RUN = 1, FLAG_HAS_WIN = 2 };

struct Counters {                     // device-side counters, one struct per pipeline call
  int cand_count, todo_msv, todo_vit, todo_ssvb, todo_vit2, todo_fwd, win_count, overflow;
  unsigned long long n_orfs, orf_res;                       // ORFs >= minlen, their total aa
  unsigned long long n_past_msv, n_past_bias, n_past_vit, n_past_fwd;
  unsigned long long pos_past_msv, pos_past_bias, pos_past_vit, pos_past_fwd;
  unsigned long long res_vit, res_fwd;                      // residues entering Viterbi / Forward
};

struct Params {
  double F1, F2, F3, F4;
  int do_bias, fs_pipe, minlen;
  float evparam[BATH_NEVPARAM];
};

// ---------------------------------------------------------------------------------------------
// 1. SSV over the length-sorted ORF list, lane per ORF (persistent waves striding over the list)
// ---------------------------------------------------------------------------------------------
template <int NR, int G>
__global__ __launch_bounds__(256) void ssv_orf_kernel(const uint8_t *__restrict__ aa, const OrfRec *__restrict__ orfs, const int *__restrict__ n_orfs_dev,
                                                      SeqView dna, const int16_t *__restrict__ cost_tab, int row_bytes,
                                                      const int16_t *__restrict__ emit_thresh, int thresh_max,
                                                      Cand cand, int cand_cap, Counters *__restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
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
  const int grank = lane % G;
  const char *tile = lds + grank * (4 * NR);
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const s16x2 fl = {(short)kSsvBegin, (short)kSsvBegin};
  for (int64_t tb = wave0 * TPW; tb < n_orfs; tb += nwaves * TPW) {
    const int64_t t = tb + lane / G;
    const bool live = t < n_orfs;
    OrfRec rec{0, 0, 0};
    if (live) rec = orfs[t];
    const int L = rec.len_sf & 0x0fffffff;
    const uint8_t *s = aa + rec.aa_off;
    const int Lw = wave_max_i32(L);
    s16x2 reg[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) reg[r] = fl;
    s16x2 xE = fl;
    uint32_t wnext = (0 < L) ? *reinterpret_cast<const uint32_t *>(s) : 0x1d1d1d1du;
    for (int i0 = 0; i0 < Lw; i0 += 4) {
      const uint32_t w4 = wnext;
      wnext = (i0 + 4 < L) ? *reinterpret_cast<const uint32_t *>(s + i0 + 4) : 0x1d1d1d1du;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int x = (w4 >> (8 * j)) & 0xff;
        x = (i0 + j < L) ? min(x, kKp - 1) : kRowReset;
        const unsigned carry = ssv_carry<NR, G>(reg, grank);
        ssv_row<NR>(reg, xE, tile + x * row_bytes, carry);
      }
    }
    const int v = ssv_group_max<G>(xE);
    if (live && grank == 0 && v >= (int)emit_thresh[min(L, thresh_max)]) {
      const int slot = atomicAdd(&ctr->cand_count, 1);
      if (slot < cand_cap) {
        const int sf = (int)((unsigned)rec.len_sf >> 28);
        const int64_t w = rec.w;
        const int64_t stream = 2 * dna.off[w] + 96 * w + (int64_t)sf * orf_stream_pitch(dna.len[w]);
        cand.window[slot] = w; cand.sf[slot] = sf; cand.startj[slot] = (int32_t)(rec.aa_off - stream); cand.len[slot] = L;
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
      filtersc = bias_forward(pool + cand.off[c], L, M, eo, p1_tab[L]);
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
    for (int i = 1; i <= L; i++) {
      const int x = min((int)s[i - 1], kKp - 1);
      const uint8_t *row = rb + (size_t)x * rb_stride;
      int prev = __shfl_up(dp[C - 1], 1, 64);
      if (lane == 0) prev = 0;
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
      }
    }
    if (lane == 0) { cand.kminmax[2 * c] = kmin; cand.kminmax[2 * c + 1] = kmax; }
  }
}

// p7_pli_ComputeLocalCompo (p7_pipeline.c:427-458) + p7_bg_SetFilter + esl_hmm_Configure for one candidate
__device__ void local_compo_eo(const uint8_t *ssv_scores, int M, int base_b, float scale_b, const float *bgf, int k_start, int k_end, float *eo /* [Kp][2] */) {
  float compo[20];
  const int k_len = k_end - k_start + 1;
  if (k_len < 20) { k_start -= (20 - k_len) / 2; k_end += (20 - k_len) / 2; }
  k_start = max(1, k_start); k_end = min(M, k_end);
  for (int x = 0; x < 20; x++) compo[x] = 0.0f;
  for (int k = k_start; k <= k_end; k++)
    for (int x = 0; x < 20; x++) {
      const float lo = ((float)base_b - (float)ssv_scores[(size_t)k * kKp + x]) / scale_b;
      compo[x] += bgf[x] * expf(lo);
    }
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
__global__ void post_vit_kernel(Cand cand, int cand_cap, Counters *__restrict__ ctr, Params p, const uint8_t *__restrict__ pool, int M,
                                const uint8_t *__restrict__ ssv_scores, int base_b, float scale_b, const float *__restrict__ bgf,
                                const float *__restrict__ p1_tab, const float *__restrict__ lt1_tab, const float *__restrict__ lt2_tab,
                                int32_t *__restrict__ todo_vit2, int32_t *__restrict__ todo_fwd) {
  const int ncand = min(ctr->cand_count, cand_cap);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncand; c += gridDim.x * blockDim.x) {
    if (cand.stage[c] != 2) continue;
    const int L = cand.len[c];
    const float usc = cand.usc[c];
    float filtersc = cand.filtersc[c];
    float vfsc = cand.vfsc[c];
    double P = cand.P[c];
    if (cand.flags[c] & FLAG_VIT_RUN) {
      const float seqsc = (float)((double)(vfsc - filtersc) / kLog2);
      P = d_gumbel_surv(seqsc, p.evparam[2], p.evparam[3]);
      cand.P[c] = P;
      if (P > p.F2) { cand.kminmax[2 * c] = 1 << 30; cand.kminmax[2 * c + 1] = 0; continue; }
    } else vfsc = -INFINITY;
    atomicAdd(&ctr->n_past_vit, 1ull); atomicAdd(&ctr->pos_past_vit, (unsigned long long)L * 3ull);
    const int kmin = cand.kminmax[2 * c], kmax = cand.kminmax[2 * c + 1];
    if (p.do_bias && kmin <= kmax) {
      cand.flags[c] |= FLAG_HAS_WIN;
      float eo[kKp * 2];
      local_compo_eo(ssv_scores, M, base_b, scale_b, bgf, kmin, kmax, eo);
      float lf = bias_forward(pool + cand.off[c], L, M, eo, p1_tab[L]);
      lf = (lf + lt1_tab[L]) + lt2_tab[L];
      if (lf > filtersc) {
        filtersc = lf;
        cand.filtersc[c] = filtersc;
        if (vfsc == -INFINITY) {
          const float seqsc = (float)((double)(usc - filtersc) / kLog2);
          P = d_gumbel_surv(seqsc, p.evparam[0], p.evparam[1]);
          cand.P[c] = P;
          if (P > p.F2) { cand.stage[c] = 5; todo_vit2[atomicAdd(&ctr->todo_vit2, 1)] = c; atomicAdd(&ctr->res_vit, (unsigned long long)L); continue; }
        } else {
          const float seqsc = (float)((double)(vfsc - filtersc) / kLog2);
          P = d_gumbel_surv(seqsc, p.evparam[2], p.evparam[3]);
          cand.P[c] = P;
          if (P > p.F2) continue;                          // rejected; stays at stage 2 (the reference has already counted it past Vit)
        }
      }
    }
    cand.stage[c] = 3;
    todo_fwd[atomicAdd(&ctr->todo_fwd, 1)] = c; atomicAdd(&ctr->res_fwd, (unsigned long long)L);
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
    for (int v = -128; v <= 127; v++) {
      float usc = 0.f;
      const int st = ssv_classify_host(v, tjb, mc, &usc);
      bool keep = (st != BATH_OK);
      if (!keep) {
        const float seqsc = (float)((double)(usc - nullsc) / kLog2);
        keep = !(gumbel_surv(seqsc, om->evparam[0], om->evparam[1]) > F1);
      }
      if (keep) { tab[len] = (int16_t)v; break; }
    }
  }
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

static size_t layout(PipelineWork &w, char *base, int cap) {
  char *p = base;
  const size_t n = (size_t)cap;
  w.cand_cap = cap; w.win_cap = 2 * cap;
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
}

extern "C" int bath_hip_pipeline_timings(const bath_hip_ctx *ctx, int max, const char **names, float *ms, int64_t *launches) {
  int n = std::min<int>(max, (int)ctx->timings.size());
  for (int i = 0; i < n; i++) { names[i] = ctx->timings[i].name; ms[i] = ctx->timings[i].ms; launches[i] = ctx->timings[i].launches; }
  return n;
}

extern "C" int bath_hip_pipeline_filters(bath_hip_ctx *ctx, const bath_hip_oprofile *om, const bath_hip_seqs *dna,
                                         const bath_pipeline_params *prm, bath_pipeline_stats *stats,
                                         const bath_orf_result **results, int64_t *n_results) {
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
  if ((st = orf_tables_upload(ctx, prm->ncbi_table, &tt)) != BATH_OK) return st;
  std::vector<int16_t> emit;
  build_emit_table(om, prm->F1, max_orf, emit);
  std::vector<uint8_t> ssv_scores((size_t)(M + 1) * kKp, 0);
  bath_hip_oprofile_get_ssv_scores(om, ssv_scores.data());

  DevBuf &b_tabs = ctx->scratch[8], &b_work = ctx->scratch[9];
  const size_t tabs_bytes = 8192 + emit.size() * 2 + 256 + ssv_scores.size() + 256 + 20 * 4 + 256;
  BATH_HIP_TRY(ctx, b_tabs.reserve(tabs_bytes));
  char *tp = b_tabs.as<char>();
  int16_t *d_emit = reinterpret_cast<int16_t *>(tp); tp += (emit.size() * 2 + 255) / 256 * 256;
  uint8_t *d_ssvsc = reinterpret_cast<uint8_t *>(tp); tp += (ssv_scores.size() + 255) / 256 * 256;
  float *d_bgf = reinterpret_cast<float *>(tp);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_emit, emit.data(), emit.size() * 2, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_ssvsc, ssv_scores.data(), ssv_scores.size(), hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_bgf, kAminoBg, 20 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors above go out of scope with this call only, but be explicit

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
      nres_c += 2 * (int64_t)n;
      max_orfs_c += 6 * (int64_t)((n / 3 + 1) / (prm->min_orf_len + 1) + 1);
    }
    dna->cache_minlen = prm->min_orf_len; dna->cache_nres = nres_c; dna->cache_max_orfs = max_orfs_c;
  }
  const int64_t nres = dna->cache_nres, max_orfs = dna->cache_max_orfs;
  if (max_orfs >= (int64_t)INT32_MAX) { ctx->set_error("DNA block too large for one pipeline call (ORF list indices are 32-bit): split it"); return BATH_EINVAL; }
  int64_t cap = std::max<int64_t>(4096, (int64_t)((double)max_orfs * std::min(1.0, prm->F1 * 4.0 + 0.01)));
  cap = std::min<int64_t>(cap, std::max<int64_t>(max_orfs, 4096));

  // ---- translation / ORF work-list buffers (bath_orfs.hip)
  if ((st = orf_tiles_ensure(ctx, dna)) != BATH_OK) return st;
  const size_t nent = (size_t)dna->ntiles * 6;
  DevBuf &b_aa = ctx->scratch[24], &b_slots = ctx->scratch[25], &b_orfs = ctx->scratch[26], &b_misc = ctx->scratch[27];
  BATH_HIP_TRY(ctx, b_aa.reserve(orf_aa_bytes(dna)));
  BATH_HIP_TRY(ctx, b_slots.reserve((nent * (size_t)orf_slot_cap(prm->min_orf_len) + 64) * 8));
  BATH_HIP_TRY(ctx, b_orfs.reserve((size_t)(max_orfs + 64) * sizeof(OrfRec)));
  BATH_HIP_TRY(ctx, b_misc.reserve((3 * nent + 2 * kOrfBins + 64) * sizeof(int32_t)));
  OrfBuffers ob{};
  ob.aa = b_aa.as<uint8_t>(); ob.slots = b_slots.p; ob.sorted = b_orfs.as<OrfRec>();
  ob.cnt = b_misc.as<int32_t>(); ob.prefix = ob.cnt + nent; ob.suffix = ob.prefix + nent;
  ob.hist = reinterpret_cast<int *>(ob.suffix + nent); ob.cursor = ob.hist + kOrfBins; ob.ntotal = ob.cursor + kOrfBins;

  const int NRk = om->NR;
  const size_t ssv_shmem = (size_t)kSsvRows * om->ssv_row_bytes;
  if (ssv_shmem > 160 * 1024) { ctx->set_error("model too long for the LDS-resident SSV cost table"); return BATH_EINVAL; }
  const int dec_blocks = ctx->prop.multiProcessorCount * 4;

  std::vector<hipEvent_t> &ev = ctx->ev_pool;
  while (ev.size() < 12) { hipEvent_t e; BATH_HIP_TRY(ctx, hipEventCreate(&e)); ev.push_back(e); }

  PipelineWork W;
  Counters hc{};
  for (int attempt = 0;; attempt++) {
    size_t need = layout(W, nullptr, (int)cap);
    BATH_HIP_TRY(ctx, b_work.reserve(need + 4096));
    layout(W, b_work.as<char>(), (int)cap);
    W.pool = b_aa.as<uint8_t>();
    BATH_HIP_TRY(ctx, hipMemsetAsync(W.ctr, 0, sizeof(Counters), ctx->stream));
    SeqView cv{W.pool, W.cand.off, W.cand.len, cap};

    int e = 0;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 1. six-frame translation, ORFs, length-sorted work list
    if ((st = launch_orf_scan(ctx, dna, tt, prm->min_orf_len, ob, &W.ctr->n_orfs, &W.ctr->orf_res)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 2. SSV + F1 threshold, lane per ORF
    {
      const int blocks = ctx->prop.multiProcessorCount * 4;
      bool launched = false;
#define BATH_ORF_CASE(N, GG)                                                                                                     \
  if (!launched && NRk == N && om->G == GG) {                                                                                    \
    if (ssv_shmem > 64 * 1024) (void)hipFuncSetAttribute((const void *)ssv_orf_kernel<N, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssv_shmem); \
    hipLaunchKernelGGL((ssv_orf_kernel<N, GG>), dim3(blocks), dim3(256), ssv_shmem, ctx->stream, W.pool, ob.sorted, ob.ntotal,                  \
                       dna->view(), om->d_ssv, om->ssv_row_bytes, d_emit, max_orf, W.cand, W.cand_cap, W.ctr);                   \
    launched = true;                                                                                                             \
  }
      BATH_SSV_SHAPES(BATH_ORF_CASE)
#undef BATH_ORF_CASE
      if (!launched) { ctx->set_error("SSV kernel: no tile shape for this model length"); return BATH_EINVAL; }
      BATH_HIP_TRY(ctx, hipGetLastError());
    }
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 3. SSV status; full MSV for the undecided
    hipLaunchKernelGGL(classify_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, om->lt.d_tjb, mc, W.todo_msv);
    BATH_HIP_TRY(ctx, hipGetLastError());
    if ((st = launch_msv_wave(ctx, om, cv, W.todo_msv, cap, W.cand.usc, W.cand.msv_status, &W.ctr->todo_msv)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 3. F1 on the MSV score, bias filter
    hipLaunchKernelGGL(f1_bias_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, P, W.pool, M, om->d_bias_eo,
                       om->lt.d_nullsc, om->lt.d_p1, om->lt.d_lt1, om->lt.d_lt2, W.todo_vit, W.todo_ssvb);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 4. Viterbi filter with windows (P > F2) / SSV windows (P <= F2)
    wa.d_filtersc = W.cand.filtersc; wa.d_ssv_scores = d_ssvsc; wa.d_wins = W.wins; wa.d_win_count = &W.ctr->win_count; wa.win_cap = W.win_cap;
    wa.d_kminmax = W.cand.kminmax;
    if (vit_lane_supported(om)) {       // lane per ORF, ORFs bucketed by length (bath_viterbi.hip)
      if ((st = launch_len_sort(ctx, W.todo_vit, &W.ctr->todo_vit, W.cand.len, W.len_bins, W.todo_sorted)) != BATH_OK) return st;
      if ((st = launch_vit_lane(ctx, om, cv, W.todo_sorted, cap, &W.ctr->todo_vit, W.cand.vfsc, W.cand.vit_status, &wa)) != BATH_OK) return st;
    } else if ((st = launch_vit_wave(ctx, om, cv, W.todo_vit, cap, W.cand.vfsc, W.cand.vit_status, &wa, &W.ctr->todo_vit)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    {
      const int Cc = (M + 63) / 64;
      int Cs = -1;
      for (int opt : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 52}) if (Cc <= opt) { Cs = opt; break; }
#define BATH_SSVB_CASE(N)                                                                                                          \
  case N:                                                                                                                          \
    hipLaunchKernelGGL(ssv_bath_kernel<N>, dim3(wave_grid_blocks(ctx) / 4), dim3(256), 0, ctx->stream, W.cand, W.ctr, W.todo_ssvb, W.pool, M, \
                       om->d_rb, om->rb_stride, d_ssvsc, om->lt.d_tjb, om->lt.d_nullsc, mc, invP_f1, W.wins, W.win_cap, W.ctr);    \
    break;
      switch (Cs) {
        BATH_SSVB_CASE(1) BATH_SSVB_CASE(2) BATH_SSVB_CASE(3) BATH_SSVB_CASE(4) BATH_SSVB_CASE(6) BATH_SSVB_CASE(8)
        BATH_SSVB_CASE(12) BATH_SSVB_CASE(16) BATH_SSVB_CASE(24) BATH_SSVB_CASE(32) BATH_SSVB_CASE(52)
        default: ctx->set_error("model too long for the SSV window kernel"); return BATH_EINVAL;
      }
#undef BATH_SSVB_CASE
    }
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 5. F2, local composition re-filter, optional plain Viterbi re-run
    hipLaunchKernelGGL(post_vit_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.cand_cap, W.ctr, P, W.pool, M, d_ssvsc, (int)om->base_b,
                       om->scale_b, d_bgf, om->lt.d_p1, om->lt.d_lt1, om->lt.d_lt2, W.todo_vit2, W.todo_fwd);
    BATH_HIP_TRY(ctx, hipGetLastError());
    if ((st = launch_vit_wave(ctx, om, cv, W.todo_vit2, cap, W.cand.vfsc, W.cand.vit_status, nullptr, &W.ctr->todo_vit2)) != BATH_OK) return st;
    hipLaunchKernelGGL(post_vit2_kernel, dim3(64), dim3(256), 0, ctx->stream, W.cand, W.ctr, P, W.todo_vit2, W.todo_fwd);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));
    // 6. Forward parser, F3/F4
    if ((st = launch_fwd_wave(ctx, om, cv, W.todo_fwd, cap, W.cand.fwdsc, W.cand.fwd_status, &W.ctr->todo_fwd)) != BATH_OK) return st;
    hipLaunchKernelGGL(final_kernel, dim3(dec_blocks), dim3(256), 0, ctx->stream, W.cand, W.ctr, P, W.todo_fwd);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipEventRecord(ev[e++], ctx->stream));

    BATH_HIP_TRY(ctx, hipMemcpyAsync(&hc, W.ctr, sizeof(Counters), hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hc.overflow || hc.cand_count > cap) {
      if (attempt >= 4) { ctx->set_error("candidate buffers overflowed repeatedly"); return BATH_EMEM; }
      cap = std::max<int64_t>(cap * 2, (int64_t)hc.cand_count + 1024);
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
    stats->cells_msv = (int64_t)hc.orf_res * M; stats->cells_vit = (int64_t)hc.res_vit * M; stats->cells_fwd = (int64_t)hc.res_fwd * M;
  }

  if (results) {
    const int nc = hc.cand_count;
    std::vector<int64_t> h_window(nc); std::vector<int32_t> h_sf(nc), h_startj(nc), h_len(nc), h_ms(nc), h_vs(nc), h_stage(nc);
    std::vector<float> h_usc(nc), h_null(nc), h_fsc(nc), h_vf(nc), h_fw(nc); std::vector<double> h_P(nc);
    auto pull = [&](void *dst, const void *src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream); };
    if (nc > 0) {
      BATH_HIP_TRY(ctx, pull(h_window.data(), W.cand.window, nc * 8)); BATH_HIP_TRY(ctx, pull(h_sf.data(), W.cand.sf, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_startj.data(), W.cand.startj, nc * 4)); BATH_HIP_TRY(ctx, pull(h_len.data(), W.cand.len, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_ms.data(), W.cand.msv_status, nc * 4)); BATH_HIP_TRY(ctx, pull(h_vs.data(), W.cand.vit_status, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_stage.data(), W.cand.stage, nc * 4)); BATH_HIP_TRY(ctx, pull(h_usc.data(), W.cand.usc, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_null.data(), W.cand.nullsc, nc * 4)); BATH_HIP_TRY(ctx, pull(h_fsc.data(), W.cand.filtersc, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_vf.data(), W.cand.vfsc, nc * 4)); BATH_HIP_TRY(ctx, pull(h_fw.data(), W.cand.fwdsc, nc * 4));
      BATH_HIP_TRY(ctx, pull(h_P.data(), W.cand.P, nc * 8));
      BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->results.clear();
    for (int c = 0; c < nc; c++) {
      if (h_stage[c] < 1) continue;
      bath_orf_result r{};
      r.window = h_window[c]; r.strand = h_sf[c] / 3; r.frame = h_sf[c] % 3;
      r.start = r.frame + 3 * h_startj[c] + 1; r.end = r.start + 3 * h_len[c] - 1; r.n = h_len[c];
      r.stage = (h_stage[c] == 5) ? 2 : h_stage[c];
      r.msv_status = h_ms[c]; r.vit_status = h_vs[c];
      r.usc = h_usc[c]; r.nullsc = h_null[c]; r.filtersc = h_fsc[c]; r.vfsc = h_vf[c]; r.fwdsc = h_fw[c]; r.P = h_P[c];
      ctx->results.push_back(r);
    }
    std::sort(ctx->results.begin(), ctx->results.end(), [](const bath_orf_result &a, const bath_orf_result &b) {
      if (a.window != b.window) return a.window < b.window;
      if (a.strand != b.strand) return a.strand < b.strand;
      if (a.frame != b.frame) return a.frame < b.frame;
      return a.start < b.start;
    });
    *results = ctx->results.data();
    if (n_results) *n_results = (int64_t)ctx->results.size();
  } else if (n_results) *n_results = (int64_t)hc.n_past_msv;
  return BATH_OK;
}
