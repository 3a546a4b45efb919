// bath_msv_lane.hip -- p7_MSVFilter's full recurrence (J state) with ONE LANE per target.
//
//   msv_lane_kernel <- p7_MSVFilter, the path after SSV returns eslENORESULT   impl_sse/msvfilter.c:106-207
//
// Every ORF that passes the F1 threshold scores above the point where p7_SSVFilter can vouch for its answer (xJ > base,
// ssvfilter.c:916), so ALL of the cascade's MSV survivors -- 576 k ORFs per 10^6 windows on the bench block -- are re-scored with
// the J state.  Rounds 1-3 did that with a wave per target (msv_wave_kernel: lanes own nodes, the row maximum is a wave
// reduction, 41 rows x ~0.6 us per target): 1.5 ms of an 10.9 ms step for 2 % of the cascade's cells.  But the MSV row has no
// dependency along the model either -- cell k reads cell k-1 of the PREVIOUS row; xE, xJ, xB are per-row scalars of the target --
// so the SSV kernel's layout applies: a lane keeps the whole row in registers, two cells per register (register r: nodes r+1 and
// NR+r+1, the diagonal shift folded into the in-place descending update), costs from an LDS table by residue, no cross-lane
// traffic.  Same number system (bath_kernels.hpp): a byte b of the reference is the binary16 number b * 2^-11, exact, and
// v_pk_add_f16 ... clamp saturates below at 0 like subs_epu8.  Per register and row: v_pk_max_f16 (the cell to the left or B),
// v_pk_add_f16 clamp (+ bias - cost), half a v_pk_maximum3_f16 (the row maximum).
// The reference saturates twice per cell, adds_epu8(sv, bias) at 255 and subs_epu8(., rsc) at 0 (msvfilter.c:160-162).  The upper
// saturation cannot act before the row's overflow test fires: every cell and B of the previous row satisfy v + bias < 255
// (that test, :172; base + bias < 255 by construction of the byte scores), so max(prev, B) + bias < 255 and one add of
// (bias - cost) with the lower clamp is the reference's arithmetic; a target whose row maximum reaches 255 - bias reports
// eslERANGE (score +inf) whatever its cells hold afterwards.
#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

namespace bath {

__device__ __forceinline__ s16x2 msv_pkmax(s16x2 a, s16x2 b) {
  s16x2 r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int NR>
__global__ __launch_bounds__(256, NR <= 76 ? 4 : 1) void msv_lane_kernel(SeqView sq, const int16_t *__restrict__ cost_tab /* the MSV increment table */, int row_bytes,
                                                                        const uint8_t *__restrict__ tjb_tab, MsvConsts c,
                                                                        const int32_t *__restrict__ todo, int64_t ntodo, const int *__restrict__ ntodo_dev,
                                                                        float *__restrict__ sc, int32_t *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  {
    const int n32 = kSsvRows * row_bytes / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(cost_tab);
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (int i = threadIdx.x; i < n32; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  if (ntodo_dev) ntodo = *ntodo_dev;
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w * 64 < ntodo; w += nwaves) {
    const int64_t job = w * 64 + lane;
    const bool live = job < ntodo;
    const int64_t sid = live ? (todo ? (int64_t)todo[job] : job) : 0;
    const int L = live ? sq.len[sid] : 0;
    const uint8_t *s = sq.data + sq.off[sid];
    const int Lw = wave_max_i32(L);
    const int tjb = tjb_tab[L];
    const int tjbm = (uint8_t)((int8_t)tjb + (int8_t)c.tbm);
    s16x2 reg[NR];
    const s16x2 zero = {0, 0};
#pragma unroll
    for (int r = 0; r < NR; r++) reg[r] = zero;
    int xJ = 0, xB = satu8(c.base - tjbm);
    bool overflow = false;
    uint2 res = make_uint2(0u, 0u);
    for (int i = 0; i < Lw; i++) {
      // residues eight at a time while at least eight remain (the candidates' residues sit in the amino-acid streams: any alignment)
      if ((i & 7) == 0) {
        if (i + 8 <= L) __builtin_memcpy(&res, s + i, 8);
        else {
          res = make_uint2(0u, 0u);
          for (int j = 0; j < 8 && i + j < L; j++) {
            const unsigned b = s[i + j];
            if (j < 4) res.x |= b << (8 * j); else res.y |= b << (8 * (j - 4));
          }
        }
      }
      const unsigned byte = (((i & 4) ? res.y : res.x) >> (8 * (i & 3))) & 0xffu;
      const int x = (i < L) ? min((int)byte, kKp - 1) : kRowReset;       // past the target's end: the reset row (every cell back to 0, xE = 0: xJ and B keep their values)
      const ssv_i4 *row = reinterpret_cast<const ssv_i4 *>(lds + (size_t)x * row_bytes);
      const h16x2 bh = {(_Float16)((float)xB * (1.0f / 2048.0f)), (_Float16)((float)xB * (1.0f / 2048.0f))};
      const s16x2 xBv = __builtin_bit_cast(s16x2, bh);
      // register 0: its low half takes node 0 (dp[0] stays 0: max(0, B) = B), its high half the old low half of register NR-1
      const s16x2 wrap = __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbit(__builtin_bit_cast(unsigned, reg[NR - 1]), 0u, 16));
      s16x2 xE = zero, xE2 = zero;
#pragma unroll
      for (int g = NR / 4 - 1; g >= 0; g--) {                           // descending, in place: reg[r] <- f(reg[r-1] of the previous row)
        const ssv_i4 inc = row[g];
        const int r = 4 * g;
        const s16x2 v3 = ssv_add(msv_pkmax(reg[r + 2], xBv), inc.w);
        const s16x2 v2 = ssv_add(msv_pkmax(reg[r + 1], xBv), inc.z);
        const s16x2 v1 = ssv_add(msv_pkmax(reg[r], xBv), inc.y);
        const s16x2 v0 = ssv_add(msv_pkmax((r > 0) ? reg[r - 1] : wrap, xBv), inc.x);
        reg[r + 3] = v3; reg[r + 2] = v2; reg[r + 1] = v1; reg[r] = v0;
        xE = ssv_max3(xE, v3, v2);
        xE2 = ssv_max3(xE2, v1, v0);
      }
      const h16x2 e1 = __builtin_bit_cast(h16x2, xE), e2 = __builtin_bit_cast(h16x2, xE2);
      int xEi = (int)(fmaxf(fmaxf((float)e1.x, (float)e1.y), fmaxf((float)e2.x, (float)e2.y)) * 2048.0f);
      if (xEi + c.bias >= 255) overflow = true;                          // msvfilter.c:172-178: sticky, the score is +inf
      xEi = max(xEi - c.tec, 0);
      xJ = max(xJ, xEi);
      xB = max(max(c.base, xJ) - tjbm, 0);
    }
    if (live) {
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

// The lane-per-target MSV for models that fit one lane's tile of at most 76 registers (152 nodes); longer models and
// BATH_HIP_MSV_WAVE=1 keep the wave-per-target kernel.  Returns BATH_ENORESULT when this kernel does not apply.
int launch_msv_lane(bath_hip_ctx *ctx, const bath_hip_oprofile *om, SeqView v, const int32_t *d_todo, int64_t ntodo, float *d_sc, int32_t *d_status, const int *ntodo_dev) {
  static const bool off = [] { const char *e = std::getenv("BATH_HIP_MSV_WAVE"); return e && e[0] == '1'; }();
  if (off || om->G != 1 || om->NR > 76 || !om->d_msv) return BATH_ENORESULT;
  if (ntodo == 0) return BATH_OK;
  const size_t shmem = (size_t)kSsvRows * om->ssv_row_bytes;
  const int64_t waves = (ntodo + 63) / 64;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((waves + 3) / 4, (int64_t)ctx->prop.multiProcessorCount * 4));
  bool launched = false;
#define BATH_MSV_CASE(N)                                                                                                               \
  if (!launched && om->NR == N) {                                                                                                      \
    hipLaunchKernelGGL((msv_lane_kernel<N>), dim3(grid), dim3(256), shmem, ctx->stream, v, om->d_msv, om->ssv_row_bytes, om->lt.d_tjb,   \
                       MsvConsts{om->tbm_b, om->tec_b, om->base_b, om->bias_b, om->scale_b}, d_todo, ntodo, ntodo_dev, d_sc, d_status); \
    launched = true;                                                                                                                   \
  }
  BATH_MSV_CASE(16) BATH_MSV_CASE(20) BATH_MSV_CASE(24) BATH_MSV_CASE(28) BATH_MSV_CASE(32) BATH_MSV_CASE(36) BATH_MSV_CASE(40) BATH_MSV_CASE(44)
  BATH_MSV_CASE(48) BATH_MSV_CASE(52) BATH_MSV_CASE(56) BATH_MSV_CASE(60) BATH_MSV_CASE(64) BATH_MSV_CASE(68) BATH_MSV_CASE(72) BATH_MSV_CASE(76)
#undef BATH_MSV_CASE
  if (!launched) return BATH_ENORESULT;
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
