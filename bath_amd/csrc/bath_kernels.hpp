// bath_kernels.hpp -- device helpers shared by the kernels (gfx950, wave64).
#pragma once
#include <climits>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "bath_hip.h"

namespace bath {

typedef short s16x2 __attribute__((ext_vector_type(2)));

// Wave maximum / minimum by DPP: a running maximum in lane order (row_shr 1/2/4/8, row_bcast 15/31: six v_max_i32_dpp, the
// identity as the value a lane without a source sees), the last lane broadcast with v_readlane.  The xor butterfly through
// ds_bpermute this replaces cost a trip through the LDS crossbar per step in the middle of the callers' dependent chains.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
// lane l <- lane l-1, lane 0 <- fill (wave_shr:1)
__device__ __forceinline__ int wave_shr1_i32(int v, int fill) { return dpp_i<0x138>(v, fill); }
__device__ __forceinline__ float wave_shr1_f32(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max_f32(float v) {                     // a maximum is exact in any order
  const int ninf = (int)0xff800000;
#define BATH_FMAX_STEP(CTRL, MASK) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v), CTRL, MASK, 0xf, false)));
  BATH_FMAX_STEP(0x111, 0xf) BATH_FMAX_STEP(0x112, 0xf) BATH_FMAX_STEP(0x114, 0xf) BATH_FMAX_STEP(0x118, 0xf) BATH_FMAX_STEP(0x142, 0xa) BATH_FMAX_STEP(0x143, 0xc)
#undef BATH_FMAX_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_i<0x111>(v, INT_MIN)); v = max(v, dpp_i<0x112>(v, INT_MIN)); v = max(v, dpp_i<0x114>(v, INT_MIN));
  v = max(v, dpp_i<0x118>(v, INT_MIN)); v = max(v, dpp_i<0x142, 0xa>(v, INT_MIN)); v = max(v, dpp_i<0x143, 0xc>(v, INT_MIN));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, dpp_i<0x111>(v, INT_MAX)); v = min(v, dpp_i<0x112>(v, INT_MAX)); v = min(v, dpp_i<0x114>(v, INT_MAX));
  v = min(v, dpp_i<0x118>(v, INT_MAX)); v = min(v, dpp_i<0x142, 0xa>(v, INT_MAX)); v = min(v, dpp_i<0x143, 0xc>(v, INT_MAX));
  return __builtin_amdgcn_readlane(v, 63);
}
// Wave sum: a running sum in lane order by DPP, the last lane broadcast.  (Another association of the 64 terms than the xor
// butterfly it replaces: a float sum may differ in its last bit.)
__device__ __forceinline__ float wave_sum_f32(float v) {
#ifdef BATH_SUM_BPERMUTE
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
#else
#define BATH_FSUM_STEP(CTRL, MASK) v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, MASK, 0xf, false));
  BATH_FSUM_STEP(0x111, 0xf) BATH_FSUM_STEP(0x112, 0xf) BATH_FSUM_STEP(0x114, 0xf) BATH_FSUM_STEP(0x118, 0xf) BATH_FSUM_STEP(0x142, 0xa) BATH_FSUM_STEP(0x143, 0xc)
#undef BATH_FSUM_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
__device__ __forceinline__ int sat16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int satu8(int v) { return min(max(v, 0), 255); }


// ---- SSV: one DP row of the lane-per-target kernels ----------------------------------------------
// Number system: the reference keeps each diagonal in bytes above a begin score with saturating subtraction
// (ssvfilter.c:130-141).  Here a cell is a binary16 number d * 2^-11, d = the score's distance above the begin score
// (d = 0: the begin score).  Every d in 0..2048 and every cost in -2047..2047 is exact in binary16 at that scale, sums
// below 1.0 are exact, and v_pk_add_f16 with the clamp modifier clamps to [0, 1]: the lower clamp IS the reference's
// "max(prev - cost, begin)", the upper one sits at d = 2048, far above where the reference reports overflow (d >= 127 - bias).
// Why floating point for integer scores: gfx950 has a three-operand packed maximum (v_pk_maximum3_f16) and no integer
// counterpart, so the running maximum costs one VALU op per FOUR cells instead of one per two:
// 1.5 VALU ops per 2 cells (v_pk_add_f16 clamp per pair + half a v_pk_maximum3_f16).
constexpr unsigned kSsvBeginPair = 0x00000000u;          // two cells at the begin score

__device__ __forceinline__ s16x2 ssv_add(s16x2 a, int inc) {
  s16x2 r;
  asm("v_pk_add_f16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(inc));
  return r;
}
__device__ __forceinline__ s16x2 ssv_max3(s16x2 a, s16x2 b, s16x2 c) {
  s16x2 r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// One DP row for this lane: residue row <rowbase> (LDS byte address of the residue's increment row, already
// offset to this lane's column tile).  Register r of a tile holds the tile's nodes r+1 (low half) and
// NR+r+1 (high half), so "the previous row's value of the node to the left" is simply the previous
// register: the diagonal shift is folded into the add's destination (registers are updated in
// place, descending).  Only register 0 needs assembling: its low half takes the node left of the tile
// (high half of <carry>: the begin score for the first tile, otherwise the last node of the neighbouring
// lane's tile when a model is split over G lanes), its high half the old low half of register NR-1.
// The LDS reads are software-pipelined by hand: NB 16-byte buffers rotate (increments of registers r .. r+3: one ds_read_b128;
// rows are 16-byte aligned, pitch/16 odd), the read of group g+NB is issued as soon as group g has been consumed, so NB-1 or NB
// reads are always in flight behind the adds (the compiler's own schedule under the
// 128-VGPR cap of 4 waves per SIMD drains the LDS queue after every pair of reads).  The reads and their waits are inline asm:
// "s_waitcnt lgkmcnt(1)" relies on LDS reads of a wave completing in order -- with at most one younger read outstanding, the older
// one has landed (a scalar load slipped in between only makes the wait longer).  The wait is tied to the buffer it guards by a
// "+v" operand, so its consumers cannot be scheduled above it.
typedef int ssv_i4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void ssv_lgkm_wait(ssv_i4 &c) {       // wait until at most N LDS reads are outstanding
  if (N <= 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c));
  else if (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(c));
  else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(c));
}
template <int NR, int NB, int g, int US = 16>
struct SsvRowPipe {
  static constexpr int NG = NR / 4;
  static __device__ __forceinline__ void run(s16x2 (&reg)[NR], s16x2 &xE, s16x2 &xE2, unsigned rowaddr, s16x2 wrap, ssv_i4 (&b)[NB]) {
    constexpr int r = NR - 4 - 4 * g;
    ssv_i4 &c = b[g % NB];
    ssv_lgkm_wait<(NG - 1 - g < NB - 1) ? NG - 1 - g : NB - 1>(c);
    const s16x2 v3 = ssv_add(reg[r + 2], c.w);
    const s16x2 v2 = ssv_add(reg[r + 1], c.z);
    reg[r + 3] = v3;
    reg[r + 2] = v2;
    xE = ssv_max3(xE, v3, v2);
    const s16x2 v1 = ssv_add(reg[r], c.y);
    const s16x2 v0 = ssv_add((r > 0) ? reg[r - 1] : wrap, c.x);
    reg[r + 1] = v1;
    reg[r] = v0;
    xE2 = ssv_max3(xE2, v1, v0);
    if (g + NB < NG) {
      // the buffer's last use is above: the asm inputs v0, v1 make the read depend on those adds, which keeps it below them
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c) : "v"(rowaddr), "n"(US * ((NR - 4 - 4 * (g + NB)) / 4)), "v"(v0), "v"(v1));
    }
    SsvRowPipe<NR, NB, g + 1, US>::run(reg, xE, xE2, rowaddr, wrap, b);
  }
};
template <int NR, int NB, int US>
struct SsvRowPipe<NR, NB, NR / 4, US> {
  static __device__ __forceinline__ void run(s16x2 (&)[NR], s16x2 &, s16x2 &, unsigned, s16x2, ssv_i4 (&)[NB]) {}
};
// US: bytes between the 16-byte units of consecutive register groups in the row (16: a plain row; 256: the replicated table)
template <int NR, int NB = 2, int US = 16>
__device__ __forceinline__ void ssv_row_pipe(s16x2 (&reg)[NR], s16x2 &xE, s16x2 &xE2, unsigned rowaddr /* LDS byte address */, unsigned carry) {
  static_assert(NR % 4 == 0 && NR >= 4 * NB && NB >= 2 && NB <= 3, "registers are consumed four at a time (one 16-byte LDS read)");
  const s16x2 wrap = __builtin_bit_cast(s16x2, __builtin_amdgcn_alignbit(__builtin_bit_cast(unsigned, reg[NR - 1]), carry, 16));
  ssv_i4 b[NB];
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[0]) : "v"(rowaddr), "n"(US * ((NR - 4) / 4)));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[1]) : "v"(rowaddr), "n"(US * ((NR - 8) / 4)));
  if (NB > 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[NB - 1]) : "v"(rowaddr), "n"(US * ((NR - 12) / 4)));
  SsvRowPipe<NR, NB, 0, US>::run(reg, xE, xE2, rowaddr, wrap, b);
}

// Per-row helpers for models split over G lanes per target.  The lanes of a group are 64/G apart (lane = grank * (64/G) + slot),
// so that the 16 lanes of an LDS cycle are 16 DIFFERENT targets reading the same tile segment of their residue rows -- the access
// pattern of G = 1.  With adjacent lanes (rounds 1-2) a group read 32-128 consecutive bytes of one row and the groups of a cycle
// collided on their banks whenever their residues differed: <128, 4> (M = 1024) 10.9 -> 6.7 ms per 125 Mb, 0.39 -> 0.63 of the
// packed-issue ceiling; <128, 2> (M = 459) 5.2 -> 4.7 ms per 200 k windows.
template <int G> struct SsvGroups {
  static constexpr int TPW = 64 / G;                       // targets per wave
  static __device__ __forceinline__ int rank(int lane) { return lane / TPW; }      // which tile of its target a lane holds
  static __device__ __forceinline__ int slot(int lane) { return lane % TPW; }      // the target's slot in the wave
};
template <int NR, int G>
__device__ __forceinline__ unsigned ssv_carry(const s16x2 (&reg)[NR], int grank) {
  if (G == 1) return kSsvBeginPair;
  const unsigned far = (unsigned)__shfl_up((int)__builtin_bit_cast(unsigned, reg[NR - 1]), SsvGroups<G>::TPW, 64);     // needed last in the row: its latency hides
  return (grank == 0) ? kSsvBeginPair : far;
}
// Maximum over the target's lanes, converted to the reference's signed-byte domain (begin score = -128).
template <int G>
__device__ __forceinline__ int ssv_group_max(s16x2 xE) {
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  const h16x2 h = __builtin_bit_cast(h16x2, xE);
  int v = (int)(fmaxf((float)h.x, (float)h.y) * 2048.0f) - 128;            // distance above the begin score, then begin = -128
#pragma unroll
  for (int d = SsvGroups<G>::TPW; d < 64; d <<= 1) v = max(v, __shfl_xor(v, d, 64));      // the target's lanes are 64/G apart
  return v;
}

// p7_SSVFilter's decision logic (ssvfilter.c:876-925) applied to the raw maximum.
struct MsvConsts {
  int tbm, tec, base, bias;
  float scale_b;
};

__device__ __forceinline__ int ssv_classify(int v, int tjb, const MsvConsts &c, float *sc) {
  if (tjb + c.tbm + c.tec + c.bias >= 127) return BATH_ENORESULT;
  unsigned xE = (v >= -1 - c.bias) ? 255u : (unsigned)(v + 256);
  if ((int)xE >= 255 - c.bias) {
    *sc = INFINITY;
    return (c.base - tjb - c.tbm < 128) ? BATH_ENORESULT : BATH_ERANGE;
  }
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

struct WindowRec { int32_t cand, n, k, length; float score; };   // P7_HMM_WINDOW fields used on the path (hmmer.h:998)

// esl_hmm_Forward on the 2-state bias-filter HMM (p7_bg_FilterScore, p7_bg.c:491): per-row max scaling,
// float accumulation in the reference's order; t[0][*] follow p7_bg_SetLength (p7_bg.c:193).
__device__ __forceinline__ float bias_forward(const uint8_t *s, int L, int M, const float *e /* [Kp][2] */, float p1) {
  const float L1 = (float)M / 8.0f;
  const float t00 = p1, t01 = 1.0f - p1, t10 = 1.0f / (L1 + 1.0f), t11 = L1 / (L1 + 1.0f);
  float logsc = 0.0f;
  if (L <= 0) return logsc;
  int x = min((int)s[0], 28);
  float d0 = e[2 * x] * 0.999f, d1 = e[2 * x + 1] * 0.001f;
  float mx = fmaxf(fmaxf(d0, 0.0f), d1);
  d0 /= mx; d1 /= mx;
  logsc += (float)log((double)mx);
  // Residues four at a time with the next four in flight: taken a byte per step, every residue is a trip to L2 (the byte, then the
  // emission pair it selects) in front of the step's arithmetic, ~0.4 us per residue with nothing to hide it behind.  The caller
  // keeps <e> in LDS where it can.  Only whole dwords inside the sequence are read (its last 1-3 residues go byte by byte).
  auto step = [&](int xx) {
    float n0 = 0.0f + d0 * t00; n0 = n0 + d1 * t10; n0 *= e[2 * xx];
    float n1 = 0.0f + d0 * t01; n1 = n1 + d1 * t11; n1 *= e[2 * xx + 1];
    mx = fmaxf(fmaxf(n0, 0.0f), n1);
    d0 = n0 / mx; d1 = n1 / mx;
    logsc += (float)log((double)mx);
  };
  int i = 1;
  uint32_t wnext = (i + 4 <= L) ? *reinterpret_cast<const uint32_t *>(s + i) : 0u;
  for (; i + 4 <= L; i += 4) {
    const uint32_t w = wnext;
    if (i + 8 <= L) wnext = *reinterpret_cast<const uint32_t *>(s + i + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) step(min((int)((w >> (8 * j)) & 0xffu), 28));
  }
  for (; i < L; i++) step(min((int)s[i], 28));
  float end = 0.0f + d0 * 1.0f; end = end + d1 * 1.0f;
  logsc += (float)log((double)end);
  return logsc;
}

}  // namespace bath
