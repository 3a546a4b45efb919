// bath_orfs.hip -- six-frame translation and ORF finding for a block of DNA windows, and the length-sorted
// ORF work list that feeds the lane-per-ORF SSV kernel.
//
// Reference behaviour (easel esl_gencode_ProcessStart/ProcessPiece/ProcessEnd as driven by
// src/bathsearch.c:384-392 with do_watson/do_crick, minimum_length = 20, no initiator requirement): every
// maximal run of non-stop codons of at least <minlen> residues in each of the six frames of a window is one
// ORF; windows shorter than 15 nt are skipped (bathsearch.c:1066).
//
// GPU formulation.  A window is cut into tiles of 384 nt; a half wave (32 lanes x 12 nt) owns a tile, so the work
// is parallel over the total sequence length whatever the shape of the block (10^6 windows of 1 kb, or a few
// windows of 256 kb), loads are contiguous across lanes and every stream's residues leave as contiguous dwords.
//   orf_tile_kernel   The three bytes at position p are codon p/3 of forward frame p%3 and, on the other strand, a codon
//                     of reverse frame (n-p)%3 (runs of non-stop codons are the same whichever way a frame is walked,
//                     so reverse frames are scanned in memory order too).  Canonical ACGT codons are decoded by one
//                     24-bit multiply + bit-field extract + a 64-entry LDS table per strand, degenerate codes by the
//                     general 18^3 table.  Per stream, the length of the stop-free run entering each lane's four codons
//                     comes from a DPP running maximum over the half wave (the last stop before the lane); runs closed by two stops inside the tile are recorded as ORFs
//                     at once, the run touching the tile's left edge ("prefix") and the one open at its right edge
//                     ("suffix") are left in the tile summary.  Positions past the end of a stream count as stops.
//   orf_stitch_kernel one lane per (window, frame) walks the tile summaries and records the ORFs that cross tiles.
//   orf_scan_bins     turns the ORF length histogram into start offsets, longest ORFs first.
//   orf_sort_kernel   compacts the per-tile records into the dense work list ordered by length (block-local counting
//                     sort: one global atomic per block and non-empty length bin).
// Amino stream of (w, sf):  aa + 2*off[w] + 96*w + sf*pitch(n),  pitch(n) = (n/3 + 16) & ~15   (closed form: no prefix sums)
#include <algorithm>
#include <climits>
#include <cstring>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

namespace bath {

constexpr int kTileLanes = 32;
constexpr int kTileCodons = 4 * kTileLanes;    // scan positions per stream and tile
constexpr int kTileNt = 12 * kTileLanes;

struct OrfScanTables {
  const uint8_t *aa_full;   // [5832] (a*18+b)*18+c -> amino code, degenerate codons resolved (esl_gencode_GetTranslation)
  const uint8_t *aa64_fwd;  // [64] a<<4|b<<2|c for canonical codons
  const uint8_t *aa64_rev;  // [64] x0<<4|x1<<2|x2 of the three bytes in memory order -> amino code of the reverse-complement codon
  const uint8_t *comp;      // [18]
};

struct OrfTiles {
  const int4 *desc;           // [ntiles] {offset of the window in the DNA block / 16, window length, window, tile index within the window}
  const int32_t *tile_first;  // [nwin] first tile of each window
  int64_t ntiles;
};

struct OrfScanOut {
  uint8_t *aa;              // amino-acid streams
  uint2 *slots;             // [ntiles][cap] {first codon index within its stream, length | (strand*3+frame) << 28}
  int32_t *cnt;             // [ntiles] ORFs recorded per tile
  uint2 *cross;             // [ntiles*6] the ORF that ends at the first stop of (tile, frame) and began in an earlier tile; length 0: none
  int32_t *prefix, *suffix; // [ntiles*6] tile summaries: stop-free run at the left / right edge (kTileCodons: no stop in the tile)
  int *hist;                // [kOrfBins] ORF length histogram (lengths above the last bin are clamped into it)
  unsigned long long *n_orfs, *orf_res;
  int cap;
  int count;                // 0: leave the histogram and the two counters alone (orf_filter_kernel recounts what it keeps)
};

// One strand only (pli->strands) and / or initiation codons (bathsearch -m / -M): applied to the records of the two scan kernels
// before the sort.  Off in bathsearch's default configuration (both strands, any codon starts an ORF), where the pass is skipped.
struct OrfFilter {
  int strands;              // BATH_STRAND_*
  int using_initiators;     // -m or -M: ORFs start at an initiation codon, translated as M
  const uint8_t *is_init;   // [64] canonical codon (16a + 4b + c) -> may initiate
  const uint8_t *comp;      // [18] complement
};

__device__ __forceinline__ int orf_bin(int len) { return min(len, kOrfBins - 1); }

__device__ __forceinline__ int half_shfl_up(int v, int d) { return __shfl_up(v, d, kTileLanes); }

// Exclusive running maximum over the lanes of a 32-lane half wave (lane 0 of a half gets <ident>), by DPP: one lane shift, then
// row_shr 1/2/4/8 inside the 16-lane rows and row_bcast:15 into the upper row of each half.  ~11 VALU instructions; the ballot
// form this replaces cost 7 per codon of the chunk.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_or_ident(int v, int ident) { return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int half_excl_max_scan(int v, int ident, bool first_lane_of_half) {
  int x = dpp_or_ident<0x138>(v, ident);                           // wave_shr:1
  x = first_lane_of_half ? ident : x;
  x = max(x, dpp_or_ident<0x111>(x, ident));                       // row_shr:1
  x = max(x, dpp_or_ident<0x112>(x, ident));                       // row_shr:2
  x = max(x, dpp_or_ident<0x114>(x, ident));                       // row_shr:4
  x = max(x, dpp_or_ident<0x118>(x, ident));                       // row_shr:8
  x = max(x, dpp_or_ident<0x142, 0xa>(x, ident));                  // row_bcast:15 into rows 1 and 3
  return x;
}

__global__ __launch_bounds__(256) void orf_tile_kernel(SeqView dna, OrfTiles tiles, OrfScanTables tabs, OrfScanOut out, int minlen) {
  __shared__ int s_hist[kOrfBins];
  __shared__ __attribute__((aligned(16))) uint8_t s_full[5832 + 8];
  __shared__ uint16_t s_fr[64];                                // canonical codon -> forward amino acid | reverse-strand amino acid << 8, bit 7 of each = stop
  __shared__ uint8_t s_comp[32];
  __shared__ unsigned s_red[2];
  __shared__ int s_cnt[8];                                     // per half wave: ORFs recorded in the current tile
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_hist[i] = 0;
  for (int i = threadIdx.x; i < 5832; i += blockDim.x) s_full[i] = tabs.aa_full[i];
  if (threadIdx.x < 64) {
    const unsigned f = tabs.aa64_fwd[threadIdx.x], r = tabs.aa64_rev[threadIdx.x];
    s_fr[threadIdx.x] = (uint16_t)((f | (f == kStop ? 0x80u : 0u)) | ((r | (r == kStop ? 0x80u : 0u)) << 8));
  }
  if (threadIdx.x < 18) s_comp[threadIdx.x] = tabs.comp[threadIdx.x];
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
  __syncthreads();
  const int hw = threadIdx.x / kTileLanes;                     // half wave within the block
  const int i = threadIdx.x % kTileLanes;                      // lane within the tile
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  unsigned my_orfs = 0, my_res = 0;
  // Software pipeline over this wave's tiles: descriptors are fetched two tiles ahead and the lane's 16 bytes of DNA one
  // tile ahead, so the dependent loads (descriptor -> sequence data) of a tile are in flight while the previous one computes.
  const int half = (threadIdx.x & 63) / kTileLanes;
  const int4 none = {0, 0, 0, 0};
  auto fetch_desc = [&](int64_t tb) { const int64_t t = tb + half; return t < tiles.ntiles ? tiles.desc[t] : none; };
  auto fetch_data = [&](const int4 &dsc, uint32_t (&D)[4]) {
    const int p0 = kTileNt * dsc.w + 12 * i;
    D[0] = D[1] = D[2] = D[3] = 0u;
    if (p0 < dsc.y) {
      const uint8_t *d = dna.data + ((int64_t)(uint32_t)dsc.x << 4) + p0;
#pragma unroll
      for (int k = 0; k < 4; k++) D[k] = *reinterpret_cast<const uint32_t *>(d + 4 * k);
    }
  };
  const int64_t tstep = nwaves * 2;
  int4 d_next = fetch_desc(wave0 * 2), d_next2 = fetch_desc(wave0 * 2 + tstep);
  uint32_t D_next[4];
  fetch_data(d_next, D_next);
  for (int64_t tb = wave0 * 2; tb < tiles.ntiles; tb += tstep) {
    const int64_t tile = tb + half;
    const int4 dsc = d_next;
    uint32_t D[4] = {D_next[0], D_next[1], D_next[2], D_next[3]};
    d_next = d_next2;
    d_next2 = fetch_desc(tb + 2 * tstep);
    fetch_data(d_next, D_next);
    const int n = dsc.y;                                       // n >= 15 for every window that has tiles; 0: no tile for this half wave
    const bool live = n > 0;
    const int w = dsc.z, T = dsc.w;
    const int64_t off = (int64_t)(uint32_t)dsc.x << 4;
    const int p0 = kTileNt * T + 12 * i;
    if (i == 0) s_cnt[hw] = 0;
    // ---- bytes past the window end read as 0 (they only feed positions that count as stops)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int rem = n - (p0 + 4 * k);
      if (rem < 4) D[k] = (rem <= 0) ? 0u : (D[k] & ((1u << (8 * rem)) - 1u));
    }
    const bool canonical = ((D[0] | D[1] | D[2] | (D[3] & 0xffffu)) & 0xfcfcfcfcu) == 0u;
    int af[12], ar[12];
    if (__all(canonical)) {
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const unsigned t = (k % 4 == 0) ? D[k / 4] : __builtin_amdgcn_alignbyte(D[k / 4 + 1], D[k / 4], k % 4);
        // bytes x0,x1,x2 < 4 at bits 0,8,16: t * (2^20 + 2^10 + 1) has x0<<4|x1<<2|x2 at bits 16..21 and nothing else there
        const unsigned idx = (__umul24(t, 0x100401u) >> 16) & 63u;
        const unsigned e = s_fr[idx];
        af[k] = (int)(e & 0xffu); ar[k] = (int)(e >> 8);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const unsigned t = (k % 4 == 0) ? D[k / 4] : __builtin_amdgcn_alignbyte(D[k / 4 + 1], D[k / 4], k % 4);
        const int x0 = min((int)(t & 0xffu), 17), x1 = min((int)((t >> 8) & 0xffu), 17), x2 = min((int)((t >> 16) & 0xffu), 17);
        const int f = s_full[(x0 * 18 + x1) * 18 + x2];
        const int r = s_full[((int)s_comp[x2] * 18 + (int)s_comp[x1]) * 18 + (int)s_comp[x0]];
        af[k] = f | (f == kStop ? 0x80 : 0); ar[k] = r | (r == kStop ? 0x80 : 0);
      }
    }
    const int nvalid = live ? min(max(n - 2 - p0, 0), 12) : 0;            // positions p0 .. p0+nvalid-1 start a codon inside the window
    const int pitch = orf_stream_pitch(n);
    uint8_t *const abase = out.aa + (2 * off + 96 * (int64_t)w);
    const int u0 = kTileCodons * T + 4 * i;                              // scan index of the lane's first codon, every stream
    // af[], ar[] carry a stop flag in bit 7.  A stream's four codons packed into a dword give its stops as the flag bits 7, 15,
    // 23, 31 with one AND; codons past the end of the stream count as stops: cv = how many of the four exist (x / 3 = x * 11 >> 5
    // for x < 15), the others' flags are forced.
    unsigned inv[3];
    int cv[3];
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
      cv[ph] = (max(nvalid - ph + 2, 0) * 11) >> 5;
      inv[ph] = cv[ph] >= 4 ? 0u : (0x80808080u << (8 * cv[ph]));
    }
    // n = 3 q3 + r3 once per tile: the reverse frame of phase ph is (r3 - ph) mod 3 and that stream has q3 - (r3 < ph) codons,
    // without a division per stream
    const int q3 = n / 3, r3 = n - 3 * q3;
    uint2 *const slots = out.slots + tile * out.cap;                     // per tile, not per stream
    int *const prefix6 = out.prefix + tile * 6, *const suffix6 = out.suffix + tile * 6;
#pragma unroll
    for (int st = 0; st < 6; st++) {                                      // st: phase 0..2 forward, 3..5 the same phases on the other strand
      const int ph = st % 3;
      const bool rev = st >= 3;
      const int fr = r3 - ph + (r3 < ph ? 3 : 0);                         // reverse frame of phase ph in this window: (n - ph) mod 3
      const int sf = rev ? 3 + fr : ph;                                   // strand*3 + frame as reported
      const int C = q3 - 1 - (r3 < ph ? 1 : 0);                           // (n - 3 - fr - ph) / 3; reverse stream: codon index j = C - u
      const int *a = rev ? ar : af;
      const unsigned pk = (unsigned)a[ph] | ((unsigned)a[ph + 3] << 8) | ((unsigned)a[ph + 6] << 16) | ((unsigned)a[ph + 9] << 24);
      const unsigned mf = (pk & 0x80808080u) | inv[ph];                   // stops, counting positions past the end of the stream
      // ---- residues out: one dword per lane when all four codons exist, bytes otherwise
      const int so = sf * pitch + (rev ? C - u0 - 3 : u0);                // 32-bit offset of the lane's dword in this tile's window
      if (cv[ph] == 4) {
        *reinterpret_cast<uint32_t *>(abase + so) = rev ? __builtin_amdgcn_perm(0u, pk & 0x7f7f7f7fu, 0x00010203u) /* bytes reversed */ : (pk & 0x7f7f7f7fu);
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) if (c < cv[ph]) abase[so + (rev ? 3 - c : c)] = (uint8_t)(a[ph + 3 * c] & 0x7f);
      }
      // ---- length of the stop-free run entering this lane's four codons: scan position of the last stop before them in the tile
      // (identity INT_MIN: the compiler then folds each step into one v_max_i32_dpp; with another identity it emits mov + dpp-mov + max)
      const int mine = mf ? 4 * i + ((31 - (int)__clz(mf)) >> 3) : INT_MIN;    // scan position of this lane's last stop
      const int last = half_excl_max_scan(mine, INT_MIN, i == 0);
      bool open = last < 0;                                               // no stop in the tile before this lane
      const int run_in = open ? 4 * i : 4 * i - 1 - last;
      auto record = [&](int u_stop, int len) {
        const int k = atomicAdd(&s_cnt[hw], 1);
        slots[k] = make_uint2((unsigned)(rev ? C - u_stop + 1 : u_stop - len), (unsigned)len | ((unsigned)sf << 28));
        atomicAdd(&s_hist[orf_bin(len)], 1);
        my_orfs++; my_res += (unsigned)len;
      };
      if (minlen > 2) {
        // a run that starts after a stop inside these four codons is at most 2 long: only the chunk's first stop can close an ORF
        if (mf != 0u) {
          const int first = (__ffs((int)mf) - 1) >> 3;
          const int len = run_in + first;
          if (open) { if (live) prefix6[sf] = len; }
          else if (live && len >= minlen) record(u0 + first, len);
        }
        if (live && i == kTileLanes - 1) {
          suffix6[sf] = mf ? (int)__clz(mf) >> 3 : run_in + 4;             // the run open at the tile's right edge
          if (open && mf == 0u) prefix6[sf] = kTileCodons;                 // no stop anywhere in the tile
        }
      } else {
        int len = run_in;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if ((mf >> (8 * c + 7)) & 1u) {
            if (open) { if (live) prefix6[sf] = len; open = false; }
            else if (live && len >= minlen) record(u0 + c, len);
            len = 0;
          } else len++;
        }
        if (live && i == kTileLanes - 1) {
          suffix6[sf] = len;
          if (open) prefix6[sf] = kTileCodons;
        }
      }
    }
    if (live && i == 0) out.cnt[tile] = s_cnt[hw];
  }
  if (my_orfs) { atomicAdd(&s_red[0], my_orfs); atomicAdd(&s_red[1], my_res); }
  __syncthreads();
  if (!out.count) return;
  for (int k = threadIdx.x; k < kOrfBins; k += blockDim.x) if (s_hist[k]) atomicAdd(&out.hist[k], s_hist[k]);
  if (threadIdx.x == 0 && s_red[0]) { atomicAdd(out.n_orfs, (unsigned long long)s_red[0]); atomicAdd(out.orf_res, (unsigned long long)s_red[1]); }
}

// ORFs that cross tile edges: one lane per (window, frame) walks the summaries of the window's tiles
__global__ __launch_bounds__(256) void orf_stitch_kernel(SeqView dna, OrfTiles tiles, OrfScanOut out, int minlen) {
  __shared__ int s_hist[kOrfBins];
  __shared__ unsigned s_red[2];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_hist[i] = 0;
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
  __syncthreads();
  unsigned my_orfs = 0, my_res = 0;
  const int64_t nstreams = dna.n * 6;
  for (int64_t sidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; sidx < nstreams; sidx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t w = sidx / 6;
    const int sf = (int)(sidx - w * 6);
    const int n = dna.len[w];
    if (n < 15) continue;
    const int nt = (n / 3) / kTileCodons + 1;
    const int64_t first = tiles.tile_first[w];
    const bool rev = sf >= 3;
    const int fr = rev ? sf - 3 : 0;
    const int ph = rev ? (n - fr) % 3 : sf;
    const int C = (n - 3 - fr - ph) / 3;
    int carry = 0;
    for (int T = 0; T < nt; T++) {
      const int64_t e = (first + T) * 6 + sf;
      const int p = out.prefix[e];
      if (p < kTileCodons) {
        const int len = carry + p;
        uint2 none = make_uint2(0u, 0u);
        if (len < minlen) out.cross[e] = none;
        else {
          const int u_stop = kTileCodons * T + p;
          out.cross[e] = make_uint2((unsigned)(rev ? C - u_stop + 1 : u_stop - len), (unsigned)len | ((unsigned)sf << 28));
          atomicAdd(&s_hist[orf_bin(len)], 1);
          my_orfs++; my_res += (unsigned)len;
        }
        carry = out.suffix[e];
      } else { carry += kTileCodons; out.cross[e] = make_uint2(0u, 0u); }
    }
  }
  if (my_orfs) { atomicAdd(&s_red[0], my_orfs); atomicAdd(&s_red[1], my_res); }
  __syncthreads();
  if (!out.count) return;
  for (int k = threadIdx.x; k < kOrfBins; k += blockDim.x) if (s_hist[k]) atomicAdd(&out.hist[k], s_hist[k]);
  if (threadIdx.x == 0 && s_red[0]) { atomicAdd(out.n_orfs, (unsigned long long)s_red[0]); atomicAdd(out.orf_res, (unsigned long long)s_red[1]); }
}

// The same for LONG windows (genome windows of 256 kb: ~680 tiles per window and frame), a WAVE per (window, frame): the lane
// walk above is a chain of dependent loads, one per tile (0.26 ms for a window whatever the block holds -- a tenth of a
// configs[3] query).  The run entering tile T comes from the nearest earlier tile that holds a stop: an exclusive running
// maximum of "index of a tile with a stop" over the wave's 64 tiles (DPP), that tile's suffix by one ds_bpermute, 128 codons
// for every stop-free tile in between; the run leaving the 64 tiles carries into the next 64.  Same records, same histogram.
__device__ __forceinline__ int wave_excl_max_scan(int v, int ident) {
  int x = dpp_or_ident<0x138>(v, ident);                           // wave_shr:1 (lane 0 gets the identity)
  x = max(x, dpp_or_ident<0x111>(x, ident));
  x = max(x, dpp_or_ident<0x112>(x, ident));
  x = max(x, dpp_or_ident<0x114>(x, ident));
  x = max(x, dpp_or_ident<0x118>(x, ident));
  x = max(x, dpp_or_ident<0x142, 0xa>(x, ident));                  // row_bcast:15
  x = max(x, dpp_or_ident<0x143, 0xc>(x, ident));                  // row_bcast:31
  return x;
}
__global__ __launch_bounds__(256) void orf_stitch_wave_kernel(SeqView dna, OrfTiles tiles, OrfScanOut out, int minlen) {
  __shared__ int s_hist[kOrfBins];
  __shared__ unsigned s_red[2];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_hist[i] = 0;
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
  __syncthreads();
  unsigned my_orfs = 0, my_res = 0;
  const int lane = threadIdx.x & 63;
  const int64_t nstreams = dna.n * 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t sidx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; sidx < nstreams; sidx += nwaves) {
    const int64_t w = sidx / 6;
    const int sf = (int)(sidx - w * 6);
    const int n = dna.len[w];
    if (n < 15) continue;
    const int nt = (n / 3) / kTileCodons + 1;
    const int64_t first = tiles.tile_first[w];
    const bool rev = sf >= 3;
    const int fr = rev ? sf - 3 : 0;
    const int ph = rev ? (n - fr) % 3 : sf;
    const int C = (n - 3 - fr - ph) / 3;
    int carry_in = 0;
    for (int T0 = 0; T0 < nt; T0 += 64) {
      const int T = T0 + lane;
      const bool valid = T < nt;
      const int64_t e = (first + (valid ? T : 0)) * 6 + sf;
      const int p = valid ? out.prefix[e] : kTileCodons;
      const bool has = p < kTileCodons;
      const int sfx = has ? out.suffix[e] : 0;
      const int j = wave_excl_max_scan(has ? lane : -1, -1);           // the nearest earlier tile of these 64 that holds a stop
      const int sfx_j = __shfl(sfx, j < 0 ? 0 : j, 64);
      const int carry = (j >= 0) ? sfx_j + kTileCodons * (lane - j - 1) : carry_in + kTileCodons * lane;
      if (valid) {
        uint2 rec = make_uint2(0u, 0u);
        if (has) {
          const int len = carry + p;
          if (len >= minlen) {
            const int u_stop = kTileCodons * T + p;
            rec = make_uint2((unsigned)(rev ? C - u_stop + 1 : u_stop - len), (unsigned)len | ((unsigned)sf << 28));
            atomicAdd(&s_hist[orf_bin(len)], 1);
            my_orfs++; my_res += (unsigned)len;
          }
        }
        out.cross[e] = rec;
      }
      const int leave = has ? sfx : carry + kTileCodons;               // the run leaving this lane's tile
      carry_in = __shfl(leave, 63, 64);
    }
  }
  if (my_orfs) { atomicAdd(&s_red[0], my_orfs); atomicAdd(&s_red[1], my_res); }
  __syncthreads();
  if (!out.count) return;
  for (int k = threadIdx.x; k < kOrfBins; k += blockDim.x) if (s_hist[k]) atomicAdd(&out.hist[k], s_hist[k]);
  if (threadIdx.x == 0 && s_red[0]) { atomicAdd(out.n_orfs, (unsigned long long)s_red[0]); atomicAdd(out.orf_res, (unsigned long long)s_red[1]); }
}

// pli->strands and the initiation codons of -m / -M, on the records the scan kernels left (a lane per tile: its slots and its
// six crossing records).  A record of an excluded strand is cleared.  With initiators an ORF begins at the first initiation
// codon of its stop-free run (esl_gencode_ProcessPiece: a stop closes the frame's ORF, any other codon extends an open one, only an
// initiator opens one) and that codon reads M: the record is cut down -- and dropped if fewer than <minlen> residues remain --
// and the M is written into the amino-acid stream (the codons before it belong to no ORF).  A degenerate codon initiates only if
// every codon it stands for does (esl_gencode_IsInitiator).  The kernel counts what it keeps (histogram, ORFs, residues).
__global__ __launch_bounds__(256) void orf_filter_kernel(SeqView dna, OrfTiles tiles, OrfScanOut out, OrfFilter flt, int minlen) {
  __shared__ int s_hist[kOrfBins];
  __shared__ unsigned s_red[2];
  __shared__ uint8_t s_init[64], s_comp[32];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_hist[i] = 0;
  if (threadIdx.x < 64) s_init[threadIdx.x] = flt.using_initiators ? flt.is_init[threadIdx.x] : 1;
  if (threadIdx.x < 18) s_comp[threadIdx.x] = flt.comp[threadIdx.x];
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
  __syncthreads();
  // members of a DNA code as a bit mask over A C G T ("ACGT-RYMKSWHBVDN*~")
  auto members = [](int x) -> unsigned {
    switch (x) {
      case 0: return 1u; case 1: return 2u; case 2: return 4u; case 3: return 8u;
      case 5: return 5u; case 6: return 10u; case 7: return 3u; case 8: return 12u; case 9: return 6u; case 10: return 9u;
      case 11: return 11u; case 12: return 14u; case 13: return 7u; case 14: return 13u; case 15: return 15u;
      default: return 0u;                                                   // gap, *, ~
    }
  };
  auto initiates = [&](int a, int b, int c) -> bool {
    if ((a | b | c) < 4) return s_init[16 * a + 4 * b + c] != 0;
    const unsigned ma = members(a), mb = members(b), mc = members(c);
    int n = 0;
    for (int x = 0; x < 4; x++) if (ma >> x & 1u)
      for (int y = 0; y < 4; y++) if (mb >> y & 1u)
        for (int z = 0; z < 4; z++) if (mc >> z & 1u) { if (!s_init[16 * x + 4 * y + z]) return false; n++; }
    return n > 0;
  };
  unsigned my_orfs = 0, my_res = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < tiles.ntiles; t += (int64_t)gridDim.x * blockDim.x) {
    const int4 dsc = tiles.desc[t];
    const int n = dsc.y, w = dsc.z;
    if (n < 15) continue;
    const int64_t off = (int64_t)(uint32_t)dsc.x << 4;
    const uint8_t *x = dna.data + off;
    uint8_t *const abase = out.aa + (2 * off + 96 * (int64_t)w);
    const int pitch = orf_stream_pitch(n);
    const int c = out.cnt[t];
    uint2 *sl = out.slots + t * out.cap;
    for (int k = 0; k < c + 6; k++) {
      uint2 *rp = k < c ? sl + k : out.cross + t * 6 + (k - c);
      uint2 r = *rp;
      int len = (int)(r.y & 0x0fffffffu);
      if (len == 0) continue;
      const int sf = (int)(r.y >> 28);
      const bool rev = sf >= 3;
      if ((flt.strands == BATH_STRAND_TOPONLY && rev) || (flt.strands == BATH_STRAND_BOTTOMONLY && !rev)) { rp->y = 0u; continue; }
      if (flt.using_initiators) {
        int j = (int)r.x;
        const int jend = j + len;
        for (; j < jend; j++) {
          int a, b, cc;
          if (!rev) { const int p = sf + 3 * j; a = x[p]; b = x[p + 1]; cc = x[p + 2]; }
          else { const int q = (sf - 3) + 3 * j; a = s_comp[min((int)x[n - 1 - q], 17)]; b = s_comp[min((int)x[n - 2 - q], 17)]; cc = s_comp[min((int)x[n - 3 - q], 17)]; }
          if (initiates(min(a, 17), min(b, 17), min(cc, 17))) break;
        }
        len = jend - j;
        if (len < minlen || len <= 0) { rp->y = 0u; continue; }
        abase[sf * pitch + j] = 10;                                          // 'M' in "ACDEFGHIKLMNPQRSTVWY": the initiation codon's residue
        r.x = (unsigned)j; r.y = (unsigned)len | ((unsigned)sf << 28);
        *rp = r;
      }
      atomicAdd(&s_hist[orf_bin(len)], 1);
      my_orfs++; my_res += (unsigned)len;
    }
  }
  if (my_orfs) { atomicAdd(&s_red[0], my_orfs); atomicAdd(&s_red[1], my_res); }
  __syncthreads();
  for (int k = threadIdx.x; k < kOrfBins; k += blockDim.x) if (s_hist[k]) atomicAdd(&out.hist[k], s_hist[k]);
  if (threadIdx.x == 0 && s_red[0]) { atomicAdd(out.n_orfs, (unsigned long long)s_red[0]); atomicAdd(out.orf_res, (unsigned long long)s_red[1]); }
}

// counts -> start offsets, longest first; cursor[] is the copy the sort kernel advances; total ORFs -> *n_total
__global__ void orf_scan_bins(const int *__restrict__ hist, int *__restrict__ cursor, int *__restrict__ n_total) {
  __shared__ int tmp[kOrfBins];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) tmp[i] = hist[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int b = kOrfBins - 1; b >= 0; b--) { const int c = tmp[b]; tmp[b] = run; run += c; }
    *n_total = run;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) cursor[i] = tmp[i];
}

constexpr int kSortTilesPerThread = 8;

__global__ __launch_bounds__(256) void orf_sort_kernel(SeqView dna, OrfTiles tiles, const uint2 *__restrict__ slots, int cap, const int32_t *__restrict__ cnt,
                                                       const uint2 *__restrict__ cross, int *__restrict__ cursor, OrfRec *__restrict__ sorted) {
  __shared__ int s_cnt[kOrfBins];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_cnt[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * kSortTilesPerThread);
  // pass 1: this block's ORFs per length bin
  for (int i = 0; i < kSortTilesPerThread; i++) {
    const int64_t t = base + (int64_t)i * 256 + threadIdx.x;
    if (t >= tiles.ntiles) break;
    const int c = cnt[t];
    const uint2 *sl = slots + t * cap;
    for (int k = 0; k < c; k++) { const unsigned y = sl[k].y & 0x0fffffffu; if (y) atomicAdd(&s_cnt[orf_bin((int)y)], 1); }   // (0: cleared by orf_filter_kernel)
    for (int f = 0; f < 6; f++) { const unsigned y = cross[t * 6 + f].y & 0x0fffffffu; if (y) atomicAdd(&s_cnt[orf_bin((int)y)], 1); }
  }
  __syncthreads();
  // reserve this block's range in every non-empty bin
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) {
    const int c = s_cnt[i];
    if (c) s_cnt[i] = atomicAdd(&cursor[i], c);
  }
  __syncthreads();
  // pass 2: scatter
  for (int i = 0; i < kSortTilesPerThread; i++) {
    const int64_t t = base + (int64_t)i * 256 + threadIdx.x;
    if (t >= tiles.ntiles) break;
    const int c = cnt[t];
    const int4 dsc = tiles.desc[t];
    const int w = dsc.z;
    const int64_t wbase = 2 * ((int64_t)(uint32_t)dsc.x << 4) + 96 * (int64_t)w;
    const int pitch = orf_stream_pitch(dsc.y);
    const uint2 *sl = slots + t * cap;
    for (int k = 0; k < c + 6; k++) {
      const uint2 r = k < c ? sl[k] : cross[t * 6 + (k - c)];
      if ((r.y & 0x0fffffffu) == 0u) continue;
      const int pos = atomicAdd(&s_cnt[orf_bin((int)(r.y & 0x0fffffffu))], 1);
      OrfRec rec;
      rec.aa_off = wbase + (int64_t)(r.y >> 28) * pitch + (int64_t)r.x; rec.w = w; rec.len_sf = (int32_t)r.y;
      sorted[pos] = rec;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
size_t orf_aa_bytes(const bath_hip_seqs *dna) { return (size_t)(2 * dna->total_aligned + 96 * dna->n + 256); }
void orf_buffers_carve(OrfBuffers *ob, void *aa, void *slots, void *sorted, void *misc, size_t nent) {
  ob->aa = static_cast<uint8_t *>(aa); ob->slots = slots; ob->sorted = static_cast<OrfRec *>(sorted);
  ob->cross = misc;                                           // nent records of 8 bytes first (keeps them 8-byte aligned)
  ob->cnt = static_cast<int32_t *>(misc) + 2 * nent; ob->prefix = ob->cnt + nent; ob->suffix = ob->prefix + nent;
  ob->hist = reinterpret_cast<int *>(ob->suffix + nent); ob->cursor = ob->hist + kOrfBins; ob->ntotal = ob->cursor + kOrfBins;
}

int orf_slot_cap(int minlen) { return 6 * (kTileCodons / (std::max(minlen, 0) + 1) + 2); }   // six frames share a tile's record region

// tile -> window map of a DNA block, built once per block and kept with it
int orf_tiles_ensure(bath_hip_ctx *ctx, const bath_hip_seqs *dna) {
  if (dna->ntiles >= 0) return BATH_OK;
  std::vector<int32_t> first((size_t)std::max<int64_t>(dna->n, 1));
  std::vector<int4> desc;
  int64_t nt = 0;
  for (int64_t w = 0; w < dna->n; w++) {
    first[(size_t)w] = (int32_t)nt;
    const int n = dna->h_len[(size_t)w];
    const int k = n >= 15 ? (n / 3) / kTileCodons + 1 : 0;
    nt += k;
    if (nt >= (int64_t)INT32_MAX / 8 || (dna->h_off[(size_t)w] >> 4) > (int64_t)UINT32_MAX) { ctx->set_error("DNA block too large for one pipeline call: split it"); return BATH_EINVAL; }
    for (int T = 0; T < k; T++) desc.push_back(int4{(int)(uint32_t)(dna->h_off[(size_t)w] >> 4), n, (int)w, T});
  }
  if (desc.empty()) desc.push_back(int4{0, 0, 0, 0});
  BATH_HIP_TRY(ctx, hipMalloc((void **)&dna->d_tile_desc, desc.size() * sizeof(int4)));
  BATH_HIP_TRY(ctx, hipMalloc((void **)&dna->d_tile_first, first.size() * sizeof(int32_t)));
  BATH_HIP_TRY(ctx, hipMemcpy(dna->d_tile_desc, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
  BATH_HIP_TRY(ctx, hipMemcpy(dna->d_tile_first, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  dna->ntiles = nt;
  return BATH_OK;
}

// ---- translation tables ------------------------------------------------------------------------------------------
static bool dna_degen_has(int x, int y) {
  static const char *members[18] = {"A", "C", "G", "T", "", "AG", "CT", "AC", "GT", "CG", "AT", "ACT", "CGT", "ACG", "AGT", "ACGT", "", ""};
  static const char nt[] = "ACGT";
  return std::strchr(members[x], nt[y]) != nullptr;
}

// codon -> amino acid for all 18^3 digital codons: canonical codons through basic[]; degenerate codons
// give the amino acid all expansions agree on, else X (easel esl_gencode_GetTranslation semantics).
static void build_codon_table(const uint8_t basic[64], std::vector<uint8_t> &tab) {
  tab.assign(18 * 18 * 18, (uint8_t)kXaa);
  for (int a = 0; a < 18; a++) for (int b = 0; b < 18; b++) for (int c = 0; c < 18; c++) {
    int aa = -1; bool mixed = false;
    for (int x = 0; x < 4 && !mixed; x++) { if (!dna_degen_has(a, x)) continue;
      for (int y = 0; y < 4 && !mixed; y++) { if (!dna_degen_has(b, y)) continue;
        for (int z = 0; z < 4; z++) { if (!dna_degen_has(c, z)) continue;
          int v = basic[16 * x + 4 * y + z];
          if (aa == -1) aa = v; else if (aa != v) { mixed = true; break; }
        } } }
    tab[(a * 18 + b) * 18 + c] = (uint8_t)((mixed || aa == -1) ? kXaa : aa);
  }
}

int orf_tables_upload(bath_hip_ctx *ctx, int ncbi_table, OrfTablesDev *t, int initiator) {
  const int key = ncbi_table | (initiator << 16);
  t->using_initiators = initiator != BATH_INIT_ANY;
  if (ctx->orf_tables_id == key && ctx->scratch[28].p) {            // still there from the last call: every query of a database pass asks again
    t->full = ctx->scratch[28].as<uint8_t>(); t->fwd = t->full + 6144; t->rev = t->fwd + 64; t->comp = t->rev + 64; t->is_init = t->comp + 32;
    return BATH_OK;
  }
  uint8_t basic[64];
  if (bath_gencode_basic(ncbi_table, basic) != BATH_OK) { ctx->set_error("unknown NCBI translation table"); return BATH_EINVAL; }
  std::vector<uint8_t> host(6144 + 256, 0);
  std::vector<uint8_t> full;
  build_codon_table(basic, full);
  std::memcpy(host.data(), full.data(), full.size());
  uint8_t *fwd = host.data() + 6144, *rev = fwd + 64, *comp = rev + 64;
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int c = 0; c < 4; c++) {
    fwd[a * 16 + b * 4 + c] = basic[16 * a + 4 * b + c];
    // memory-order bytes (x0,x1,x2) = (a,b,c): the reverse-strand codon is comp(x2), comp(x1), comp(x0)
    rev[a * 16 + b * 4 + c] = basic[16 * (3 - c) + 4 * (3 - b) + (3 - a)];
  }
  static const uint8_t kComp[18] = {3, 2, 1, 0, 4, 6, 5, 8, 7, 9, 10, 14, 13, 12, 11, 15, 16, 17};   // ACGT-RYMKSWHBVDN*~
  std::memcpy(comp, kComp, 18);
  if (bath_gencode_initiators(ncbi_table, initiator, comp + 32) != BATH_OK) { ctx->set_error("no initiation codons for this table / initiator mode"); return BATH_EINVAL; }
  DevBuf &b = ctx->scratch[28];
  BATH_HIP_TRY(ctx, b.reserve(host.size()));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(b.p, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->orf_tables_id = key;
  t->full = b.as<uint8_t>(); t->fwd = t->full + 6144; t->rev = t->fwd + 64; t->comp = t->rev + 64; t->is_init = t->comp + 32;
  return BATH_OK;
}

int launch_orf_scan(bath_hip_ctx *ctx, const bath_hip_seqs *dna, const OrfTablesDev &tt, int minlen, const OrfBuffers &b,
                    unsigned long long *d_n_orfs, unsigned long long *d_orf_res, int strands) {
  const int64_t ntiles = dna->ntiles;
  BATH_HIP_TRY(ctx, hipMemsetAsync(b.hist, 0, kOrfBins * sizeof(int), ctx->stream));
  OrfScanTables tabs{tt.full, tt.fwd, tt.rev, tt.comp};
  OrfTiles tiles{reinterpret_cast<const int4 *>(dna->d_tile_desc), dna->d_tile_first, ntiles};
  const bool filtered = strands != BATH_STRAND_BOTH || tt.using_initiators;   // not bathsearch's defaults: orf_filter_kernel edits the records and counts
  OrfScanOut out{b.aa, reinterpret_cast<uint2 *>(b.slots), b.cnt, reinterpret_cast<uint2 *>(b.cross), b.prefix, b.suffix, b.hist, d_n_orfs, d_orf_res, orf_slot_cap(minlen), filtered ? 0 : 1};
  const int cus = ctx->prop.multiProcessorCount;
  // persistent blocks: exactly as many as are resident at once, so that every block gets the same share of the tiles
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, orf_tile_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + 7) / 8, (int64_t)cus * per_cu));
  hipLaunchKernelGGL(orf_tile_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dna->view(), tiles, tabs, out, minlen);
  static const int stitch_env = [] { const char *e = std::getenv("BATH_HIP_STITCH_WAVE"); return e ? std::atoi(e) : -1; }();   // 1 / 0: always / never the wave kernel (tests, A/B)
  const bool long_windows = stitch_env >= 0 ? stitch_env == 1 : ntiles > 32 * dna->n;                   // more than 32 tiles (12 kb) per window on average
  if (long_windows) {
    const int wblocks = (int)std::max<int64_t>(1, std::min<int64_t>((dna->n * 6 + 3) / 4, (int64_t)cus * 8));
    hipLaunchKernelGGL(orf_stitch_wave_kernel, dim3(wblocks), dim3(256), 0, ctx->stream, dna->view(), tiles, out, minlen);
  } else {
  const int sblocks = (int)std::max<int64_t>(1, std::min<int64_t>((dna->n * 6 + 255) / 256, (int64_t)cus * 8));
  hipLaunchKernelGGL(orf_stitch_kernel, dim3(sblocks), dim3(256), 0, ctx->stream, dna->view(), tiles, out, minlen);
  }
  if (filtered) {
    OrfFilter flt{strands, tt.using_initiators ? 1 : 0, tt.is_init, tt.comp};
    const int fblocks = (int)std::max<int64_t>(1, std::min<int64_t>((ntiles + 255) / 256, (int64_t)cus * 8));
    hipLaunchKernelGGL(orf_filter_kernel, dim3(fblocks), dim3(256), 0, ctx->stream, dna->view(), tiles, out, flt, minlen);
  }
  hipLaunchKernelGGL(orf_scan_bins, dim3(1), dim3(256), 0, ctx->stream, b.hist, b.cursor, b.ntotal);
  const int per_block = 256 * kSortTilesPerThread;
  hipLaunchKernelGGL(orf_sort_kernel, dim3((unsigned)std::max<int64_t>(1, (ntiles + per_block - 1) / per_block)), dim3(256), 0, ctx->stream, dna->view(), tiles,
                     reinterpret_cast<const uint2 *>(b.slots), out.cap, b.cnt, reinterpret_cast<const uint2 *>(b.cross), b.cursor, b.sorted);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath

using namespace bath;

// esl_gencode_Process* as driven by bathsearch.c:384-392, for a whole block: the ORF list and residues, on the host.
extern "C" int bath_hip_translate_orfs(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int ncbi_table, int min_orf_len,
                                       const bath_orf **orfs, int64_t *n_orfs, const uint8_t **aa) {
  return bath_hip_translate_orfs_opts(ctx, dna, ncbi_table, min_orf_len, BATH_STRAND_BOTH, BATH_INIT_ANY, orfs, n_orfs, aa);
}

extern "C" int bath_hip_translate_orfs_opts(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int ncbi_table, int min_orf_len, int strands, int initiator,
                                            const bath_orf **orfs, int64_t *n_orfs, const uint8_t **aa) {
  if (!ctx || !dna || !orfs || !n_orfs || min_orf_len < 0 || strands < 0 || strands > 2 || initiator < 0 || initiator > 2) return BATH_EINVAL;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  *orfs = nullptr; *n_orfs = 0;
  if (aa) *aa = nullptr;
  ctx->orfs.clear(); ctx->orf_aa.clear();
  if (dna->n == 0) return BATH_OK;
  int st;
  OrfTablesDev tt{};
  if ((st = orf_tables_upload(ctx, ncbi_table, &tt, initiator)) != BATH_OK) return st;
  if ((st = orf_tiles_ensure(ctx, dna)) != BATH_OK) return st;
  int64_t max_orfs = 0;
  for (int64_t i = 0; i < dna->n; i++) if (dna->h_len[i] >= 15) max_orfs += 6 * (int64_t)((dna->h_len[i] / 3 + 1) / (min_orf_len + 1) + 1);
  const size_t nent = (size_t)dna->ntiles * 6;
  DevBuf &b_aa = ctx->scratch[24], &b_slots = ctx->scratch[25], &b_orfs = ctx->scratch[26], &b_misc = ctx->scratch[27];
  BATH_HIP_TRY(ctx, b_aa.reserve(orf_aa_bytes(dna)));
  BATH_HIP_TRY(ctx, b_slots.reserve(((size_t)dna->ntiles * (size_t)orf_slot_cap(min_orf_len) + 64) * 8));
  BATH_HIP_TRY(ctx, b_orfs.reserve((size_t)(max_orfs + 64) * sizeof(OrfRec)));
  BATH_HIP_TRY(ctx, b_misc.reserve((5 * nent + 2 * kOrfBins + 64) * sizeof(int32_t) + 64));
  OrfBuffers ob{};
  orf_buffers_carve(&ob, b_aa.p, b_slots.p, b_orfs.p, b_misc.p, nent);
  unsigned long long *d_ctr = reinterpret_cast<unsigned long long *>(ob.ntotal + 2);   // two counters nobody reads here
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_ctr, 0, 16, ctx->stream));
  if ((st = launch_orf_scan(ctx, dna, tt, min_orf_len, ob, d_ctr, d_ctr + 1, strands)) != BATH_OK) return st;
  int total = 0;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(&total, ob.ntotal, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<OrfRec> recs((size_t)total);
  std::vector<uint8_t> pool(orf_aa_bytes(dna));
  if (total > 0) BATH_HIP_TRY(ctx, hipMemcpy(recs.data(), ob.sorted, recs.size() * sizeof(OrfRec), hipMemcpyDeviceToHost));
  BATH_HIP_TRY(ctx, hipMemcpy(pool.data(), ob.aa, pool.size(), hipMemcpyDeviceToHost));
  ctx->orfs.resize((size_t)total);
  for (int i = 0; i < total; i++) {
    const OrfRec &r = recs[(size_t)i];
    const int64_t w = r.w;
    const int sf = (int)((unsigned)r.len_sf >> 28), len = r.len_sf & 0x0fffffff;
    const int64_t stream = 2 * dna->h_off[w] + 96 * w + (int64_t)sf * orf_stream_pitch(dna->h_len[w]);
    bath_orf &o = ctx->orfs[(size_t)i];
    o.window = w; o.strand = sf / 3; o.frame = sf % 3; o.n = len;
    o.start = o.frame + 3 * (int32_t)(r.aa_off - stream) + 1; o.end = o.start + 3 * len - 1;
    o.aa_off = r.aa_off;                                     // device pool offset for now; rewritten below
  }
  std::sort(ctx->orfs.begin(), ctx->orfs.end(), [](const bath_orf &a, const bath_orf &b) {
    if (a.window != b.window) return a.window < b.window;
    if (a.strand != b.strand) return a.strand < b.strand;
    if (a.frame != b.frame) return a.frame < b.frame;
    return a.start < b.start;
  });
  for (bath_orf &o : ctx->orfs) {
    const int64_t src = o.aa_off;
    o.aa_off = (int64_t)ctx->orf_aa.size();
    ctx->orf_aa.insert(ctx->orf_aa.end(), pool.begin() + src, pool.begin() + src + o.n);
  }
  *orfs = ctx->orfs.data(); *n_orfs = total;
  if (aa) *aa = ctx->orf_aa.data();
  return BATH_OK;
}
