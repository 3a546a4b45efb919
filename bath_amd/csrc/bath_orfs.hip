// bath_orfs.hip -- six-frame translation and ORF finding for a block of DNA windows, and the length-sorted
// ORF work list that feeds the lane-per-ORF SSV kernel.
//
// Reference behaviour (easel esl_gencode_ProcessStart/ProcessPiece/ProcessEnd as driven by
// src/bathsearch.c:384-392 with do_watson/do_crick, minimum_length = 20, no initiator requirement): every
// maximal run of non-stop codons of at least <minlen> residues in each of the six frames of a window is one
// ORF; windows shorter than 15 nt are skipped (bathsearch.c:1066).
//
// GPU formulation.
//   orf_scan_kernel   one lane per window.  The lane reads its window once, 12 nt per step, and feeds all six frames:
//                     the three bytes at position p are codon p/3 of forward frame p%3 and, on the other strand,
//                     a codon of reverse frame (n-p)%3 (runs of non-stop codons are the same whichever way a frame is
//                     walked, so the reverse frames are scanned in memory order too).  Canonical ACGT codons are
//                     decoded by one 24-bit multiply + bit-field extract + a 64-entry LDS table per strand, degenerate
//                     codes by the general 18^3 table.  Each stream gets one dword of residues per step, each ORF
//                     >= minlen one record in the stream's own slot range (no atomics), and ORF lengths are counted
//                     in an LDS histogram flushed once per block.
//   orf_scan_bins     turns the global histogram into start offsets, longest ORFs first.
//   orf_sort_kernel   compacts the sparse slots into the dense work list ordered by length (block-local counting
//                     sort: one global atomic per block and non-empty length bin).
// Memory layout (all offsets are closed forms of the window's offset in the DNA block, so no prefix sums):
//   amino stream of (w, sf):  aa + 2*off[w] + 96*w + sf*pitch(n),    pitch(n) = (n/3 + 16) & ~15
//   ORF slots of (w, sf):     slots + 6*(off[w]/q + 2*w) + sf*(n/q + 1),   q = 3*(minlen+1)
#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

namespace bath {

struct OrfScanTables {
  const uint8_t *aa_full;   // [5832] (a*18+b)*18+c -> amino code, degenerate codons resolved (esl_gencode_GetTranslation)
  const uint8_t *aa64_fwd;  // [64] a<<4|b<<2|c for canonical codons
  const uint8_t *aa64_rev;  // [64] x0<<4|x1<<2|x2 of the three bytes in memory order -> amino code of the reverse-complement codon
  const uint8_t *comp;      // [18]
};

struct OrfScanOut {
  uint8_t *aa;              // amino-acid streams
  uint2 *slots;             // {first codon index within the stream, length}
  int32_t *cnt;             // [6*nwin] ORFs recorded per stream
  int *hist;                // [kOrfBins] ORF length histogram (lengths above the last bin are clamped into it)
  unsigned long long *n_orfs, *orf_res;
};

__device__ __forceinline__ int orf_bin(int len) { return min(len, kOrfBins - 1); }

__device__ __forceinline__ void orf_record(uint2 *slots, int &nrec, int start, int len, unsigned &res, int *s_hist) {
  slots[nrec++] = make_uint2((unsigned)start, (unsigned)len);
  res += (unsigned)len;
  atomicAdd(&s_hist[orf_bin(len)], 1);
}

// One 12-nt chunk of one window: positions p = 12J+k, k = 0..11.  The codon starting at p is codon 4J+k/3 of forward
// frame k%3 and, read downwards on the other strand, codon C[k%3]-4J-k/3 of the reverse frame that phase k%3 maps to in
// this window ((n-p) mod 3; the caller resolved it into the rev_* pointers).  D[0..3] are the dwords at 12J .. 12J+15.
// FAST: every position of the chunk is inside the window and holds canonical nucleotides, for every active lane of the
// wave: codons are decoded with one 24-bit multiply + bit-field extract + two 64-entry LDS tables, and each of the six
// streams receives one dword.  Otherwise: general 18^3 table, per-position bounds, byte stores.
template <bool FAST>
__device__ __forceinline__ void orf_chunk(const uint32_t (&D)[4], int J, int n, const int (&C)[3], int minlen, int (&len_f)[3], int (&len_r)[3],
                                          int (&nrec_f)[3], int (&nrec_r)[3], uint2 *const (&slots_f)[3], uint2 *const (&slots_r)[3],
                                          uint8_t *const (&aa_f)[3], uint8_t *const (&aa_r)[3], unsigned &res, int *s_hist,
                                          const uint8_t *s_full, const uint8_t *s_fwd, const uint8_t *s_rev, const uint8_t *s_comp) {
  unsigned acc_f[3] = {0u, 0u, 0u}, acc_r[3] = {0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const int ph = k % 3, i = k / 3;
    const unsigned t = (k % 4 == 0) ? D[k / 4] : __builtin_amdgcn_alignbyte(D[k / 4 + 1], D[k / 4], k % 4);   // bytes p, p+1, p+2 (+1 ignored)
    int af, ar;
    if (FAST) {
      // bytes x0,x1,x2 < 4 at bits 0,8,16: t * (2^20 + 2^10 + 1) has x0<<4|x1<<2|x2 at bits 16..21 and nothing else there
      const unsigned idx = (__umul24(t, 0x100401u) >> 16) & 63u;
      af = s_fwd[idx]; ar = s_rev[idx];
    } else {
      const int x0 = min((int)(t & 0xffu), 17), x1 = min((int)((t >> 8) & 0xffu), 17), x2 = min((int)((t >> 16) & 0xffu), 17);
      af = s_full[(x0 * 18 + x1) * 18 + x2];
      ar = s_full[((int)s_comp[x2] * 18 + (int)s_comp[x1]) * 18 + (int)s_comp[x0]];
    }
    const int jf = 4 * J + i, jr = C[ph] - 4 * J - i;
    if (FAST || 12 * J + k + 2 < n) {
      const bool stop_f = (af == kStop), stop_r = (ar == kStop);
      if (stop_f && len_f[ph] >= minlen) orf_record(slots_f[ph], nrec_f[ph], jf - len_f[ph], len_f[ph], res, s_hist);
      if (stop_r && len_r[ph] >= minlen) orf_record(slots_r[ph], nrec_r[ph], jr + 1, len_r[ph], res, s_hist);
      len_f[ph] = stop_f ? 0 : len_f[ph] + 1;
      len_r[ph] = stop_r ? 0 : len_r[ph] + 1;
      if (FAST) { acc_f[ph] |= (unsigned)af << (8 * i); acc_r[ph] |= (unsigned)ar << (8 * (3 - i)); }
      else { aa_f[ph][jf] = (uint8_t)af; aa_r[ph][jr] = (uint8_t)ar; }
    }
  }
  if (FAST) {
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
      *reinterpret_cast<uint32_t *>(aa_f[ph] + 4 * J) = acc_f[ph];
      *reinterpret_cast<uint32_t *>(aa_r[ph] + (C[ph] - 4 * J - 3)) = acc_r[ph];      // codons C-4J-3 .. C-4J; unaligned by the lane's (C+1)%4
    }
  }
}

// One lane per window: the lane walks its window once and feeds all six frames.
__global__ __launch_bounds__(256) void orf_scan_kernel(SeqView dna, OrfScanTables tabs, OrfScanOut out, int minlen) {
  __shared__ int s_hist[kOrfBins];
  __shared__ __attribute__((aligned(16))) uint8_t s_full[5832 + 8];
  __shared__ uint8_t s_fwd[64], s_rev[64], s_comp[32];
  __shared__ unsigned s_red[2];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_hist[i] = 0;
  for (int i = threadIdx.x; i < 5832; i += blockDim.x) s_full[i] = tabs.aa_full[i];
  if (threadIdx.x < 64) { s_fwd[threadIdx.x] = tabs.aa64_fwd[threadIdx.x]; s_rev[threadIdx.x] = tabs.aa64_rev[threadIdx.x]; }
  if (threadIdx.x < 18) s_comp[threadIdx.x] = tabs.comp[threadIdx.x];
  if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
  __syncthreads();
  const int q = 3 * (minlen + 1);
  unsigned my_orfs = 0, my_res = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t wb = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); wb < dna.n; wb += stride) {
    const int64_t w = wb + (threadIdx.x & 63);
    const bool live = w < dna.n;
    const int nraw = live ? dna.len[w] : 0;
    const int n = nraw >= 15 ? nraw : 0;                      // windows < 15 nt are skipped, bathsearch.c:1066
    const int64_t off = live ? dna.off[w] : 0;
    const uint8_t *d = dna.data + off;
    const int npos = max(n - 2, 0);                           // codon start positions 0 .. n-3
    const int nchunks = (npos + 11) / 12;
    const int nfull = n >= 14 ? (n - 14) / 12 + 1 : 0;        // chunks whose 12 positions all fit: 12J+14 <= n
    const int Jw = wave_max_i32(nchunks);
    const int pitch = orf_stream_pitch(nraw), cap = nraw / q + 1;
    uint8_t *const abase = out.aa + (2 * off + 96 * w);
    uint2 *const sbase = out.slots + 6 * (off / q + 2 * w);
    int C[3];
    uint8_t *aa_f[3], *aa_r[3];
    uint2 *slots_f[3], *slots_r[3];
    int len_f[3] = {0, 0, 0}, len_r[3] = {0, 0, 0}, nrec_f[3] = {0, 0, 0}, nrec_r[3] = {0, 0, 0};
#pragma unroll
    for (int ph = 0; ph < 3; ph++) {
      const int fr = (n - ph + 3) % 3;                        // the reverse frame whose codons start (in memory order) at p = ph mod 3
      C[ph] = (n - 3 - fr - ph) / 3;                          // its codon index at p = ph; one less every 3 nt
      aa_f[ph] = abase + (int64_t)ph * pitch;       aa_r[ph] = abase + (int64_t)(3 + fr) * pitch;
      slots_f[ph] = sbase + (int64_t)ph * cap;      slots_r[ph] = sbase + (int64_t)(3 + fr) * cap;
    }
    uint32_t N[4] = {0, 0, 0, 0};
    if (0 < nchunks) {
#pragma unroll
      for (int i = 0; i < 4; i++) N[i] = *reinterpret_cast<const uint32_t *>(d + 4 * i);
    }
    for (int J = 0; J < Jw; J++) {
      const bool active = J < nchunks;
      const uint32_t D[4] = {N[0], N[1], N[2], N[3]};
      if (J + 1 < nchunks) {
#pragma unroll
        for (int i = 0; i < 4; i++) N[i] = *reinterpret_cast<const uint32_t *>(d + 12 * (J + 1) + 4 * i);
      }
      const bool fast = (J < nfull) && (((D[0] | D[1] | D[2] | (D[3] & 0xffffu)) & 0xfcfcfcfcu) == 0u);
      if (__all(fast || !active)) {
        if (active) orf_chunk<true>(D, J, n, C, minlen, len_f, len_r, nrec_f, nrec_r, slots_f, slots_r, aa_f, aa_r, my_res, s_hist, s_full, s_fwd, s_rev, s_comp);
      } else if (active) {
        orf_chunk<false>(D, J, n, C, minlen, len_f, len_r, nrec_f, nrec_r, slots_f, slots_r, aa_f, aa_r, my_res, s_hist, s_full, s_fwd, s_rev, s_comp);
      }
    }
    // the ends of the streams close the open ORFs: forward frames at their last codon, reverse frames at codon 0
    if (n > 0) {
#pragma unroll
      for (int ph = 0; ph < 3; ph++) {
        if (len_f[ph] >= minlen) orf_record(slots_f[ph], nrec_f[ph], (n - ph) / 3 - len_f[ph], len_f[ph], my_res, s_hist);
        if (len_r[ph] >= minlen) orf_record(slots_r[ph], nrec_r[ph], 0, len_r[ph], my_res, s_hist);
      }
    }
    if (live) {
#pragma unroll
      for (int ph = 0; ph < 3; ph++) {
        const int fr = (n - ph + 3) % 3;
        out.cnt[w * 6 + ph] = nrec_f[ph];
        out.cnt[w * 6 + 3 + fr] = nrec_r[ph];
        my_orfs += (unsigned)(nrec_f[ph] + nrec_r[ph]);
      }
    }
  }
  if (my_orfs) { atomicAdd(&s_red[0], my_orfs); atomicAdd(&s_red[1], my_res); }
  __syncthreads();
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) if (s_hist[i]) atomicAdd(&out.hist[i], s_hist[i]);
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

constexpr int kSortStreamsPerThread = 16;

__global__ __launch_bounds__(256) void orf_sort_kernel(SeqView dna, const uint2 *__restrict__ slots, const int32_t *__restrict__ cnt, int *__restrict__ cursor,
                                                       OrfRec *__restrict__ sorted, int minlen, int64_t nstreams) {
  __shared__ int s_cnt[kOrfBins];
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) s_cnt[i] = 0;
  __syncthreads();
  const int q = 3 * (minlen + 1);
  const int64_t base = (int64_t)blockIdx.x * (256 * kSortStreamsPerThread);
  // pass 1: this block's ORFs per length bin
  for (int i = 0; i < kSortStreamsPerThread; i++) {
    const int64_t s = base + (int64_t)i * 256 + threadIdx.x;
    if (s >= nstreams) break;
    const int c = cnt[s];
    if (c == 0) continue;
    const int64_t w = s / 6;
    const int sf = (int)(s - w * 6);
    const int n = dna.len[w];
    const uint2 *sl = slots + (6 * (dna.off[w] / q + 2 * w) + (int64_t)sf * (n / q + 1));
    for (int k = 0; k < c; k++) atomicAdd(&s_cnt[orf_bin((int)sl[k].y)], 1);
  }
  __syncthreads();
  // reserve this block's range in every non-empty bin
  for (int i = threadIdx.x; i < kOrfBins; i += blockDim.x) {
    const int c = s_cnt[i];
    if (c) s_cnt[i] = atomicAdd(&cursor[i], c);
  }
  __syncthreads();
  // pass 2: scatter
  for (int i = 0; i < kSortStreamsPerThread; i++) {
    const int64_t s = base + (int64_t)i * 256 + threadIdx.x;
    if (s >= nstreams) break;
    const int c = cnt[s];
    if (c == 0) continue;
    const int64_t w = s / 6;
    const int sf = (int)(s - w * 6);
    const int n = dna.len[w];
    const int64_t off = dna.off[w];
    const uint2 *sl = slots + (6 * (off / q + 2 * w) + (int64_t)sf * (n / q + 1));
    const int64_t stream = 2 * off + 96 * w + (int64_t)sf * orf_stream_pitch(n);
    for (int k = 0; k < c; k++) {
      const uint2 r = sl[k];
      const int pos = atomicAdd(&s_cnt[orf_bin((int)r.y)], 1);
      OrfRec rec;
      rec.aa_off = stream + (int64_t)r.x; rec.w = (int32_t)w; rec.len_sf = (int32_t)(r.y | ((unsigned)sf << 28));
      sorted[pos] = rec;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
size_t orf_aa_bytes(const bath_hip_seqs *dna) { return (size_t)(2 * dna->total_aligned + 96 * dna->n + 256); }
size_t orf_slot_count(const bath_hip_seqs *dna, int minlen) { return (size_t)(6 * (dna->total_aligned / (3 * (minlen + 1)) + 2 * dna->n + 2)); }

void build_codon64(const uint8_t basic[64], uint8_t fwd[64], uint8_t rev[64]) {
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) for (int c = 0; c < 4; c++) {
    fwd[a * 16 + b * 4 + c] = basic[16 * a + 4 * b + c];
    // memory-order bytes (x0,x1,x2) = (a,b,c): the reverse-strand codon is comp(x2), comp(x1), comp(x0)
    rev[a * 16 + b * 4 + c] = basic[16 * (3 - c) + 4 * (3 - b) + (3 - a)];
  }
}

int launch_orf_scan(bath_hip_ctx *ctx, const bath_hip_seqs *dna, const uint8_t *d_aa_full, const uint8_t *d_aa64_fwd, const uint8_t *d_aa64_rev,
                    const uint8_t *d_comp, int minlen, uint8_t *d_aa, void *d_slots, int32_t *d_cnt, int *d_hist, int *d_cursor, int *d_ntotal,
                    unsigned long long *d_n_orfs, unsigned long long *d_orf_res, OrfRec *d_sorted) {
  const int64_t nwin = dna->n;
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, kOrfBins * sizeof(int), ctx->stream));
  OrfScanTables tabs{d_aa_full, d_aa64_fwd, d_aa64_rev, d_comp};
  OrfScanOut out{d_aa, reinterpret_cast<uint2 *>(d_slots), d_cnt, d_hist, d_n_orfs, d_orf_res};
  const int blocks = (int)std::min<int64_t>((nwin + 255) / 256, (int64_t)ctx->prop.multiProcessorCount * 8);
  hipLaunchKernelGGL(orf_scan_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dna->view(), tabs, out, minlen);
  hipLaunchKernelGGL(orf_scan_bins, dim3(1), dim3(256), 0, ctx->stream, d_hist, d_cursor, d_ntotal);
  const int64_t nstreams = nwin * 6;
  const int per_block = 256 * kSortStreamsPerThread;
  hipLaunchKernelGGL(orf_sort_kernel, dim3((unsigned)((nstreams + per_block - 1) / per_block)), dim3(256), 0, ctx->stream, dna->view(),
                     reinterpret_cast<const uint2 *>(d_slots), d_cnt, d_cursor, d_sorted, minlen, nstreams);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
