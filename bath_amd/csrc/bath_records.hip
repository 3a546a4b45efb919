// bath_records.hip -- the cascade's per-ORF records (bath_orf_result), assembled and ordered on the device.
//
// What p7_Pipeline_BATH leaves per ORF that passed the MSV filter is scattered over the candidate arrays (structure of arrays,
// in the order the SSV kernel's lanes happened to append them).  The caller wants one record per ORF, ordered by window,
// strand, frame and start.  Doing that on the host cost 30-60 ms for the 576 k survivors of the bench block (13 pageable
// copies, a gather and a sort of 72-byte records); here a key per candidate is radix-sorted on the device (rocPRIM), a kernel
// writes the records in that order, and one copy moves them into page-locked memory.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

namespace bath {

namespace {
// key: window (32 bits) | strand*3+frame (4) | first codon (28): ascending = the order of the reference's ORF loop within a window
__global__ void record_keys_kernel(Cand cand, int nc, unsigned long long *__restrict__ keys, unsigned *__restrict__ idx, unsigned *__restrict__ count) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
    if (cand.stage[c] < 1) continue;
    const unsigned slot = atomicAdd(count, 1u);
    keys[slot] = ((unsigned long long)cand.window[c] << 32) | ((unsigned long long)(unsigned)cand.sf[c] << 28) | (unsigned long long)((unsigned)cand.startj[c] & 0x0fffffffu);
    idx[slot] = (unsigned)c;
  }
}
__global__ void record_build_kernel(Cand cand, const unsigned *__restrict__ idx, unsigned n, long long window_offset, bath_orf_result *__restrict__ out) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = (int)idx[i];
    bath_orf_result r{};
    const int sf = cand.sf[c], len = cand.len[c], stage = cand.stage[c];
    r.window = cand.window[c] + window_offset; r.strand = sf / 3; r.frame = sf % 3;
    r.start = r.frame + 3 * cand.startj[c] + 1; r.end = r.start + 3 * len - 1; r.n = len;
    r.stage = (stage == 5) ? 2 : stage;
    r.msv_status = cand.msv_status[c]; r.vit_status = cand.vit_status[c];
    r.usc = cand.usc[c]; r.nullsc = cand.nullsc[c]; r.filtersc = cand.filtersc[c]; r.vfsc = cand.vfsc[c]; r.fwdsc = cand.fwdsc[c]; r.P = cand.P[c];
    out[i] = r;
  }
}
}  // namespace

int build_orf_records(bath_hip_ctx *ctx, const Cand &cand, int nc, int64_t window_offset, bath_orf_result **d_out, int64_t *n_out) {
  *d_out = nullptr; *n_out = 0;
  if (nc <= 0) return BATH_OK;
  DevBuf &b_keys = ctx->scratch[33], &b_tmp = ctx->scratch[34], &b_rec = ctx->scratch[35];
  const size_t n = (size_t)nc;
  const size_t o_k1 = 256, o_k2 = o_k1 + (n * 8 + 255) / 256 * 256, o_i1 = o_k2 + (n * 8 + 255) / 256 * 256, o_i2 = o_i1 + (n * 4 + 255) / 256 * 256;
  BATH_HIP_TRY(ctx, b_keys.reserve(o_i2 + n * 4 + 256));
  char *p = b_keys.as<char>();
  unsigned *d_cnt = reinterpret_cast<unsigned *>(p);
  unsigned long long *k1 = reinterpret_cast<unsigned long long *>(p + o_k1), *k2 = reinterpret_cast<unsigned long long *>(p + o_k2);
  unsigned *i1 = reinterpret_cast<unsigned *>(p + o_i1), *i2 = reinterpret_cast<unsigned *>(p + o_i2);
  BATH_HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 256, ctx->stream));
  const int blocks = ctx->prop.multiProcessorCount * 4;
  hipLaunchKernelGGL(record_keys_kernel, dim3(blocks), dim3(256), 0, ctx->stream, cand, nc, k1, i1, d_cnt);
  BATH_HIP_TRY(ctx, hipGetLastError());
  unsigned h_cnt = 0;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(&h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (h_cnt == 0) return BATH_OK;
  size_t tmp_bytes = 0;
  BATH_HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, k1, k2, i1, i2, (size_t)h_cnt, 0, 64, ctx->stream));
  BATH_HIP_TRY(ctx, b_tmp.reserve(tmp_bytes + 256));
  BATH_HIP_TRY(ctx, rocprim::radix_sort_pairs(b_tmp.p, tmp_bytes, k1, k2, i1, i2, (size_t)h_cnt, 0, 64, ctx->stream));
  BATH_HIP_TRY(ctx, b_rec.reserve((size_t)h_cnt * sizeof(bath_orf_result) + 256));
  hipLaunchKernelGGL(record_build_kernel, dim3(blocks), dim3(256), 0, ctx->stream, cand, i2, h_cnt, (long long)window_offset, b_rec.as<bath_orf_result>());
  BATH_HIP_TRY(ctx, hipGetLastError());
  *d_out = b_rec.as<bath_orf_result>(); *n_out = (int64_t)h_cnt;
  return BATH_OK;
}

}  // namespace bath
