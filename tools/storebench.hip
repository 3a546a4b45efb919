// storebench.hip -- what a wave's store costs by the shape of its addresses (gfx950): the envelope wavefront kernels
// (bath_fs_wavefront.hip) write a cell per lane and step, every lane in a row of its own.
//   hipcc --offload-arch=gfx950 -O3 -o storebench tools/storebench.hip && ./storebench
// Patterns, all writing the same bytes (64 rows x 4 steps x 32 B per wave and round):
//   0  lane-per-row, per step 2 x 16 B to 64 different rows            (today's kernel)
//   1  lane-per-row, 4 steps gathered: 8 x 16 B to the lane's own 128 B (same requests, whole lines)
//   2  quads transposed: per instruction the 4 lanes of a quad write 64 contiguous bytes of one row (16 rows per instruction)
//   3  octets: 8 lanes write the 128 B of one row (8 rows per instruction)            (what an LDS transpose would give)
//   4  fully coalesced: 64 lanes x 16 B contiguous                                   (the ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int P>
__global__ __launch_bounds__(128) void k(float4 *out, int rows_per_wave_round, int rounds, int row_floats4 /* float4 per row */, int nodes) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float4 v = make_float4(lane, wave, 1.f, 2.f);
  float4 *base = out + wave * (size_t)64 * row_floats4;      // this wave's 64 rows
  for (int r = 0; r < rounds; r++) {
    for (int k0 = 0; k0 + 4 <= nodes; k0 += 4) {
      if (P == 0) {
#pragma unroll
        for (int s = 0; s < 4; s++) { float4 *c = base + (size_t)lane * row_floats4 + (size_t)(k0 + s) * 2; c[0] = v; c[1] = v; v.x += 1.f; }
      } else if (P == 1) {
        float4 *c = base + (size_t)lane * row_floats4 + (size_t)k0 * 2;
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = v;
        v.x += 1.f;
      } else if (P == 2) {
        // instruction j (0..7): quad g = lane / 4 serves row 4 * (g % ... ) -- 16 quads x 8 instructions cover 64 rows x 2 halves of 64 B
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int row = (lane >> 2) * 4 + (j >> 1), piece = (j & 1) * 4 + (lane & 3);
          base[(size_t)row * row_floats4 + (size_t)k0 * 2 + piece] = v;
        }
        v.x += 1.f;
      } else if (P == 3) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int row = (lane >> 3) * 8 + j, piece = lane & 7;
          base[(size_t)row * row_floats4 + (size_t)k0 * 2 + piece] = v;
        }
        v.x += 1.f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) base[((size_t)(k0 / 4) * 8 + j) * 64 + lane] = v;
        v.x += 1.f;
      }
    }
  }
}

template <int P>
void run(float4 *d, int blocks, int nodes, int row_f4) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int rounds = 12;
  hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(128), 0, 0, d, 64, 1, row_f4, nodes);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(128), 0, 0, d, 64, rounds, row_f4, nodes);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * 2 * 64 * (nodes / 4 * 4) * 32.0 * rounds;
  printf("pattern %d: %7.3f ms  %8.1f GB/s\n", P, ms, bytes / ms / 1e6);
}

int main() {
  const int blocks = 512, nodes = 144, row_f4 = 146 * 2;       // M = 145: rows of 146 cells x 32 B
  float4 *d; hipMalloc(&d, (size_t)blocks * 2 * 64 * row_f4 * sizeof(float4));
  run<0>(d, blocks, nodes, row_f4); run<1>(d, blocks, nodes, row_f4); run<2>(d, blocks, nodes, row_f4); run<3>(d, blocks, nodes, row_f4); run<4>(d, blocks, nodes, row_f4);
  run<0>(d, blocks, nodes, row_f4); run<2>(d, blocks, nodes, row_f4);
  return 0;
}
