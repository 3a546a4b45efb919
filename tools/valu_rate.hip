// Issue rate of the packed binary16 ops of the SSV row on gfx950: v_pk_add_f16 (clamp), v_pk_max_f16, v_pk_maximum3_f16.
// Each wave runs a long chain-free stream of one op on 16 independent registers; 4 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
  uint32_t r[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 16 + i;
  uint32_t a = 0x3c003c00u, b = threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) asm volatile("v_pk_add_f16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(a));
      if (OP == 1) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 2) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (OP == 3) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 4) asm volatile("v_max3_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (OP == 7) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 8) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, uint32_t *d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = (double)blocks * 4 * iters * 16;           // per-wave instructions
  // a SIMD issues one 64-lane instruction per 4 cycles at full rate: cycles per instruction per SIMD
  printf("%-22s %8.3f ms   %.2f Ginst/s (wave-instructions)\n", name, ms, wave_insts / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int per_cu = argc > 1 ? atoi(argv[1]) : 4;              // blocks of 4 waves per CU = waves per SIMD
  const int blocks = p.multiProcessorCount * per_cu, iters = 20000;
  printf("%d waves per SIMD\n", per_cu);
  uint32_t *d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  printf("%s: %d CUs, %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  run<0>("v_pk_add_f16 clamp", d, blocks, iters);
  run<1>("v_pk_max_f16", d, blocks, iters);
  run<2>("v_pk_maximum3_f16", d, blocks, iters);
  run<3>("v_pk_max_i16", d, blocks, iters);
  run<4>("v_max3_f16", d, blocks, iters);
  run<5>("v_add_u32", d, blocks, iters);
  run<6>("v_fma_f32", d, blocks, iters);
  run<7>("v_pk_add_u16", d, blocks, iters);
  run<8>("v_max_u32", d, blocks, iters);
  printf("one instruction per 4 cycles per SIMD = CUs x 4 SIMDs x clock / 4 = %.2f Ginst/s\n", p.multiProcessorCount * 4.0 * (p.clockRate / 1e6) / 4);
  return 0;
}
