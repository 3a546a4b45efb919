// Micro-benchmark: sustained issue rate of the VALU instructions the SSV kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
#define N_ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed) {
  unsigned a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
  const unsigned c = seed | 1;
  for (int it = 0; it < N_ITERS; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) a[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, c)));
      if (OP == 1) a[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, c)));
      if (OP == 2) a[i] = __builtin_amdgcn_alignbit(a[i], c, 16);
      if (OP == 3) a[i] = (unsigned)max((int)a[i], (int)c);
      if (OP == 4) a[i] = a[i] - c;
      if (OP == 5) a[i] = __builtin_bit_cast(unsigned, fmaf(__builtin_bit_cast(float, a[i]), 1.0001f, 0.5f));
      if (OP == 6) a[i] = (unsigned)__builtin_elementwise_sub_sat((int)a[i], (int)c);
      if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], c, 0x05040100u);
      asm volatile("" : "+v"(a[i]));
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, unsigned *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU: 8 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waveinst = (double)blocks * 4 * N_ITERS * 16;
  double per_simd = waveinst / (256.0 * 4);
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
  unsigned *d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_pk_max_i16", d); run<1>("v_pk_sub_i16 clamp", d); run<2>("v_alignbit_b32", d); run<3>("v_max_i32", d);
  run<4>("v_sub_u32", d); run<5>("v_fma_f32", d); run<6>("v_sub_i32 clamp", d); run<7>("v_perm_b32", d);
  return 0;
}
