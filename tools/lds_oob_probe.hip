// probe: what a ds_read_b32 beyond the block's LDS allocation returns on gfx950 (the log-sum's index clamp could be the bounds check itself)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, const unsigned *addrs, int n, int words) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xabcd0000u + i;
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    unsigned a = addrs[i], v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[i] = v;
  }
}
int main() {
  for (int bytes : {1024, 65536, 66000, 110 * 1024, 160 * 1024}) {
    std::vector<unsigned> a;
    for (unsigned off : {0u, 4u, 16u, 64u, 252u, 256u, 508u, 512u, 1020u, 1024u, 2048u, 4096u, 65536u, 1u << 20, 1u << 24, 0x7ffffffcu, 0xfffffffcu})
      a.push_back((unsigned)bytes - 4 + off);
    a.push_back(0x7fffffffu * 4u);                      // what (int)(inf * 1000) << 2 gives
    unsigned *da, *dout;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dout, a.size() * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), bytes, 0, dout, da, (int)a.size(), bytes / 4);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned> o(a.size());
    hipMemcpy(o.data(), dout, a.size() * 4, hipMemcpyDeviceToHost);
    printf("LDS %d bytes (%s):", bytes, hipGetErrorString(e));
    for (size_t i = 0; i < a.size(); i++) printf(" [+%lld]=%08x", (long long)a[i] - (bytes - 4), o[i]);
    printf("\n");
  }
  return 0;
}
