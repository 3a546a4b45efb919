// probe: what decides the speed of a dependent LDS log-sum chain that runs alone on its CU?  One chain wave per block, a block per CU
// (64 KB of LDS each), every block times the Forward chain's node loop of bath_fs_chain.hip (clock64) and reports its hardware id.
// The other waves of the block (SPINTHREADS = 64 .. 256: none, one, three) wait for the chain in one of four ways (SPINKIND):
//   0 a VALU loop, 1 s_nop loops, 2 `s_sleep 1` between polls of an LDS word, 3 v_mov + s_nop; SPINTHREADS=64: no other waves.
// Measured on MI355X, clocks per node over 256 blocks (min / median / p95 / max):
//   no other waves (or waves parked at s_barrier)  198 / 213 / 251 / 267   -- and which blocks are slow changes from launch to launch
//   VALU loop on the other three SIMDs             226 / 229 / 231 / 231
//   s_nop loops                                    199 / 201 / 203 / 221
//   s_sleep 1 between polls                        199 / 202 / 203 / 203   <- what chain_keepalive() in bath_fs_chain.hip does
//   one other wave only (s_sleep polls)            208 / 223 / 225 / 225
//   hipcc --offload-arch=gfx950 -O3 -DSPINKIND=2 -DSPINTHREADS=256 -o tools/_ab/lsbench7 tools/lsbench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <map>
#include <algorithm>
#define N 2048
#ifndef SPINKIND
#define SPINKIND 0
#endif
#ifndef SPINTHREADS
#define SPINTHREADS 256
#endif
#define BATH_LS_INDEX(a, x, y)                      \
  "v_sub_f32 " a ", " x ", " y "\n\t"               \
  "v_min_f32_e64 " a ", |" a "|, %[c15]\n\t"        \
  "v_mul_f32 " a ", 0x447a0000, " a "\n\t"          \
  "v_cvt_i32_f32 " a ", " a "\n\t"                  \
  "v_lshl_add_u32 " a ", " a ", 2, %[tbl]\n\t"      \
  "ds_read_b32 " a ", " a "\n\t"
#define BATH_FWD_NODE(MK, TX, TY, MN, UX, UY)                                                                 \
  BATH_LS_INDEX("%[a1]", "%[d]", "%[e]")                                                                      \
  "s_waitcnt lgkmcnt(1)\n\t"                                                                                  \
  "v_add_f32 %[u], " MK ", " TX "\n\t"                                                                        \
  "v_add_f32 %[w], %[d], " TY "\n\t"                                                                          \
  "ds_write_b32 %[st], %[d]\n\t"                                                                              \
  BATH_LS_INDEX("%[a2]", "%[u]", "%[w]")                                                                      \
  "v_max_f32 %[mx1], %[d], %[e]\n\t"                                                                          \
  "v_max_f32 %[mxd], %[u], %[w]\n\t"                                                                          \
  "s_waitcnt lgkmcnt(2)\n\t"                                                                                  \
  "v_add_f32 %[x], %[mx1], %[a1]\n\t"                                                                         \
  BATH_LS_INDEX("%[a1]", MK, "%[x]")                                                                          \
  "ds_read_b32 " MN ", %[st] offset:4\n\t"                                                                    \
  "ds_read_b32 " UX ", %[tp]\n\t"                                                                             \
  "ds_read_b32 " UY ", %[tp] offset:4\n\t"                                                                    \
  "v_max_f32 %[mx1], " MK ", %[x]\n\t"                                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                                                                  \
  "v_add_f32 %[d], %[mxd], %[a2]\n\t"                                                                         \
  "v_add_u32 %[st], 4, %[st]\n\t"                                                                             \
  "v_add_u32 %[tp], 32, %[tp]\n\t"                                                                            \
  "s_waitcnt lgkmcnt(3)\n\t"                                                                                  \
  "v_add_f32 %[e], %[mx1], %[a1]\n\t"

__global__ __launch_bounds__(256) void k(const float *tblg, const float *xs, float *out, unsigned *ids, long long *cyc) {
  extern __shared__ float tbl[];
  __shared__ float sx[N + 8]; __shared__ float stf[(N + 8) * 8];
  for (int i = threadIdx.x; i < 16000; i += blockDim.x) tbl[i] = i < 15700 ? tblg[i] : 0.f;
  for (int i = threadIdx.x; i < (N + 8) * 8; i += blockDim.x) stf[i] = -0.5f - (i % 7) * 0.1f;
  for (int i = threadIdx.x; i < N + 8; i += blockDim.x) sx[i] = xs[i % 1024];
  __syncthreads();
  float e = -3.0f + (threadIdx.x & 15) * 0.37f;
  const float c15 = 15.999f;
  unsigned tb = (unsigned)(size_t)tbl, xa = (unsigned)(size_t)sx, tpa = (unsigned)(size_t)stf;
  float x = sx[0], dd = -4.f, tx = -1.f, ty = -0.5f;
  __shared__ volatile int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (threadIdx.x >= 64) {                                       // waves 1..3: keep their SIMDs' VALUs busy while wave 0 runs its chain
    float a = threadIdx.x * 0.001f, b = 1.0001f;
#if SPINKIND == 0
    while (!done) { for (int j = 0; j < 64; j++) { a = a * b + 0.5f; b = b * 0.99999f + 0.00001f; } }
#elif SPINKIND == 1
    while (!done) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }
#elif SPINKIND == 2
    while (!done) { asm volatile("s_sleep 1" ::: "memory"); }
#elif SPINKIND == 3
    while (!done) { asm volatile("v_mov_b32 %0, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a) :: "memory"); }
#endif
    if (a == 12345.f) out[0] = a + b;
    return;
  }
  if (threadIdx.x >= 16) return;
  long long c0 = clock64();
  for (int i = 0; i < N; i += 2) {
    float Mn, ux, uy, a1, a2, u, w, mx1, mxd, xx;
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_FWD_NODE("%[Mk]", "%[tx]", "%[ty]", "%[Mn]", "%[ux]", "%[uy]")
                 BATH_FWD_NODE("%[Mn]", "%[ux]", "%[uy]", "%[Mk]", "%[tx]", "%[ty]")
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(e), [d] "+v"(dd), [Mk] "+v"(x), [tx] "+v"(tx), [ty] "+v"(ty), [st] "+v"(xa), [tp] "+v"(tpa),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(xx)
                 : [tbl] "s"(tb), [c15] "s"(c15)
                 : "memory");
  }
  long long c1 = clock64();
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  out[blockIdx.x * 16 + threadIdx.x] = e + x;
  if (threadIdx.x == 0) { done = 1; cyc[blockIdx.x] = c1 - c0; ids[blockIdx.x * 2] = hwid; ids[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
  std::vector<float> t(16000), xs(1024);
  for (int i = 0; i < 16000; i++) t[i] = (float)log(1.0 + exp(-i / 1000.0));
  for (int i = 0; i < 1024; i++) xs[i] = -5.0f + (float)((i * 7919) % 1000) * 0.004f;
  const int B = 256;
  float *dt, *dx, *dout; long long *dc; unsigned *did;
  hipMalloc(&dt, 64000); hipMalloc(&dx, 4096); hipMalloc(&dout, B * 16 * 4); hipMalloc(&dc, B * 8); hipMalloc(&did, B * 8);
  hipMemcpy(dt, t.data(), 64000, hipMemcpyHostToDevice); hipMemcpy(dx, xs.data(), 4096, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64000);
  std::map<unsigned, std::vector<double>> by_cu;
  for (int rep = 0; rep < 6; rep++) {
    hipLaunchKernelGGL(k, dim3(B), dim3(SPINTHREADS), 64000, 0, dt, dx, dout, did, dc);
    hipDeviceSynchronize();
    std::vector<long long> c(B); std::vector<unsigned> id(B * 2);
    hipMemcpy(c.data(), dc, B * 8, hipMemcpyDeviceToHost); hipMemcpy(id.data(), did, B * 8, hipMemcpyDeviceToHost);
    std::vector<double> v(B);
    for (int b = 0; b < B; b++) { v[b] = (double)c[b] / N; const unsigned key = ((id[2 * b + 1] & 0xf) << 16) | ((id[2 * b] >> 8) & 0xfff0) | ((id[2 * b] >> 4) & 0x3); by_cu[key].push_back(v[b]); }
    std::vector<double> s = v; std::sort(s.begin(), s.end());
    printf("launch %d: ticks per node min %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  max %.1f\n", rep, s[0], s[B / 4], s[B / 2], s[3 * B / 4], s[B * 95 / 100], s[B - 1]);
  }
  // per (XCC, SE/CU, SIMD): spread across launches
  int stable_slow = 0, n = 0; double worst_spread = 0;
  for (auto &kv : by_cu) { if (kv.second.size() < 3) continue; n++; double lo = *std::min_element(kv.second.begin(), kv.second.end()), hi = *std::max_element(kv.second.begin(), kv.second.end()); if (lo > 225) stable_slow++; worst_spread = std::max(worst_spread, hi - lo); }
  printf("%zu distinct (xcc, cu, simd) keys, %d seen >= 3 times; always slower than 225 ticks: %d; largest spread on one key %.1f ticks\n", by_cu.size(), n, stable_slow, worst_spread);
  int shown = 0;
  for (auto &kv : by_cu) { if (kv.second.size() >= 4 && shown < 12) { printf("  key %05x:", kv.first); for (double x : kv.second) printf(" %.0f", x); printf("\n"); shown++; } }
  return 0;
}
