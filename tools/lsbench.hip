// micro-benchmark: cycles per dependent table log-sum on one wave, variants of the instruction sequence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define N 1024
#define BATH_LS_INDEX(a, x, y)                      \
  "v_sub_f32 " a ", " x ", " y "\n\t"               \
  "v_min_f32_e64 " a ", |" a "|, %[c15]\n\t"        \
  "v_mul_f32 " a ", 0x447a0000, " a "\n\t"          \
  "v_cvt_i32_f32 " a ", " a "\n\t"                  \
  "v_lshl_add_u32 " a ", " a ", 2, %[tbl]\n\t"      \
  "ds_read_b32 " a ", " a "\n\t"

#define BATH_FWD_NODE(MK, TX, TY, MN, UX, UY)                                                                 \
  BATH_LS_INDEX("%[a1]", "%[d]", "%[e]")                        /* L1: LS(D_k, E) */                          \
  "s_waitcnt lgkmcnt(1)\n\t"                                    /* the loads of this node's M, tMD, tDD */    \
  "v_add_f32 %[u], " MK ", " TX "\n\t"                                                                        \
  "v_add_f32 %[w], %[d], " TY "\n\t"                                                                          \
  "ds_write_b32 %[st], %[d]\n\t"                                /* W: D_k */                                  \
  BATH_LS_INDEX("%[a2]", "%[u]", "%[w]")                        /* L2: D_{k+1} */                             \
  "v_max_f32 %[mx1], %[d], %[e]\n\t"                                                                          \
  "v_max_f32 %[mxd], %[u], %[w]\n\t"                                                                          \
  "s_waitcnt lgkmcnt(2)\n\t"                                    /* L1 */                                      \
  "v_add_f32 %[x], %[mx1], %[a1]\n\t"                                                                         \
  BATH_LS_INDEX("%[a1]", MK, "%[x]")                            /* L3: LS(M_k, .) */                          \
  "ds_read_b32 " MN ", %[st] offset:4\n\t"                                                                    \
  "ds_read_b32 " UX ", %[tp]\n\t"                                                                             \
  "ds_read_b32 " UY ", %[tp] offset:4\n\t"                                                                    \
  "v_max_f32 %[mx1], " MK ", %[x]\n\t"                                                                        \
  "s_waitcnt lgkmcnt(4)\n\t"                                    /* W, L2 */                                   \
  "v_add_f32 %[d], %[mxd], %[a2]\n\t"                                                                         \
  "v_add_u32 %[st], 4, %[st]\n\t"                                                                             \
  "v_add_u32 %[tp], 32, %[tp]\n\t"                                                                            \
  "s_waitcnt lgkmcnt(3)\n\t"                                    /* L3 */                                      \
  "v_add_f32 %[e], %[mx1], %[a1]\n\t"


// the Backward chains of bath_fs_chain.hip (round 4: why is a Backward node 190-220 ns when its three dependent log-sums are 3 x 40?)
#define BATH_BSUM_NODE(V, VN)                                                          \
  BATH_LS_INDEX("%[a1]", "%[b]", V)                                                    \
  "s_waitcnt lgkmcnt(1)\n\t"                                                           \
  "v_add_f32 " VN ", %[sN], %[tN]\n\t"                                                 \
  "ds_read_b32 %[sN], %[st] offset:8\n\t"                                              \
  "ds_read_b32 %[tN], %[tp] offset:64\n\t"                                             \
  "v_max_f32 %[mx], %[b], " V "\n\t"                                                   \
  "v_add_u32 %[st], 4, %[st]\n\t"                                                      \
  "v_add_u32 %[tp], 32, %[tp]\n\t"                                                     \
  "s_waitcnt lgkmcnt(2)\n\t"                                                           \
  "v_add_f32 %[b], %[mx], %[a1]\n\t"

#define BATH_BWD_D_NODE(IVN, TX, TY, IVQ, UX, UY)                                      \
  "v_add_f32 %[u], %[dn], " TX "\n\t"                                                  \
  "v_add_f32 %[bs], " IVN ", " TY "\n\t"                                               \
  "v_cndmask_b32_e64 %[p1], %[u], %[bs], %[mid]\n\t"                                   \
  "v_cndmask_b32_e64 %[p2], %[bs], %[u], %[mid]\n\t"                                   \
  BATH_LS_INDEX("%[a1]", "%[xE]", "%[p1]")                                             \
  "ds_read_b32 " IVQ ", %[st]\n\t"                                                     \
  "ds_read_b32 " UX ", %[tp]\n\t"                                                      \
  "ds_read_b32 " UY ", %[tp] offset:4\n\t"                                             \
  "v_max_f32 %[mx1], %[xE], %[p1]\n\t"                                                 \
  "s_waitcnt lgkmcnt(3)\n\t"                                                           \
  "v_add_f32 %[x], %[mx1], %[a1]\n\t"                                                  \
  BATH_LS_INDEX("%[a1]", "%[x]", "%[p2]")                                              \
  "v_max_f32 %[mx1], %[x], %[p2]\n\t"                                                  \
  "v_add_u32 %[st], -4, %[st]\n\t"                                                     \
  "v_add_u32 %[tp], -32, %[tp]\n\t"                                                    \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "v_add_f32 %[dn], %[mx1], %[a1]\n\t"                                                 \
  "ds_write_b32 %[st], %[dn] offset:8\n\t"

template <int V>
__global__ void k(const float *tblg, const float *xs, float *out, long long *cyc, int lanes) {
  extern __shared__ float tbl[];
  for (int i = threadIdx.x; i < 16000; i += blockDim.x) tbl[i] = i < 15700 ? tblg[i] : 0.f;
  __shared__ float sx[N + 8]; __shared__ float stf[(N + 8) * 8];
  for (int i = threadIdx.x; i < (N + 8) * 8; i += blockDim.x) stf[i] = -0.5f - (i % 7) * 0.1f;
  for (int i = threadIdx.x; i < N + 8; i += blockDim.x) sx[i] = xs[i % N];
  __syncthreads();
  if ((int)threadIdx.x >= lanes) return;
  float e = -3.0f + threadIdx.x * 0.01f;
  const float c15 = 15.999f;
  unsigned tb = (unsigned)(size_t)tbl, xa = (unsigned)(size_t)sx;
  float x = sx[0], xn, a, mx; float dd = -4.f, tx = -1.f, ty = -0.5f; unsigned tpa = (unsigned)(size_t)stf;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int i = 0; i < N; i++) {
    if (V == 0) {        // full LS, next x prefetched in the shadow
      asm volatile("v_sub_f32 %[a], %[x], %[e]\n\tv_min_f32_e64 %[a], |%[a]|, %[c15]\n\tv_mul_f32 %[a], 0x447a0000, %[a]\n\tv_cvt_i32_f32 %[a], %[a]\n\t"
                   "v_lshl_add_u32 %[a], %[a], 2, %[tb]\n\tds_read_b32 %[a], %[a]\n\tds_read_b32 %[xn], %[xa] offset:4\n\tv_max_f32 %[mx], %[x], %[e]\n\tv_add_u32 %[xa], 4, %[xa]\n\t"
                   "s_waitcnt lgkmcnt(1)\n\tv_add_f32 %[e], %[mx], %[a]\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %[x], %[xn]"
                   : [e] "+v"(e), [x] "+v"(x), [xa] "+v"(xa), [a] "=&v"(a), [mx] "=&v"(mx), [xn] "=&v"(xn) : [c15] "s"(c15), [tb] "s"(tb) : "memory");
    } else if (V == 1) { // no table read: a stays the index (as float bits) -> measures VALU chain only
      asm volatile("v_sub_f32 %[a], %[x], %[e]\n\tv_min_f32_e64 %[a], |%[a]|, %[c15]\n\tv_mul_f32 %[a], 0x447a0000, %[a]\n\tv_cvt_i32_f32 %[a], %[a]\n\t"
                   "v_lshl_add_u32 %[a], %[a], 2, %[tb]\n\tv_cvt_f32_i32 %[a], %[a]\n\tv_max_f32 %[mx], %[x], %[e]\n\t"
                   "v_fma_f32 %[e], %[a], 0, %[mx]\n\t"
                   : [e] "+v"(e), [x] "+v"(x), [xa] "+v"(xa), [a] "=&v"(a), [mx] "=&v"(mx), [xn] "=&v"(xn) : [c15] "s"(c15), [tb] "s"(tb) : "memory");
    } else if (V == 2) { // dependent ds_read chain only: address from previous value
      asm volatile("v_and_b32 %[a], 0xfffc, %[e]\n\tds_read_b32 %[e], %[a]\n\ts_waitcnt lgkmcnt(0)"
                   : [e] "+v"(e), [a] "=&v"(a) :: "memory");
    } else if (V == 3) { // 6 dependent v_add_f32
      asm volatile("v_add_f32 %[e], 1.0, %[e]\n\tv_add_f32 %[e], 1.0, %[e]\n\tv_add_f32 %[e], 1.0, %[e]\n\tv_add_f32 %[e], 1.0, %[e]\n\tv_add_f32 %[e], 1.0, %[e]\n\tv_add_f32 %[e], 1.0, %[e]"
                   : [e] "+v"(e) :: "memory");
    } else if (V == 4) { // sub, min, mul, cvt, lshl_add dependent only (5 ops), result fed back through cvt
      asm volatile("v_sub_f32 %[a], %[x], %[e]\n\tv_min_f32_e64 %[a], |%[a]|, %[c15]\n\tv_mul_f32 %[a], 0x447a0000, %[a]\n\tv_cvt_i32_f32 %[a], %[a]\n\t"
                   "v_lshl_add_u32 %[e], %[a], 2, %[tb]\n\t"
                   : [e] "+v"(e), [a] "=&v"(a) : [x] "v"(x), [c15] "s"(c15), [tb] "s"(tb) : "memory");
    } else if (V == 5) { // 6 dependent v_cvt
      asm volatile("v_cvt_i32_f32 %[e], %[e]\n\tv_cvt_f32_i32 %[e], %[e]\n\tv_cvt_i32_f32 %[e], %[e]\n\tv_cvt_f32_i32 %[e], %[e]\n\tv_cvt_i32_f32 %[e], %[e]\n\tv_cvt_f32_i32 %[e], %[e]"
                   : [e] "+v"(e) :: "memory");
    } else if (V == 6) { // 6 dependent v_mul with literal
      asm volatile("v_mul_f32 %[e], 0x3f800001, %[e]\n\tv_mul_f32 %[e], 0x3f800001, %[e]\n\tv_mul_f32 %[e], 0x3f800001, %[e]\n\tv_mul_f32 %[e], 0x3f800001, %[e]\n\tv_mul_f32 %[e], 0x3f800001, %[e]\n\tv_mul_f32 %[e], 0x3f800001, %[e]"
                   : [e] "+v"(e) :: "memory");

    } else if (V == 8) {
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
      i++;
    } else if (V == 9) {   // Backward's B sum, two nodes per iteration
      float vn, a1, mx; static_cast<void>(a);
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   BATH_BSUM_NODE("%[v]", "%[vn]")
                   BATH_BSUM_NODE("%[vn]", "%[v]")
                   "s_waitcnt lgkmcnt(0)"
                   : [b] "+v"(e), [v] "+v"(x), [sN] "+v"(tx), [tN] "+v"(ty), [st] "+v"(xa), [tp] "+v"(tpa), [vn] "=&v"(vn), [a1] "=&v"(a1), [mx] "=&v"(mx)
                   : [tbl] "s"(tb), [c15] "s"(c15)
                   : "memory");
      i++;
    } else if (V == 10) {  // Backward's D chain, two nodes per iteration (addresses walk down; restarted every 256 nodes)
      float ivq, ux, uy, u, bs, p1, p2, a1, mx1, xx;
      const unsigned long long mid = 0ull;
      if ((i & 255) == 0) { xa = (unsigned)(size_t)(sx + 600); tpa = (unsigned)(size_t)(stf + 600 * 8); }
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                   BATH_BWD_D_NODE("%[ivn]", "%[tx]", "%[ty]", "%[ivq]", "%[ux]", "%[uy]")
                   BATH_BWD_D_NODE("%[ivk]", "%[ux]", "%[uy]", "%[ivk]", "%[tx]", "%[ty]")
                   "v_mov_b32 %[ivn], %[ivq]\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : [dn] "+v"(dd), [ivn] "+v"(x), [ivk] "+v"(e), [tx] "+v"(tx), [ty] "+v"(ty), [st] "+v"(xa), [tp] "+v"(tpa),
                     [ivq] "=&v"(ivq), [ux] "=&v"(ux), [uy] "=&v"(uy), [u] "=&v"(u), [bs] "=&v"(bs), [p1] "=&v"(p1), [p2] "=&v"(p2),
                     [a1] "=&v"(a1), [mx1] "=&v"(mx1), [x] "=&v"(xx)
                   : [xE] "v"(-2.5f), [mid] "s"(mid), [tbl] "s"(tb), [c15] "s"(c15)
                   : "memory");
      i++;
    } else if (V == 7) { // 6 dependent VOP3 min with abs + sgpr
      asm volatile("v_min_f32_e64 %[e], |%[e]|, %[c15]\n\tv_min_f32_e64 %[e], |%[e]|, %[c15]\n\tv_min_f32_e64 %[e], |%[e]|, %[c15]\n\tv_min_f32_e64 %[e], |%[e]|, %[c15]\n\tv_min_f32_e64 %[e], |%[e]|, %[c15]\n\tv_min_f32_e64 %[e], |%[e]|, %[c15]"
                   : [e] "+v"(e) : [c15] "s"(c15) : "memory");
    }
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  out[threadIdx.x] = e + x;
  if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
template <int V> void run(const char *name, const float *tbl, const float *xs, float *out, long long *cyc, int lanes) {
  hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 64000);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 64000, 0, tbl, xs, out, cyc, lanes); hipDeviceSynchronize(); }
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("%-44s lanes %2d: %7.1f clock64 ticks/iter, %7.1f ns/iter (wall clock %d kHz)\n", name, lanes, (double)h[0] / N, (double)h[1] / N * 1e6 / rate, rate);
}
int main() {
  std::vector<float> t(16000), xs(N);
  for (int i = 0; i < 16000; i++) t[i] = (float)log(1.0 + exp(-i / 1000.0));
  for (int i = 0; i < N; i++) xs[i] = -5.0f + (float)((i * 7919) % 1000) * 0.004f;
  float *dt, *dx, *dout; long long *dc;
  hipMalloc(&dt, 64000); hipMalloc(&dx, N * 4); hipMalloc(&dout, 256); hipMalloc(&dc, 16);
  hipMemcpy(dt, t.data(), 64000, hipMemcpyHostToDevice); hipMemcpy(dx, xs.data(), N * 4, hipMemcpyHostToDevice);
  for (int lanes : {64, 16, 1}) {
    run<0>("full LS (prefetch in shadow)", dt, dx, dout, dc, lanes);
    run<1>("LS without the table read", dt, dx, dout, dc, lanes);
    run<2>("dependent and+ds_read", dt, dx, dout, dc, lanes);
    run<3>("6 dependent v_add_f32", dt, dx, dout, dc, lanes);
    run<4>("sub,min,mul,cvt,lshl_add dependent", dt, dx, dout, dc, lanes);
    run<5>("6 dependent v_cvt", dt, dx, dout, dc, lanes);
    run<6>("6 dependent v_mul literal", dt, dx, dout, dc, lanes);
    run<7>("6 dependent v_min_e64 |.|,sgpr", dt, dx, dout, dc, lanes);
    run<8>("FWD_NODE (per node)", dt, dx, dout, dc, lanes);
    run<9>("BSUM_NODE (per node)", dt, dx, dout, dc, lanes);
    run<10>("BWD_D_NODE (per node)", dt, dx, dout, dc, lanes);
  }
  return 0;
}
