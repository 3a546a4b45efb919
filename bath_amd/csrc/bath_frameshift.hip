// bath_frameshift.hip -- frameshift-aware Forward/Backward, posterior decoding, optimal accuracy and
// null2 for gfx950, batched over DNA windows / envelopes.
//
//   fs3_fwd_kernel   <- p7_ForwardParser_Frameshift_3Codons  impl_sse/fwdback_fs.c:97
//                       (scalar twin p7_GForwardParser_Frameshift_3Codons generic_fwdback_frameshift.c:451)
//   fs_bwd_kernel<3> <- p7_BackwardParser_Frameshift_3Codons impl_sse/fwdback_fs.c:565  / generic :1422
//   fs5_fwd_kernel   <- p7_Forward_Frameshift                impl_sse/fwdback_fs.c:2054 / generic :64
//   fs_bwd_kernel<5> <- p7_Backward_Frameshift               impl_sse/fwdback_fs.c:2634 / generic :1035
//   fs5_decode_kernel<- p7_Decoding_Frameshift               generic_decoding_frameshift.c:36
//   fs5_oa_kernel    <- p7_OptimalAccuracy_Frameshift (fill) generic_optacc_frameshift.c:53
//   fs5_null2_kernel <- p7_Null2_fs_ByExpectation            generic_null2_frameshift.c:46
//
// One wavefront owns one DNA window; lanes own contiguous blocks of model nodes; the nucleotide
// recurrence (rows i-1..i-5) lives in registers as ring buffers; the D->D chain along the model and
// the E-state sum are wavefront scans/reductions.  Arithmetic is the log-space arithmetic of the
// generic reference with p7_FLogsum's 16000-entry table (logsum.c:105) held in LDS, so every
// individual log-sum is bit-identical to the reference's; only the ASSOCIATION of the sums along the
// model differs (scan instead of a serial loop), which with a 0.001-nat table moves scores by
// O(1e-3) nats (the reference's own SIMD-vs-generic tolerance is 1.0 nat, fwdback_fs.c:3189).
#include <cmath>
#include <cstring>
#include <vector>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

#include "bath_fs_device.hpp"

namespace bath {

// D(i,k) = LS(M(i,k-1)+tMD(k-1), D(i,k-1)+tDD(k-1)) for this lane's nodes, chained across lanes.
// md[c] = M(i,node_c)+tMD(node_c) and dd[c] = tDD(node_c) describe the step OUT of node c.
// Returns D at the lane's nodes in Dout[]; the step into the lane's first node comes from the previous lane.
template <int C, bool EXACT>
__device__ __forceinline__ void d_chain_fwd(const float (&md)[C], const float (&dd)[C], float (&Dout)[C], int lane, const float *tbl) {
  // lane function f(x) = LS(A, x + B): value handed to the next lane's first node
  float A = -INFINITY, B = 0.f;
#pragma unroll
  for (int c = 0; c < C; c++) { A = flogsum<EXACT>(md[c], A + dd[c], tbl); B += dd[c]; }
#ifdef BATH_FS_BPERMUTE
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float Ap = __shfl_up(A, d, 64), Bp = __shfl_up(B, d, 64);
    const float An = flogsum<EXACT>(A, Ap + B, tbl);
    A = (lane >= d) ? An : A; B = (lane >= d) ? B + Bp : B;
  }
  float din = __shfl_up(A, 1, 64);
  if (lane == 0) din = -INFINITY;
#else
  // lanes without a source see the identity map (A = -inf, B = 0): LS(A, -inf + B) = A, B + 0 = B
#define BATH_DCHAIN_STEP(CTRL, MASK) { const float Ap = dpp_f<CTRL, MASK>(A, -INFINITY), Bp = dpp_f<CTRL, MASK>(B, 0.f); A = flogsum<EXACT>(A, Ap + B, tbl); B = B + Bp; }
  BATH_DCHAIN_STEP(0x111, 0xf) BATH_DCHAIN_STEP(0x112, 0xf) BATH_DCHAIN_STEP(0x114, 0xf) BATH_DCHAIN_STEP(0x118, 0xf)
  BATH_DCHAIN_STEP(0x142, 0xa) BATH_DCHAIN_STEP(0x143, 0xc)
#undef BATH_DCHAIN_STEP
  const float din = wave_shr1(A, -INFINITY);
  (void)lane;
#endif
  Dout[0] = din;
#pragma unroll
  for (int c = 1; c < C; c++) Dout[c] = flogsum<EXACT>(md[c - 1], Dout[c - 1] + dd[c - 1], tbl);
}

// BATH_LOGSUM_TABLE_SERIAL ("strict"): the reference's own order of the sums along the model (generic_fwdback_frameshift.c:
// 340-365, 577-590): D(i,k) from D(i,k-1) node by node and E(i) accumulated in the same walk, one lane after the other.
// With the truncating table every log-sum is then the reference's log-sum of the reference's operands: results are
// bit-identical to the generic reference, at the cost of a 64-step hand-off per row.  Returns E(i).
template <int C>
__device__ __forceinline__ float fwd_chain_strict(const float (&Mc)[C], const float (&md)[C], const float (&dd)[C], float (&Dout)[C], int lane, int M,
                                                  bool pair_first_at_M, const float *tbl) {
  const int nl = (M + C - 1) / C;                         // lanes that hold nodes
  float dnext = -INFINITY, e = -INFINITY;                 // D at the first node of the next lane; E after this lane's nodes
#pragma unroll
  for (int c = 0; c < C; c++) Dout[c] = -INFINITY;
  for (int l = 0; l < nl; l++) {
    const float din = (l == 0) ? -INFINITY : __shfl(dnext, l - 1, 64);
    const float ein = (l == 0) ? -INFINITY : __shfl(e, l - 1, 64);
    if (lane == l) {
      float dcur = din, ecur = ein;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = l * C + c + 1;
        if (node <= M) {
          Dout[c] = dcur;
          if (node == M && pair_first_at_M) ecur = flogsum<false>(flogsum<false>(Mc[c], dcur, tbl), ecur, tbl);   // :392
          else ecur = flogsum<false>(Mc[c], flogsum<false>(dcur, ecur, tbl), tbl);
          dcur = flogsum<false>(md[c], dcur + dd[c], tbl);
        }
      }
      dnext = dcur; e = ecur;
    }
  }
  return __shfl(e, nl - 1, 64);
}

// Backward, strict order: B(i) = a(1)+tBM(0), then LS(B, a(k)+tBM(k-1)) for k = 2..M (generic_fwdback_frameshift.c:1279-1283)
template <int C>
__device__ __forceinline__ float bwd_bsum_strict(const float (&term)[C], int lane, int M, const float *tbl) {
  const int nl = (M + C - 1) / C;
  float b = -INFINITY;
  for (int l = 0; l < nl; l++) {
    const float bin = (l == 0) ? -INFINITY : __shfl(b, 63 - (l - 1), 64);      // (Backward's lanes are in descending order: logical l = physical 63 - l)
    if (lane == l) {
      float cur = bin;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = l * C + c + 1;
        if (node <= M) cur = (node == 1) ? term[c] : flogsum<false>(cur, term[c], tbl);
      }
      b = cur;
    }
  }
  return __shfl(b, 63 - (nl - 1), 64);
}

// ... and the descending D chain: dstep(c, node, D(node+1)) -> D(node) is the row's own formula; returns D at the first node of
// the NEXT lane (what the lane's last node needs), every value computed in the reference's order.
template <int C, class F>
__device__ __forceinline__ float bwd_dnext_strict(F &&dstep, int lane, int M) {
  const int nl = (M + C - 1) / C;
  float dfirst = -INFINITY;
  for (int l = nl - 1; l >= 0; l--) {
    const float dn_in = (l == nl - 1) ? -INFINITY : __shfl(dfirst, 63 - (l + 1), 64);
    if (lane == l) {
      float dn = dn_in;
#pragma unroll
      for (int c = C - 1; c >= 0; c--) {
        const int node = l * C + c + 1;
        if (node <= M) dn = dstep(c, node, dn);
      }
      dfirst = dn;
    }
  }
  float dnext = wave_shr1(dfirst, -INFINITY);                                 // from the logical lane above = the physical lane below
  if (lane >= nl - 1) dnext = -INFINITY;
  return dnext;
}

// ---------------------------------------------------------------------------------------------
// 3-codon Forward parser.  Row convention of the reference's parser: IVX(i,k) collects the paths
// leaving row i-2 (generic_fwdback_frameshift.c:562-569), so codon lengths 2,3,4 read IVX(i), IVX(i-1), IVX(i-2).
// tf[node] = {tMM(k-1), tIM(k-1), tDM(k-1), tBM(k-1), tMD(k), tDD(k), tMI(k), tII(k)}
// ---------------------------------------------------------------------------------------------
template <int C, int MODE>
__global__ __launch_bounds__(kFsBlock, fs_min_waves(C)) void fs3_fwd_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                      float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs) {
  constexpr bool EXACT = (MODE == 1), STRICT = (MODE == 2);   // 0: table + scans, 1: exact log-sums, 2: table in the reference's serial order
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (p.M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int M = p.M;
  const int lane = threadIdx.x & 63;
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  for (int64_t job = fs_next_job(jobs, dna.n, lane); job >= 0; job = fs_next_job(jobs, dna.n, lane)) {
    const int L = dna.len[job];
    const uint8_t *d = dna.data + dna.off[job];
    float *xo = xmx ? xmx + xmx_off[job] : nullptr;
    if (L < 3) { if (lane == 0) sc[job] = -INFINITY; continue; }
    const float tNL = loop_tab[L / 3], tNM = move_tab[L / 3], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    // rows i-1, i-2, i-3 of M/I/D (index 0 = most recent) and IVX(i-1), IVX(i-2)
    float Mr[3][C], Ir[3][C], Dr[3][C], iv1[C], iv2[C];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Mr[r][c] = Ir[r][c] = Dr[r][c] = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; c++) iv1[c] = iv2[c] = -INFINITY;
    // specials of rows i-1, i-2, i-3
    float xN[3] = {0.f, 0.f, 0.f}, xJ[3] = {-INFINITY, -INFINITY, -INFINITY}, xC[3] = {-INFINITY, -INFINITY, -INFINITY}, xB[3] = {tNM, tNM, tNM};
    if (xo && lane == 0) for (int i = 0; i < 2; i++) { xo[i * 5 + 0] = -INFINITY; xo[i * 5 + 1] = 0.f; xo[i * 5 + 2] = -INFINITY; xo[i * 5 + 3] = tNM; xo[i * 5 + 4] = -INFINITY; }
    int v = 338, w = 338, x = (d[0] < 4) ? d[0] : 338;                // p7P_MAXCODONS3 marks a degenerate nucleotide
    float cL = -INFINITY, cL1 = -INFINITY, cL2 = -INFINITY;           // C(L), C(L-1), C(L-2) for the final score

    // The emission scores of a row depend on the nucleotides only, not on the DP state: they are fetched one row ahead, so
    // that their trip to L2 (the table has 0.2-1.5 MB) overlaps the previous row's log-sum chains instead of heading the
    // row's dependency chain.
    float e2n[C], e3n[C], e4n[C];
    auto fetch = [&](int xx, int ww, int vv, int uu) {
      const float *q2 = p.rsc + (size_t)imin(xx * 84 + ww * 21, 337) * p.pitch;
      const float *q3 = p.rsc + (size_t)imin(xx * 84 + ww * 21 + vv * 5 + 1, 336) * p.pitch;
      const float *q4 = p.rsc + (size_t)imin(xx * 84 + ww * 21 + vv * 5 + uu + 2, 337) * p.pitch;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node <= M) { e2n[c] = q2[node]; e3n[c] = q3[node]; e4n[c] = q4[node]; } else { e2n[c] = e3n[c] = e4n[c] = -INFINITY; }
      }
    };
    int xn = (d[1] < 4) ? d[1] : 338;                                  // nucleotide of row 2
    fetch(xn, x, w, v);
    for (int i = 2; i <= L; i++) {
      v = w; w = x; x = xn;
      float e2[C], e3[C], e4[C];
#pragma unroll
      for (int c = 0; c < C; c++) { e2[c] = e2n[c]; e3[c] = e3n[c]; e4[c] = e4n[c]; }
      if (i < L) { xn = (d[i] < 4) ? d[i] : 338; fetch(xn, x, w, v); }
      // values of row i-2 at node-1 for the lane's first node
      const float mIn = wave_shr1(Mr[1][C - 1], -INFINITY), iIn = wave_shr1(Ir[1][C - 1], -INFINITY), dIn = wave_shr1(Dr[1][C - 1], -INFINITY);
      float Mc[C], Ic[C], ivc[C], md[C], dd[C];
      float eloc = -INFINITY;
      // nodes beyond M read the all -inf transition row M+1 and -inf emissions: every value below comes out -inf without a
      // branch (a per-node `if (node <= M)` splits the row into exec-masked blocks and serialises the lane's log-sums)
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1, nd = imin(node, M + 1);
        const float4 ta = *reinterpret_cast<const float4 *>(s_tf + nd * 8);
        const float4 tb = *reinterpret_cast<const float4 *>(s_tf + nd * 8 + 4);
        const float m1 = (c == 0) ? mIn : Mr[1][c - 1], i1 = (c == 0) ? iIn : Ir[1][c - 1], d1 = (c == 0) ? dIn : Dr[1][c - 1];
        float iv;
        if (i == 2) iv = xB[1] + ta.w;                                           // row 2: IVX3(2,k) = B(0) + tBM (:503)
        else iv = LS(m1 + ta.x, LS(i1 + ta.y, LS(d1 + ta.z, xB[1] + ta.w)));     // from row i-2, B(i-2)
        ivc[c] = iv;
        float mv = iv + e2[c];
        if (i > 2) { mv = LS(mv, iv1[c] + e3[c]); mv = LS(mv, iv2[c] + e4[c]); }
        Mc[c] = mv;
        const float ins = LS(Mr[2][c] + tb.z, Ir[2][c] + tb.w);
        Ic[c] = (i > 2 && node < M) ? ins : -INFINITY;
        md[c] = mv + tb.x; dd[c] = tb.y;
      }
      float Dc[C];
      float xE;
      if constexpr (STRICT) xE = fwd_chain_strict<C>(Mc, md, dd, Dc, lane, M, false, s_tbl);
      else {
        d_chain_fwd<C, EXACT>(md, dd, Dc, lane, s_tbl);
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          Dc[c] = (node <= M) ? Dc[c] : -INFINITY;
          eloc = LS(Mc[c], LS(Dc[c], eloc));                        // nodes beyond M hold -inf: the sum is unchanged
        }
        xE = wave_logsum<EXACT>(eloc, s_tbl);
      }
      float nN, nJ, nC, nB;
      if (i == 2) { nN = 0.f; nJ = xE + tEL; nC = xE + tEM; }
      else { nN = xN[2] + tNL; nJ = LS(xJ[2] + tJL, xE + tEL); nC = LS(xC[2] + tCL, xE + tEM); }
      nB = LS(nN + tNM, nJ + tJM);
      if (xo && lane == 0) { xo[i * 5 + 0] = xE; xo[i * 5 + 1] = nN; xo[i * 5 + 2] = nJ; xo[i * 5 + 3] = nB; xo[i * 5 + 4] = nC; }
      xN[2] = xN[1]; xN[1] = xN[0]; xN[0] = nN;
      xJ[2] = xJ[1]; xJ[1] = xJ[0]; xJ[0] = nJ;
      xC[2] = xC[1]; xC[1] = xC[0]; xC[0] = nC;
      xB[2] = xB[1]; xB[1] = xB[0]; xB[0] = nB;
      cL2 = cL1; cL1 = cL; cL = nC;
#pragma unroll
      for (int c = 0; c < C; c++) {
        Mr[2][c] = Mr[1][c]; Mr[1][c] = Mr[0][c]; Mr[0][c] = Mc[c];
        Ir[2][c] = Ir[1][c]; Ir[1][c] = Ir[0][c]; Ir[0][c] = Ic[c];
        Dr[2][c] = Dr[1][c]; Dr[1][c] = Dr[0][c]; Dr[0][c] = Dc[c];
        iv2[c] = iv1[c]; iv1[c] = ivc[c];
      }
    }
    if (lane == 0) sc[job] = LS(cL, LS(cL1 + tCL, cL2 + tCL)) + tCM;
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------
// 5-codon Forward, full matrix: fwd[(i*(M+1)+k)*8 + {D,I,C0..C5}], xmx[i*5 + {E,N,J,B,C}].
// IVX(i,k) collects the paths leaving row i-1; codon length c reads IVX(i-c+1).
// c5_compat selects the ring slot the generic reference reads for 5-nt codons.
// ---------------------------------------------------------------------------------------------
template <int C, int MODE, bool UNIHIT = false>
__global__ __launch_bounds__(kFsBlock) void fs5_fwd_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                      float tEL, float tEM, int c5_compat, float *__restrict__ sc,
                                                      float *__restrict__ fwd, const int64_t *__restrict__ fwd_off, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off,
                                                      int cfg_len /* >= 0: the amino length the model is configured for, instead of L/3 */, FsJobs jobs,
                                                      int *__restrict__ done = nullptr /* host-visible: done[job] = 1 once the job's matrix and score have landed */) {
  constexpr bool EXACT = (MODE == 1), STRICT = (MODE == 2);   // 0: table + scans, 1: exact log-sums, 2: table in the reference's serial order
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (p.M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int M = p.M;
  const int lane = threadIdx.x & 63;
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  for (int64_t job = fs_next_job(jobs, dna.n, lane); job >= 0; job = fs_next_job(jobs, dna.n, lane)) {
    const int L = dna.len[job];
    const uint8_t *d = dna.data + dna.off[job];
    float *fo = fwd + fwd_off[job];
    float *xo = xmx + xmx_off[job];
    if (L < 5) {
      if (lane == 0) sc[job] = -INFINITY;
      if (done) { __threadfence_system(); if (lane == 0) __hip_atomic_store(done + job, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      continue;
    }
    const int Lc = cfg_len >= 0 ? cfg_len : L / 3;
    const float tNL = loop_tab[Lc], tNM = move_tab[Lc], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    float Mr[3][C], Ir[3][C], Dr1[C], iv[4][C];       // M,I of rows i-1..i-3 (C0 totals); D of row i-1; IVX(i-1..i-4)
#pragma unroll
    for (int c = 0; c < C; c++) {
      Mr[0][c] = Mr[1][c] = Mr[2][c] = Ir[0][c] = Ir[1][c] = Ir[2][c] = Dr1[c] = -INFINITY;
      iv[0][c] = iv[1][c] = iv[2][c] = iv[3][c] = -INFINITY;
    }
    // row 0
    for (int k = lane; k <= M; k += 64)
#pragma unroll
      for (int s = 0; s < 8; s++) fo[(size_t)k * 8 + s] = -INFINITY;
    if (lane == 0) { xo[0] = -INFINITY; xo[1] = 0.f; xo[2] = -INFINITY; xo[3] = tNM; xo[4] = -INFINITY; }
    float xN[3] = {0.f, 0.f, 0.f}, xJ[3] = {-INFINITY, -INFINITY, -INFINITY}, xC[3] = {-INFINITY, -INFINITY, -INFINITY};
    float xBprev = tNM;
    int t = 1367, u = 1367, v = 1367, w = 1367, x = 1367;
    float cL = -INFINITY, cL1 = -INFINITY, cL2 = -INFINITY;

    for (int i = 1; i <= L; i++) {
      t = u; u = v; v = w; w = x; x = (d[i - 1] < 4) ? d[i - 1] : 1367;
      const float *r1 = p.rsc + (size_t)imin(x * 341, 1366) * p.pitch;
      const float *r2 = p.rsc + (size_t)imin(x * 341 + w * 85 + 1, 1365) * p.pitch;
      const float *r3 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + 2, 1364) * p.pitch;
      const float *r4 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + u * 5 + 3, 1365) * p.pitch;
      const float *r5 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + u * 5 + t + 4, 1366) * p.pitch;
      const float mIn = wave_shr1(Mr[0][C - 1], -INFINITY), iIn = wave_shr1(Ir[0][C - 1], -INFINITY), dIn = wave_shr1(Dr1[C - 1], -INFINITY);
      float Mc[C], Ic[C], ivc[C], md[C], dd[C];
      float *row = fo + (size_t)i * (M + 1) * 8;
      if (lane == 0) {
#pragma unroll
        for (int s = 0; s < 8; s++) row[s] = -INFINITY;
      }
      // straight-line per node (see fs3_fwd_kernel): nodes beyond M read transition row M+1 (-inf) and emission column M
      // (finite, but added to a -inf IVX), so their values come out -inf; only the stores are guarded
      float cc[C][5];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1, nd = imin(node, M + 1), ne = imin(node, M);
        const float4 ta = *reinterpret_cast<const float4 *>(s_tf + nd * 8);
        const float4 tb = *reinterpret_cast<const float4 *>(s_tf + nd * 8 + 4);
        const float m1 = (c == 0) ? mIn : Mr[0][c - 1], i1 = (c == 0) ? iIn : Ir[0][c - 1], d1 = (c == 0) ? dIn : Dr1[c - 1];
        float ivn;
        if (i <= 2) ivn = xBprev + ta.w;                                          // rows 1,2: only B(i-1) enters (:109,:150)
        else ivn = LS(m1 + ta.x, LS(i1 + ta.y, LS(d1 + ta.z, xBprev + ta.w)));
        ivc[c] = ivn;
        const float c1 = ivn + r1[ne];
        const float c2 = (i >= 2) ? iv[0][c] + r2[ne] : -INFINITY;
        const float c3 = (i >= 3) ? iv[1][c] + r3[ne] : -INFINITY;
        const float c4 = (i >= 4) ? iv[2][c] + r4[ne] : -INFINITY;
        const float c5 = (i >= 5) ? (c5_compat ? ivn : iv[3][c]) + r5[ne] : -INFINITY;
        float c0;
        if (i == 1) c0 = c1;
        else if (i == 2) c0 = LS(c1, c2);
        else if (i < 5) c0 = LS(c1, LS(c2, LS(c3, c4)));
        else c0 = LS(LS(c1, LS(c2, c3)), LS(c4, c5));
        Mc[c] = c0;
        const float ins = LS(Mr[2][c] + tb.z, Ir[2][c] + tb.w);
        Ic[c] = (i >= 3 && node < M) ? ins : -INFINITY;
        md[c] = c0 + tb.x; dd[c] = tb.y;
        cc[c][0] = c1; cc[c][1] = c2; cc[c][2] = c3; cc[c][3] = c4; cc[c][4] = c5;
      }
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node <= M) {
          float *cell = row + (size_t)node * 8;
          cell[1] = Ic[c]; cell[2] = Mc[c]; cell[3] = cc[c][0]; cell[4] = cc[c][1]; cell[5] = cc[c][2]; cell[6] = cc[c][3]; cell[7] = cc[c][4];
        }
      }
      float Dc[C];
      float xE;
      if constexpr (STRICT) {
        xE = fwd_chain_strict<C>(Mc, md, dd, Dc, lane, M, i >= 5, s_tbl);
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          if (node <= M) row[(size_t)node * 8] = Dc[c];
        }
      } else {
        d_chain_fwd<C, EXACT>(md, dd, Dc, lane, s_tbl);
        float eloc = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          Dc[c] = (node <= M) ? Dc[c] : -INFINITY;
          eloc = LS(Mc[c], LS(Dc[c], eloc));
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          if (node <= M) row[(size_t)node * 8] = Dc[c];
        }
        xE = wave_logsum<EXACT>(eloc, s_tbl);
      }
      float nN, nJ, nC, nB;
      if (i <= 2) { nN = 0.f; nJ = xE + tEL; nC = xE + tEM; nB = tNM; }           // :126-132, :166-167
      else {
        nN = xN[2] + tNL; nJ = LS(xJ[2] + tJL, xE + tEL); nC = LS(xC[2] + tCL, xE + tEM);
        nB = LS(nN + tNM, nJ + tJM);
      }
      if constexpr (UNIHIT) {
        // unihit (tEL = -inf): J is unreachable, J(i) = -inf and B(i) = N(i) + tNM -- the values above, said without reading E(i).
        // The row's E reduction (C log-sums per lane + 6 across the wave) then feeds C(i) only and leaves the chain that the
        // next row waits for.
        nJ = -INFINITY; nB = nN + tNM;
      }
      if (lane == 0) { xo[i * 5 + 0] = xE; xo[i * 5 + 1] = nN; xo[i * 5 + 2] = nJ; xo[i * 5 + 3] = nB; xo[i * 5 + 4] = nC; }
      xN[2] = xN[1]; xN[1] = xN[0]; xN[0] = nN;
      xJ[2] = xJ[1]; xJ[1] = xJ[0]; xJ[0] = nJ;
      xC[2] = xC[1]; xC[1] = xC[0]; xC[0] = nC;
      xBprev = nB;
      cL2 = cL1; cL1 = cL; cL = nC;
#pragma unroll
      for (int c = 0; c < C; c++) {
        Mr[2][c] = Mr[1][c]; Mr[1][c] = Mr[0][c]; Mr[0][c] = Mc[c];
        Ir[2][c] = Ir[1][c]; Ir[1][c] = Ir[0][c]; Ir[0][c] = Ic[c];
        Dr1[c] = Dc[c];
        iv[3][c] = iv[2][c]; iv[2][c] = iv[1][c]; iv[1][c] = iv[0][c]; iv[0][c] = ivc[c];
      }
    }
    if (lane == 0) sc[job] = LS(cL, LS(cL1 + tCL, cL2 + tCL)) + tCM;
    if (done) {                                                  // every lane's stores first (system scope), then the flag
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(done + job, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------
// Backward, both codon systems (NCOD = 3: parser, no matrix stored; NCOD = 5: full, 3 cells per node).
// Only NCOD = 3 is instantiated since round 4: the envelopes' Backward is fs5_bwd_wf_kernel (bath_fs_wavefront.hip) in every mode.
// tb[node] = {tMD(k), tMI(k), tMM(k), tDD(k), tDM(k), tII(k), tIM(k), tBM(k-1)}
// Row types follow the reference: rows without an emitted codon, "tail" rows with no i+3 row,
// accumulate-left-to-right rows (L-3, L-4) and the main recursion (generic_fwdback_frameshift.c:1054-1323, 1442-1677).
// ---------------------------------------------------------------------------------------------
template <int C, int NCOD, int MODE>
__global__ __launch_bounds__(kFsBlock, fs_min_waves(C)) void fs_bwd_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                     float tEL, float tEM, float *__restrict__ sc,
                                                     float *__restrict__ bck, const int64_t *__restrict__ bck_off, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs) {
  constexpr bool EXACT = (MODE == 1), STRICT = (MODE == 2);   // 0: table + scans, 1: exact log-sums, 2: table in the reference's serial order
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tb = s_tbl + kLogsumTbl;
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (p.M + 2) * 8; i += blockDim.x) s_tb[i] = p.tb[i];
  __syncthreads();
  constexpr bool FIVE = (NCOD == 5);
  constexpr int DEG = FIVE ? 1367 : 338;
  constexpr int NR = 5;                             // rows i+1..i+5 of M kept in registers
  const int M = p.M;
  // Lanes own their nodes in DESCENDING order (logical lane = 63 - physical lane): Backward's chains run from node M down, and with
  // this order "the lane holding the next nodes" is the physical lane below, so the chains are the same upward DPP scans as Forward's
  const int plane = threadIdx.x & 63;
  const int lane = 63 - plane;
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  for (int64_t job = fs_next_job(jobs, dna.n, plane); job >= 0; job = fs_next_job(jobs, dna.n, plane)) {
    const int L = dna.len[job];
    const uint8_t *d = dna.data + dna.off[job];
    float *bo = bck ? bck + bck_off[job] : nullptr;
    float *xo = xmx ? xmx + xmx_off[job] : nullptr;
    if (L < 5) { if (lane == 0) sc[job] = -INFINITY; continue; }
    const float tNL = loop_tab[L / 3], tNM = move_tab[L / 3], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    float Mr[NR][C], Ir3[3][C];                     // M(i+1..i+5); I(i+1..i+3)
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Mr[r][c] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Ir3[r][c] = -INFINITY;
    float xN3[3] = {-INFINITY, -INFINITY, -INFINITY}, xJ3[3] = {-INFINITY, -INFINITY, -INFINITY}, xC3[3] = {-INFINITY, -INFINITY, -INFINITY};
    float n0 = -INFINITY, n1 = -INFINITY, n2 = -INFINITY;           // N(0), N(1), N(2)
    const int first_emit = FIVE ? L - 1 : L - 2;
    int t = DEG, u = DEG, v = DEG, w = DEG, x = DEG;
    if (!FIVE) w = (d[L - 1] < 4) ? d[L - 1] : DEG;

    for (int i = L; i >= 0; i--) {
      float Mc[C], Ic[C], Dc[C];
      float xE, xNn, xJn, xCn, xBn = -INFINITY;
      if (i > first_emit) {
        // rows before any codon can be emitted (:1054-1073, :1442-1465)
        xCn = (i == L) ? tCM : tCL + tCM;
        xJn = xNn = -INFINITY;
        xE = xCn + tEM;
        // D(i,k) = LS(E, D(i,k+1)+tDD(k)), M(i,k) = LS(E, D(i,k+1)+tMD(k)): reverse chain
        float dnext;
        if constexpr (STRICT) {
          dnext = bwd_dnext_strict<C>([&](int, int node, float dn) { return (node == M) ? xE : LS(xE, dn + s_tb[node * 8 + 3]); }, lane, M);
        } else {
          float A = -INFINITY, B = 0.f;              // lane function applied to D(i, last node of lane + 1)
          // node M's and node M+1's transition rows are -inf: node M falls out of the general formula (LS(E, -inf) = E) and
          // nodes beyond M are deselected, without a branch per node
#pragma unroll
          for (int c = C - 1; c >= 0; c--) {
            const int node = lane * C + c + 1;
            const float tdd = s_tb[imin(node, M + 1) * 8 + 3];
            const float An = LS(xE, A + tdd);
            A = (node <= M) ? An : A; B = (node <= M) ? B + tdd : B;
          }
          // lanes without a source see the identity map (A = -inf, B = 0)
#define BATH_BCHAIN_STEP(CTRL, MASK) { const float An = dpp_f<CTRL, MASK>(A, -INFINITY), Bn = dpp_f<CTRL, MASK>(B, 0.f); A = LS(A, An + B); B = B + Bn; }
          BATH_BCHAIN_STEP(0x111, 0xf) BATH_BCHAIN_STEP(0x112, 0xf) BATH_BCHAIN_STEP(0x114, 0xf) BATH_BCHAIN_STEP(0x118, 0xf)
          BATH_BCHAIN_STEP(0x142, 0xa) BATH_BCHAIN_STEP(0x143, 0xc)
#undef BATH_BCHAIN_STEP
          dnext = wave_shr1(A, -INFINITY);
        }
#pragma unroll
        for (int c = C - 1; c >= 0; c--) {
          const int node = lane * C + c + 1, nd = imin(node, M + 1);
          const float dn = (c == C - 1) ? dnext : Dc[c + 1];
          const float mv = LS(xE, dn + s_tb[nd * 8 + 0]), dv = LS(xE, dn + s_tb[nd * 8 + 3]);
          Mc[c] = (node <= M) ? mv : -INFINITY; Dc[c] = (node <= M) ? dv : -INFINITY;
          Ic[c] = -INFINITY;
        }
      } else {
        if (FIVE || i < first_emit) { t = u; u = v; v = w; w = x; }
        x = (d[i] < 4) ? d[i] : DEG;                // x_{i+1}
        const int avail = L - i;
        const bool mainrow = (i <= L - 5);
        const bool tail = (avail < 3);
        const float *r1 = nullptr, *r2 = nullptr, *r3 = nullptr, *r4 = nullptr, *r5 = nullptr;
        if (FIVE) {
          r1 = p.rsc + (size_t)imin(x * 341, 1366) * p.pitch;                                          // C1(x): x is the only (and last) base
          // reversed argument order: the codon's LAST base is the oldest one in the window (:1260-1270)
          if (avail >= 2) r2 = p.rsc + (size_t)imin(w * 341 + x * 85 + 1, 1365) * p.pitch;
          if (avail >= 3) r3 = p.rsc + (size_t)imin(v * 341 + w * 85 + x * 21 + 2, 1364) * p.pitch;
          if (avail >= 4) r4 = p.rsc + (size_t)imin(u * 341 + v * 85 + w * 21 + x * 5 + 3, 1365) * p.pitch;
          if (avail >= 5) r5 = p.rsc + (size_t)imin(t * 341 + u * 85 + v * 21 + w * 5 + x + 4, 1366) * p.pitch;
        } else {
          if (avail >= 2) r2 = p.rsc + (size_t)imin(w * 84 + x * 21, 337) * p.pitch;
          if (avail >= 3) r3 = p.rsc + (size_t)imin(v * 84 + w * 21 + x * 5 + 1, 336) * p.pitch;
          if (avail >= 4) r4 = p.rsc + (size_t)imin(u * 84 + v * 21 + w * 5 + x + 2, 337) * p.pitch;
        }
        // ivx[k] = logsum_c M(i+c,k)+e_c(k);  B(i) = logsum_k ivx[k]+tBM(k-1)
        float ivx[C];
        [[maybe_unused]] float bterm[C];
        float bloc = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; c++) {
          // nodes beyond M: their M rows are -inf, so ivx comes out -inf and the B term (transition row M+1 = -inf) drops out
          const int node = lane * C + c + 1, ne = imin(node, M), nd = imin(node, M + 1);
          float a;
          if (FIVE) {
            if (mainrow) a = LS(Mr[0][c] + r1[ne], LS(Mr[1][c] + r2[ne], LS(Mr[2][c] + r3[ne], LS(Mr[3][c] + r4[ne], Mr[4][c] + r5[ne]))));
            else {
              a = Mr[0][c] + r1[ne];
              if (avail >= 2) a = LS(a, Mr[1][c] + r2[ne]);
              if (avail >= 3) a = LS(a, Mr[2][c] + r3[ne]);
              if (avail >= 4) a = LS(a, Mr[3][c] + r4[ne]);
            }
          } else {
            if (mainrow) a = LS(Mr[1][c] + r2[ne], LS(Mr[2][c] + r3[ne], Mr[3][c] + r4[ne]));
            else {
              a = Mr[1][c] + r2[ne];
              if (avail >= 3) a = LS(a, Mr[2][c] + r3[ne]);
              if (avail >= 4) a = LS(a, Mr[3][c] + r4[ne]);
            }
          }
          ivx[c] = a;
          if constexpr (STRICT) bterm[c] = a + s_tb[nd * 8 + 7];
          else bloc = LS(bloc, a + s_tb[nd * 8 + 7]);
        }
        if constexpr (STRICT) xBn = bwd_bsum_strict<C>(bterm, lane, M, s_tbl);
        else xBn = wave_logsum<EXACT>(bloc, s_tbl);
        if (i == 0) {
          n0 = LS(xN3[2] + tNL, xBn + tNM);
          if (xo && lane == 0) { xo[0] = -INFINITY; xo[1] = n0; xo[2] = -INFINITY; xo[3] = xBn; xo[4] = -INFINITY; }
          if (bo) for (int k = lane; k <= M; k += 64) { bo[(size_t)k * 3] = bo[(size_t)k * 3 + 1] = bo[(size_t)k * 3 + 2] = -INFINITY; }
          break;
        }
        if (tail) { xJn = xBn + tJM; xNn = xBn + tNM; xCn = tCL + tCM; }
        else { xJn = LS(xJ3[2] + tJL, xBn + tJM); xCn = xC3[2] + tCL; xNn = LS(xN3[2] + tNL, xBn + tNM); }
        // NCOD = 5 is the envelopes' kernel, always unihit (tEL = -inf): E(i) = C(i) + tEM, said without reading J(i) -- so the row's
        // B reduction feeds N(i) and J(i) only and leaves the chain M/D/I wait for
        if constexpr (FIVE) xE = xCn + tEM;
        else xE = LS(xJn + tEL, xCn + tEM);
        // ivx at node+1 for every node of the lane
        const float ivNext = wave_shr1(ivx[0], -INFINITY);
        // D chain (descending): D(k) = LS(LS(E, D(k+1)+tDD(k)), ivx(k+1)+tDM(k)); a(k) := LS(E, ivx(k+1)+tDM(k)) up to association
        float A = -INFINITY, B = 0.f;
        float base[C];
        // node M needs no special case: its transitions out are -inf, so base = -inf and D(M) = M(M) = E fall out of the general
        // formulas (LS(x, -inf) = x); nodes beyond M are deselected
#pragma unroll
        for (int c = C - 1; c >= 0; c--) {
          const int node = lane * C + c + 1, nd = imin(node, M + 1);
          const float ivn = (c == C - 1) ? ivNext : ivx[c + 1];
          const float tdd = s_tb[nd * 8 + 3], tdm = s_tb[nd * 8 + 4];
          base[c] = ivn + tdm;
          if constexpr (!STRICT) {
            const float An = (!FIVE && !mainrow && !tail) ? LS(A + tdd, LS(xE, base[c])) : LS(LS(xE, A + tdd), base[c]);
            A = (node <= M) ? An : A; B = (node <= M) ? B + tdd : B;
          }
        }
        float dnext;
        if constexpr (STRICT) {
          dnext = bwd_dnext_strict<C>([&](int c, int node, float dn) {
            const float tdd = s_tb[node * 8 + 3];                // node M: tdd = -inf, base = -inf: the formula gives E
            return (!FIVE && !mainrow && !tail) ? LS(dn + tdd, LS(xE, base[c])) : LS(LS(xE, dn + tdd), base[c]);
          }, lane, M);
        } else {
          // lanes without a source see the identity map (A = -inf, B = 0)
#define BATH_BCHAIN_STEP(CTRL, MASK) { const float An = dpp_f<CTRL, MASK>(A, -INFINITY), Bn = dpp_f<CTRL, MASK>(B, 0.f); A = LS(A, An + B); B = B + Bn; }
          BATH_BCHAIN_STEP(0x111, 0xf) BATH_BCHAIN_STEP(0x112, 0xf) BATH_BCHAIN_STEP(0x114, 0xf) BATH_BCHAIN_STEP(0x118, 0xf)
          BATH_BCHAIN_STEP(0x142, 0xa) BATH_BCHAIN_STEP(0x143, 0xc)
#undef BATH_BCHAIN_STEP
          dnext = wave_shr1(A, -INFINITY);
        }
#pragma unroll
        for (int c = C - 1; c >= 0; c--) {
          const int node = lane * C + c + 1, nd = imin(node, M + 1);
          const float dn = (c == C - 1) ? dnext : Dc[c + 1];
          const float ivn = (c == C - 1) ? ivNext : ivx[c + 1];
          const float4 t0 = *reinterpret_cast<const float4 *>(s_tb + nd * 8);          // tMD tMI tMM tDD
          const float tii = s_tb[nd * 8 + 5], tim = s_tb[nd * 8 + 6];
          const float tmd = t0.x, tmi = t0.y, tmm = t0.z, tdd = t0.w;
          float mv, dv, iv_;
          if (tail) {
            mv = LS(dn + tmd, LS(ivn + tmm, xE));
            dv = LS(LS(xE, dn + tdd), base[c]);
            iv_ = ivn + tim;
          } else if (!FIVE && !mainrow) {
            mv = LS(dn + tmd, LS(Ir3[2][c] + tmi, LS(ivn + tmm, xE)));
            dv = LS(dn + tdd, LS(xE, base[c]));
            iv_ = LS(Ir3[2][c] + tii, ivn + tim);
          } else {
            mv = LS(LS(dn + tmd, LS(Ir3[2][c] + tmi, ivn + tmm)), xE);
            dv = LS(LS(xE, dn + tdd), base[c]);
            iv_ = LS(Ir3[2][c] + tii, ivn + tim);
          }
          // node M: -inf transitions make mv = dv = E and iv_ = -inf by themselves; nodes beyond M hold -inf
          Mc[c] = (node <= M) ? mv : -INFINITY; Dc[c] = (node <= M) ? dv : -INFINITY; Ic[c] = (node <= M) ? iv_ : -INFINITY;
        }
      }
      // store row i
      if (bo) {
        float *row = bo + (size_t)i * (M + 1) * 3;
        if (lane == 0) row[0] = row[1] = row[2] = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          if (node <= M) { row[(size_t)node * 3 + 0] = Dc[c]; row[(size_t)node * 3 + 1] = Ic[c]; row[(size_t)node * 3 + 2] = Mc[c]; }
        }
      }
      if (xo && lane == 0) { xo[i * 5 + 0] = xE; xo[i * 5 + 1] = xNn; xo[i * 5 + 2] = xJn; xo[i * 5 + 3] = xBn; xo[i * 5 + 4] = xCn; }
      if (i == 2) n2 = xNn;
      if (i == 1) n1 = xNn;
      xN3[2] = xN3[1]; xN3[1] = xN3[0]; xN3[0] = xNn;
      xJ3[2] = xJ3[1]; xJ3[1] = xJ3[0]; xJ3[0] = xJn;
      xC3[2] = xC3[1]; xC3[1] = xC3[0]; xC3[0] = xCn;
#pragma unroll
      for (int c = 0; c < C; c++) {
#pragma unroll
        for (int r = NR - 1; r > 0; r--) Mr[r][c] = Mr[r - 1][c];
        Mr[0][c] = Mc[c];
        Ir3[2][c] = Ir3[1][c]; Ir3[1][c] = Ir3[0][c]; Ir3[0][c] = Ic[c];
      }
    }
    if (lane == 0) sc[job] = LS(n0, LS(n1, n2));
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------
// Posterior decoding in place on the Forward matrix (generic_decoding_frameshift.c:36-156), plus the
// column sums null2 needs (generic_null2_frameshift.c:62-68) accumulated in the same pass.
// One wave per envelope, rows in order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fs5_decode_kernel(SeqView dna, int M, const float *__restrict__ loop_tab, const float *__restrict__ bcksc,
                                                         float *__restrict__ fwd, const int64_t *__restrict__ fwd_off, float *__restrict__ fx, const int64_t *__restrict__ fx_off,
                                                         const float *__restrict__ bck, const int64_t *__restrict__ bck_off, const float *__restrict__ bx,
                                                         float *__restrict__ colsum /* [n][(M+1)*8 + 8] */) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t job = wid; job < dna.n; job += nw) {
    const int L = dna.len[job];
    if (L < 5) continue;
    float *f = fwd + fwd_off[job];
    float *x = fx + fx_off[job];
    const float *b = bck + bck_off[job];
    const float *y = bx + fx_off[job];
    float *cs = colsum + (size_t)job * ((size_t)(M + 1) * 8 + 8);
    const float overall = bcksc[job];
    const float tL = loop_tab[L / 3];
    float N0 = x[1], J0 = x[2], C0 = x[4], N1 = 0, N2 = 0, N3 = 0, J1 = 0, J2 = 0, J3 = 0, C1 = 0, C2 = 0, C3 = 0;
    for (int k = lane; k < (M + 1) * 8; k += 64) f[k] = 0.f;
    if (lane < 5) x[lane] = 0.f;
    for (int i = 1; i <= L; i++) {
      N3 = N2; N2 = N1; N1 = N0; J3 = J2; J2 = J1; J1 = J0; C3 = C2; C2 = C1; C1 = C0;
      float *fr = f + (size_t)i * (M + 1) * 8;
      const float *br = b + (size_t)i * (M + 1) * 3;
      float dloc = 0.f;
      if (lane == 0) for (int s = 0; s < 8; s++) fr[s] = 0.f;
      for (int k = 1 + lane; k <= M; k += 64) {
        const float bm = br[(size_t)k * 3 + 2], bi = br[(size_t)k * 3 + 1];
        float *cell = fr + (size_t)k * 8;
#pragma unroll
        for (int c = 2; c < 8; c++) cell[c] = expf(cell[c] + bm - overall);
        dloc += cell[2];
        if (k < M) { cell[1] = expf(cell[1] + bi - overall); dloc += cell[1]; } else cell[1] = 0.f;
        cell[0] = 0.f;
      }
      N0 = x[i * 5 + 1]; J0 = x[i * 5 + 2]; C0 = x[i * 5 + 4];
      float pn, pc, pj;
      if (i > 2) {
        pn = expf(N3 + y[i * 5 + 1] + tL - overall);
        pc = expf(C3 + y[i * 5 + 4] + tL - overall);
        pj = expf(J3 + y[i * 5 + 2] + tL - overall);
      } else { pn = expf(y[i * 5 + 1] - overall); pc = 0.f; pj = 0.f; }
      float denom = wave_sum_f32(dloc) + ((i > 2) ? (pn + pj + pc) : pn);
      denom = (float)(1.0 / (double)denom);
      for (int k = 1 + lane; k <= M; k += 64) {
        float *cell = fr + (size_t)k * 8;
#pragma unroll
        for (int c = 2; c < 8; c++) { cell[c] *= denom; cs[(size_t)k * 8 + c] += cell[c]; }
        if (k < M) { cell[1] *= denom; cs[(size_t)k * 8 + 1] += cell[1]; }
      }
      pn *= denom; pc *= denom; pj *= denom;
      if (lane == 0) {
        x[i * 5 + 0] = 0.f; x[i * 5 + 3] = 0.f; x[i * 5 + 1] = pn; x[i * 5 + 4] = pc; x[i * 5 + 2] = pj;
        float *xs = cs + (size_t)(M + 1) * 8;
        xs[1] += pn; xs[2] += pj; xs[4] += pc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Optimal-accuracy fill (generic_optacc_frameshift.c:53-324): max-sum over posteriors.  TSCDELTA is 1 for a
// possible transition and FLT_MIN for an impossible one.  oa[(i*(M+1)+k)*3 + {D,I,M}], xmx in ox.
// The D row is a running maximum along the model: an exact wavefront scan (max is associative).
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void fs5_oa_kernel(SeqView dna, int M, const float *__restrict__ tf /* forward-ordered log transitions */,
                                                     const float *__restrict__ pp, const int64_t *__restrict__ pp_off, const float *__restrict__ px, const int64_t *__restrict__ px_off,
                                                     float *__restrict__ oa, const int64_t *__restrict__ oa_off, float *__restrict__ oasc, float ej, float ec,
                                                     float *__restrict__ ox /* optional: OA special-state rows (L+1) x {E,N,J,B,C} at px_off */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_dl = reinterpret_cast<float *>(lds);                 // [(M+2)][8] deltas, same order as tf
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_dl[i] = (tf[i] == -INFINITY) ? 1.17549435e-38f : 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t job = wid; job < dna.n; job += nw) {
    const int L = dna.len[job];
    if (L < 5) { if (lane == 0) oasc[job] = -INFINITY; continue; }
    const float *P = pp + pp_off[job];
    const float *X = px + px_off[job];
    float *O = oa + oa_off[job];
    float *OX = ox ? ox + px_off[job] : nullptr;
    if (OX && lane == 0) { OX[0] = -INFINITY; OX[1] = 0.f; OX[2] = -INFINITY; OX[3] = 0.f; OX[4] = -INFINITY; }
    // rows i-1..i-5 of M, I, D and B; rows i-1..i-3 of N, J, C
    float Mr[5][C], Ir[5][C], Dr[5][C];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Mr[r][c] = Ir[r][c] = Dr[r][c] = -INFINITY;
    float Bh[5] = {0.f, -INFINITY, -INFINITY, -INFINITY, -INFINITY};      // B(i-1) ... ; B(0) = 0
    float Nh[3] = {0.f, 0.f, 0.f}, Jh[3] = {-INFINITY, -INFINITY, -INFINITY}, Ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    float cL = -INFINITY, cL1 = -INFINITY, cL2 = -INFINITY;
    for (int k = lane; k <= M; k += 64) { O[(size_t)k * 3] = O[(size_t)k * 3 + 1] = O[(size_t)k * 3 + 2] = -INFINITY; }
    for (int i = 1; i <= L; i++) {
      const float *pr = P + (size_t)i * (M + 1) * 8;
      float *orow = O + (size_t)i * (M + 1) * 3;
      if (lane == 0) orow[0] = orow[1] = orow[2] = -INFINITY;
      float mIn[5], iIn[5], dIn[5];
#pragma unroll
      for (int r = 0; r < 5; r++) {
        mIn[r] = wave_shr1(Mr[r][C - 1], -INFINITY); iIn[r] = wave_shr1(Ir[r][C - 1], -INFINITY); dIn[r] = wave_shr1(Dr[r][C - 1], -INFINITY);
      }
      float Mc[C], Ic[C], am[C], bm[C];
      float eloc = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) { Mc[c] = Ic[c] = -INFINITY; am[c] = -INFINITY; bm[c] = 1.0f; continue; }
        const float dMM = s_dl[node * 8 + 0], dIM = s_dl[node * 8 + 1], dDM = s_dl[node * 8 + 2], dBM = s_dl[node * 8 + 3];
        const float dMD = s_dl[node * 8 + 4], dDD = s_dl[node * 8 + 5], dMI = s_dl[node * 8 + 6], dII = s_dl[node * 8 + 7];
        const float *cell = pr + (size_t)node * 8;
        float best;
        if (i == 1) best = dBM * cell[3];
        else {
          float mx[6];
          const int cmax = (i >= 5) ? 5 : (i == 2 ? 2 : (i == 4 ? 4 : 3));
#pragma unroll
          for (int cl = 1; cl <= 5; cl++) {
            if (cl > cmax) { mx[cl] = -INFINITY; continue; }
            const float pv = cell[2 + cl];
            if ((i == 2 && cl == 2) || (i == 4 && cl == 4)) mx[cl] = dBM * (0.0f + pv);     // only B(0)=0 can precede
            else {
              const int r = cl - 1;
              const float m1 = (c == 0) ? mIn[r] : Mr[r][c - 1], i1 = (c == 0) ? iIn[r] : Ir[r][c - 1], d1 = (c == 0) ? dIn[r] : Dr[r][c - 1];
              mx[cl] = fmaxf(dMM * (m1 + pv), fmaxf(dIM * (i1 + pv), fmaxf(dDM * (d1 + pv), dBM * (Bh[r] + pv))));
            }
          }
          if (i == 2) best = fmaxf(mx[1], mx[2]);
          else if (i < 5) best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(mx[3], mx[4]));
          else best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(fmaxf(mx[3], mx[4]), mx[5]));
        }
        Mc[c] = best;
        Ic[c] = (i >= 3 && node < M) ? fmaxf(dMI * (Mr[2][c] + cell[1]), dII * (Ir[2][c] + cell[1])) : -INFINITY;
        am[c] = dMD * best;                       // contribution to D(node+1) from M(node)
        bm[c] = dDD;                              // multiplier on D(node) into D(node+1)
        // NB tf[node] holds the transitions OUT of node for MD,DD (k) and INTO node for MM.. (k-1); the reference's
        // D(i,k) uses TSCDELTA(MD,k-1), (DD,k-1): i.e. the out-of-(k-1) deltas, which is what am/bm of node k-1 are.
      }
      // D(node+1) = max(am(node), bm(node) * D(node)): scan of x -> max(A, B*x)
      float A = -INFINITY, Bm = 1.0f;
#pragma unroll
      for (int c = 0; c < C; c++) { A = fmaxf(am[c], bm[c] * A); Bm *= bm[c]; }
      // lanes without a source see the identity map (A = -inf, B = 1); fmaxf ignores the NaN of 0 * -inf when B has underflowed
#define BATH_OA_STEP(CTRL, MASK) { const float Ap = dpp_f<CTRL, MASK>(A, -INFINITY), Bp = dpp_f<CTRL, MASK>(Bm, 1.0f); A = fmaxf(A, Bm * Ap); Bm *= Bp; }
      BATH_OA_STEP(0x111, 0xf) BATH_OA_STEP(0x112, 0xf) BATH_OA_STEP(0x114, 0xf) BATH_OA_STEP(0x118, 0xf) BATH_OA_STEP(0x142, 0xa) BATH_OA_STEP(0x143, 0xc)
#undef BATH_OA_STEP
      const float din = wave_shr1(A, -INFINITY);
      float Dc[C];
      Dc[0] = din;
#pragma unroll
      for (int c = 1; c < C; c++) Dc[c] = fmaxf(am[c - 1], bm[c - 1] * Dc[c - 1]);
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) { Dc[c] = -INFINITY; continue; }
        orow[(size_t)node * 3 + 0] = Dc[c]; orow[(size_t)node * 3 + 1] = Ic[c]; orow[(size_t)node * 3 + 2] = Mc[c];
        eloc = fmaxf(eloc, (node < M) ? Mc[c] : fmaxf(Mc[c], Dc[c]));
      }
      float xE = eloc;                                      // wave maximum: running maximum by DPP, last lane broadcast (max is exact in any order)
      xE = fmaxf(xE, dpp_f<0x111>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x112>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x114>(xE, -INFINITY));
      xE = fmaxf(xE, dpp_f<0x118>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x142, 0xa>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x143, 0xc>(xE, -INFINITY));
      xE = wave_bcast_last(xE);
      float nN, nJ, nC;
      if (i <= 2) { nJ = ej * xE; nC = ec * xE; nN = X[i * 5 + 1]; }
      else { nJ = fmaxf(Jh[2] + X[i * 5 + 2], ej * xE); nC = fmaxf(Ch[2] + X[i * 5 + 4], ec * xE); nN = Nh[2] + X[i * 5 + 1]; }
      const float nB = fmaxf(nN, nJ);
      if (OX && lane == 0) { float *r = OX + (size_t)i * 5; r[0] = xE; r[1] = nN; r[2] = nJ; r[3] = nB; r[4] = nC; }
      Nh[2] = Nh[1]; Nh[1] = Nh[0]; Nh[0] = nN;
      Jh[2] = Jh[1]; Jh[1] = Jh[0]; Jh[0] = nJ;
      Ch[2] = Ch[1]; Ch[1] = Ch[0]; Ch[0] = nC;
      Bh[4] = Bh[3]; Bh[3] = Bh[2]; Bh[2] = Bh[1]; Bh[1] = Bh[0]; Bh[0] = nB;
      cL2 = cL1; cL1 = cL; cL = nC;
#pragma unroll
      for (int c = 0; c < C; c++) {
#pragma unroll
        for (int r = 4; r > 0; r--) { Mr[r][c] = Mr[r - 1][c]; Ir[r][c] = Ir[r - 1][c]; Dr[r][c] = Dr[r - 1][c]; }
        Mr[0][c] = Mc[c]; Ir[0][c] = Ic[c]; Dr[0][c] = Dc[c];
      }
    }
    if (lane == 0) oasc[job] = cL + cL1 + cL2;
  }
}

// ---------------------------------------------------------------------------------------------
// Posterior decoding AND the optimal-accuracy fill in one walk over the rows (fs5_decode_kernel + fs5_oa_kernel fused).
// Both go through the rows in ascending order and the OA row needs exactly the posteriors decoding has just produced, so the
// posteriors are taken from registers instead of being written (32 B/cell) and read back (32 B/cell) by a second kernel: the
// pass reads Forward (32 B) and Backward (12 B) and writes the posteriors (32 B, for the traceback) and the OA cells (12 B),
// 88 B/cell instead of 120, and -- what matters more for these latency-bound kernels -- one chain of dependent rows instead of
// two.  Lanes own C contiguous nodes (the OA recursion's layout); the row's normalising sum is a wave reduction; null2's
// column sums stay in registers (C <= 4) and are written once per envelope.
// ---------------------------------------------------------------------------------------------
#ifndef BATH_FS_OA_WAVES
#define BATH_FS_OA_WAVES 2
#endif
template <int C>
__global__ __launch_bounds__(256, (C <= 3 ? BATH_FS_OA_WAVES : 1)) void fs5_decode_oa_kernel(SeqView dna, int M, const float *__restrict__ tf, const float *__restrict__ loop_tab, const float *__restrict__ bcksc,
                                                            float *__restrict__ fwd, const int64_t *__restrict__ fwd_off, float *__restrict__ fx, const int64_t *__restrict__ x_off,
                                                            const float *__restrict__ bck, const int64_t *__restrict__ bck_off, const float *__restrict__ bx,
                                                            float *__restrict__ colsum /* [n][(M+1)*8 + 8], zeroed */, float *__restrict__ oa, float *__restrict__ oasc,
                                                            float ej, float ec, float *__restrict__ ox, FsJobs jobs,
                                                            int store_pp /* 0: the posterior matrix is NOT written (the pipeline reads it along the trace only: fs5_trace_kernel
                                                                            recomputes those cells from Forward, Backward and rowden) */,
                                                            float *__restrict__ rowden /* [rows]: 1 / (row sum) of every row, at x_off / 5; or null */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_dl = reinterpret_cast<float *>(lds);                 // [(M+2)][8] TSCDELTA, same order as tf
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_dl[i] = (tf[i] == -INFINITY) ? 1.17549435e-38f : 1.0f;
  __syncthreads();
  constexpr bool REGSUM = (C <= 4);
  const int lane = threadIdx.x & 63;
  for (int64_t job = fs_next_job(jobs, dna.n, lane); job >= 0; job = fs_next_job(jobs, dna.n, lane)) {
    const int L = dna.len[job];
    if (L < 5) { if (lane == 0) oasc[job] = -INFINITY; continue; }
    float *F = fwd + fwd_off[job];
    float *X = fx + x_off[job];
    const float *Bk = bck + bck_off[job];
    const float *Y = bx + x_off[job];
    float *O = oa + bck_off[job];
    float *OX = ox ? ox + x_off[job] : nullptr;
    float *cs = colsum + (size_t)job * ((size_t)(M + 1) * 8 + 8);
    const float overall = bcksc[job];
    const float tL = loop_tab[L / 3];
    // Forward special states of rows i, i-1, i-2, i-3 (read before their rows are overwritten with posteriors)
    float N0 = X[1], J0 = X[2], C0 = X[4], N1 = 0, N2 = 0, N3 = 0, J1 = 0, J2 = 0, J3 = 0, C1 = 0, C2 = 0, C3 = 0;
    // row 0: posteriors 0, OA cells -inf
    float *RD = rowden ? rowden + x_off[job] / 5 : nullptr;
    if (store_pp) for (int k = lane; k < (M + 1) * 8; k += 64) F[k] = 0.f;
    for (int k = lane; k <= M; k += 64) { O[(size_t)k * 3] = O[(size_t)k * 3 + 1] = O[(size_t)k * 3 + 2] = -INFINITY; }
    if (lane < 5) X[lane] = 0.f;
    if (OX && lane == 0) { OX[0] = -INFINITY; OX[1] = 0.f; OX[2] = -INFINITY; OX[3] = 0.f; OX[4] = -INFINITY; }
    float Mr[5][C], Ir[5][C], Dr[5][C];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Mr[r][c] = Ir[r][c] = Dr[r][c] = -INFINITY;
    float Bh[5] = {0.f, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float Nh[3] = {0.f, 0.f, 0.f}, Jh[3] = {-INFINITY, -INFINITY, -INFINITY}, Ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    float cL = -INFINITY, cL1 = -INFINITY, cL2 = -INFINITY;
    float csum[REGSUM ? C : 1][7];
    float sN = 0.f, sJ = 0.f, sC = 0.f;
    if (REGSUM) {
#pragma unroll
      for (int c = 0; c < C; c++)
#pragma unroll
        for (int q = 0; q < 7; q++) csum[c][q] = 0.f;
    }
    // Row i+1's Forward and Backward cells and special states are fetched while row i is worked on: a row's trip to HBM then
    // overlaps the previous row's chain (exp, the row sum, the optimal-accuracy scan) instead of heading this one's
    float4 fa_n[C], fb_n[C]; float bm_n[C], bi_n[C];
    float xn1 = 0.f, xn2 = 0.f, xn4 = 0.f, yn1 = 0.f, yn2 = 0.f, yn4 = 0.f;
    auto fetch_row = [&](int r) {
      const float *fq = F + (size_t)r * (M + 1) * 8;
      const float *bq = Bk + (size_t)r * (M + 1) * 3;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = imin(lane * C + c + 1, M);
        fa_n[c] = *reinterpret_cast<const float4 *>(fq + (size_t)node * 8);
        fb_n[c] = *reinterpret_cast<const float4 *>(fq + (size_t)node * 8 + 4);
        bm_n[c] = bq[(size_t)node * 3 + 2]; bi_n[c] = bq[(size_t)node * 3 + 1];
      }
      xn1 = X[r * 5 + 1]; xn2 = X[r * 5 + 2]; xn4 = X[r * 5 + 4]; yn1 = Y[r * 5 + 1]; yn2 = Y[r * 5 + 2]; yn4 = Y[r * 5 + 4];
    };
    fetch_row(1);
    for (int i = 1; i <= L; i++) {
      N3 = N2; N2 = N1; N1 = N0; J3 = J2; J2 = J1; J1 = J0; C3 = C2; C2 = C1; C1 = C0;
      float *fr = F + (size_t)i * (M + 1) * 8;
      float *orow = O + (size_t)i * (M + 1) * 3;
      float4 fa[C], fb[C]; float bmv[C], biv[C];
#pragma unroll
      for (int c = 0; c < C; c++) { fa[c] = fa_n[c]; fb[c] = fb_n[c]; bmv[c] = bm_n[c]; biv[c] = bi_n[c]; }
      const float x1 = xn1, x2 = xn2, x4 = xn4, y1 = yn1, y2 = yn2, y4 = yn4;
      if (i < L) fetch_row(i + 1);
      // ---- decoding of row i (generic_decoding_frameshift.c:62-150): exp(F + B - overall), then the row normalised to sum 1
      float pI[C], pC[C][6];
      float dloc = 0.f;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node <= M) {
          const float4 a = fa[c];
          const float4 b = fb[c];
          const float bm = bmv[c], bi = biv[c];
          pC[c][0] = expf(a.z + bm - overall); pC[c][1] = expf(a.w + bm - overall);
          pC[c][2] = expf(b.x + bm - overall); pC[c][3] = expf(b.y + bm - overall); pC[c][4] = expf(b.z + bm - overall); pC[c][5] = expf(b.w + bm - overall);
          dloc += pC[c][0];
          if (node < M) { pI[c] = expf(a.y + bi - overall); dloc += pI[c]; } else pI[c] = 0.f;
        } else {
          pI[c] = 0.f;
#pragma unroll
          for (int q = 0; q < 6; q++) pC[c][q] = 0.f;
        }
      }
      N0 = x1; J0 = x2; C0 = x4;
      float pn, pc, pj;
      if (i > 2) {
        pn = expf(N3 + y1 + tL - overall);
        pc = expf(C3 + y4 + tL - overall);
        pj = expf(J3 + y2 + tL - overall);
      } else { pn = expf(y1 - overall); pc = 0.f; pj = 0.f; }
      float denom = wave_sum_f32(dloc) + ((i > 2) ? (pn + pj + pc) : pn);
      denom = (float)(1.0 / (double)denom);
      pn *= denom; pc *= denom; pj *= denom;
      if (lane == 0) {
        if (store_pp) {
#pragma unroll
          for (int q = 0; q < 8; q++) fr[q] = 0.f;
        }
        if (RD) RD[i] = denom;
        X[i * 5 + 0] = 0.f; X[i * 5 + 3] = 0.f; X[i * 5 + 1] = pn; X[i * 5 + 4] = pc; X[i * 5 + 2] = pj;
        orow[0] = orow[1] = orow[2] = -INFINITY;
      }
      sN += pn; sJ += pj; sC += pc;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) continue;
        pI[c] *= denom;
#pragma unroll
        for (int q = 0; q < 6; q++) pC[c][q] *= denom;
        if (store_pp) {
          *reinterpret_cast<float4 *>(fr + (size_t)node * 8) = make_float4(0.f, pI[c], pC[c][0], pC[c][1]);
          *reinterpret_cast<float4 *>(fr + (size_t)node * 8 + 4) = make_float4(pC[c][2], pC[c][3], pC[c][4], pC[c][5]);
        }
        if (REGSUM) {
          csum[c][0] += pI[c];
#pragma unroll
          for (int q = 0; q < 6; q++) csum[c][1 + q] += pC[c][q];
        } else {
          if (node < M) cs[(size_t)node * 8 + 1] += pI[c];
#pragma unroll
          for (int q = 0; q < 6; q++) cs[(size_t)node * 8 + 2 + q] += pC[c][q];
        }
      }
      // ---- optimal-accuracy row i (generic_optacc_frameshift.c:53-324) on the posteriors in registers
      float mIn[5], iIn[5], dIn[5];
#pragma unroll
      for (int r = 0; r < 5; r++) {
        mIn[r] = wave_shr1(Mr[r][C - 1], -INFINITY); iIn[r] = wave_shr1(Ir[r][C - 1], -INFINITY); dIn[r] = wave_shr1(Dr[r][C - 1], -INFINITY);
      }
      float Mc[C], Ic[C], am[C], bmul[C];
      float eloc = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) { Mc[c] = Ic[c] = -INFINITY; am[c] = -INFINITY; bmul[c] = 1.0f; continue; }
        const float dMM = s_dl[node * 8 + 0], dIM = s_dl[node * 8 + 1], dDM = s_dl[node * 8 + 2], dBM = s_dl[node * 8 + 3];
        const float dMD = s_dl[node * 8 + 4], dDD = s_dl[node * 8 + 5], dMI = s_dl[node * 8 + 6], dII = s_dl[node * 8 + 7];
        float best;
        if (i == 1) best = dBM * pC[c][1];
        else {
          float mx[6];
          const int cmax = (i >= 5) ? 5 : (i == 2 ? 2 : (i == 4 ? 4 : 3));
#pragma unroll
          for (int cl = 1; cl <= 5; cl++) {
            if (cl > cmax) { mx[cl] = -INFINITY; continue; }
            const float pv = pC[c][cl];
            if ((i == 2 && cl == 2) || (i == 4 && cl == 4)) mx[cl] = dBM * (0.0f + pv);
            else {
              const int r = cl - 1;
              const float m1 = (c == 0) ? mIn[r] : Mr[r][c - 1], i1 = (c == 0) ? iIn[r] : Ir[r][c - 1], d1 = (c == 0) ? dIn[r] : Dr[r][c - 1];
              mx[cl] = fmaxf(dMM * (m1 + pv), fmaxf(dIM * (i1 + pv), fmaxf(dDM * (d1 + pv), dBM * (Bh[r] + pv))));
            }
          }
          if (i == 2) best = fmaxf(mx[1], mx[2]);
          else if (i < 5) best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(mx[3], mx[4]));
          else best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(fmaxf(mx[3], mx[4]), mx[5]));
        }
        Mc[c] = best;
        Ic[c] = (i >= 3 && node < M) ? fmaxf(dMI * (Mr[2][c] + pI[c]), dII * (Ir[2][c] + pI[c])) : -INFINITY;
        am[c] = dMD * best;
        bmul[c] = dDD;
      }
      float A = -INFINITY, Bm = 1.0f;
#pragma unroll
      for (int c = 0; c < C; c++) { A = fmaxf(am[c], bmul[c] * A); Bm *= bmul[c]; }
      // lanes without a source see the identity map (A = -inf, B = 1); fmaxf ignores the NaN of 0 * -inf when B has underflowed
#define BATH_OA_STEP(CTRL, MASK) { const float Ap = dpp_f<CTRL, MASK>(A, -INFINITY), Bp = dpp_f<CTRL, MASK>(Bm, 1.0f); A = fmaxf(A, Bm * Ap); Bm *= Bp; }
      BATH_OA_STEP(0x111, 0xf) BATH_OA_STEP(0x112, 0xf) BATH_OA_STEP(0x114, 0xf) BATH_OA_STEP(0x118, 0xf) BATH_OA_STEP(0x142, 0xa) BATH_OA_STEP(0x143, 0xc)
#undef BATH_OA_STEP
      const float din = wave_shr1(A, -INFINITY);
      float Dc[C];
      Dc[0] = din;
#pragma unroll
      for (int c = 1; c < C; c++) Dc[c] = fmaxf(am[c - 1], bmul[c - 1] * Dc[c - 1]);
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) { Dc[c] = -INFINITY; continue; }
        orow[(size_t)node * 3 + 0] = Dc[c]; orow[(size_t)node * 3 + 1] = Ic[c]; orow[(size_t)node * 3 + 2] = Mc[c];
        eloc = fmaxf(eloc, (node < M) ? Mc[c] : fmaxf(Mc[c], Dc[c]));
      }
      float xE = eloc;                                      // wave maximum: running maximum by DPP, last lane broadcast (max is exact in any order)
      xE = fmaxf(xE, dpp_f<0x111>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x112>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x114>(xE, -INFINITY));
      xE = fmaxf(xE, dpp_f<0x118>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x142, 0xa>(xE, -INFINITY)); xE = fmaxf(xE, dpp_f<0x143, 0xc>(xE, -INFINITY));
      xE = wave_bcast_last(xE);
      float nN, nJ, nC;
      if (i <= 2) { nJ = ej * xE; nC = ec * xE; nN = pn; }
      else { nJ = fmaxf(Jh[2] + pj, ej * xE); nC = fmaxf(Ch[2] + pc, ec * xE); nN = Nh[2] + pn; }
      const float nB = fmaxf(nN, nJ);
      if (OX && lane == 0) { float *r = OX + (size_t)i * 5; r[0] = xE; r[1] = nN; r[2] = nJ; r[3] = nB; r[4] = nC; }
      Nh[2] = Nh[1]; Nh[1] = Nh[0]; Nh[0] = nN;
      Jh[2] = Jh[1]; Jh[1] = Jh[0]; Jh[0] = nJ;
      Ch[2] = Ch[1]; Ch[1] = Ch[0]; Ch[0] = nC;
      Bh[4] = Bh[3]; Bh[3] = Bh[2]; Bh[2] = Bh[1]; Bh[1] = Bh[0]; Bh[0] = nB;
      cL2 = cL1; cL1 = cL; cL = nC;
#pragma unroll
      for (int c = 0; c < C; c++) {
#pragma unroll
        for (int r = 4; r > 0; r--) { Mr[r][c] = Mr[r - 1][c]; Ir[r][c] = Ir[r - 1][c]; Dr[r][c] = Dr[r - 1][c]; }
        Mr[0][c] = Mc[c]; Ir[0][c] = Ic[c]; Dr[0][c] = Dc[c];
      }
    }
    if (REGSUM) {
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1;
        if (node > M) continue;
        if (node < M) cs[(size_t)node * 8 + 1] = csum[c][0];
#pragma unroll
        for (int q = 0; q < 6; q++) cs[(size_t)node * 8 + 2 + q] = csum[c][1 + q];
      }
    }
    if (lane == 0) {
      float *xs = cs + (size_t)(M + 1) * 8;
      xs[1] = sN; xs[2] = sJ; xs[4] = sC;
      oasc[job] = cL + cL1 + cL2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// null2 (generic_null2_frameshift.c:70-125) from the column sums: 20 lanes, each runs the reference's serial
// log-sum over the model for one residue, so the association is the reference's.
// ---------------------------------------------------------------------------------------------
// p7_OATrace_Frameshift (generic form: generic_optacc_frameshift.c:373-588) and the null2 score of the aligned residues
// (rescore_isolated_domain_frameshift, p7_domaindef.c:1086-1147), one LANE per envelope: the walk is serial (at most L+M
// steps of a few dependent loads), envelopes are independent, and doing it here means the posterior and OA matrices
// (>1 MB per envelope) never leave the device.  The trace is written backwards into <tbuf> and read forwards again.
#ifndef BATH_TRACE_SPREAD
#define BATH_TRACE_SPREAD 64
#endif
constexpr int kTraceSpread = BATH_TRACE_SPREAD;
__global__ void fs5_trace_kernel(SeqView dna, int M, int maxcodons, const float *__restrict__ tf, const uint8_t *__restrict__ codons,
                                 const float *__restrict__ pp, const int64_t *__restrict__ pp_off, const float *__restrict__ px, const int64_t *__restrict__ x_off,
                                 const float *__restrict__ oa, const int64_t *__restrict__ oa_off, const float *__restrict__ ox,
                                 const float *__restrict__ null2 /* [n][Kp] */, uint2 *__restrict__ tbuf, const int64_t *__restrict__ t_off, FsTraceOut *__restrict__ out,
                                 const uint8_t *__restrict__ indel_tab, const uint8_t *__restrict__ cons, uint16_t *__restrict__ steps,
                                 const float *__restrict__ amino /* rsc + maxcodons*pitch: the amino rows */, int pitch,
                                 float *__restrict__ step_pp /* posterior of every column's state (tr->pp), parallel to steps */, int *__restrict__ col_cursor,
                                 const float *__restrict__ bck /* null: <pp> holds the posterior matrix.  Else <pp> is the FORWARD matrix, untouched, and the
                                                                    posteriors the walk reads -- a match cell's five codon lengths, a column's state -- are formed here
                                                                    exactly as the decoding kernels form them: expf(F + B - overall) * rowden(i) */,
                                 const int64_t *__restrict__ bck_off, const float *__restrict__ bcksc, const float *__restrict__ rowden) {
  // One envelope per WAVE (kTraceSpread = 64 lanes apart): the walk is a state machine whose lanes diverge at every step (each
  // state's loads and compares run under their own exec mask, one after the other), so a wave holding 64 envelopes pays for
  // every state present among them at every step.  The chip has room for a wave per envelope: 2.77 -> 1.15 ms on the bench's
  // 4.8 k envelopes (8 per wave: 1.53, 4 per wave: 1.36).
  if (threadIdx.x % kTraceSpread) return;
  const int64_t job = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kTraceSpread;
  if (job >= dna.n) return;
  enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC, sT };
  enum { XE = 0, XN, XJ, XB, XC };
  const float kTiny = 1.17549435e-38f;                          // FLT_MIN: TSCDELTA of an impossible transition
  const int L = dna.len[job];
  const uint8_t *dsq = dna.data + dna.off[job] - 1;             // dsq[1..L]
  const float *P = pp + pp_off[job], *PX = px + x_off[job], *O = oa + oa_off[job], *OX = ox + x_off[job];
  const float *BK = bck ? bck + bck_off[job] : nullptr, *RD = bck ? rowden + x_off[job] / 5 : nullptr;
  const float overall = bck ? bcksc[job] : 0.f;
  // posterior of cell (i, k): q = 1: the insert state, q = 2: the match state (all codon lengths), q = 3..7: codon lengths 1..5
  auto post = [&](int i, int k, int q) -> float {
    const float f = P[((size_t)i * (M + 1) + k) * 8 + q];
    if (!BK) return f;
    if (q == 1 && k >= M) return 0.f;
    const float b = BK[((size_t)i * (M + 1) + k) * 3 + (q == 1 ? 1 : 2)];
    return expf(f + b - overall) * RD[i];
  };
  uint2 *T = tbuf + t_off[job];
  const int cap = (int)(t_off[job + 1] - t_off[job]);
  FsTraceOut r{0, 0, 0, 0, 0, 0, 0.f, 0, 0, 0, 0.f, 0};
  auto dl = [&](int node, int s) { return (node >= 1 && node <= M && tf[(size_t)node * 8 + s] != -INFINITY) ? 1.0f : kTiny; };
  auto OM = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 2]; };
  auto OI = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 1]; };
  auto OD = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 0]; };
  int n = 0;
  auto push = [&](int st, int k, int i, int c) { if (n < cap) T[n] = make_uint2((unsigned)st | ((unsigned)c << 8) | ((unsigned)k << 16), (unsigned)i); n++; };
  int i = L, k = 0, c = 0, prev = sC;
  bool bad = (L < 5);
  push(sT, k, i, c); push(sC, k, i, c);
  while (!bad && prev != sS) {
    int cur = -1;
    switch (prev) {
    case sM: {          // transitions into node k: MM, IM, DM, BM are tf[k][0..3]
      const float p0 = dl(k, 0) * OM(i, k - 1), p1 = dl(k, 1) * OI(i, k - 1), p2 = dl(k, 2) * OD(i, k - 1), p3 = dl(k, 3) * OX[(size_t)i * 5 + XB];
      cur = sM; float b = p0;
      if (p1 > b) { b = p1; cur = sI; }
      if (p2 > b) { b = p2; cur = sD; }
      if (p3 > b) { b = p3; cur = sB; }
      k--; break; }
    case sD: {          // transitions leaving node k-1: MD, DD are tf[k-1][4..5]
      const float p0 = dl(k - 1, 4) * OM(i, k - 1), p1 = dl(k - 1, 5) * OD(i, k - 1);
      cur = p0 >= p1 ? sM : sD; k--; break; }
    case sI: {          // MI, II of node k: tf[k][6..7]
      const float p0 = dl(k, 6) * OM(i - 3, k), p1 = dl(k, 7) * OI(i - 3, k);
      cur = p0 >= p1 ? sM : sI; i -= 3; break; }
    case sN: cur = (i == 0) ? sS : sN; break;
    case sC: {
      if (i < 4) { cur = sE; break; }
      const float p0 = OX[(size_t)(i - 3) * 5 + XC] + PX[(size_t)i * 5 + XC];
      const float p1 = (i < L) ? OX[(size_t)(i - 2) * 5 + XC] + PX[(size_t)(i + 1) * 5 + XC] : kTiny;
      const float p2 = (i < L - 1) ? OX[(size_t)(i - 1) * 5 + XC] + PX[(size_t)(i + 2) * 5 + XC] : kTiny;
      const float p3 = OX[(size_t)i * 5 + XE];
      cur = sC; float b = p0;
      if (p1 > b) b = p1;
      if (p2 > b) b = p2;
      if (p3 > b) { b = p3; cur = sE; }
      break; }
    case sJ: {
      if (i <= 5) { cur = sE; break; }
      const float p0 = OX[(size_t)i * 5 + XJ] + PX[(size_t)i * 5 + XJ], p1 = kTiny * OX[(size_t)i * 5 + XE];      // unihit: E->J impossible
      cur = (p1 > p0) ? sE : sJ; break; }
    case sE: {
      float mx = -INFINITY; int smax = -1, kmax = -1;
      for (int q = 1; q <= M; q++) {
        const float m = OM(i, q), d = OD(i, q);
        if (m > mx) { mx = m; smax = sM; kmax = q; }
        if (d > mx) { mx = d; smax = sD; kmax = q; }
      }
      k = kmax; cur = smax; break; }
    case sB: cur = (OX[(size_t)i * 5 + XN] > OX[(size_t)i * 5 + XJ]) ? sN : sJ; break;
    default: bad = true; break;
    }
    if (bad || cur < 0 || k < 0 || i < 0) { bad = true; break; }
    if (cur == sM) {
      c = 1; float b = post(i, k, 3);
#pragma unroll
      for (int q = 1; q < 5; q++) { const float v = post(i, k, 3 + q); if (v > b) { b = v; c = q + 1; } }
    } else c = 0;
    push(cur, k, i, c);
    if ((cur == sN || cur == sC || cur == sJ) && cur == prev) i--;
    prev = cur;
    i -= c;
    if (n > cap) bad = true;
  }
  if (!bad) {
    // forward order = T[n-1] .. T[0]
    const float *n2 = null2 + (size_t)job * kKp;
    auto st_of = [&](int z) { return (int)(T[n - 1 - z].x & 0xffu); };
    auto c_of = [&](int z) { return (int)((T[n - 1 - z].x >> 8) & 0xffu); };
    auto k_of = [&](int z) { return (int)(T[n - 1 - z].x >> 16); };
    auto i_of = [&](int z) { return (int)T[n - 1 - z].y; };
    int z1 = 0, z2 = n - 1;
    while (z1 < n && st_of(z1) != sM) z1++;
    while (z2 >= 0 && st_of(z2) != sM) z2--;
    if (z1 < n && z2 >= 0) {
      r.ok = 1;
      r.ihmm = k_of(z1); r.jhmm = k_of(z2);
      r.iali = i_of(z1) - (c_of(z1) - 1); r.jali = i_of(z2);
      float corr = 0.f;
      int t = -1, u = -1, v = -1, w = -1, x = -1, pos = 1, z = 0;
      while (pos <= L && z < n) {
        x = dsq[pos] < 4 ? (int)dsq[pos] : 1367;
        const int s = st_of(z);
        if (s == sN || s == sC || s == sJ) { if (i_of(z) == pos && pos > 2) pos++; z++; }
        else if (s == sM) {
          if (i_of(z) == pos) {
            int ci, capi;
            switch (c_of(z)) {
            case 1: ci = x * 341; capi = 1366; break;
            case 2: ci = x * 341 + w * 85 + 1; capi = 1365; break;
            case 3: ci = x * 341 + w * 85 + v * 21 + 2; capi = 1364; break;
            case 4: ci = x * 341 + w * 85 + v * 21 + u * 5 + 3; capi = 1365; break;
            default: ci = x * 341 + w * 85 + v * 21 + u * 5 + t + 4; capi = 1366; break;
            }
            if (c_of(z) != 3) r.nshift++;
            const float sc = logf(n2[codons[(size_t)k_of(z) * maxcodons + min(ci, capi)]]);
            if (sc != -INFINITY) corr += sc;
            z++;
          }
          pos++;
        } else if (s == sI) {
          if (i_of(z) == pos) {
            const float sc = logf(n2[codons[(size_t)k_of(z) * maxcodons + min(x * 341 + w * 85 + v * 21 + 2, 1364)]]);
            if (sc != -INFINITY) corr += sc;
            z++;
          }
          pos++;
        } else z++;
        t = u; u = v; v = w; w = x;
      }
      r.domcorrection = corr;
      // what the alignment display keeps of the trace (p7_alidisplay_fs_Create, p7_alidisplay.c:700-925): per column the
      // state, codon length and indel type (for --cigar), identities with the consensus, stop codons
      r.ncol = z2 - z1 + 1;
      // the columns of all envelopes lie DENSELY in <steps> / <step_pp>, each envelope's range reserved here (the host copies the
      // columns that exist -- a third of a nucleotide per column -- instead of every envelope's worst case of L + M + 16)
      r.col_off = steps ? atomicAdd(col_cursor, r.ncol) : 0;
      uint16_t *S = steps ? steps + r.col_off : nullptr;
      float *SP = (steps && step_pp) ? step_pp + r.col_off : nullptr;
      // ... and p7_pli_computeAliScores_BATH (p7_pipeline.c:781-979): per column the amino row score of the quasi-codon's best
      // amino acid plus the transition that entered the state (the last match state gets no MM: inner loops stop at z1 < z2)
      float ali = 0.f;
      int prevs = sB;
      for (int zz = z1; zz <= z2; zz++) {
        const int s = st_of(zz), cc = (s == sM) ? c_of(zz) : (s == sI ? 3 : 0), kk = k_of(zz), ii = i_of(zz);
        unsigned code = (unsigned)s;
        float colsc = 0.f;
        if (s == sI) colsc = tf[(size_t)kk * 8 + (prevs == sI ? 7 : 6)];
        else if (s == sD) colsc = tf[(size_t)(kk - 1) * 8 + (prevs == sD ? 5 : 4)];
        if (cc > 0 && ii - cc + 1 >= 1 && ii <= L) {
          int nn[5] = {0, 0, 0, 0, 0};
          bool degen = false;
          for (int q = 0; q < cc; q++) { nn[q] = dsq[ii - cc + 1 + q]; degen |= nn[q] >= 4; }
          int ci;
          switch (cc) {                                       // p7P_CODON{1..5}_FS5, hmmer.h:292-316; the last nucleotide is the most significant
          case 1: ci = degen ? 1366 : nn[0] * 341; break;
          case 2: ci = degen ? 1365 : nn[1] * 341 + nn[0] * 85 + 1; break;
          case 3: ci = degen ? 1364 : nn[2] * 341 + nn[1] * 85 + nn[0] * 21 + 2; break;
          case 4: ci = degen ? 1365 : nn[3] * 341 + nn[2] * 85 + nn[1] * 21 + nn[0] * 5 + 3; break;
          default: ci = degen ? 1366 : nn[4] * 341 + nn[3] * 85 + nn[2] * 21 + nn[1] * 5 + nn[0] + 4; break;
          }
          const int indel = indel_tab ? indel_tab[(size_t)kk * maxcodons + ci] : 5;
          const bool stop = (cc == 3) && (indel == 6 || indel == 7 || indel == 8);      // p7P_XXx, p7P_XxX, p7P_xXX
          if (stop) r.nstops++;
          if (s == sM && cons && codons[(size_t)kk * maxcodons + ci] == cons[kk]) r.exact++;
          code |= (unsigned)cc << 4 | (unsigned)indel << 8;
          if (s == sM && amino) {
            colsc = amino[(size_t)codons[(size_t)kk * maxcodons + ci] * pitch + kk];
            if (prevs == sI) colsc += tf[(size_t)kk * 8 + 1];
            else if (prevs == sD) colsc += tf[(size_t)kk * 8 + 2];
            else if (prevs == sM && zz < z2) colsc += tf[(size_t)kk * 8 + 0];
          }
        }
        ali += colsc;
        prevs = s;
        if (S) S[zz - z1] = (uint16_t)code;
        // get_postprob (generic_optacc_frameshift.c:425-440): a match state's total posterior (all codon lengths), an insert state's; 0 for D
        if (SP) SP[zz - z1] = (s == sM) ? post(ii, kk, 2) : (s == sI ? post(ii, kk, 1) : 0.0f);
      }
      r.aliscore = ali;
    }
  }
  out[job] = r;
}

// The same trace, correction and columns with the WAVE walking (round 5; fs5_trace_kernel above stays as the one-lane restatement the
// tests compare with, BATH_HIP_FS_TRACE_LANE=1).  A lane on its own pays a trip to memory for every decision -- two per match step
// (the optimal-accuracy cells, then the five codon-length posteriors of the cell it chose), one per residue of the correction loop,
// several per column: ~2 ms for the longest envelope of a pass, at the tail of the pass.  Here all 64 lanes follow the same state
// machine on uniform state and fill look-ahead buffers together:
//   * a match run of whole codons walks the slope-3 diagonal: on entering cell (i, k) lane d fetches cell (i - 3d, k - d) -- its
//     optimal-accuracy cells, B of its row, which transitions into node k - d + 1 exist, and its five codon-length posteriors
//     (formed like the decoding kernels form them when the matrix was not stored) -- and the steps take them by v_readlane until
//     a quasi-codon, an insert or a delete leaves the diagonal;
//   * the C and N flanks are decided / written 64 rows at a time; the E state's first maximum is a lane-parallel scan + a wave
//     reduction over (value, scan position);
//   * the correction loop (rescore_isolated_domain_frameshift's walk over the residues, whose nucleotide window also shifts on
//     iterations that do not advance -- kept as it is) runs serially on chunks of the trace and of the residues held a lane each,
//     its table look-ups deferred to 64 at a time, its sum in the reference's order;
//   * the columns are a lane each.
__global__ __launch_bounds__(64) void fs5_trace_wave_kernel(SeqView dna, int M, int maxcodons, const float *__restrict__ tf, const uint8_t *__restrict__ codons,
                                 const float *__restrict__ pp, const int64_t *__restrict__ pp_off, const float *__restrict__ px, const int64_t *__restrict__ x_off,
                                 const float *__restrict__ oa, const int64_t *__restrict__ oa_off, const float *__restrict__ ox,
                                 const float *__restrict__ null2 /* [n][Kp] */, uint2 *__restrict__ tbuf, const int64_t *__restrict__ t_off, FsTraceOut *__restrict__ out,
                                 const uint8_t *__restrict__ indel_tab, const uint8_t *__restrict__ cons, uint16_t *__restrict__ steps,
                                 const float *__restrict__ amino, int pitch, float *__restrict__ step_pp, int *__restrict__ col_cursor,
                                 const float *__restrict__ bck, const int64_t *__restrict__ bck_off, const float *__restrict__ bcksc, const float *__restrict__ rowden) {
  const int lane = threadIdx.x;
  const int64_t job = blockIdx.x;
  enum { sS = 0, sN, sB, sM, sD, sI, sE, sJ, sC, sT };
  enum { XE = 0, XN, XJ, XB, XC };
  const float kTiny = 1.17549435e-38f;
  const int L = dna.len[job];
  const uint8_t *dsq = dna.data + dna.off[job] - 1;             // dsq[1..L]
  const float *P = pp + pp_off[job], *PX = px + x_off[job], *O = oa + oa_off[job], *OX = ox + x_off[job];
  const float *BK = bck ? bck + bck_off[job] : nullptr, *RD = bck ? rowden + x_off[job] / 5 : nullptr;
  const float overall = bck ? bcksc[job] : 0.f;
  auto post = [&](int i, int k, int q) -> float {
    const float f = P[((size_t)i * (M + 1) + k) * 8 + q];
    if (!BK) return f;
    if (q == 1 && k >= M) return 0.f;
    const float b = BK[((size_t)i * (M + 1) + k) * 3 + (q == 1 ? 1 : 2)];
    return expf(f + b - overall) * RD[i];
  };
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto rlf = [](float v, int d) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), d)); };
  uint2 *T = tbuf + t_off[job];
  const int cap = (int)(t_off[job + 1] - t_off[job]);
  FsTraceOut r{0, 0, 0, 0, 0, 0, 0.f, 0, 0, 0, 0.f, 0};
  auto dl = [&](int node, int s) { return (node >= 1 && node <= M && tf[(size_t)node * 8 + s] != -INFINITY) ? 1.0f : kTiny; };
  auto OM = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 2]; };
  auto OI = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 1]; };
  auto OD = [&](int i, int k) { return O[((size_t)i * (M + 1) + k) * 3 + 0]; };
  auto entry = [](int st, int k, int i, int c) { return make_uint2((unsigned)st | ((unsigned)c << 8) | ((unsigned)k << 16), (unsigned)i); };
  int n = 0;
  auto push = [&](int st, int k, int i, int c) { if (n < cap && lane == 0) T[n] = entry(st, k, i, c); n++; };
  int i = L, k = 0, c = 0, prev = sC;
  bool bad = (L < 5);
  push(sT, k, i, c); push(sC, k, i, c);
  // the diagonal look-ahead: lane d holds cell (bi0 - 3 d, bk0 - d)
  float bM = 0.f, bI = 0.f, bD = 0.f, bXB = 0.f, bd0 = 0.f, bd1 = 0.f, bd2 = 0.f, bd3 = 0.f, bp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int bi0 = -(1 << 28), bk0 = -(1 << 28);
  while (!bad && prev != sS) {
    i = uni(i); k = uni(k); prev = uni(prev); n = uni(n);
    int cur = -1, dd = -1;
    if (prev == sC) {
      // rows i, i-1, ...: lane d decides row i - d; the rows that stay in C are written and skipped together
      const int rr = i - lane;
      bool stay = false;
      if (rr >= 4) {
        const float p0 = OX[(size_t)(rr - 3) * 5 + XC] + PX[(size_t)rr * 5 + XC];
        const float p1 = (rr < L) ? OX[(size_t)(rr - 2) * 5 + XC] + PX[(size_t)(rr + 1) * 5 + XC] : kTiny;
        const float p2 = (rr < L - 1) ? OX[(size_t)(rr - 1) * 5 + XC] + PX[(size_t)(rr + 2) * 5 + XC] : kTiny;
        const float p3 = OX[(size_t)rr * 5 + XE];
        float b = p0;
        if (p1 > b) b = p1;
        if (p2 > b) b = p2;
        stay = !(p3 > b);
      }
      const unsigned long long m = __ballot(stay);
      const int ns = (~m == 0ull) ? 64 : (__ffsll((long long)~m) - 1);
      if (ns > 0) {
        if (lane < ns && n + lane < cap) T[n + lane] = entry(sC, k, i - lane, 0);
        n += ns; i -= ns;
        if (n > cap) bad = true;
        continue;
      }
      cur = sE;
    } else if (prev == sN) {
      // N(i) <- N(i-1) ... <- N(1), then S at row 0: one entry per row, nothing to read
      for (int b0 = 0; b0 < i; b0 += 64) { const int rr = i - b0 - lane; if (rr >= 1 && n + b0 + lane < cap) T[n + b0 + lane] = entry(sN, k, rr, 0); }
      n += i; i = 0;
      push(sS, k, 0, 0);
      if (n > cap) bad = true;
      prev = sS;
      continue;
    } else if (prev == sM) {
      dd = bk0 - (k - 1);
      if (dd < 0 || dd >= 64 || bi0 - 3 * dd != i) {
        bi0 = i; bk0 = k - 1; dd = 0;
        const int ri = max(i - 3 * lane, 0), rk = max(k - 1 - lane, 0);
        bM = OM(ri, rk); bI = OI(ri, rk); bD = OD(ri, rk); bXB = OX[(size_t)ri * 5 + XB];
        bd0 = dl(rk + 1, 0); bd1 = dl(rk + 1, 1); bd2 = dl(rk + 1, 2); bd3 = dl(rk + 1, 3);
#pragma unroll
        for (int q = 0; q < 5; q++) bp[q] = post(ri, rk, 3 + q);
      }
      dd = uni(dd);
      const float p0 = rlf(bd0, dd) * rlf(bM, dd), p1 = rlf(bd1, dd) * rlf(bI, dd), p2 = rlf(bd2, dd) * rlf(bD, dd), p3 = rlf(bd3, dd) * rlf(bXB, dd);
      cur = sM; float b = p0;
      if (p1 > b) { b = p1; cur = sI; }
      if (p2 > b) { b = p2; cur = sD; }
      if (p3 > b) { b = p3; cur = sB; }
      k--;
    } else if (prev == sD) {
      const float p0 = dl(k - 1, 4) * OM(i, max(k - 1, 0)), p1 = dl(k - 1, 5) * OD(i, max(k - 1, 0));
      cur = p0 >= p1 ? sM : sD; k--;
    } else if (prev == sI) {
      const float p0 = dl(k, 6) * OM(max(i - 3, 0), k), p1 = dl(k, 7) * OI(max(i - 3, 0), k);
      cur = p0 >= p1 ? sM : sI; i -= 3;
    } else if (prev == sJ) {
      if (i <= 5) cur = sE;
      else { const float p0 = OX[(size_t)i * 5 + XJ] + PX[(size_t)i * 5 + XJ], p1 = kTiny * OX[(size_t)i * 5 + XE]; cur = (p1 > p0) ? sE : sJ; }
    } else if (prev == sE) {
      // the first maximum of M(i,1), D(i,1), M(i,2), D(i,2), ... (a later cell must be strictly larger)
      float mx = -INFINITY; int pos = 1 << 30;
      for (int q = 1 + lane; q <= M; q += 64) {
        const float m = OM(i, q), d = OD(i, q);
        if (m > mx) { mx = m; pos = 2 * q; }
        if (d > mx) { mx = d; pos = 2 * q + 1; }
      }
      const float best = wave_max_f32(mx);
      const int bpos = wave_min_i32((mx == best && best > -INFINITY) ? pos : (1 << 30));
      if (bpos >= (1 << 30)) bad = true;
      else { k = bpos >> 1; cur = (bpos & 1) ? sD : sM; }
    } else if (prev == sB) {
      cur = (OX[(size_t)i * 5 + XN] > OX[(size_t)i * 5 + XJ]) ? sN : sJ;
    } else bad = true;
    if (bad || cur < 0 || k < 0 || i < 0) { bad = true; break; }
    if (cur == sM) {
      float v0, v1, v2, v3, v4;
      if (prev == sM) { v0 = rlf(bp[0], dd); v1 = rlf(bp[1], dd); v2 = rlf(bp[2], dd); v3 = rlf(bp[3], dd); v4 = rlf(bp[4], dd); }
      else { v0 = post(i, k, 3); v1 = post(i, k, 4); v2 = post(i, k, 5); v3 = post(i, k, 6); v4 = post(i, k, 7); }
      c = 1; float b = v0;
      if (v1 > b) { b = v1; c = 2; }
      if (v2 > b) { b = v2; c = 3; }
      if (v3 > b) { b = v3; c = 4; }
      if (v4 > b) { b = v4; c = 5; }
    } else c = 0;
    push(cur, k, i, c);
    if ((cur == sN || cur == sC || cur == sJ) && cur == prev) i--;
    prev = cur;
    i -= c;
    if (n > cap) bad = true;
  }
  if (bad) { if (lane == 0) out[job] = r; return; }
  __threadfence();                                                            // the trace entries, written by lane 0 or a lane each, read below by every lane
  {
    // forward order = T[n-1] .. T[0]
    const float *n2 = null2 + (size_t)job * kKp;
    // the first and the last match state
    int z1 = n, z2 = -1;
    for (int zb = 0; zb < n; zb += 64) {
      const int z = zb + lane;
      const bool isM = z < n && (int)(T[n - 1 - z].x & 0xffu) == sM;
      const unsigned long long m = __ballot(isM);
      if (m) { if (z1 == n) z1 = zb + __ffsll((long long)m) - 1; z2 = zb + 63 - __clzll((long long)m); }
    }
    if (z1 < n && z2 >= 0) {
      r.ok = 1;
      const uint2 e1 = T[n - 1 - z1], e2 = T[n - 1 - z2];
      r.ihmm = (int)(e1.x >> 16); r.jhmm = (int)(e2.x >> 16);
      r.iali = (int)e1.y - ((int)((e1.x >> 8) & 0xffu) - 1); r.jali = (int)e2.y;
      // ---- the correction: the reference's loop over the residues, on chunks of the trace and of the residues held a lane each
      float corr = 0.f;
      int nshift = 0;
      {
        int t = -1, u = -1, v = -1, w = -1, x = -1, pos = 1, z = 0;
        int zb = 0, pb = 1;                                                   // the chunks: lane l holds entry zb + l, residue pb + l
        uint2 te = (zb + lane < n) ? T[n - 1 - (zb + lane)] : make_uint2(0u, 0u);
        int re = (pb + lane <= L) ? (int)dsq[pb + lane] : 0;
        int pk = 0, pc = 0, npend = 0;                                        // pending look-ups: lane e holds (node, codon index) of the e-th
        auto flush = [&]() {
          float sc = 0.f;
          if (lane < npend) sc = logf(n2[codons[(size_t)pk * maxcodons + pc]]);
          for (int l = 0; l < npend; l++) { const float q = rlf(sc, l); if (q != -INFINITY) corr += q; }
          npend = 0;
        };
        while (pos <= L && z < n) {
          pos = uni(pos); z = uni(z);
          if (z - zb >= 64) { zb = z; te = (zb + lane < n) ? T[n - 1 - (zb + lane)] : make_uint2(0u, 0u); }
          if (pos - pb >= 64) { pb = pos; re = (pb + lane <= L) ? (int)dsq[pb + lane] : 0; }
          const int rv = __builtin_amdgcn_readlane(re, pos - pb);
          x = rv < 4 ? rv : 1367;
          const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)te.x, z - zb);
          const int ei = __builtin_amdgcn_readlane((int)te.y, z - zb);
          const int s = (int)(ex & 0xffu), cz = (int)((ex >> 8) & 0xffu), kz = (int)(ex >> 16);
          if (s == sN || s == sC || s == sJ) { if (ei == pos && pos > 2) pos++; z++; }
          else if (s == sM || s == sI) {
            if (ei == pos) {
              int ci, capi;
              if (s == sI) { ci = x * 341 + w * 85 + v * 21 + 2; capi = 1364; }
              else {
                switch (cz) {
                case 1: ci = x * 341; capi = 1366; break;
                case 2: ci = x * 341 + w * 85 + 1; capi = 1365; break;
                case 3: ci = x * 341 + w * 85 + v * 21 + 2; capi = 1364; break;
                case 4: ci = x * 341 + w * 85 + v * 21 + u * 5 + 3; capi = 1365; break;
                default: ci = x * 341 + w * 85 + v * 21 + u * 5 + t + 4; capi = 1366; break;
                }
                if (cz != 3) nshift++;
              }
              if (lane == npend) { pk = kz; pc = min(ci, capi); }
              if (++npend == 64) flush();
              z++;
            }
            pos++;
          } else z++;
          t = u; u = v; v = w; w = x;
        }
        flush();
      }
      r.nshift = nshift;
      r.domcorrection = corr;
      // ---- the columns (p7_alidisplay_fs_Create, p7_pli_computeAliScores_BATH), a lane each; the score summed in column order
      r.ncol = z2 - z1 + 1;
      int col_off = 0;
      if (steps) { if (lane == 0) col_off = atomicAdd(col_cursor, r.ncol); col_off = uni(col_off); }
      r.col_off = col_off;
      uint16_t *S = steps ? steps + col_off : nullptr;
      float *SP = (steps && step_pp) ? step_pp + col_off : nullptr;
      float ali = 0.f;
      int nstops = 0, exact = 0;
      for (int zb = z1; zb <= z2; zb += 64) {
        const int zz = zb + lane;
        const bool in = zz <= z2;
        float colsc = 0.f;
        bool stop = false, same = false;
        if (in) {
          const uint2 e = T[n - 1 - zz];
          const int s = (int)(e.x & 0xffu), cc0 = (int)((e.x >> 8) & 0xffu), kk = (int)(e.x >> 16), ii = (int)e.y;
          const int prevs = (zz == z1) ? (int)sB : (int)(T[n - zz].x & 0xffu);
          const int cc = (s == sM) ? cc0 : (s == sI ? 3 : 0);
          unsigned code = (unsigned)s;
          if (s == sI) colsc = tf[(size_t)kk * 8 + (prevs == sI ? 7 : 6)];
          else if (s == sD) colsc = tf[(size_t)(kk - 1) * 8 + (prevs == sD ? 5 : 4)];
          if (cc > 0 && ii - cc + 1 >= 1 && ii <= L) {
            int nn[5] = {0, 0, 0, 0, 0};
            bool degen = false;
            for (int q = 0; q < cc; q++) { nn[q] = dsq[ii - cc + 1 + q]; degen |= nn[q] >= 4; }
            int ci;
            switch (cc) {
            case 1: ci = degen ? 1366 : nn[0] * 341; break;
            case 2: ci = degen ? 1365 : nn[1] * 341 + nn[0] * 85 + 1; break;
            case 3: ci = degen ? 1364 : nn[2] * 341 + nn[1] * 85 + nn[0] * 21 + 2; break;
            case 4: ci = degen ? 1365 : nn[3] * 341 + nn[2] * 85 + nn[1] * 21 + nn[0] * 5 + 3; break;
            default: ci = degen ? 1366 : nn[4] * 341 + nn[3] * 85 + nn[2] * 21 + nn[1] * 5 + nn[0] + 4; break;
            }
            const int indel = indel_tab ? indel_tab[(size_t)kk * maxcodons + ci] : 5;
            stop = (cc == 3) && (indel == 6 || indel == 7 || indel == 8);
            same = s == sM && cons && codons[(size_t)kk * maxcodons + ci] == cons[kk];
            code |= (unsigned)cc << 4 | (unsigned)indel << 8;
            if (s == sM && amino) {
              colsc = amino[(size_t)codons[(size_t)kk * maxcodons + ci] * pitch + kk];
              if (prevs == sI) colsc += tf[(size_t)kk * 8 + 1];
              else if (prevs == sD) colsc += tf[(size_t)kk * 8 + 2];
              else if (prevs == sM && zz < z2) colsc += tf[(size_t)kk * 8 + 0];
            }
          }
          if (S) S[zz - z1] = (uint16_t)code;
          if (SP) SP[zz - z1] = (s == sM) ? post(ii, kk, 2) : (s == sI ? post(ii, kk, 1) : 0.0f);
        }
        nstops += __popcll(__ballot(stop)); exact += __popcll(__ballot(same));
        const int nin = min(64, z2 - zb + 1);
        for (int l = 0; l < nin; l++) ali += rlf(colsc, l);
      }
      r.nstops = nstops; r.exact = exact;
      r.aliscore = ali;
    }
  }
  if (lane == 0) out[job] = r;
}

__global__ void fs5_null2_kernel(int64_t n, const int32_t *__restrict__ len, int M, int pitch, const float *__restrict__ amino /* rsc + maxcodons*pitch */,
                                 const float *__restrict__ logsum, const float *__restrict__ colsum, float *__restrict__ null2 /* [n][Kp] */) {
  const int64_t job = blockIdx.x;
  const int x = threadIdx.x;
  if (job >= n) return;
  const int Ld = len[job];
  float *out = null2 + (size_t)job * kKp;
  if (Ld < 5) { if (x < kKp) out[x] = 1.0f; return; }
  const float *cs = colsum + (size_t)job * ((size_t)(M + 1) * 8 + 8);
  const float *xs = cs + (size_t)(M + 1) * 8;
  const float lld = (float)-log((double)(float)Ld);
  __shared__ float res[20];
  if (x < 20) {
    float xf = (float)log((double)xs[1]) + lld;
    xf = flogsum_g(xf, (float)log((double)xs[4]) + lld, logsum);
    xf = flogsum_g(xf, (float)log((double)xs[2]) + lld, logsum);
    float v = -INFINITY;
    for (int k = 1; k < M; k++) {
      v = flogsum_g(v, ((float)log((double)cs[(size_t)k * 8 + 2]) + lld) + amino[(size_t)x * pitch + k], logsum);
      v = flogsum_g(v, (float)log((double)cs[(size_t)k * 8 + 1]) + lld, logsum);
    }
    v = flogsum_g(v, ((float)log((double)cs[(size_t)M * 8 + 2]) + lld) + amino[(size_t)x * pitch + M], logsum);
    v = flogsum_g(v, xf, logsum);
    res[x] = expf(v);
    out[x] = res[x];
  }
  __syncthreads();
  if (x == 0) {
    // esl_abc_FAvgScVec: degenerate residues = plain mean over members; gap, '*', '~' = 1
    const int mem[5][2] = {{2, 11}, {7, 9}, {3, 13}, {8, 8}, {1, 1}};
    for (int dx = 0; dx < 5; dx++) {
      const int a = mem[dx][0], b = mem[dx][1];
      out[21 + dx] = (a == b) ? res[a] / 1.0f : ((a < b ? res[a] + res[b] : res[b] + res[a]) / 2.0f);
    }
    float s = 0.f;
    for (int y = 0; y < 20; y++) s += res[y];
    out[26] = s / 20.0f;
    out[20] = 1.0f; out[27] = 1.0f; out[28] = 1.0f;
  }
}

}  // namespace bath

// =================================================================================================
// host side
// =================================================================================================

int bath_hip_fsprofile::ensure_len(int maxL_amino) const {
  std::lock_guard<std::mutex> lock(grow_mu);
  if (maxL_amino <= maxL) return BATH_OK;
  const int n = std::max(maxL_amino, 4096) + 1;
  for (int h = 0; h < 2; h++) {
    const float nj = (h == 0) ? 1.0f : 0.0f;                       // 0: multihit (parsers), 1: unihit (envelopes)
    std::vector<float> lo(n), mv(n);
    for (int L = 0; L < n; L++) {
      const float pmove = (2.0f + nj) / ((float)L + 2.0f + nj);    // p7_fs_ReconfigLength, modelconfig.c:767-770
      const float ploop = 1.0f - pmove;
      lo[L] = (float)std::log((double)ploop); mv[L] = (float)std::log((double)pmove);
    }
    float *nl = nullptr, *nm = nullptr;                            // complete before they are published; the old tables are freed with the profile
    BATH_HIP_TRY(ctx, hipMalloc((void **)&nl, n * sizeof(float)));
    BATH_HIP_TRY(ctx, hipMalloc((void **)&nm, n * sizeof(float)));
    BATH_HIP_TRY(ctx, hipMemcpy(nl, lo.data(), n * sizeof(float), hipMemcpyHostToDevice));
    BATH_HIP_TRY(ctx, hipMemcpy(nm, mv.data(), n * sizeof(float), hipMemcpyHostToDevice));
    if (d_loop[h]) retired.push_back(d_loop[h]);
    if (d_move[h]) retired.push_back(d_move[h]);
    d_loop[h] = nl; d_move[h] = nm;
  }
  maxL = n - 1;
  return BATH_OK;
}

extern "C" void bath_hip_fsprofile_destroy(bath_hip_fsprofile *om) {
  if (!om) return;
  for (void *p : {(void *)om->d_codons, (void *)om->d_indel, (void *)om->d_rsc, (void *)om->d_tf, (void *)om->d_tb, (void *)om->d_logsum, (void *)om->d_loop[0], (void *)om->d_loop[1],
                  (void *)om->d_move[0], (void *)om->d_move[1]})
    if (p) (void)hipFree(p);
  for (void *p : om->retired) (void)hipFree(p);
  delete om;
}

extern "C" int bath_hip_fsprofile_convert(bath_hip_ctx *ctx, const bath_fs_profile *gm, bath_hip_fsprofile **ret) {
  *ret = nullptr;
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int M = gm->M;
  bath_hip_fsprofile *om = new bath_hip_fsprofile();
  om->ctx = ctx; om->M = M; om->codon_lengths = gm->codon_lengths; om->maxcodons = gm->maxcodons; om->max_length = gm->max_length;
  om->fsprob = gm->fsprob;
  std::memcpy(om->evparam, gm->evparam, sizeof om->evparam);
  om->h_tsc.assign(gm->tsc, gm->tsc + (size_t)M * 8);
  if (gm->codons) {
    om->h_codons.assign(gm->codons, gm->codons + (size_t)(M + 1) * gm->maxcodons);
    BATH_HIP_TRY(ctx, hipMalloc((void **)&om->d_codons, om->h_codons.size() + 64));
    BATH_HIP_TRY(ctx, hipMemcpy(om->d_codons, om->h_codons.data(), om->h_codons.size(), hipMemcpyHostToDevice));
    if (gm->indel_pos) {
      BATH_HIP_TRY(ctx, hipMalloc((void **)&om->d_indel, om->h_codons.size() + 64));
      BATH_HIP_TRY(ctx, hipMemcpy(om->d_indel, gm->indel_pos, om->h_codons.size(), hipMemcpyHostToDevice));
    }
  }
  om->pitch = (M + 1 + 3) / 4 * 4;
  const int nrows = gm->maxcodons + kKp;
  std::vector<float> rsc((size_t)nrows * om->pitch, -INFINITY);
  for (int r = 0; r < nrows; r++) std::memcpy(&rsc[(size_t)r * om->pitch], gm->rsc + (size_t)r * (M + 1), sizeof(float) * (M + 1));
  enum { MM, IM, DM, BM, MD, DD, MI, II };
  auto tsc = [&](int k, int s) -> float { return (k >= 0 && k < M) ? gm->tsc[(size_t)k * 8 + s] : -INFINITY; };
  std::vector<float> tf((size_t)(M + 2) * 8, -INFINITY), tb((size_t)(M + 2) * 8, -INFINITY);
  for (int k = 1; k <= M; k++) {
    float *f = &tf[(size_t)k * 8];
    f[0] = tsc(k - 1, MM); f[1] = tsc(k - 1, IM); f[2] = tsc(k - 1, DM); f[3] = tsc(k - 1, BM);
    f[4] = tsc(k, MD); f[5] = tsc(k, DD); f[6] = tsc(k, MI); f[7] = tsc(k, II);
    float *b = &tb[(size_t)k * 8];
    b[0] = tsc(k, MD); b[1] = tsc(k, MI); b[2] = tsc(k, MM); b[3] = tsc(k, DD); b[4] = tsc(k, DM); b[5] = tsc(k, II); b[6] = tsc(k, IM);
    b[7] = tsc(k - 1, BM);
  }
  std::vector<float> tbl(kLogsumTbl);
  for (int i = 0; i < kLogsumTbl; i++) tbl[i] = (float)std::log(1. + std::exp((double)-i / 1000.f));     // p7_FLogsumInit, logsum.c:89
  auto up = [&](float **dst, const std::vector<float> &src) -> hipError_t {
    hipError_t e = hipMalloc((void **)dst, src.size() * sizeof(float) + 64);
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
  };
  BATH_HIP_TRY(ctx, up(&om->d_rsc, rsc));
  BATH_HIP_TRY(ctx, up(&om->d_tf, tf));
  BATH_HIP_TRY(ctx, up(&om->d_tb, tb));
  BATH_HIP_TRY(ctx, up(&om->d_logsum, tbl));
  int st = om->ensure_len(4096);
  if (st != BATH_OK) { bath_hip_fsprofile_destroy(om); return st; }
  *ret = om;
  return BATH_OK;
}

namespace bath {

static int fs_columns(int M) {
  const int c = (M + 63) / 64;
  for (int opt : {1, 2, 3, 4, 6, 8, 12, 16, 20}) if (c <= opt) return opt;
  return -1;
}

#define BATH_FS_SWITCH(Cv, BODY)                          \
  switch (Cv) {                                           \
    case 1: { constexpr int CC = 1; BODY } break;         \
    case 2: { constexpr int CC = 2; BODY } break;         \
    case 3: { constexpr int CC = 3; BODY } break;         \
    case 4: { constexpr int CC = 4; BODY } break;         \
    case 6: { constexpr int CC = 6; BODY } break;         \
    case 8: { constexpr int CC = 8; BODY } break;         \
    case 12: { constexpr int CC = 12; BODY } break;       \
    case 16: { constexpr int CC = 16; BODY } break;       \
    case 20: { constexpr int CC = 20; BODY } break;       \
    default: ctx->set_error("frameshift kernels support models up to 1280 nodes"); return BATH_EINVAL; \
  }

// logsum_mode -> kernel MODE: 0 table + wavefront scans, 1 exact log-sums, 2 table in the reference's serial order ("strict")
#define BATH_FS_MODE(modev, BODY)                         \
  switch (modev) {                                        \
    case 1: { constexpr int MD = 1; BODY } break;         \
    case 2: { constexpr int MD = 2; BODY } break;         \
    default: { constexpr int MD = 0; BODY } break;        \
  }

static FsDev fsdev(const bath_hip_fsprofile *om) { return FsDev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum}; }

static int fs_grid(bath_hip_ctx *ctx, int64_t n) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->prop.multiProcessorCount * 2));
}
static int fs_grid_dp(bath_hip_ctx *ctx, int64_t n) {           // blocks of kFsBlock threads, two per CU
  const int wpb = kFsBlock / 64;
  return (int)std::max<int64_t>(1, std::min<int64_t>((n + wpb - 1) / wpb, (int64_t)ctx->prop.multiProcessorCount * 2));
}

template <class K>
static int fs_set_shmem(bath_hip_ctx *ctx, K kernel, size_t shmem) {
  if (shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)kernel));
  return BATH_OK;
}

// xmx rows live in a device buffer laid out like the host array the caller passes (offsets in floats)
static int upload_offsets(bath_hip_ctx *ctx, DevBuf &buf, const int64_t *off, int64_t n) {
  BATH_HIP_TRY(ctx, buf.reserve((size_t)(n + 1) * sizeof(int64_t)));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(buf.p, off, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  return BATH_OK;
}

// the windows of <dna> by decreasing length and <k> zeroed job counters (one per kernel launch that will draw from the list),
// queued on ctx->stream; jobs[i] is what launch i passes to its kernel
static int fs_schedule(bath_hip_ctx *ctx, const bath_hip_seqs *dna, int k, FsJobs *jobs) {
  const int64_t n = dna->n;
  std::vector<int32_t> order;
  fs_order_by_length_desc(dna->h_len.data(), n, &order);
  if (k > 64) { ctx->set_error("fs_schedule: more than 64 job counters"); return BATH_EINVAL; }
  DevBuf &b = ctx->scratch[40];                                    // its own slot: the cascade's local-composition terms live in 36
  BATH_HIP_TRY(ctx, b.reserve(256 + (size_t)n * sizeof(int32_t) + 64));
  BATH_HIP_TRY(ctx, hipMemsetAsync(b.p, 0, 256, ctx->stream));
  if (ctx->stage_upload(0, b.as<char>() + 256, order.data(), (size_t)n, ctx->stream) != BATH_OK) return BATH_EFAIL;   // through page-locked staging: no synchronize
  for (int i = 0; i < k; i++) jobs[i] = FsJobs{reinterpret_cast<const int32_t *>(b.as<char>() + 256), b.as<unsigned>() + i};
  return BATH_OK;
}

}  // namespace bath

static int fs3_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int logsum_mode, float *sc, float *xmx,
                      const int64_t *xmx_off, bool backward, DevBuf *keep = nullptr /* the rows stay in this device buffer instead of going to <xmx> */,
                      const std::function<int()> *after_launch = nullptr /* runs between the kernel's launch and the wait for it */) {
  if (!ctx || !om || !dna || om->codon_lengths != 3) { if (ctx) ctx->set_error("fs3 parser needs a 3-codon profile"); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (logsum_mode == BATH_LOGSUM_CONTEXT) logsum_mode = ctx->fs_strict ? BATH_LOGSUM_TABLE_SERIAL : BATH_LOGSUM_TABLE;
  const int64_t n = dna->n;
  if (n == 0) return BATH_OK;
  int st = om->ensure_len(dna->maxlen / 3 + 1);
  if (st != BATH_OK) return st;
  DevBuf &b_sc = ctx->scratch[12], &b_x = ctx->scratch[13], &b_off = ctx->scratch[14];
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * sizeof(float)));
  float *d_x = nullptr;
  if (xmx || keep) {
    DevBuf &bx = keep ? *keep : b_x;
    BATH_HIP_TRY(ctx, bx.reserve((size_t)xmx_off[n] * sizeof(float) + 64));
    if ((st = upload_offsets(ctx, b_off, xmx_off, n)) != BATH_OK) return st;
    d_x = bx.as<float>();
  }
  const int Cv = fs_columns(om->M);
  const size_t shmem = (size_t)(kLogsumTbl + (om->M + 2) * 8) * sizeof(float);
  const float tE = (float)-0.69314718055994529;
  [[maybe_unused]] const int grid = fs_grid(ctx, n);
  const int grid_dp = fs_grid_dp(ctx, n);
  FsJobs jq[1];
  if ((st = fs_schedule(ctx, dna, 1, jq)) != BATH_OK) return st;
  const bool chain = logsum_mode == BATH_LOGSUM_TABLE_SERIAL && fs_chain_enabled();
  StageGate gate(chain ? ctx->device : -1, backward ? StageGate::kBwdChain : StageGate::kFwdChain);   // held until this stage's kernels have finished (the synchronize below)
  const int sp = ctx->span_begin(backward ? "fs_bwd_kernel<3>" : "fs3_fwd_kernel", ctx->stream, (double)dna->total * om->M, (double)dna->total * ((xmx || keep) ? 21.0 : 1.0));
  if (logsum_mode == BATH_LOGSUM_TABLE_SERIAL && fs_chain_enabled()) {
    if (!backward) st = launch_fs3_fwd_chain(ctx, ctx->stream, om, dna, Cv, tE, tE, b_sc.as<float>(), d_x, b_off.as<int64_t>(), jq[0]);
    else st = launch_fs3_bwd_chain(ctx, ctx->stream, om, dna, Cv, tE, tE, b_sc.as<float>(), d_x, b_off.as<int64_t>(), jq[0]);
    if (st != BATH_OK) return st;
  } else
  BATH_FS_SWITCH(Cv, BATH_FS_MODE(logsum_mode, {
    if (!backward) {
      if ((st = fs_set_shmem(ctx, fs3_fwd_kernel<CC, MD>, shmem)) != BATH_OK) return st;
      hipLaunchKernelGGL((fs3_fwd_kernel<CC, MD>), dim3(grid_dp), dim3(kFsBlock), shmem, ctx->stream, dna->view(), fsdev(om), om->d_loop[0], om->d_move[0], tE, tE, b_sc.as<float>(), d_x, b_off.as<int64_t>(), jq[0]);
    } else {
      if ((st = fs_set_shmem(ctx, fs_bwd_kernel<CC, 3, MD>, shmem)) != BATH_OK) return st;
      hipLaunchKernelGGL((fs_bwd_kernel<CC, 3, MD>), dim3(grid_dp), dim3(kFsBlock), shmem, ctx->stream, dna->view(), fsdev(om), om->d_loop[0], om->d_move[0], tE, tE, b_sc.as<float>(), (float *)nullptr, (const int64_t *)nullptr, d_x, b_off.as<int64_t>(), jq[0]);
    }
  }))
  ctx->span_end(sp, ctx->stream);
  BATH_HIP_TRY(ctx, hipGetLastError());
  if (after_launch && *after_launch && (st = (*after_launch)()) != BATH_OK) return st;
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc, b_sc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (xmx) BATH_HIP_TRY(ctx, hipMemcpyAsync(xmx, d_x, (size_t)xmx_off[n] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}

// p7_DomainDecoding_Frameshift (generic form, generic_decoding_frameshift.c:204-290) and the region heuristics of
// p7_domaindef_ByPosteriorHeuristics_Frameshift_BATH (p7_domaindef.c:328-392, is_multidomain_region_frameshift :684-714)
// on the parsers' special-state rows, one WAVE per window: the posterior terms (11 expf per position) are elementwise and
// computed by all 64 lanes; the two running sums and the scan for regions are serial and short, lane 0 does them with the
// reference's own order of additions.  Only the regions (a few integers per window) travel to the host.
constexpr int kMaxRegions = 64;       // regions kept per DNA window; a window with more reports the true count and the call fails loudly (the reference has no cap)
__global__ __launch_bounds__(64) void fs_regions_kernel(int64_t n, const int32_t *__restrict__ len, const float *__restrict__ fx, const float *__restrict__ bx,
                                  const int64_t *__restrict__ x_off, const int64_t *__restrict__ fx_off /* where the Forward rows start (they may live in another buffer) */,
                                  const float *__restrict__ tbl, float loop, float *__restrict__ work /* 3 floats per xmx row */,
                                  int32_t *__restrict__ regions /* [n][1 + 3*kMaxRegions]: count, then {i, j, multidomain} */) {
  const int64_t w = blockIdx.x;
  const int lane = threadIdx.x;
  if (w >= n) return;
  enum { XE = 0, XN, XJ, XB, XC };
  const int L = len[w];
  const float *F = fx + fx_off[w], *B = bx + x_off[w];
  float *btot = work + (x_off[w] / 5) * 3, *etot = btot + (L + 1), *mocc = etot + (L + 1);
  int32_t *out = regions + w * (1 + 3 * kMaxRegions);
  int nreg = 0;
  if (L < 6) { if (lane == 0) out[0] = 0; return; }
  const float Z = flogsum_g(B[0 * 5 + XN], flogsum_g(B[1 * 5 + XN], B[2 * 5 + XN], tbl), tbl);
  if (!(Z > -INFINITY)) { if (lane == 0) out[0] = -1; return; }               // Backward underflow: the window is skipped (p7_pipeline.c:1471)
  auto em = [&](int s, int a, int b) { return expf(F[(size_t)a * 5 + s] + B[(size_t)b * 5 + s] + loop - Z); };
  for (int i = lane; i <= L; i += 64) {
    if (i < 3) { btot[i] = etot[i] = mocc[i] = 0.f; continue; }
    btot[i] = expf(F[(size_t)(i - 3) * 5 + XB] + B[(size_t)(i - 3) * 5 + XB] - Z);      // the terms; summed below
    etot[i] = expf(F[(size_t)i * 5 + XE] + B[(size_t)i * 5 + XE] - Z);
    float p = 0.0f;
    if (i < L - 1) {
      p += em(XN, i - 3, i); p += em(XN, i - 2, i + 1); p += em(XN, i - 1, i + 2);
      p += em(XC, i - 3, i); p += em(XC, i - 2, i + 1); p += em(XC, i - 1, i + 2);
      p += em(XJ, i - 3, i); p += em(XJ, i - 2, i + 1); p += em(XJ, i - 1, i + 2);
    } else if (i == L - 1) {
      p += em(XN, L - 4, L - 1); p += em(XN, L - 3, L); p += em(XC, L - 4, L - 1); p += em(XC, L - 3, L); p += em(XJ, L - 4, L - 1); p += em(XJ, L - 3, L);
    } else {
      p += em(XN, L - 3, L); p += em(XC, L - 3, L); p += em(XJ, L - 3, L);
    }
    mocc[i] = (float)(1. - p);
  }
  __syncthreads();
  if (lane != 0) return;
  out[0] = 0;
  for (int i = 3; i <= L; i++) { btot[i] = btot[i - 3] + btot[i]; etot[i] = etot[i - 3] + etot[i]; }
  const float rt1 = 0.25f, rt2 = 0.10f, rt3 = 0.20f;                         // p7_domaindef.c:80-82
  bool triggered = false;
  int d = 0;
  for (int j = 1; j < L; j++) {
    if (!triggered) { if (mocc[j] >= rt1) triggered = true; d = j; continue; }
    bool found = false;
    while (d > 1 && !found) {                                                // the start must show in all three frames
      int run = 0;
      d--;
      while (run < 3 && d > 3 && mocc[d] - (btot[d] - btot[d - 3]) < rt2) { d--; run++; }
      if (run == 3) found = true;
    }
    const int i = max(1, d - 3);
    d = j + 1;
    found = false;
    while (d < L && !found) {
      int run = 0;
      d++;
      while (run < 3 && d < L && mocc[d] - (etot[d] - etot[d - 3]) < rt2) { d++; run++; }
      if (run == 3) found = true;
    }
    j = min(L, d + 3);
    if (j - i + 1 >= 12) {
      float best = -1.0f;                                                    // is_multidomain_region_frameshift
      for (int ph = 0; ph < 3; ph++) {
        const int f = (j - i + 1 - ph) % 3;
        for (int z = i + 2 + ph; z <= j - f; z += 3) best = fmaxf(best, fminf(etot[z] - etot[i - 1 + ph], btot[j - f] - btot[z - 3]));
      }
      if (nreg < kMaxRegions) { out[1 + 3 * nreg] = i; out[2 + 3 * nreg] = j; out[3 + 3 * nreg] = best >= rt3 ? 1 : 0; }
      nreg++;
    }
    triggered = false;
  }
  out[0] = nreg;                        // may exceed kMaxRegions: the host checks
}

namespace bath {
// Both 3-codon parsers and the regions of every window of <dna>; regions_out[i] = {count (or -1: Backward underflow),
// then count x {i, j, multidomain}}, 1 + 3*fs_max_regions() ints per window.
int fs_max_regions() { return kMaxRegions; }
// Forward and Backward of a batch are independent and, with few windows, each is bound by the latency of its longest window's
// row chain: run Backward on the context's side stream while Forward runs on the main one.
int fs_fork(bath_hip_ctx *ctx) {
  if (!ctx->side_stream) {
    static const bool lowprio = [] { const char *e = std::getenv("BATH_HIP_FS_SIDE_LOWPRIO"); return e && e[0] == '1'; }();   // (probe)
    if (lowprio) {
      int lo = 0, hi = 0;
      BATH_HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
      BATH_HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, lo));
    } else
    BATH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
  return BATH_OK;
}
int fs_join(bath_hip_ctx *ctx) {
  BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
  BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return BATH_OK;
}

// rows of the selected windows from one buffer into another: (len + 1) x 5 floats each, a block per window
__global__ void fs_rows_copy_kernel(int n, const int32_t *__restrict__ len, const int64_t *__restrict__ src_off /* -1: nothing to copy */, const float *__restrict__ src,
                                    const int64_t *__restrict__ dst_off, float *__restrict__ dst) {
  const int w = blockIdx.x;
  if (w >= n || src_off[w] < 0) return;
  const int cnt = (len[w] + 1) * 5;
  const float *a = src + src_off[w];
  float *b = dst + dst_off[w];
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) b[i] = a[i];
}

// Speculative Backward (strict mode, a host with ONE context at work): the 3-codon Backward parser of the <k> longest windows of
// <dna> on the context's speculation stream, beside whatever follows on the main stream -- the Forward parser of all windows.
// The pass is a chain of stages each as long as its longest window (Forward 9.4 ms, then Backward 14.2 ms on the bench block);
// the two parsers of a window do not read each other, only the DECISION to run Backward reads Forward's score (p7_pipeline.c:1464-1470).
// Rows go to scratch[53] at the offsets fs_spec_rows records; fs3_regions (the domain stage) waits for the stream, runs the parser
// for the frameshift-branch windows that were not among the <k>, and takes the others' rows from here.  What the speculation computes
// for windows that turn out to take the standard branch is thrown away: CU time nothing else of this context could use, which is
// why a host running several worker contexts does not speculate (bath_pipeline.hip).
int fs3_backward_spec(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int k) {
  ctx->fs_spec_valid = false;
  const int64_t n = dna->n;
  if (n == 0 || k <= 0 || !ctx->fs_strict || !fs_chain_enabled() || om->codon_lengths != 3) return BATH_OK;
  int st = om->ensure_len(dna->maxlen / 3 + 1);
  if (st != BATH_OK) return st;
  if (!ctx->spec_stream) {
    BATH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->spec_stream, hipStreamNonBlocking));
    BATH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_spec, hipEventDisableTiming));
  }
  std::vector<int32_t> order((size_t)n);
  for (int64_t i = 0; i < n; i++) order[(size_t)i] = (int32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return dna->h_len[(size_t)a] > dna->h_len[(size_t)b]; });
  k = (int)std::min<int64_t>(k, n);
  order.resize((size_t)k);
  // rows of the chosen windows back to back; every other window: -1
  ctx->fs_spec_rows.assign((size_t)n, -1);
  std::vector<int64_t> xoff((size_t)n, 0);
  int64_t tot = 0;
  for (int32_t w : order) { ctx->fs_spec_rows[(size_t)w] = tot; xoff[(size_t)w] = tot; tot += ((int64_t)dna->h_len[(size_t)w] + 1) * 5; }
  DevBuf &b_rows = ctx->scratch[53], &b_aux = ctx->scratch[54];
  BATH_HIP_TRY(ctx, b_rows.reserve((size_t)tot * sizeof(float) + 64));
  // aux: job counter (256 B), order list, per-window row offsets, scores
  const size_t o_ord = 256, o_xoff = o_ord + (((size_t)k * 4 + 255) & ~(size_t)255), o_sc = o_xoff + (((size_t)n * 8 + 255) & ~(size_t)255);
  BATH_HIP_TRY(ctx, b_aux.reserve(o_sc + (size_t)n * 4 + 64));
  hipStream_t s = ctx->spec_stream;
  // (the windows' pool was gathered on ctx->stream BEFORE the Forward parser's launch, and the side stream was forked behind the
  // gather: ev_fork is that point, so the speculation does not wait for Forward)
  BATH_HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_fork, 0));
  BATH_HIP_TRY(ctx, hipMemsetAsync(b_aux.p, 0, 256, s));
  if (ctx->stage[8].reserve((size_t)k * 4 + (size_t)n * 8 + 64) != hipSuccess) { ctx->set_error("cannot allocate page-locked staging memory"); return BATH_EFAIL; }
  char *hs = static_cast<char *>(ctx->stage[8].p);
  std::memcpy(hs, order.data(), (size_t)k * 4);
  std::memcpy(hs + (size_t)k * 4, xoff.data(), (size_t)n * 8);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(b_aux.as<char>() + o_ord, hs, (size_t)k * 4, hipMemcpyHostToDevice, s));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(b_aux.as<char>() + o_xoff, hs + (size_t)k * 4, (size_t)n * 8, hipMemcpyHostToDevice, s));
  // the launcher sizes its blocks and batches from a block's host arrays: a shadow of <dna> that holds the chosen windows only
  // (the kernel addresses windows by their index in <dna> through the order list; n = the length of that list)
  bath_hip_seqs shadow;
  shadow.ctx = ctx; shadow.is_part = true; shadow.n = k; shadow.d_data = dna->d_data; shadow.d_off = dna->d_off; shadow.d_len = dna->d_len;
  shadow.h_len.resize((size_t)k);
  shadow.maxlen = 0; shadow.total = 0;
  for (int i = 0; i < k; i++) { const int32_t L = dna->h_len[(size_t)order[(size_t)i]]; shadow.h_len[(size_t)i] = L; shadow.maxlen = std::max(shadow.maxlen, L); shadow.total += L; }
  const FsJobs jq{reinterpret_cast<const int32_t *>(b_aux.as<char>() + o_ord), b_aux.as<unsigned>()};
  const float tE = (float)-0.69314718055994529;
  const int sp = ctx->span_begin("fs_bwd_kernel<3> (speculative)", s, (double)(tot / 5) * om->M, (double)(tot / 5) * 21.0);
  st = launch_fs3_bwd_chain(ctx, s, om, &shadow, fs_columns(om->M), tE, tE, reinterpret_cast<float *>(b_aux.as<char>() + o_sc), b_rows.as<float>(),
                            reinterpret_cast<const int64_t *>(b_aux.as<char>() + o_xoff), jq, 4, 57, 9);
  ctx->span_end(sp, s);
  shadow.d_data = nullptr; shadow.d_off = nullptr; shadow.d_len = nullptr;   // borrowed
  if (st != BATH_OK) return st;
  ctx->fs_spec_valid = true;
  return BATH_OK;
}

int fs3_regions(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, float loop, int32_t *regions_out, float *fwd_sc_out, const int32_t *kept) {
  if (!ctx || !om || !dna || om->codon_lengths != 3) { if (ctx) ctx->set_error("fs3 parser needs a 3-codon profile"); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t n = dna->n;
  if (n == 0) return BATH_OK;
  int st = om->ensure_len(dna->maxlen / 3 + 1);
  if (st != BATH_OK) return st;
  std::vector<int64_t> xoff((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; i++) xoff[(size_t)i + 1] = xoff[(size_t)i] + ((int64_t)dna->h_len[(size_t)i] + 1) * 5;
  DevBuf &b_sc = ctx->scratch[12], &b_fx = ctx->scratch[13], &b_off = ctx->scratch[14], &b_bx = ctx->scratch[11], &b_work = ctx->scratch[10], &b_reg = ctx->scratch[15];
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * sizeof(float)));
  BATH_HIP_TRY(ctx, b_fx.reserve((size_t)xoff[(size_t)n] * sizeof(float) + 64));
  BATH_HIP_TRY(ctx, b_bx.reserve((size_t)xoff[(size_t)n] * sizeof(float) + 64));
  BATH_HIP_TRY(ctx, b_work.reserve((size_t)xoff[(size_t)n] / 5 * 3 * sizeof(float) + 64));
  const size_t reg_ints = (size_t)n * (1 + 3 * kMaxRegions);
  BATH_HIP_TRY(ctx, b_reg.reserve(reg_ints * sizeof(int32_t) + 64));
  if ((st = upload_offsets(ctx, b_off, xoff.data(), n)) != BATH_OK) return st;
  // <kept>: window q of <dna> is window kept[q] of the block whose Forward rows fs3_forward_scores left on the device -- the
  // reference runs the Forward parser once per window too (p7_pipeline.c:1450, rows kept in pli->oxf)
  const bool reuse = kept && !ctx->fs_keep_xoff.empty() && fwd_sc_out == nullptr;
  DevBuf &b_foff = ctx->scratch[46];
  const float *d_fx = b_fx.as<float>();
  const int64_t *d_fxoff = b_off.as<int64_t>();
  std::vector<int64_t> fsel;
  if (reuse) {
    fsel.resize((size_t)n);
    for (int64_t q = 0; q < n; q++) fsel[(size_t)q] = ctx->fs_keep_xoff[(size_t)kept[q]];
    BATH_HIP_TRY(ctx, b_foff.reserve((size_t)n * sizeof(int64_t)));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(b_foff.p, fsel.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    d_fx = ctx->scratch[45].as<float>(); d_fxoff = b_foff.as<int64_t>();
  }
  const int Cv = fs_columns(om->M);
  const size_t shmem = (size_t)(kLogsumTbl + (om->M + 2) * 8) * sizeof(float);
  const float tE = (float)-0.69314718055994529;
  [[maybe_unused]] const int grid = fs_grid(ctx, n);
  const int grid_dp = fs_grid_dp(ctx, n);
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 2 * sizeof(float)));
  FsJobs jq[2];
  if ((st = fs_schedule(ctx, dna, 2, jq)) != BATH_OK) return st;
  // Speculative rows (fs3_backward_spec): window q of <dna> is window kept[q] of the decision stage's block; when its Backward rows are
  // already in scratch[53] the parser below skips it (a job list of the OTHER windows, longest first; the launcher sees a shadow of
  // <dna> that holds only those) and a copy kernel puts the rows where the region heuristics read them
  const bool spec = reuse && ctx->fs_spec_valid && ctx->fs_strict && fs_chain_enabled();
  std::vector<int64_t> spec_src;
  bath_hip_seqs rest;
  const bath_hip_seqs *bwd_dna = dna;
  int64_t n_rest = n;
  if (spec) {
    spec_src.assign((size_t)n, -1);
    std::vector<int32_t> ord;
    for (int64_t q = 0; q < n; q++) {
      const int64_t at = ctx->fs_spec_rows[(size_t)kept[q]];
      if (at >= 0) spec_src[(size_t)q] = at; else ord.push_back((int32_t)q);
    }
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return dna->h_len[(size_t)a] > dna->h_len[(size_t)b]; });
    n_rest = (int64_t)ord.size();
    DevBuf &b_sp = ctx->scratch[55];                                          // counter, the rest's job list, the speculative rows' offsets
    const size_t o_ord = 256, o_src = o_ord + (((size_t)std::max<int64_t>(n_rest, 1) * 4 + 255) & ~(size_t)255);
    BATH_HIP_TRY(ctx, b_sp.reserve(o_src + (size_t)n * 8 + 64));
    BATH_HIP_TRY(ctx, hipMemsetAsync(b_sp.p, 0, 256, ctx->stream));
    if (ctx->stage[10].reserve((size_t)n_rest * 4 + (size_t)n * 8 + 64) != hipSuccess) { ctx->set_error("cannot allocate page-locked staging memory"); return BATH_EFAIL; }
    char *hs = static_cast<char *>(ctx->stage[10].p);
    if (n_rest) std::memcpy(hs, ord.data(), (size_t)n_rest * 4);
    std::memcpy(hs + (size_t)n_rest * 4, spec_src.data(), (size_t)n * 8);
    if (n_rest) BATH_HIP_TRY(ctx, hipMemcpyAsync(b_sp.as<char>() + o_ord, hs, (size_t)n_rest * 4, hipMemcpyHostToDevice, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(b_sp.as<char>() + o_src, hs + (size_t)n_rest * 4, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    jq[1] = FsJobs{reinterpret_cast<const int32_t *>(b_sp.as<char>() + o_ord), b_sp.as<unsigned>()};
    rest.ctx = ctx; rest.is_part = true; rest.n = n_rest; rest.d_data = dna->d_data; rest.d_off = dna->d_off; rest.d_len = dna->d_len;
    rest.h_len.resize((size_t)n_rest); rest.maxlen = 0; rest.total = 0;
    for (int64_t i = 0; i < n_rest; i++) { const int32_t L = dna->h_len[(size_t)ord[(size_t)i]]; rest.h_len[(size_t)i] = L; rest.maxlen = std::max(rest.maxlen, L); rest.total += L; }
    bwd_dna = &rest;
  }
  struct Borrowed { bath_hip_seqs &v; ~Borrowed() { v.d_data = nullptr; v.d_off = nullptr; v.d_len = nullptr; } } rest_guard{rest};
  const int mode = ctx->fs_strict ? BATH_LOGSUM_TABLE_SERIAL : BATH_LOGSUM_TABLE;
  static const int spec_share = [] { const char *e = std::getenv("BATH_HIP_FS_SPEC_SHARE"); return e ? std::max(1, std::atoi(e)) : 2; }();   // (probe) CUs each parser's launch is sized for: all / this
  StageGate gate((mode == BATH_LOGSUM_TABLE_SERIAL && fs_chain_enabled()) ? ctx->device : -1, reuse ? StageGate::kBwdChain : StageGate::kFwdChain);   // held until the synchronize below
  if ((st = fs_fork(ctx)) != BATH_OK) return st;
  const double cells3 = (double)(xoff[(size_t)n] / 5) * om->M;                // rows x nodes; algorithmic HBM bytes: 1 B/nt in + 20 B/row out
  const double bytes3 = (double)(xoff[(size_t)n] / 5) * 21.0;
  BATH_FS_SWITCH(Cv, BATH_FS_MODE(mode, {
    if ((st = fs_set_shmem(ctx, fs3_fwd_kernel<CC, MD>, shmem)) != BATH_OK) return st;
    const int s1 = reuse ? -1 : ctx->span_begin("fs3_fwd_kernel", ctx->stream, cells3, bytes3);
    if (reuse) {
    } else if (MD == 2 && fs_chain_enabled()) {
      if ((st = launch_fs3_fwd_chain(ctx, ctx->stream, om, dna, Cv, tE, tE, b_sc.as<float>(), b_fx.as<float>(), b_off.as<int64_t>(), jq[0], spec_share)) != BATH_OK) return st;
    } else
    hipLaunchKernelGGL((fs3_fwd_kernel<CC, MD>), dim3(grid_dp), dim3(kFsBlock), shmem, ctx->stream, dna->view(), fsdev(om), om->d_loop[0], om->d_move[0], tE, tE, b_sc.as<float>(), b_fx.as<float>(), b_off.as<int64_t>(), jq[0]);
    ctx->span_end(s1, ctx->stream);
    if ((st = fs_set_shmem(ctx, fs_bwd_kernel<CC, 3, MD>, shmem)) != BATH_OK) return st;
    const int s2 = ctx->span_begin("fs_bwd_kernel<3>", ctx->side_stream, cells3, bytes3);
    if (MD == 2 && fs_chain_enabled()) {
      if (n_rest > 0 && (st = launch_fs3_bwd_chain(ctx, ctx->side_stream, om, bwd_dna, Cv, tE, tE, b_sc.as<float>() + n, b_bx.as<float>(), b_off.as<int64_t>(), jq[1], reuse ? 1 : spec_share)) != BATH_OK) return st;
    } else
    hipLaunchKernelGGL((fs_bwd_kernel<CC, 3, MD>), dim3(grid_dp), dim3(kFsBlock), shmem, ctx->side_stream, dna->view(), fsdev(om), om->d_loop[0], om->d_move[0], tE, tE, b_sc.as<float>() + n, (float *)nullptr, (const int64_t *)nullptr, b_bx.as<float>(), b_off.as<int64_t>(), jq[1]);
    ctx->span_end(s2, ctx->side_stream);
  }))
  if ((st = fs_join(ctx)) != BATH_OK) return st;
  if (spec) {                                                                  // the speculative rows of this call's windows into place
    BATH_HIP_TRY(ctx, hipEventRecord(ctx->ev_spec, ctx->spec_stream));
    BATH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_spec, 0));
    const size_t o_src = 256 + (((size_t)std::max<int64_t>(n_rest, 1) * 4 + 255) & ~(size_t)255);
    hipLaunchKernelGGL(fs_rows_copy_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, (int)n, dna->d_len, reinterpret_cast<const int64_t *>(ctx->scratch[55].as<char>() + o_src),
                       ctx->scratch[53].as<float>(), b_off.as<int64_t>(), b_bx.as<float>());
    BATH_HIP_TRY(ctx, hipGetLastError());
  }
  const int s3 = ctx->span_begin("fs_regions_kernel", ctx->stream, (double)(xoff[(size_t)n] / 5), (double)(xoff[(size_t)n] / 5) * 52.0);
  hipLaunchKernelGGL(fs_regions_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, dna->d_len, d_fx, b_bx.as<float>(), b_off.as<int64_t>(), d_fxoff, om->d_logsum, loop,
                     b_work.as<float>(), b_reg.as<int32_t>());
  ctx->span_end(s3, ctx->stream);
  BATH_HIP_TRY(ctx, hipGetLastError());
  BATH_HIP_TRY(ctx, hipMemcpyAsync(regions_out, b_reg.p, reg_ints * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (fwd_sc_out) BATH_HIP_TRY(ctx, hipMemcpyAsync(fwd_sc_out, b_sc.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return BATH_OK;
}
}  // namespace bath

namespace bath {
// used by the pipeline's frameshift stage (bath_pipeline.hip)
int fs3_forward_scores(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, float *sc, const std::function<int()> *after_launch) {
  ctx->fs_keep_xoff.clear();
  const int mode = ctx->fs_strict ? BATH_LOGSUM_TABLE_SERIAL : BATH_LOGSUM_TABLE;
  if (!ctx->fs_want_regions) return fs3_parser(ctx, om3, dna, mode, sc, nullptr, nullptr, false, nullptr, after_launch);
  // the domain stage follows: the special-state rows of every window stay on the device (20 B per nucleotide), so that the
  // windows that take the frameshift branch need the Backward parser only
  std::vector<int64_t> xoff((size_t)dna->n + 1, 0);
  for (int64_t i = 0; i < dna->n; i++) xoff[(size_t)i + 1] = xoff[(size_t)i] + ((int64_t)dna->h_len[(size_t)i] + 1) * 5;
  const int st = fs3_parser(ctx, om3, dna, mode, sc, nullptr, xoff.data(), false, &ctx->scratch[45], after_launch);
  if (st == BATH_OK) ctx->fs_keep_xoff = std::move(xoff);
  return st;
}
const float *fsprofile_evparam(const bath_hip_fsprofile *om) { return om->evparam; }
int fsprofile_codon_lengths(const bath_hip_fsprofile *om) { return om->codon_lengths; }
}  // namespace bath

extern "C" int bath_hip_fs3_forward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, int logsum_mode,
                                           float *sc, float *xmx, const int64_t *xmx_offsets) {
  return fs3_parser(ctx, om3, dna, logsum_mode, sc, xmx, xmx_offsets, false);
}
extern "C" int bath_hip_fs3_backward_parser(bath_hip_ctx *ctx, const bath_hip_fsprofile *om3, const bath_hip_seqs *dna, int logsum_mode,
                                            float *sc, float *xmx, const int64_t *xmx_offsets) {
  return fs3_parser(ctx, om3, dna, logsum_mode, sc, xmx, xmx_offsets, true);
}

namespace bath {
const FsHostTables fsprofile_host(const bath_hip_fsprofile *om) {
  return FsHostTables{om->M, om->max_length, om->maxcodons, om->h_tsc.data(), om->h_codons.empty() ? nullptr : om->h_codons.data(), om->evparam};
}
}  // namespace bath

extern "C" int bath_hip_fs5_envelopes(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                                      bath_fs5_result *res, float *pp, const int64_t *pp_off_h, float *oa, const int64_t *oa_off_h) {
  (void)pp_off_h; (void)oa_off_h;
  return bath::fs5_envelopes_ex(ctx, om, dna, logsum_mode, c5_compat, res, pp, oa, nullptr, nullptr, nullptr);
}

extern "C" int bath_hip_fs5_envelopes_x(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                                        bath_fs5_result *res, float *pp, float *oa, float *ppx, float *oax) {
  return bath::fs5_envelopes_ex(ctx, om, dna, logsum_mode, c5_compat, res, pp, oa, ppx, oax, nullptr);
}

extern "C" int bath_hip_fs5_forward_full(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int cfg_len_amino,
                                         float *sc, float *fwd, float *xmx) {
  if (!ctx || !om || !dna || !sc || om->codon_lengths != 5) { if (ctx) ctx->set_error("needs a 5-codon profile"); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (dna->n == 0) return BATH_OK;
  const float *h_f = nullptr, *h_x = nullptr;
  std::vector<int64_t> foff, xoff;
  std::vector<float> h_sc;
  const int st = bath::fs5_region_forward(ctx, om, dna, cfg_len_amino, &h_f, &foff, &h_x, &xoff, &h_sc, nullptr, nullptr);
  if (st != BATH_OK) return st;
  std::memcpy(sc, h_sc.data(), (size_t)dna->n * sizeof(float));
  if (fwd) std::memcpy(fwd, h_f, (size_t)foff[(size_t)dna->n] * sizeof(float));
  if (xmx) std::memcpy(xmx, h_x, (size_t)xoff[(size_t)dna->n] * sizeof(float));
  return BATH_OK;
}

// Envelope rescoring; layouts per envelope i, rows = L_i+1: pp rows*(M+1)*8, oa rows*(M+1)*3, ppx / oax rows*5
// {E,N,J,B,C} (posterior and OA special-state rows), each packed back to back in envelope order.
int bath::fs5_envelopes_ex(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int logsum_mode, int c5_compat,
                           bath_fs5_result *res, float *pp, float *oa, float *ppx, float *oax, FsTraceOut *trace,
                           const uint8_t *cons, std::vector<uint16_t> *steps, std::vector<int64_t> *step_off, std::vector<float> *step_pp) {
  if (!ctx || !om || !dna || om->codon_lengths != 5) { if (ctx) ctx->set_error("fs5 envelopes need a 5-codon profile"); return BATH_EINVAL; }
  BATH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int64_t n = dna->n;
  if (n == 0) return BATH_OK;
  const int M = om->M;
  int st = om->ensure_len(dna->maxlen / 3 + 1);
  if (st != BATH_OK) return st;
  // matrix layouts: fwd (L+1)*(M+1)*8, bck/oa (L+1)*(M+1)*3, xmx (L+1)*5
  std::vector<int64_t> foff(n + 1, 0), boff(n + 1, 0), xoff(n + 1, 0);
  for (int64_t i = 0; i < n; i++) {
    const int64_t rows = (int64_t)dna->h_len[i] + 1;
    foff[i + 1] = foff[i] + rows * (M + 1) * 8; boff[i + 1] = boff[i] + rows * (M + 1) * 3; xoff[i + 1] = xoff[i] + rows * 5;
  }
  DevBuf &b_f = ctx->scratch[15], &b_b = ctx->scratch[16], &b_o = ctx->scratch[17], &b_fx = ctx->scratch[18], &b_bx = ctx->scratch[19];
  DevBuf &b_off = ctx->scratch[20], &b_sc = ctx->scratch[21], &b_cs = ctx->scratch[22], &b_n2 = ctx->scratch[23], &b_ox = ctx->scratch[11];
  if (oax || trace) BATH_HIP_TRY(ctx, b_ox.reserve((size_t)xoff[n] * 4 + 64));
  BATH_HIP_TRY(ctx, b_f.reserve((size_t)foff[n] * 4 + 64)); BATH_HIP_TRY(ctx, b_b.reserve((size_t)boff[n] * 4 + 64)); BATH_HIP_TRY(ctx, b_o.reserve((size_t)boff[n] * 4 + 64));
  BATH_HIP_TRY(ctx, b_fx.reserve((size_t)xoff[n] * 4 + 64)); BATH_HIP_TRY(ctx, b_bx.reserve((size_t)xoff[n] * 4 + 64));
  BATH_HIP_TRY(ctx, b_off.reserve((size_t)(n + 1) * 3 * sizeof(int64_t)));
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 3 * sizeof(float)));
  const size_t cs_stride = (size_t)(M + 1) * 8 + 8;
  BATH_HIP_TRY(ctx, b_cs.reserve((size_t)n * cs_stride * sizeof(float)));
  BATH_HIP_TRY(ctx, b_n2.reserve((size_t)n * kKp * sizeof(float)));
  int64_t *d_foff = b_off.as<int64_t>(), *d_boff = d_foff + (n + 1), *d_xoff = d_boff + (n + 1);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_foff, foff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_boff, boff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_xoff, xoff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  float *d_fsc = b_sc.as<float>(), *d_bsc = d_fsc + n, *d_osc = d_bsc + n;
  const int Cv = fs_columns(M);
  const size_t oa_shmem = (size_t)(M + 2) * 8 * sizeof(float);
  [[maybe_unused]] const int grid = fs_grid(ctx, n);
  if (logsum_mode == BATH_LOGSUM_CONTEXT) logsum_mode = ctx->fs_strict ? BATH_LOGSUM_TABLE_SERIAL : BATH_LOGSUM_TABLE;
  const double cells5 = (double)(foff[(size_t)n] / 8);                          // (L+1) x (M+1) cells of all envelopes
  // The posterior matrix (32 B per cell) is written only when the caller wants it back (<pp>) or the unfused A/B kernels run: the
  // pipeline reads posteriors along the optimal-accuracy trace only (O(L + M) cells of (L+1)(M+1)), and fs5_trace_kernel forms those
  // from the Forward and Backward matrices -- which then stay as they are -- and the rows' normalising factors (4 B per ROW) with the
  // decoding kernels' own arithmetic, value for value
  static const bool unfused_env = [] { const char *e = std::getenv("BATH_HIP_FS_UNFUSED"); return e && e[0] == '1'; }();
  static const bool always_pp = [] { const char *e = std::getenv("BATH_HIP_FS_STORE_PP"); return e && e[0] == '1'; }();          // A/B: the round-4 behaviour
  const bool store_pp = pp != nullptr || unfused_env || always_pp;
  DevBuf &b_rowden = ctx->scratch[37];
  BATH_HIP_TRY(ctx, b_rowden.reserve((size_t)(xoff[(size_t)n] / 5 + 8) * sizeof(float)));
  float *d_rowden = store_pp ? nullptr : b_rowden.as<float>();
  FsJobs jq[4];
  if ((st = fs_schedule(ctx, dna, 4, jq)) != BATH_OK) return st;
  StageGate gate(ctx->device, StageGate::kEnvelopes);                           // (BATH_HIP_FS_GATE=3: not while another worker's Forward parser has the chip)
  if ((st = fs_fork(ctx)) != BATH_OK) return st;                                // Backward on the side stream, concurrently with Forward
  BATH_FS_SWITCH(Cv, {
    BATH_FS_MODE(logsum_mode, {
      // Forward and Backward: the row-per-lane wavefronts (bath_fs_wavefront.hip), the reference's order of every sum, in EVERY mode --
      // the unihit recursion has no sum that a scan could shorten, so the fast mode's envelopes are the strict ones (the node-per-lane
      // scan kernels this replaced in fast mode were 13 times slower at 1024 nodes and are gone)
      const int s1 = ctx->span_begin("fs5_fwd_kernel", ctx->stream, cells5, cells5 * 32.0);
      if ((st = launch_fs5_fwd_wf(ctx, ctx->stream, om, dna, MD == 1, c5_compat, d_fsc, b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff, ctx->scratch[41], jq[0])) != BATH_OK) return st;
      ctx->span_end(s1, ctx->stream);
      static const bool serial_env = [] { const char *e = std::getenv("BATH_HIP_FS_SERIAL"); return e && e[0] == '1'; }();   // timing probes: Backward after Forward
      const bool serial = ctx->fs_serial >= 0 ? ctx->fs_serial != 0 : serial_env;                                           // (bath_hip_set_fs_serial)
      hipStream_t bs = serial ? ctx->stream : ctx->side_stream;
      const int s2 = ctx->span_begin("fs_bwd_kernel<5>", bs, cells5, cells5 * 12.0);
      if ((st = launch_fs5_bwd_wf(ctx, bs, om, dna, MD == 1, d_bsc, b_b.as<float>(), d_boff, b_bx.as<float>(), d_xoff, ctx->scratch[42], ctx->scratch[43], ctx->scratch[44], jq[1], jq[3])) != BATH_OK) return st;
      ctx->span_end(s2, bs);
    })
    if ((st = fs_join(ctx)) != BATH_OK) return st;
    BATH_HIP_TRY(ctx, hipMemsetAsync(b_cs.p, 0, (size_t)n * cs_stride * sizeof(float), ctx->stream));
    // decoding + optimal-accuracy fill, one walk over the rows (BATH_HIP_FS_UNFUSED=1: the two separate kernels, for A/B runs)
    static const bool unfused = [] { const char *e = std::getenv("BATH_HIP_FS_UNFUSED"); return e && e[0] == '1'; }();
    int mw_nodes = 0;
    // reads Forward 32 + Backward 12, writes OA 12 B/cell -- and the posteriors, 32 more, only when the caller asked for the matrix
    const double oa_bytes = cells5 * (store_pp ? 88.0 : 56.0);
    if (!unfused && fs5_decode_oa_mw_shape(M, &mw_nodes) > 0) {        // long models: a block of waves per envelope (bath_fs_decode.hip)
      const int s3 = ctx->span_begin("fs5_decode_oa_kernel", ctx->stream, cells5, oa_bytes);
      if ((st = launch_fs5_decode_oa_mw(ctx, ctx->stream, om, dna, d_bsc, b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff, b_b.as<float>(), d_boff, b_bx.as<float>(),
                                        b_cs.as<float>(), b_o.as<float>(), d_osc, (oax || trace) ? b_ox.as<float>() : nullptr, jq[2], store_pp ? 1 : 0, d_rowden)) != BATH_OK) return st;
      ctx->span_end(s3, ctx->stream);
    } else if (!unfused) {
      if ((st = fs_set_shmem(ctx, fs5_decode_oa_kernel<CC>, oa_shmem)) != BATH_OK) return st;
      const int s3 = ctx->span_begin("fs5_decode_oa_kernel", ctx->stream, cells5, oa_bytes);
      const int oa_grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + 3) / 4, (int64_t)ctx->prop.multiProcessorCount * (CC <= 3 ? BATH_FS_OA_WAVES : 1)));
      hipLaunchKernelGGL((fs5_decode_oa_kernel<CC>), dim3(oa_grid), dim3(256), oa_shmem, ctx->stream, dna->view(), M, om->d_tf, om->d_loop[1], d_bsc, b_f.as<float>(), d_foff,
                         b_fx.as<float>(), d_xoff, b_b.as<float>(), d_boff, b_bx.as<float>(), b_cs.as<float>(), b_o.as<float>(), d_osc,
                         1.17549435e-38f /* E->J impossible in unihit mode: TSCDELTA = FLT_MIN */, 1.0f, (oax || trace) ? b_ox.as<float>() : nullptr, jq[2],
                         store_pp ? 1 : 0, d_rowden);
      ctx->span_end(s3, ctx->stream);
    } else {
    const int s3 = ctx->span_begin("fs5_decode_kernel", ctx->stream, cells5, cells5 * 76.0);     // reads Forward 32 + Backward 12, rewrites 32 B/cell
    hipLaunchKernelGGL(fs5_decode_kernel, dim3(grid), dim3(256), 0, ctx->stream, dna->view(), M, om->d_loop[1], d_bsc, b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff,
                       b_b.as<float>(), d_boff, b_bx.as<float>(), b_cs.as<float>());
    ctx->span_end(s3, ctx->stream);
    if ((st = fs_set_shmem(ctx, fs5_oa_kernel<CC>, oa_shmem)) != BATH_OK) return st;
    const int s4 = ctx->span_begin("fs5_oa_kernel", ctx->stream, cells5, cells5 * 44.0);         // reads posteriors 32, writes OA 12 B/cell
    hipLaunchKernelGGL((fs5_oa_kernel<CC>), dim3(grid), dim3(256), oa_shmem, ctx->stream, dna->view(), M, om->d_tf, b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff, b_o.as<float>(), d_boff, d_osc,
                       1.17549435e-38f /* E->J impossible in unihit mode: TSCDELTA = FLT_MIN */, 1.0f, (oax || trace) ? b_ox.as<float>() : nullptr);
    ctx->span_end(s4, ctx->stream);
    }
  })
  const int s5 = ctx->span_begin("fs5_null2_kernel", ctx->stream, (double)n * M, (double)n * M * 32.0);
  hipLaunchKernelGGL(fs5_null2_kernel, dim3((unsigned)n), dim3(32), 0, ctx->stream, n, dna->d_len, M, om->pitch, om->d_rsc + (size_t)om->maxcodons * om->pitch, om->d_logsum,
                     b_cs.as<float>(), b_n2.as<float>());
  ctx->span_end(s5, ctx->stream);
  BATH_HIP_TRY(ctx, hipGetLastError());
  if (trace) {                                                  // optimal-accuracy traceback + null2 along the trace, on the device
    if (!om->d_codons) { ctx->set_error("traceback needs the profile's codon table"); return BATH_EINVAL; }
    std::vector<int64_t> toff((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; i++) toff[(size_t)i + 1] = toff[(size_t)i] + dna->h_len[(size_t)i] + M + 16;
    DevBuf &b_tb = ctx->scratch[10], &b_to = ctx->scratch[13], &b_steps = ctx->scratch[14];
    BATH_HIP_TRY(ctx, b_tb.reserve((size_t)toff[(size_t)n] * sizeof(uint2) + (size_t)(n + 1) * sizeof(int64_t) + 256));
    // dense columns: [cursor (256 B)] [pp: 4 B per column] [codes: 2 B per column], at most toff[n] columns
    const size_t col_cap = (size_t)toff[(size_t)n];
    if (col_cap >= (size_t)INT32_MAX) { ctx->set_error("envelope batch too large for the trace columns' 32-bit offsets: lower BATH_HIP_ENV_MB"); return BATH_ERANGE; }
    if (steps) BATH_HIP_TRY(ctx, b_steps.reserve(256 + col_cap * (sizeof(float) + sizeof(uint16_t)) + 64));
    int *d_cursor = steps ? b_steps.as<int>() : nullptr;
    float *d_step_pp = steps ? reinterpret_cast<float *>(b_steps.as<char>() + 256) : nullptr;
    uint16_t *d_steps = steps ? reinterpret_cast<uint16_t *>(d_step_pp + col_cap) : nullptr;
    if (steps) BATH_HIP_TRY(ctx, hipMemsetAsync(d_cursor, 0, sizeof(int), ctx->stream));
    BATH_HIP_TRY(ctx, b_to.reserve((size_t)n * sizeof(FsTraceOut) + 64));
    int64_t *d_toff = reinterpret_cast<int64_t *>(b_tb.as<char>() + ((size_t)toff[(size_t)n] * sizeof(uint2) + 255) / 256 * 256);
    BATH_HIP_TRY(ctx, hipMemcpyAsync(d_toff, toff.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    const int s6 = ctx->span_begin("fs5_trace_kernel", ctx->stream, (double)toff[(size_t)n], 0.0);
    const bool lane_trace = [] { const char *e = std::getenv("BATH_HIP_FS_TRACE_LANE"); return e && e[0] == '1'; }();      // A/B and tests: the walk by one lane
    if (lane_trace)
    hipLaunchKernelGGL(fs5_trace_kernel, dim3((unsigned)((n * kTraceSpread + 63) / 64)), dim3(64), 0, ctx->stream, dna->view(), M, om->maxcodons, om->d_tf, om->d_codons,
                       b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff, b_o.as<float>(), d_boff, b_ox.as<float>(), b_n2.as<float>(), b_tb.as<uint2>(), d_toff,
                       b_to.as<FsTraceOut>(), om->d_indel, cons, d_steps,
                       om->d_rsc + (size_t)om->maxcodons * om->pitch, om->pitch, step_pp ? d_step_pp : nullptr, d_cursor,
                       store_pp ? nullptr : b_b.as<float>(), d_boff, d_bsc, d_rowden);
    else
    hipLaunchKernelGGL(fs5_trace_wave_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, dna->view(), M, om->maxcodons, om->d_tf, om->d_codons,
                       b_f.as<float>(), d_foff, b_fx.as<float>(), d_xoff, b_o.as<float>(), d_boff, b_ox.as<float>(), b_n2.as<float>(), b_tb.as<uint2>(), d_toff,
                       b_to.as<FsTraceOut>(), om->d_indel, cons, d_steps,
                       om->d_rsc + (size_t)om->maxcodons * om->pitch, om->pitch, step_pp ? d_step_pp : nullptr, d_cursor,
                       store_pp ? nullptr : b_b.as<float>(), d_boff, d_bsc, d_rowden);
    ctx->span_end(s6, ctx->stream);
    BATH_HIP_TRY(ctx, hipGetLastError());
    BATH_HIP_TRY(ctx, hipMemcpyAsync(trace, b_to.p, (size_t)n * sizeof(FsTraceOut), hipMemcpyDeviceToHost, ctx->stream));
    if (steps) {                                                // columns z1..z2 of envelope e: (*steps)[step_off[e] .. +trace[e].ncol)
      int total = 0;
      BATH_HIP_TRY(ctx, hipMemcpyAsync(&total, d_cursor, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // (also: toff is a local)
      steps->resize((size_t)total);
      if (total > 0) BATH_HIP_TRY(ctx, hipMemcpyAsync(steps->data(), d_steps, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
      if (step_pp) {
        step_pp->resize((size_t)total);
        if (total > 0) BATH_HIP_TRY(ctx, hipMemcpyAsync(step_pp->data(), d_step_pp, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      }
      step_off->assign((size_t)n + 1, (int64_t)total);
      for (int64_t e = 0; e < n; e++) (*step_off)[(size_t)e] = trace[e].ok ? (int64_t)trace[e].col_off : (int64_t)total;
    }
    BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // toff is a local
  }
  std::vector<float> h_sc((size_t)n * 3), h_n2((size_t)n * kKp);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h_sc.data(), d_fsc, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(h_n2.data(), b_n2.p, (size_t)n * kKp * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (pp) BATH_HIP_TRY(ctx, hipMemcpyAsync(pp, b_f.p, (size_t)foff[n] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (oa) BATH_HIP_TRY(ctx, hipMemcpyAsync(oa, b_o.p, (size_t)boff[n] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (ppx) BATH_HIP_TRY(ctx, hipMemcpyAsync(ppx, b_fx.p, (size_t)xoff[n] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (oax) BATH_HIP_TRY(ctx, hipMemcpyAsync(oax, b_ox.p, (size_t)xoff[n] * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int64_t i = 0; i < n; i++) {
    res[i].fwdsc = h_sc[i]; res[i].bcksc = h_sc[n + i]; res[i].oasc = h_sc[2 * n + i];
    std::memcpy(res[i].null2, &h_n2[(size_t)i * kKp], sizeof(float) * kKp);
  }
  return BATH_OK;
}

// p7_Forward_Frameshift of regions in the MULTIHIT configuration of a fixed amino length (p7_domaindef.c:411-414: the model's
// saved length), matrices and special-state rows copied to the host for the stochastic-trace ensemble (bath_ensemble.hip).
// fwd: (L+1) x (M+1) x {D, I, M_C0, M_C1..M_C5}; xmx: (L+1) x {E,N,J,B,C}.  sc[e] = -inf: no path (the region is dropped).
int bath::fs5_region_forward(bath_hip_ctx *ctx, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int cfg_len_amino,
                             const float **fwd, std::vector<int64_t> *fwd_off, const float **xmx, std::vector<int64_t> *xmx_off, std::vector<float> *sc,
                             const int **done_flags, const float **sc_live) {
  // done_flags != nullptr: return right after the launch.  *done_flags (page-locked, one int per region) turns 1 when that region's
  // matrix, rows and score (*sc_live) have landed in host memory; the caller synchronises the stream before it reuses the buffers.
  const int64_t n = dna->n;
  const int M = om->M;
  int st = om->ensure_len(std::max(dna->maxlen / 3 + 1, cfg_len_amino));
  if (st != BATH_OK) return st;
  std::vector<int64_t> &foff = *fwd_off, &xoff = *xmx_off;
  foff.assign((size_t)n + 1, 0); xoff.assign((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; i++) {
    const int64_t rows = (int64_t)dna->h_len[(size_t)i] + 1;
    foff[(size_t)i + 1] = foff[(size_t)i] + rows * (M + 1) * 8; xoff[(size_t)i + 1] = xoff[(size_t)i] + rows * 5;
  }
  DevBuf &b_f = ctx->scratch[15], &b_fx = ctx->scratch[18], &b_off = ctx->scratch[20], &b_sc = ctx->scratch[21];
  // The matrices are for the host (the ensembles' tracebacks).  The kernel is bound by its row chain, not by where its stores
  // go, so it writes them straight into page-locked host memory: the transfer rides along with the computation (32 B/cell,
  // ~27 GB/s on the bench block) instead of following it as a copy of its own.  BATH_HIP_FS_REGION_COPY=1: HBM first, then copy.
  const size_t x_bytes = ((size_t)xoff[(size_t)n] * 4 + 255) / 256 * 256;     // then: done flags [n], scores [n]
  BATH_HIP_TRY(ctx, ctx->pinned[2].reserve((size_t)foff[(size_t)n] * 4 + 64, true)); BATH_HIP_TRY(ctx, ctx->pinned[3].reserve(x_bytes + (size_t)n * 8 + 64, true));
  float *d_f = nullptr, *d_fx = nullptr;
  {
    const char *e = std::getenv("BATH_HIP_FS_REGION_COPY");
    void *pf = nullptr, *px = nullptr;
    if (!(e && e[0] == '1') && hipHostGetDevicePointer(&pf, ctx->pinned[2].p, 0) == hipSuccess && hipHostGetDevicePointer(&px, ctx->pinned[3].p, 0) == hipSuccess) {
      d_f = static_cast<float *>(pf); d_fx = static_cast<float *>(px);
    } else (void)hipGetLastError();
  }
  const bool direct = d_f != nullptr;
  int *h_done = reinterpret_cast<int *>(static_cast<char *>(ctx->pinned[3].p) + x_bytes);
  float *h_sc = reinterpret_cast<float *>(h_done + n);
  const bool live = direct && done_flags != nullptr;
  if (done_flags) { *done_flags = h_done; *sc_live = h_sc; }
  for (int64_t i = 0; i < n; i++) h_done[i] = 0;
  BATH_HIP_TRY(ctx, b_sc.reserve((size_t)n * 3 * sizeof(float)));            // BEFORE its pointer is taken: on a fresh context it is null, and a growing buffer moves
  int *d_done = live ? reinterpret_cast<int *>(reinterpret_cast<char *>(d_fx) + x_bytes) : nullptr;
  float *d_sc_out = live ? reinterpret_cast<float *>(d_done + n) : b_sc.as<float>();
  if (!direct) {
    BATH_HIP_TRY(ctx, b_f.reserve((size_t)foff[(size_t)n] * 4 + 64)); BATH_HIP_TRY(ctx, b_fx.reserve((size_t)xoff[(size_t)n] * 4 + 64));
    d_f = b_f.as<float>(); d_fx = b_fx.as<float>();
  }
  BATH_HIP_TRY(ctx, b_off.reserve((size_t)(n + 1) * 3 * sizeof(int64_t)));
  int64_t *d_foff = b_off.as<int64_t>(), *d_xoff = d_foff + (n + 1);
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_foff, foff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  BATH_HIP_TRY(ctx, hipMemcpyAsync(d_xoff, xoff.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  const int Cv = fs_columns(M);
  const size_t shmem = (size_t)(kLogsumTbl + (M + 2) * 8) * sizeof(float);
  [[maybe_unused]] const int grid = fs_grid(ctx, n);
  const int grid_dp = fs_grid_dp(ctx, n);
  const float tE = (float)-0.69314718055994529;                               // multihit: E->C and E->J both log 1/2 (modelconfig.c:825-831)
  const int mode = ctx->fs_strict ? BATH_LOGSUM_TABLE_SERIAL : BATH_LOGSUM_TABLE;
  FsJobs jq[1];
  if ((st = fs_schedule(ctx, dna, 1, jq)) != BATH_OK) return st;
  if (mode == BATH_LOGSUM_TABLE_SERIAL && fs_chain_enabled()) {
    const int s1 = ctx->span_begin("fs5_fwd_kernel(regions)", ctx->stream, (double)(foff[(size_t)n] / 8), (double)(foff[(size_t)n] / 8) * 32.0);
    if ((st = launch_fs5_fwd_chain(ctx, ctx->stream, om, dna, Cv, tE, tE, 0, d_sc_out, d_f, d_foff, d_fx, d_xoff, cfg_len_amino, jq[0], d_done)) != BATH_OK) return st;
    ctx->span_end(s1, ctx->stream);
  } else
  BATH_FS_SWITCH(Cv, BATH_FS_MODE(mode, {
    if ((st = fs_set_shmem(ctx, fs5_fwd_kernel<CC, MD>, shmem)) != BATH_OK) return st;
    const int s1 = ctx->span_begin("fs5_fwd_kernel(regions)", ctx->stream, (double)(foff[(size_t)n] / 8), (double)(foff[(size_t)n] / 8) * 32.0);
    hipLaunchKernelGGL((fs5_fwd_kernel<CC, MD>), dim3(grid_dp), dim3(kFsBlock), shmem, ctx->stream, dna->view(), fsdev(om), om->d_loop[0], om->d_move[0], tE, tE, 0, d_sc_out,
                       d_f, d_foff, d_fx, d_xoff, cfg_len_amino, jq[0], d_done);
    ctx->span_end(s1, ctx->stream);
  }))
  BATH_HIP_TRY(ctx, hipGetLastError());
  sc->resize((size_t)n);
  *fwd = ctx->pinned[2].as<float>(); *xmx = ctx->pinned[3].as<float>();      // page-locked: the matrices are a few MB per region
  if (live) return BATH_OK;                                                  // the flags tell the rest
  if (!direct) {
    BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned[2].p, b_f.p, (size_t)foff[(size_t)n] * 4, hipMemcpyDeviceToHost, ctx->stream));
    BATH_HIP_TRY(ctx, hipMemcpyAsync(ctx->pinned[3].p, b_fx.p, (size_t)xoff[(size_t)n] * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  BATH_HIP_TRY(ctx, hipMemcpyAsync(sc->data(), b_sc.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  BATH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (done_flags) for (int64_t i = 0; i < n; i++) { h_sc[i] = (*sc)[(size_t)i]; h_done[i] = 1; }
  return BATH_OK;
}
