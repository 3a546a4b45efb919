// bath_fs_chain.hip -- the MULTIHIT frameshift recursions with every sum along the model in the reference's serial order
// ("strict": bit-identical to generic_fwdback_frameshift.c), organised so that the serial part costs little.
//
//   fs3_fwd_chain_kernel <- p7_ForwardParser_Frameshift_3Codons   generic_fwdback_frameshift.c:451 (impl_sse/fwdback_fs.c:97)
//   fs3_bwd_chain_kernel <- p7_BackwardParser_Frameshift_3Codons  generic_fwdback_frameshift.c:1422 (impl_sse/fwdback_fs.c:565)
//   fs5_fwd_chain_kernel <- p7_Forward_Frameshift, multihit       generic_fwdback_frameshift.c:64 (the regions' matrices)
//
// In the multihit configuration B(i) reads E(i) through J(i), so a row cannot start before the row that feeds its B is
// complete: the rows of a window are a chain and the row-per-lane wavefront of the envelopes (bath_fs_wavefront.hip) does not
// apply.  What the recursion does allow: the 3-codon parsers' rows i and i+1 are independent of each other (IVX(i) collects the
// paths leaving row i-2: codon lengths 2..4), so rows go through in PAIRS; and within a row only two things are serial along the
// model -- the D chain and the E sum (Backward: the B sum and the D chain).  So a block of W waves owns W windows:
//   1. every wave computes, for its window and both rows of the pair, everything that is parallel over the nodes (IVX, M, I:
//      lanes own contiguous nodes, rows i-1..i-4 in registers) and leaves M(i,k) in LDS;
//   2. ONE wave runs the serial chains of all 2W rows at once, a lane per row, node by node in the reference's order
//      (generic :577-590) -- 2 dependent table log-sums per node for E, the D chain beside it -- and leaves D(i,k) and E(i);
//   3. every wave picks up its D and E, does the special states and moves on.
// The pair's duration is the chain's: 2M dependent log-sums; the 64-step lane hand-off this replaces paid a trip through the
// LDS crossbar per lane and ran every window's chain as its own wave-wide instruction stream.
#include <cstring>

#include "bath_fs_device.hpp"

namespace bath {

constexpr int kChainMaxWaves = 16;
// threads per block by nodes per lane: a lane holds both rows of the pair and rows i-1..i-3 for its C nodes, so the register
// budget per wave grows with C while LDS (two staged rows per window) limits the windows per block anyway
constexpr int chain_threads(int C) { return C <= 3 ? 1024 : (C <= 8 ? 512 : 256); }

// staged rows: [wave][row of the pair][stride]; stride odd, so that the chain lanes (one row each) read different banks
__host__ __device__ inline int fs_chain_stride(int C) { return C * 64 + 1; }

// ---------------------------------------------------------------------------------------------------------------------------
// 3-codon Forward parser, multihit, strict.  tf[node] = {tMM(k-1), tIM(k-1), tDM(k-1), tBM(k-1), tMD(k), tDD(k), tMI(k), tII(k)}
// xmx (optional): (L+1) x {E,N,J,B,C}
// ---------------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(chain_threads(C)) void fs3_fwd_chain_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                             float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  const int M = p.M;
  const int W = blockDim.x >> 6;
  const int stride = fs_chain_stride(C);
  float *s_stage = s_tf + (M + 2) * 8;                          // [W][2][stride]
  float *s_e = s_stage + (size_t)W * 2 * stride;                // [W][2] E(i) of the pair's rows
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * kChainMaxWaves);   // [0]: first job of the block's batch
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  for (;;) {
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, (unsigned)W);
    __syncthreads();
    const int64_t base = s_ctl[0];
    if (base >= dna.n) break;
    const int64_t job = (base + wv < dna.n) ? (int64_t)jobs.order[base + wv] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];                 // the batch's longest window (the list is sorted by length)
    const int L = job >= 0 ? dna.len[job] : 0;
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    float *xo = (xmx && job >= 0) ? xmx + xmx_off[job] : nullptr;
    const int Lc = L / 3;
    const float tNL = loop_tab[Lc], tNM = move_tab[Lc], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    // rows i-1 ("1"), i-2 ("2"), i-3 ("3") of the pair (i, i+1)
    float M1[C], I1[C], D1[C], M2[C], I2[C], D2[C], M3[C], I3[C], iv_1[C], iv_2[C];
#pragma unroll
    for (int c = 0; c < C; c++) M1[c] = I1[c] = D1[c] = M2[c] = I2[c] = D2[c] = M3[c] = I3[c] = iv_1[c] = iv_2[c] = -INFINITY;
    float N1 = 0.f, N2 = 0.f, N3 = 0.f, J1 = -INFINITY, J2 = -INFINITY, J3 = -INFINITY, C1 = -INFINITY, C2 = -INFINITY, C3 = -INFINITY;
    float B1 = tNM, B2 = tNM;                                   // B(i-1), B(i-2)
    if (xo && lane == 0 && L >= 3)
      for (int i = 0; i < 2; i++) { xo[i * 5 + 0] = -INFINITY; xo[i * 5 + 1] = 0.f; xo[i * 5 + 2] = -INFINITY; xo[i * 5 + 3] = tNM; xo[i * 5 + 4] = -INFINITY; }
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 338) : 338; };   // x_i; 338 = p7P_MAXCODONS3 (degenerate / outside)
    for (int i = 2; i <= Lmax; i += 2) {
      const bool actA = job >= 0 && L >= 3 && i <= L, actB = job >= 0 && L >= 3 && i + 1 <= L;
      // ---- emission rows of the pair: codon lengths 2, 3, 4 ending at x_i (row A) and x_{i+1} (row B)
      const int xa = nuc(i), wa = nuc(i - 1), va = nuc(i - 2), ua = nuc(i - 3), xb = nuc(i + 1);
      const float *qa2 = p.rsc + (size_t)imin(xa * 84 + wa * 21, 337) * p.pitch;
      const float *qa3 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + 1, 336) * p.pitch;
      const float *qa4 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + ua + 2, 337) * p.pitch;
      const float *qb2 = p.rsc + (size_t)imin(xb * 84 + xa * 21, 337) * p.pitch;
      const float *qb3 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + 1, 336) * p.pitch;
      const float *qb4 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + va + 2, 337) * p.pitch;
      // ---- 1. everything of both rows that is parallel over the nodes
      const float mInA = wave_shr1(M2[C - 1], -INFINITY), iInA = wave_shr1(I2[C - 1], -INFINITY), dInA = wave_shr1(D2[C - 1], -INFINITY);
      const float mInB = wave_shr1(M1[C - 1], -INFINITY), iInB = wave_shr1(I1[C - 1], -INFINITY), dInB = wave_shr1(D1[C - 1], -INFINITY);
      float MA[C], IA[C], ivA[C], MB[C], IB[C], ivB[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1, nd = imin(node, M + 1), ne = imin(node, M);
        const float4 ta = *reinterpret_cast<const float4 *>(s_tf + nd * 8);
        const float4 tb = *reinterpret_cast<const float4 *>(s_tf + nd * 8 + 4);
        const bool in = node <= M;
        const float ea2 = in ? qa2[ne] : -INFINITY, ea3 = in ? qa3[ne] : -INFINITY, ea4 = in ? qa4[ne] : -INFINITY;
        const float eb2 = in ? qb2[ne] : -INFINITY, eb3 = in ? qb3[ne] : -INFINITY, eb4 = in ? qb4[ne] : -INFINITY;
        // row A = i: from row i-2 and B(i-2) (:562-569); row 2 takes B(0) only (:503)
        const float mA = (c == 0) ? mInA : M2[c - 1], iA = (c == 0) ? iInA : I2[c - 1], dA = (c == 0) ? dInA : D2[c - 1];
        float a = LS(mA + ta.x, LS(iA + ta.y, LS(dA + ta.z, B2 + ta.w)));
        if (i == 2) a = B2 + ta.w;
        ivA[c] = a;
        float mv = a + ea2;
        if (i > 2) { mv = LS(mv, iv_1[c] + ea3); mv = LS(mv, iv_2[c] + ea4); }   // :571-574
        MA[c] = mv;
        const float insA = LS(M3[c] + tb.z, I3[c] + tb.w);
        IA[c] = (i > 2 && node < M) ? insA : -INFINITY;
        // row B = i+1: from row i-1 and B(i-1); its 3- and 4-nucleotide codons start in rows i and i-1
        const float mB = (c == 0) ? mInB : M1[c - 1], iB = (c == 0) ? iInB : I1[c - 1], dB = (c == 0) ? dInB : D1[c - 1];
        const float b = LS(mB + ta.x, LS(iB + ta.y, LS(dB + ta.z, B1 + ta.w)));
        ivB[c] = b;
        float mw = b + eb2;
        mw = LS(mw, a + eb3); mw = LS(mw, iv_1[c] + eb4);
        MB[c] = mw;
        const float insB = LS(M2[c] + tb.z, I2[c] + tb.w);
        IB[c] = (node < M) ? insB : -INFINITY;
        if (in) { s_stage[((size_t)wv * 2 + 0) * stride + node] = mv; s_stage[((size_t)wv * 2 + 1) * stride + node] = mw; }
      }
      __syncthreads();
      // ---- 2. the serial part, a lane per row: D(i,k) = LS(M(i,k-1) + tMD, D(i,k-1) + tDD), E(i) = LS(M(i,k), LS(D(i,k), E)) (:577-590)
      if (wv == 0 && lane < 2 * W) {
        float *st = s_stage + (size_t)lane * stride;
        float dch = -INFINITY, ech = -INFINITY;
        for (int k = 1; k <= M; k++) {
          const float Mk = st[k];
          const float2 t = *reinterpret_cast<const float2 *>(s_tf + k * 8 + 4);   // tMD(k), tDD(k)
          st[k] = dch;
          ech = LS(Mk, LS(dch, ech));
          dch = LS(Mk + t.x, dch + t.y);
        }
        s_e[lane] = ech;
      }
      __syncthreads();
      // ---- 3. D and E back to the window's wave; special states of both rows (:592-603)
      float DA[C], DB[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1, ne = imin(node, M);
        const float da = s_stage[((size_t)wv * 2 + 0) * stride + ne], db = s_stage[((size_t)wv * 2 + 1) * stride + ne];
        DA[c] = (node <= M) ? da : -INFINITY; DB[c] = (node <= M) ? db : -INFINITY;
      }
      const float EA = s_e[wv * 2 + 0], EB = s_e[wv * 2 + 1];
      float NA, JA, CA;
      if (i == 2) { NA = 0.f; JA = EA + tEL; CA = EA + tEM; }
      else { NA = N3 + tNL; JA = LS(J3 + tJL, EA + tEL); CA = LS(C3 + tCL, EA + tEM); }
      const float BA = LS(NA + tNM, JA + tJM);
      const float NB = N2 + tNL, JB = LS(J2 + tJL, EB + tEL), CB = LS(C2 + tCL, EB + tEM);
      const float BB = LS(NB + tNM, JB + tJM);
      if (xo && lane == 0) {
        if (actA) { float *r = xo + (size_t)i * 5; r[0] = EA; r[1] = NA; r[2] = JA; r[3] = BA; r[4] = CA; }
        if (actB) { float *r = xo + (size_t)(i + 1) * 5; r[0] = EB; r[1] = NB; r[2] = JB; r[3] = BB; r[4] = CB; }
      }
      if (actB) {                                               // both rows exist: the rings move by two
        N3 = N1; N2 = NA; N1 = NB; J3 = J1; J2 = JA; J1 = JB; C3 = C1; C2 = CA; C1 = CB; B2 = BA; B1 = BB;
#pragma unroll
        for (int c = 0; c < C; c++) {
          M3[c] = M1[c]; I3[c] = I1[c]; M2[c] = MA[c]; I2[c] = IA[c]; D2[c] = DA[c]; M1[c] = MB[c]; I1[c] = IB[c]; D1[c] = DB[c];
          iv_2[c] = ivA[c]; iv_1[c] = ivB[c];
        }
      } else if (actA) {                                        // the window's last row: only C(L), C(L-1), C(L-2) are still needed
        C3 = C2; C2 = C1; C1 = CA;
      }
    }
    if (job >= 0 && lane == 0) sc[job] = (L >= 3) ? LS(C1, LS(C2 + tCL, C3 + tCL)) + tCM : -INFINITY;
    __syncthreads();                                            // s_ctl is rewritten at the top
  }
#undef LS
}

static int chain_waves(int M, int C, size_t *shmem_out) {
  // as many windows per block as LDS holds next to the table and the transitions (one block per CU)
  const size_t fixed = (size_t)(kLogsumTbl + (M + 2) * 8 + 2 * kChainMaxWaves + 16) * sizeof(float);
  int W = chain_threads(C) / 64;
  while (W > 1 && fixed + (size_t)W * 2 * fs_chain_stride(C) * sizeof(float) > 160 * 1024) W >>= 1;
  *shmem_out = fixed + (size_t)W * 2 * fs_chain_stride(C) * sizeof(float);
  return W;
}

#define BATH_CHAIN_SWITCH(Cv, BODY)                       \
  switch (Cv) {                                           \
    case 1: { constexpr int CC = 1; BODY } break;         \
    case 2: { constexpr int CC = 2; BODY } break;         \
    case 3: { constexpr int CC = 3; BODY } break;         \
    case 4: { constexpr int CC = 4; BODY } break;         \
    case 6: { constexpr int CC = 6; BODY } break;         \
    case 8: { constexpr int CC = 8; BODY } break;         \
    case 12: { constexpr int CC = 12; BODY } break;       \
    case 16: { constexpr int CC = 16; BODY } break;       \
    default: ctx->set_error("frameshift kernels support models up to 1024 nodes"); return BATH_EINVAL; \
  }

int launch_fs3_fwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM,
                         float *d_sc, float *d_xmx, const int64_t *d_xoff, FsJobs jobs) {
  const int M = om->M;
  size_t shmem = 0;
  const int W = chain_waves(M, Cv, &shmem);
  const int64_t n = dna->n;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + W - 1) / W, (int64_t)ctx->prop.multiProcessorCount));
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  BATH_CHAIN_SWITCH(Cv, {
    BATH_HIP_TRY(ctx, hipFuncSetAttribute((const void *)fs3_fwd_chain_kernel<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((fs3_fwd_chain_kernel<CC>), dim3(grid), dim3(64 * W), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, d_sc, d_xmx, d_xoff, jobs);
  })
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
