// bath_fs_decode.hip -- posterior decoding + optimal-accuracy fill of an envelope with SEVERAL waves per envelope, for long models.
//
//   fs5_decode_oa_mw_kernel <- p7_Decoding_Frameshift          generic_decoding_frameshift.c:36-156
//                              p7_OptimalAccuracy_Frameshift   generic_optacc_frameshift.c:53-324 (the fill; the traceback is fs5_trace_kernel)
//
// The one-wave kernel (fs5_decode_oa_kernel, bath_frameshift.hip) gives a lane C = ceil(M / 64) consecutive nodes and walks the
// rows of its envelope one after the other: the optimal-accuracy recursion reads E(i) through J and B, so the rows are a chain.
// At M = 1024 that is 16 nodes per lane -- five rows of {M, I, D} history, the posteriors and the next row's Forward / Backward
// cells do not fit a lane's registers (the kernel spilled), a row took ~60 us, and a launch lasted as long as its longest
// envelope: 193 ms per launch on BASELINE configs[4]'s one-GPU slice, 14 times the Forward wavefront beside it.
//
// Here a BLOCK of W = ceil(M / (64 C)) waves owns the envelope, C = 2 (3 beyond 1024 nodes) nodes per lane, and the row's
// cross-lane steps go through the waves with ONE LDS barrier per row:
//   * the row's normalising sum is needed before anything else of the row, so every wave computes exp(F + B - total) of row
//     i + 1 while it works on row i (decoding does not depend on the optimal-accuracy recursion) and publishes its partial sum
//     with row i's other values; the special states' terms are wave 0's (it is the only reader of the rows it overwrites);
//   * the D chain D(k+1) = max(dMD M(k), dDD D(k)) is a scan over maps x -> max(a, b x), closed under composition: DPP scan inside
//     a wave, and every wave publishes its aggregate split as (all nodes but the last, the last node) -- folding the waves in
//     order then gives each wave the D entering its first node, the D of its left neighbour node (which the next rows' M cells
//     read) and, with identity maps from node M on, D(M) for the E state;
//   * E(i) = max over the nodes is a maximum of the waves' maxima.  max and multiplication by {1, FLT_MIN} are exact in any
//     association, so the optimal-accuracy cells are those of a serial fill; the posteriors differ from the reference's by the
//     rounding of the row sum's association (as in the one-wave kernel), <= 2e-5 in the tests.
// Published per row and wave: 8 floats; double-buffered by row parity, so a wave that is a row ahead never overwrites what a
// slower wave still reads.  The barrier waits for LDS only (the rows' global stores and the prefetch of row i + 2 stay in flight).
#include <cmath>
#include <vector>

#include "bath_common.hpp"
#include "bath_kernels.hpp"
#include "bath_launch.hpp"

using namespace bath;

#include "bath_fs_device.hpp"

namespace bath {

constexpr int kOaMaxWaves = 8;

__device__ __forceinline__ void oa_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int C>
__global__ __launch_bounds__(64 * kOaMaxWaves) void fs5_decode_oa_mw_kernel(SeqView dna, int M, const float *__restrict__ tf, const float *__restrict__ loop_tab, const float *__restrict__ bcksc,
                                                                            float *__restrict__ fwd, const int64_t *__restrict__ fwd_off, float *__restrict__ fx, const int64_t *__restrict__ x_off,
                                                                            const float *__restrict__ bck, const int64_t *__restrict__ bck_off, const float *__restrict__ bx,
                                                                            float *__restrict__ colsum /* [n][(M+1)*8 + 8] */, float *__restrict__ oa, float *__restrict__ oasc,
                                                                            float ej, float ec, float *__restrict__ ox, FsJobs jobs,
                                                                            int store_pp /* 0: no posterior matrix, see fs5_decode_oa_kernel */, float *__restrict__ rowden) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_dl = reinterpret_cast<float *>(lds);                 // [(M+2)][8] TSCDELTA, same order as tf
  float *s_pub = s_dl + (size_t)(M + 2) * 8;                    // [2][kOaMaxWaves][8]: {A', B', a_last, b_last, den, wmax, M_last, I_last}
  float *s_spec = s_pub + 2 * kOaMaxWaves * 8;                  // [2][4]: wave 0's unnormalised posteriors of N, J, C of the next row
  int *s_job = reinterpret_cast<int *>(s_spec + 8);
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_dl[i] = (tf[i] == -INFINITY) ? 1.17549435e-38f : 1.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int gl = wv * 64 + lane;                                // lane index over the block: nodes gl * C + 1 .. gl * C + C
  for (;;) {
    if (threadIdx.x == 0) { const unsigned j = atomicAdd(jobs.counter, 1u); s_job[0] = (int64_t)j < dna.n ? jobs.order[j] : -1; }
    __syncthreads();
    const int64_t job = s_job[0];
    __syncthreads();
    if (job < 0) break;
    const int L = dna.len[job];
    if (L < 5) { if (threadIdx.x == 0) oasc[job] = -INFINITY; continue; }
    float *F = fwd + fwd_off[job];
    float *X = fx + x_off[job];
    const float *Bk = bck + bck_off[job];
    const float *Y = bx + x_off[job];
    float *O = oa + bck_off[job];
    float *OX = ox ? ox + x_off[job] : nullptr;
    float *cs = colsum + (size_t)job * ((size_t)(M + 1) * 8 + 8);
    const float overall = bcksc[job];
    const float tL = loop_tab[L / 3];
    // Forward special states of rows r, r-1, r-2, r-3: wave 0 only (it overwrites those rows with the posteriors; no other wave reads them)
    float N0 = 0.f, J0 = 0.f, C0 = 0.f, N1 = 0.f, N2 = 0.f, N3 = 0.f, J1 = 0.f, J2 = 0.f, J3 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f;
    if (wv == 0) { N0 = X[1]; J0 = X[2]; C0 = X[4]; }
    // row 0: posteriors 0, OA cells -inf
    if (store_pp) for (int k = threadIdx.x + 8; k < (M + 1) * 8; k += blockDim.x) F[k] = 0.f;          // (node 0 of row 0 below, after wave 0 has read the row's special states)
    for (int k = threadIdx.x; k <= M; k += blockDim.x) { O[(size_t)k * 3] = O[(size_t)k * 3 + 1] = O[(size_t)k * 3 + 2] = -INFINITY; }
    if (wv == 0) {
      if (store_pp && lane < 8) F[lane] = 0.f;
      if (lane < 5) X[lane] = 0.f;
      if (OX && lane == 0) { OX[0] = -INFINITY; OX[1] = 0.f; OX[2] = -INFINITY; OX[3] = 0.f; OX[4] = -INFINITY; }
    }
    float Mr[5][C], Ir[5][C], Dr[5][C];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
      for (int c = 0; c < C; c++) Mr[r][c] = Ir[r][c] = Dr[r][c] = -INFINITY;
    float mH[5], iH[5], dH[5];                                  // M, I, D of the node left of this lane's first, rows i-1 .. i-5
#pragma unroll
    for (int r = 0; r < 5; r++) mH[r] = iH[r] = dH[r] = -INFINITY;
    float Bh[5] = {0.f, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float Nh[3] = {0.f, 0.f, 0.f}, Jh[3] = {-INFINITY, -INFINITY, -INFINITY}, Ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    float cL = -INFINITY, cL1 = -INFINITY, cL2 = -INFINITY;
    float csum[C][7];
#pragma unroll
    for (int c = 0; c < C; c++)
#pragma unroll
      for (int q = 0; q < 7; q++) csum[c][q] = 0.f;
    float sN = 0.f, sJ = 0.f, sC = 0.f;
    // raw Forward / Backward cells of the row after next, fetched while this row is worked on
    float4 fa_n[C], fb_n[C]; float bm_n[C], bi_n[C];
    float xn1 = 0.f, xn2 = 0.f, xn4 = 0.f, yn1 = 0.f, yn2 = 0.f, yn4 = 0.f;
    auto fetch_row = [&](int r) {
      const float *fq = F + (size_t)r * (M + 1) * 8;
      const float *bq = Bk + (size_t)r * (M + 1) * 3;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = imin(gl * C + c + 1, M);
        fa_n[c] = *reinterpret_cast<const float4 *>(fq + (size_t)node * 8);
        fb_n[c] = *reinterpret_cast<const float4 *>(fq + (size_t)node * 8 + 4);
        bm_n[c] = bq[(size_t)node * 3 + 2]; bi_n[c] = bq[(size_t)node * 3 + 1];
      }
      if (wv == 0) { xn1 = X[r * 5 + 1]; xn2 = X[r * 5 + 2]; xn4 = X[r * 5 + 4]; yn1 = Y[r * 5 + 1]; yn2 = Y[r * 5 + 2]; yn4 = Y[r * 5 + 4]; }
    };
    // exp(F + B - total) of the fetched row (generic_decoding_frameshift.c:62-150), this lane's part of its sum, and -- wave 0 -- the
    // special states' terms; the row is normalised when its turn comes
    float eI[C], eC[C][6];
    float pnU = 0.f, pjU = 0.f, pcU = 0.f;
    auto exp_row = [&](int r) -> float {
      float dloc = 0.f;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = gl * C + c + 1;
        if (node <= M) {
          const float4 a = fa_n[c];
          const float4 b = fb_n[c];
          const float bm = bm_n[c], bi = bi_n[c];
          eC[c][0] = expf(a.z + bm - overall); eC[c][1] = expf(a.w + bm - overall);
          eC[c][2] = expf(b.x + bm - overall); eC[c][3] = expf(b.y + bm - overall); eC[c][4] = expf(b.z + bm - overall); eC[c][5] = expf(b.w + bm - overall);
          dloc += eC[c][0];
          if (node < M) { eI[c] = expf(a.y + bi - overall); dloc += eI[c]; } else eI[c] = 0.f;
        } else {
          eI[c] = 0.f;
#pragma unroll
          for (int q = 0; q < 6; q++) eC[c][q] = 0.f;
        }
      }
      if (wv == 0) {
        N3 = N2; N2 = N1; N1 = N0; J3 = J2; J2 = J1; J1 = J0; C3 = C2; C2 = C1; C1 = C0;
        if (r > 2) { pnU = expf(N3 + yn1 + tL - overall); pcU = expf(C3 + yn4 + tL - overall); pjU = expf(J3 + yn2 + tL - overall); }
        else { pnU = expf(yn1 - overall); pcU = 0.f; pjU = 0.f; }
        N0 = xn1; J0 = xn2; C0 = xn4;
      }
      return wave_sum_f32(dloc);
    };
    // "row 0" of the pipeline: row 1's exponentials and sums, published in slot 0
    fetch_row(1);
    {
      const float den = exp_row(1);
      if (lane == 63) s_pub[(0 * kOaMaxWaves + wv) * 8 + 4] = den;
      if (wv == 0 && lane == 0) { s_spec[0] = pnU; s_spec[1] = pjU; s_spec[2] = pcU; }
    }
    if (L >= 2) fetch_row(2);
    oa_lds_barrier();
    for (int i = 1; i <= L; i++) {
      const int prv = (i - 1) & 1, cur = i & 1;
      float *fr = F + (size_t)i * (M + 1) * 8;
      float *orow = O + (size_t)i * (M + 1) * 3;
      // ---- the row's normalising sum: the waves' parts in wave order, then the special states' terms (as the one-wave kernel)
      float dsum = s_pub[(prv * kOaMaxWaves + 0) * 8 + 4];
      for (int v = 1; v < W; v++) dsum += s_pub[(prv * kOaMaxWaves + v) * 8 + 4];
      float pn = s_spec[prv * 4 + 0], pj = s_spec[prv * 4 + 1], pc = s_spec[prv * 4 + 2];
      float denom = dsum + ((i > 2) ? (pn + pj + pc) : pn);
      denom = (float)(1.0 / (double)denom);
      pn *= denom; pc *= denom; pj *= denom;
      if (wv == 0 && lane == 0) {
        if (store_pp) {
#pragma unroll
          for (int q = 0; q < 8; q++) fr[q] = 0.f;
        }
        if (rowden) rowden[x_off[job] / 5 + i] = denom;
        X[i * 5 + 0] = 0.f; X[i * 5 + 3] = 0.f; X[i * 5 + 1] = pn; X[i * 5 + 4] = pc; X[i * 5 + 2] = pj;
        orow[0] = orow[1] = orow[2] = -INFINITY;
      }
      sN += pn; sJ += pj; sC += pc;
      float pI[C], pC[C][6];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = gl * C + c + 1;
        pI[c] = eI[c] * denom;
#pragma unroll
        for (int q = 0; q < 6; q++) pC[c][q] = eC[c][q] * denom;
        if (node > M) continue;
        if (store_pp) {
          *reinterpret_cast<float4 *>(fr + (size_t)node * 8) = make_float4(0.f, pI[c], pC[c][0], pC[c][1]);
          *reinterpret_cast<float4 *>(fr + (size_t)node * 8 + 4) = make_float4(pC[c][2], pC[c][3], pC[c][4], pC[c][5]);
        }
        csum[c][0] += pI[c];
#pragma unroll
        for (int q = 0; q < 6; q++) csum[c][1 + q] += pC[c][q];
      }
      // ---- next row's exponentials (its cells arrived while the previous row was worked on), then the fetch of the row after it
      float den_next = 0.f;
      if (i < L) {
        den_next = exp_row(i + 1);
        if (i + 2 <= L) fetch_row(i + 2);
      }
      // ---- optimal-accuracy row i (generic_optacc_frameshift.c:53-324) on the posteriors in registers
      float Mc[C], Ic[C], am[C], bmul[C];
      float eloc = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = gl * C + c + 1;
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
              const float m1 = (c == 0) ? mH[r] : Mr[r][c - 1], i1 = (c == 0) ? iH[r] : Ir[r][c - 1], d1 = (c == 0) ? dH[r] : Dr[r][c - 1];
              mx[cl] = fmaxf(dMM * (m1 + pv), fmaxf(dIM * (i1 + pv), fmaxf(dDM * (d1 + pv), dBM * (Bh[r] + pv))));
            }
          }
          if (i == 2) best = fmaxf(mx[1], mx[2]);
          else if (i < 5) best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(mx[3], mx[4]));
          else best = fmaxf(fmaxf(mx[1], mx[2]), fmaxf(fmaxf(mx[3], mx[4]), mx[5]));
        }
        Mc[c] = best;
        Ic[c] = (i >= 3 && node < M) ? fmaxf(dMI * (Mr[2][c] + pI[c]), dII * (Ir[2][c] + pI[c])) : -INFINITY;
        // the map out of node k: D(k+1) = max(dMD M(k), dDD D(k)); identity from node M on (D(M+1) is no cell), so that folding
        // every wave's maps gives D(M)
        am[c] = (node < M) ? dMD * best : -INFINITY;
        bmul[c] = (node < M) ? dDD : 1.0f;
        eloc = fmaxf(eloc, best);
      }
      // this lane's nodes but the last / all of them, composed
      float Ap = -INFINITY, Bp = 1.0f;
#pragma unroll
      for (int c = 0; c < C - 1; c++) { Ap = fmaxf(am[c], bmul[c] * Ap); Bp *= bmul[c]; }
      float A = fmaxf(am[C - 1], bmul[C - 1] * Ap), Bm = Bp * bmul[C - 1];
      // lanes without a source see the identity map (A = -inf, B = 1); fmaxf ignores the NaN of 0 * -inf when B has underflowed
#define BATH_OA_STEP(CTRL, MASK) { const float Aq = dpp_f<CTRL, MASK>(A, -INFINITY), Bq = dpp_f<CTRL, MASK>(Bm, 1.0f); A = fmaxf(A, Bm * Aq); Bm *= Bq; }
      BATH_OA_STEP(0x111, 0xf) BATH_OA_STEP(0x112, 0xf) BATH_OA_STEP(0x114, 0xf) BATH_OA_STEP(0x118, 0xf) BATH_OA_STEP(0x142, 0xa) BATH_OA_STEP(0x143, 0xc)
#undef BATH_OA_STEP
      const float Aex = wave_shr1(A, -INFINITY), Bex = wave_shr1(Bm, 1.0f);          // the lanes before this one
      float wmax = eloc;
      wmax = fmaxf(wmax, dpp_f<0x111>(wmax, -INFINITY)); wmax = fmaxf(wmax, dpp_f<0x112>(wmax, -INFINITY)); wmax = fmaxf(wmax, dpp_f<0x114>(wmax, -INFINITY));
      wmax = fmaxf(wmax, dpp_f<0x118>(wmax, -INFINITY)); wmax = fmaxf(wmax, dpp_f<0x142, 0xa>(wmax, -INFINITY)); wmax = fmaxf(wmax, dpp_f<0x143, 0xc>(wmax, -INFINITY));
      if (lane == 63) {                                          // (the running maximum is complete in the last lane)
        float *pb = s_pub + (cur * kOaMaxWaves + wv) * 8;
        // the wave's nodes but the last one: the lanes before this one, then this lane's first C - 1 nodes
        *reinterpret_cast<float4 *>(pb) = make_float4(fmaxf(Ap, Bp * Aex), Bp * Bex, am[C - 1], bmul[C - 1]);
        *reinterpret_cast<float4 *>(pb + 4) = make_float4(den_next, wmax, Mc[C - 1], Ic[C - 1]);
      }
      if (wv == 0 && lane == 0) { s_spec[cur * 4 + 0] = pnU; s_spec[cur * 4 + 1] = pjU; s_spec[cur * 4 + 2] = pcU; }
      oa_lds_barrier();
      // ---- fold the waves in order: D entering this wave, M / I / D of the node left of it, D(M), E(i)
      float x = -INFINITY, dl = -INFINITY, Din = -INFINITY, Dnb = -INFINITY, Mnb = -INFINITY, Inb = -INFINITY, xE = -INFINITY;
      for (int v = 0; v < W; v++) {
        const float4 q0 = *reinterpret_cast<const float4 *>(s_pub + (cur * kOaMaxWaves + v) * 8);
        const float4 q1 = *reinterpret_cast<const float4 *>(s_pub + (cur * kOaMaxWaves + v) * 8 + 4);
        if (v == wv) { Din = x; Dnb = dl; }
        if (v + 1 == wv) { Mnb = q1.z; Inb = q1.w; }
        dl = fmaxf(q0.x, q0.y * x);                              // D at wave v's last node
        x = fmaxf(q0.z, q0.w * dl);                              // D entering wave v + 1
        xE = fmaxf(xE, q1.y);
      }
      xE = fmaxf(xE, x);                                         // x: D(M) (identity maps from node M on)
      float Dc[C];
      Dc[0] = (lane == 0) ? Din : fmaxf(Aex, Bex * Din);
#pragma unroll
      for (int c = 1; c < C; c++) Dc[c] = fmaxf(am[c - 1], bmul[c - 1] * Dc[c - 1]);
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = gl * C + c + 1;
        if (node > M) { Dc[c] = -INFINITY; continue; }
        orow[(size_t)node * 3 + 0] = Dc[c]; orow[(size_t)node * 3 + 1] = Ic[c]; orow[(size_t)node * 3 + 2] = Mc[c];
      }
      float nN, nJ, nC;
      if (i <= 2) { nJ = ej * xE; nC = ec * xE; nN = pn; }
      else { nJ = fmaxf(Jh[2] + pj, ej * xE); nC = fmaxf(Ch[2] + pc, ec * xE); nN = Nh[2] + pn; }
      const float nB = fmaxf(nN, nJ);
      if (OX && wv == 0 && lane == 0) { float *r = OX + (size_t)i * 5; r[0] = xE; r[1] = nN; r[2] = nJ; r[3] = nB; r[4] = nC; }
      Nh[2] = Nh[1]; Nh[1] = Nh[0]; Nh[0] = nN;
      Jh[2] = Jh[1]; Jh[1] = Jh[0]; Jh[0] = nJ;
      Ch[2] = Ch[1]; Ch[1] = Ch[0]; Ch[0] = nC;
      Bh[4] = Bh[3]; Bh[3] = Bh[2]; Bh[2] = Bh[1]; Bh[1] = Bh[0]; Bh[0] = nB;
      cL2 = cL1; cL1 = cL; cL = nC;
      // the left neighbour node of this lane's first, row i: the previous lane's last node, or the previous wave's
      const float mL = wave_shr1(Mc[C - 1], -INFINITY), iL = wave_shr1(Ic[C - 1], -INFINITY), dLn = wave_shr1(Dc[C - 1], -INFINITY);
#pragma unroll
      for (int r = 4; r > 0; r--) { mH[r] = mH[r - 1]; iH[r] = iH[r - 1]; dH[r] = dH[r - 1]; }
      mH[0] = (lane == 0) ? Mnb : mL; iH[0] = (lane == 0) ? Inb : iL; dH[0] = (lane == 0) ? Dnb : dLn;
#pragma unroll
      for (int c = 0; c < C; c++) {
#pragma unroll
        for (int r = 4; r > 0; r--) { Mr[r][c] = Mr[r - 1][c]; Ir[r][c] = Ir[r - 1][c]; Dr[r][c] = Dr[r - 1][c]; }
        Mr[0][c] = Mc[c]; Ir[0][c] = Ic[c]; Dr[0][c] = Dc[c];
      }
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
      const int node = gl * C + c + 1;
      if (node > M) continue;
      if (node < M) cs[(size_t)node * 8 + 1] = csum[c][0];
#pragma unroll
      for (int q = 0; q < 6; q++) cs[(size_t)node * 8 + 2 + q] = csum[c][1 + q];
    }
    if (threadIdx.x == 0) {
      float *xs = cs + (size_t)(M + 1) * 8;
      xs[1] = sN; xs[2] = sJ; xs[4] = sC;
      oasc[job] = cL + cL1 + cL2;
    }
  }
}

// Waves per envelope and nodes per lane for a model of M nodes; 0 waves: use the one-wave kernel.  BATH_HIP_FS_OA_MW=1 forces the
// multi-wave kernel for every model (tests: the small models against the oracle), =0 switches it off (A/B runs).
int fs5_decode_oa_mw_shape(int M, int *nodes_per_lane) {
  static const int force = [] { const char *e = std::getenv("BATH_HIP_FS_OA_MW"); return e ? std::atoi(e) : -1; }();
  if (force == 0) return 0;
  // up to 2 nodes per lane the one-wave kernel; beyond, a block of waves: at 145 nodes (3 per lane -> two waves of 2) a pass's two
  // launches take 5.9 instead of 7.1 ms and the strict pass 1-1.5 ms less (median of 30 passes, twice: 61.9 / 63.0 against 63.2 / 64.6)
  if (force != 1 && M <= 128) return 0;
  const int C = M <= 64 * 2 * kOaMaxWaves ? 2 : 3;
  const int W = (M + 64 * C - 1) / (64 * C);
  if (W > kOaMaxWaves) return 0;
  *nodes_per_lane = C;
  return W;
}

int launch_fs5_decode_oa_mw(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, const float *d_bsc,
                            float *d_fwd, const int64_t *d_foff, float *d_fx, const int64_t *d_xoff, const float *d_bck, const int64_t *d_boff, const float *d_bx,
                            float *d_colsum, float *d_oa, float *d_osc, float *d_ox, FsJobs jobs, int store_pp, float *d_rowden) {
  int C = 0;
  const int W = fs5_decode_oa_mw_shape(om->M, &C);
  if (W <= 0) { ctx->set_error("multi-wave optimal-accuracy kernel: no shape for this model"); return BATH_EINVAL; }
  const int M = om->M;
  const size_t shmem = ((size_t)(M + 2) * 8 + 2 * kOaMaxWaves * 8 + 8 + 4) * sizeof(float);
  // a block per envelope; as many blocks as the chip holds at this kernel's registers (the work list hands the longest out first)
  const int64_t n = dna->n;
  const int per_cu = std::max(1, 16 / W);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)ctx->prop.multiProcessorCount * per_cu));
#define BATH_OA_MW(CC)                                                                                                              \
  {                                                                                                                                 \
    if (shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs5_decode_oa_mw_kernel<CC>)); \
    hipLaunchKernelGGL((fs5_decode_oa_mw_kernel<CC>), dim3(grid), dim3(64 * W), shmem, stream, dna->view(), M, om->d_tf, om->d_loop[1], d_bsc, d_fwd, d_foff, d_fx, d_xoff,      \
                       d_bck, d_boff, d_bx, d_colsum, d_oa, d_osc, 1.17549435e-38f /* E->J impossible in unihit mode: TSCDELTA = FLT_MIN */, 1.0f, d_ox, jobs, store_pp, d_rowden); \
  }
  if (C == 2) BATH_OA_MW(2) else BATH_OA_MW(3)
#undef BATH_OA_MW
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
