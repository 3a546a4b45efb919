// bath_fs_wavefront.hip -- the 5-codon Forward / Backward of the ENVELOPES (unihit configuration) as a systolic wavefront:
// one wave per envelope, one LANE PER ROW, every lane sweeping the model's nodes one after the other.
//
//   fs5_fwd_wf_kernel <- p7_Forward_Frameshift   generic_fwdback_frameshift.c:64   (SIMD twin impl_sse/fwdback_fs.c:2054)
//   fs5_bwd_wf_kernel <- p7_Backward_Frameshift  generic_fwdback_frameshift.c:1035 (SIMD twin impl_sse/fwdback_fs.c:2634)
//   fs5_bwd_x_kernel  <- the B(i) sums and the N / J rows of the same function (:1279-1283, :1290-1300)
//
// Why this shape.  p7_FLogsum's table truncates, so log-sum is not associative and a score is the reference's only if every
// sum along the model -- the D chain D(i,k) <- D(i,k-1), the E sum, Backward's D chain and B sum -- runs node by node in the
// reference's order.  With lanes owning NODES (bath_frameshift.hip) that order is a 64-step hand-off per row; with lanes owning
// ROWS it is a lane's own loop and costs nothing.  What makes rows independent enough: in the unihit configuration
// (p7_fs_ReconfigUnihit, modelconfig.c:868: E->J impossible) B(i) = N(i) + tNM does not read E(i), so cell (i,k) needs only
// (i,k-1) and cells of rows i-1..i-5 at nodes k-1 and k: lane l works on row i at node k while lane l+1 works on row i+1 at
// node k-1, and everything a lane needs from the rows above arrives from its neighbour lane by one DPP move per value:
//   IVX(i+1,k)            "paths leaving row i" (generic :332-335), computed by row i's lane from its node k-1 cells;
//   IVX(i..i-3,k)         passed on, a shift register along the lanes (codon lengths 2..5 read IVX(i-1..i-4,k));
//   I(i+3,k) = LS(M(i,k)+tMI, I(i,k)+tII)  computed by row i's lane, passed on twice.
// Lane l owns rows l+1, l+65, l+129, ...; lane 0 picks up what lane 63 left M-63 steps earlier from a ring indexed by node
// (LDS when it fits, global memory otherwise).  Every log-sum has the reference's operands in the reference's order, in every
// mode: the envelope scores, matrices and special-state rows are BIT-IDENTICAL to generic_fwdback_frameshift.c, and the pass is
// bound by instructions issued (about 11 table log-sums per cell) instead of by a 30-log-sum dependent chain per row.
#include <cstring>

#include "bath_fs_device.hpp"

namespace bath {

#ifndef BATH_WF_BLOCK
#define BATH_WF_BLOCK 1024
#endif
constexpr int kWfBlock = BATH_WF_BLOCK;        // 1024: one block per CU, 16 waves share one 64 KB log-sum table
constexpr int kWfWaves = kWfBlock / 64;
// a barrier that waits for LDS operations only: the waves of a multi-wave sweep exchange through LDS; their stores to the
// matrices need not have landed
__device__ __forceinline__ void lds_wf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------------------
// Forward.  fwd[(i*(M+1)+k)*8 + {D,I,C0..C5}], xmx[i*5 + {E,N,J,B,C}] as fs5_fwd_kernel writes them.
// tf[node] = {tMM(k-1), tIM(k-1), tDM(k-1), tBM(k-1), tMD(k), tDD(k), tMI(k), tII(k)}
// ---------------------------------------------------------------------------------------------------------------------------
// W: waves per envelope.  W = 1: every wave of a 1024-thread block draws its own envelopes (the throughput configuration: thousands
// of envelopes).  W > 1: a block of W waves works on ONE envelope with 64 W rows in flight -- for a few long envelopes (the
// clusters' batch of a pass; the 1024-node model of configs[4]) the sweep is L / (64 W) rounds instead of L / 64.  Lane 0 of wave
// w takes over from lane 63 of wave w-1 through a two-slot LDS mailbox, the last wave's lane 63 feeds the ring; one LDS-only
// barrier per step keeps the waves in step.
template <bool EXACT, bool RING_G, int W>
__global__ __launch_bounds__(W == 1 ? kWfBlock : 64 * W) void fs5_fwd_wf_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                              int c5_compat, float *__restrict__ sc, float *__restrict__ fwd, const int64_t *__restrict__ fwd_off,
                                                              float *__restrict__ xmx, const int64_t *__restrict__ xmx_off,
                                                              float *ring_g /* [waves][(M+2)*8] or null: the ring lives in LDS */, FsJobs jobs, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
#ifdef BATH_WF_PROBES
  // (dbg & 8, a timing probe with wrong results: a table of 8000 entries, so that three blocks fit a CU -- what a compressed table would buy)
  const int tbl_n = (dbg & 8) ? 8000 : kLogsumTbl;
  for (int i = threadIdx.x; i < tbl_n; i += blockDim.x) s_tbl[i] = (i < 15700) ? p.logsum[i] : 0.f;
#else
  dbg = 0;                                        // the timing probes (WRONG results) exist only in a -DBATH_WF_PROBES build: their branches fold away here
  constexpr int tbl_n = kLogsumTbl;
  fs_load_logsum_table(s_tbl, p.logsum);
#endif
  float *s_tf = s_tbl + tbl_n;
  const int M = p.M;
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int RW = 64 * W;                      // rows in flight
  const int gl = (W == 1) ? lane : (int)threadIdx.x;   // the lane's place in the pipeline of rows
  const int rstride = (M + 2) * 8;
  // the ring: the last lane leaves what lane 0 will need for the next row, indexed by node.  W = 1: one wave writes and reads it,
  // in program order (LDS operations of a wave execute in order; global memory is coherent within a CU), so plain accesses do
  const int nrings = (W == 1) ? kWfWaves : 1;
  float4 *ring_l = reinterpret_cast<float4 *>(s_tf + (M + 2) * 8 + (size_t)(W == 1 ? wv : 0) * rstride);
  float4 *ring_gl = RING_G ? reinterpret_cast<float4 *>(ring_g + ((size_t)blockIdx.x * nrings + (W == 1 ? wv : 0)) * rstride) : nullptr;
  // W > 1: mailboxes between consecutive waves [W][2 slots][2 float4], the C values of every wave's lanes 61..63, all lanes' last C, the job
  float4 *s_mb = reinterpret_cast<float4 *>(s_tf + (M + 2) * 8 + (RING_G ? 0 : (size_t)nrings * rstride));
  float *s_cpub = reinterpret_cast<float *>(s_mb + (size_t)W * 4);
  float *s_cfin = s_cpub + W * 4;
  int *s_job = reinterpret_cast<int *>(s_cfin + RW);
  const int Mp = M > RW ? M : RW;                 // steps between two rows of a lane
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  auto next_job = [&]() -> int64_t {                // W > 1: one envelope per block
    if (W == 1) return fs_next_job(jobs, dna.n, lane);
    if (threadIdx.x == 0) { const unsigned q = atomicAdd(jobs.counter, 1u); s_job[0] = (int64_t)q < dna.n ? (int)jobs.order[q] : -1; }
    __syncthreads();
    const int64_t job = s_job[0];
    __syncthreads();
    return job;
  };
  for (int64_t job = next_job(); job >= 0; job = next_job()) {
    const int L = dna.len[job];
    const uint8_t *d = dna.data + dna.off[job];
    float *fo = static_cast<float *>(__builtin_assume_aligned(fwd + fwd_off[job], 32));       // rows of (M+1) x 8 floats: every cell is 32-byte aligned
    float *xo = xmx + xmx_off[job];
    if (L < 5) { if (gl == 0) sc[job] = -INFINITY; continue; }
    const float tNL = loop_tab[L / 3], tNM = move_tab[L / 3], tCL = tNL, tCM = tNM;
    // ---- row 0 and the rows of N, J, B: N(i) = N(i-3) + tNL (a chain of float additions per residue class), J = -inf,
    //      B(i) = N(i) + tNM (unihit; generic :265-277 with tEL = -inf)
    for (int k = gl; k <= M; k += RW) {
      float4 *c = reinterpret_cast<float4 *>(fo + (size_t)k * 8);
      c[0] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); c[1] = c[0];
    }
    if (gl == 0) { xo[0] = -INFINITY; xo[4] = -INFINITY; }
    if (gl < 3) {
      float n = 0.f;
      for (int i = gl; i <= L; i += 3) {
        if (i >= 3) n += tNL;
        xo[i * 5 + 1] = n; xo[i * 5 + 2] = -INFINITY; xo[i * 5 + 3] = n + tNM;
      }
    }
    if (W == 1) __threadfence_block(); else __syncthreads();            // the other waves read these rows from global memory
    // ---- per-lane state
    int row = 1 + gl, k = 1 - gl;                                       // k <= 0: the lane has not started yet
    float dch = -INFINITY, ech = -INFINITY, pM = -INFINITY, pI = -INFINITY, pD = -INFINITY;
    float oP = -INFINITY, o0 = -INFINITY, o1 = -INFINITY, o2 = -INFINITY, o3 = -INFINITY, oQ = -INFINITY, oqa = -INFINITY, oqb = -INFINITY;
    float cfin = -INFINITY;                                             // C of the lane's last finished row
    if (W > 1 && lane >= 61) s_cpub[wv * 4 + (lane - 61)] = -INFINITY;
    int r1, r2, r3, r4, r5;                                             // emission rows of the lane's current row (times pitch)
    float Bcur;
    // x_i as the kernels index codons: 0..3, or 1367 = p7P_MAXCODONS5 for a degenerate nucleotide / a position before the start
    auto code = [](int byte, bool inside) -> int { return (inside && byte < 4) ? byte : 1367; };
    auto codon_rows = [&](int x, int w, int v, int u, int t, int &q1, int &q2, int &q3, int &q4, int &q5) {
      q1 = imin(x * 341, 1366) * p.pitch;
      q2 = imin(x * 341 + w * 85 + 1, 1365) * p.pitch;
      q3 = imin(x * 341 + w * 85 + v * 21 + 2, 1364) * p.pitch;
      q4 = imin(x * 341 + w * 85 + v * 21 + u * 5 + 3, 1365) * p.pitch;
      q5 = imin(x * 341 + w * 85 + v * 21 + u * 5 + t + 4, 1366) * p.pitch;
    };
    {
      const int rr = imin(row, L);
      auto at = [&](int i) { return code((int)d[(i >= 1 ? i : 1) - 1], i >= 1); };
      codon_rows(at(rr), at(rr - 1), at(rr - 2), at(rr - 3), at(rr - 4), r1, r2, r3, r4, r5);
      Bcur = xo[(size_t)rr * 5 + 3];
    }
    // what the lane's NEXT row needs is fetched a round ahead, raw: nothing waits for these loads where they are issued
    int nb0; unsigned nbw; float Bnext;                                  // x_i, and x_{i-4}..x_{i-1} as one (unaligned) dword, unpacked where they are used
    auto prefetch_row = [&](int nrow) {                                 // nrow >= 65
      const int rr = imin(nrow, L);
      nb0 = d[rr - 1]; __builtin_memcpy(&nbw, d + rr - 5, 4);
      Bnext = xo[(size_t)rr * 5 + 3];
    };
    prefetch_row(row + RW);
    const int T = ((L - 1) / RW) * Mp + ((L - 1) % RW) + M;
    // The emission scores, FOUR steps at a time: a lane walks along its row, so the scores of its next four nodes are 16
    // contiguous bytes of each of its five codon rows -- one request per codon row and four steps instead of four (the kernel
    // is bound by the number of memory requests, not by bytes).  A block of four steps is loaded while the block before runs.
    // A lane whose four steps straddle the end of its row (or its start) fetches the four scores one by one: a couple of
    // lanes per block.  <kb>: the lane's node at the block's first step, counted on from the lane's state at the time of the
    // call (past Mp: the lane's next row, whose nucleotides are already here).
    float4 cu1, cu2, cu3, cu4, cu5, nx1, nx2, nx3, nx4, nx5;
    auto load_block = [&](int kb, float4 &d1, float4 &d2, float4 &d3, float4 &d4, float4 &d5) {
      int b1, b2, b3, b4, b5;                                          // codon rows of the lane's next row
      codon_rows(code(nb0, true), code((int)(nbw >> 24), true), code((int)((nbw >> 16) & 255u), true), code((int)((nbw >> 8) & 255u), true), code((int)(nbw & 255u), true), b1, b2, b3, b4, b5);
      const bool inA = kb >= 1 && kb + 3 <= imin(Mp, M), inB = kb > Mp && kb - Mp + 3 <= M;
      if (inA || inB) {
        const int n0 = inA ? kb : kb - Mp;
        const float *s1 = p.rsc + (size_t)(inA ? r1 : b1) + n0, *s2 = p.rsc + (size_t)(inA ? r2 : b2) + n0, *s3 = p.rsc + (size_t)(inA ? r3 : b3) + n0;
        const float *s4 = p.rsc + (size_t)(inA ? r4 : b4) + n0, *s5 = p.rsc + (size_t)(inA ? r5 : b5) + n0;
        __builtin_memcpy(&d1, s1, 16); __builtin_memcpy(&d2, s2, 16); __builtin_memcpy(&d3, s3, 16); __builtin_memcpy(&d4, s4, 16); __builtin_memcpy(&d5, s5, 16);
      } else {
        float o1[4], o2[4], o3[4], o4[4], o5[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int kj = kb + j;
          const bool nxt = kj > Mp;
          const int nd = nxt ? kj - Mp : kj, nq = nd < 1 ? 1 : (nd > M ? M : nd);
          o1[j] = p.rsc[(size_t)(nxt ? b1 : r1) + nq]; o2[j] = p.rsc[(size_t)(nxt ? b2 : r2) + nq]; o3[j] = p.rsc[(size_t)(nxt ? b3 : r3) + nq];
          o4[j] = p.rsc[(size_t)(nxt ? b4 : r4) + nq]; o5[j] = p.rsc[(size_t)(nxt ? b5 : r5) + nq];
        }
        d1 = make_float4(o1[0], o1[1], o1[2], o1[3]); d2 = make_float4(o2[0], o2[1], o2[2], o2[3]); d3 = make_float4(o3[0], o3[1], o3[2], o3[3]);
        d4 = make_float4(o4[0], o4[1], o4[2], o4[3]); d5 = make_float4(o5[0], o5[1], o5[2], o5[3]);
      }
    };
    cu1 = cu2 = cu3 = cu4 = cu5 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(dbg & 1)) load_block(k, cu1, cu2, cu3, cu4, cu5);
    for (int t0 = 0; t0 < T; t0 += 4) {
     nx1 = cu1; nx2 = cu2; nx3 = cu3; nx4 = cu4; nx5 = cu5;
     if (!(dbg & 1)) load_block(k + 4, nx1, nx2, nx3, nx4, nx5);
#pragma unroll
     for (int ts = 0; ts < 4; ts++) {
      const int t = t0 + ts;
      const float e1 = (ts == 0) ? cu1.x : (ts == 1) ? cu1.y : (ts == 2) ? cu1.z : cu1.w, e2 = (ts == 0) ? cu2.x : (ts == 1) ? cu2.y : (ts == 2) ? cu2.z : cu2.w;
      const float e3 = (ts == 0) ? cu3.x : (ts == 1) ? cu3.y : (ts == 2) ? cu3.z : cu3.w, e4 = (ts == 0) ? cu4.x : (ts == 1) ? cu4.y : (ts == 2) ? cu4.z : cu4.w;
      const float e5 = (ts == 0) ? cu5.x : (ts == 1) ? cu5.y : (ts == 2) ? cu5.z : cu5.w;
      const bool act = (k >= 1) && (k <= M) && (row <= L);
      const int kk = k < 1 ? 1 : (k > M ? M : k);
      const float4 ta = *reinterpret_cast<const float4 *>(s_tf + kk * 8);
      const float4 tb = *reinterpret_cast<const float4 *>(s_tf + kk * 8 + 4);
      // ---- what the row above hands down (lane 0: from the ring, or the boundary of row 1)
      float v0 = wave_shr1(oP, -INFINITY), v1 = wave_shr1(o0, -INFINITY), v2 = wave_shr1(o1, -INFINITY), v3 = wave_shr1(o2, -INFINITY), v4 = wave_shr1(o3, -INFINITY);
      float Ik = wave_shr1(oqb, -INFINITY), qa = wave_shr1(oQ, -INFINITY), qb = wave_shr1(oqa, -INFINITY);
      if (lane == 0) {
        if (W > 1 && wv > 0) {                                          // from the wave above: what its lane 63 left at the previous step
          const float4 a = s_mb[((size_t)(wv - 1) * 2 + ((t + 1) & 1)) * 2], b = s_mb[((size_t)(wv - 1) * 2 + ((t + 1) & 1)) * 2 + 1];
          v0 = a.x; v1 = a.y; v2 = a.z; v3 = a.w; v4 = b.x; Ik = b.y; qa = b.z; qb = b.w;
        } else if (row == 1) v0 = tNM + ta.w;                            // IVX(1,k) = B(0) + tBM(k-1) (:109)
        else {
          float4 a, b;
          if constexpr (RING_G) { a = ring_gl[(size_t)kk * 2]; b = ring_gl[(size_t)kk * 2 + 1]; } else { a = ring_l[(size_t)kk * 2]; b = ring_l[(size_t)kk * 2 + 1]; }
          v0 = a.x; v1 = a.y; v2 = a.z; v3 = a.w; v4 = b.x; Ik = b.y; qa = b.z; qb = b.w;
        }
      }
      // ---- cell (row, k)
      const float c1 = v0 + e1, c2 = v1 + e2, c3 = v2 + e3, c4 = v3 + e4;
      const float c5 = (c5_compat ? (row >= 5 ? v0 : -INFINITY) : v4) + e5;
      float Mk = LS(LS(c1, LS(c2, c3)), LS(c4, c5));                    // :337-339
      if (row == 4) Mk = LS(c1, LS(c2, LS(c3, c4)));                    // rows 3, 4 associate differently (:222-225); for row 3 (c4 = -inf) the two forms agree
      const float Dk = dch;
      // ---- where the lane will be at the next step, and that step's emission scores (in flight while the chains below run)
      int kn = k + 1, rown = row, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5;
      float Bn = Bcur;
      const bool wrap = kn > Mp;
      if (wrap) {
        kn = 1; rown = row + RW;
        codon_rows(code(nb0, true), code((int)(nbw >> 24), true), code((int)((nbw >> 16) & 255u), true), code((int)((nbw >> 8) & 255u), true), code((int)(nbw & 255u), true), q1, q2, q3, q4, q5);
        Bn = Bnext;
        prefetch_row(rown + RW);
      }
      // the cell goes out after the loads above were issued: what the next step waits for is then a step old
      if (act && !(dbg & 2)) {
        float4 *cell = reinterpret_cast<float4 *>(fo + ((size_t)row * (M + 1) + k) * 8);
        if (k == 1) { cell[-2] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); cell[-1] = cell[-2]; }
        if (dbg & 4) {
          // BATH_HIP_WF_DBG=4, a timing probe (the results are wrong): the four lanes of a quad write 64 contiguous bytes instead of 16
          // bytes in four rows.  Round 4, 4800 envelopes, a wave each (16 waves per CU): 4.7 -> 2.5 ms (1.9 without any store) --
          // the address shape is most of the stores' cost there; 2400 envelopes, two waves each (4 waves per CU): 2.49 -> 2.47, the
          // step's latency is what counts.  Staging the cells through LDS to get that shape (80 B per lane: the ring must then move
          // to global memory) was built and measured: 4.6 -> 4.1 ms a wave each, 2.5 -> 2.85 two waves each; not kept.
          float4 *q = reinterpret_cast<float4 *>(fo + ((size_t)(row - (lane & 3)) * (M + 1) + (k + (lane & 3))) * 8) + (lane & 3);
          q[0] = make_float4(Dk, Ik, Mk, c1); q[4] = make_float4(c2, c3, c4, c5);
        } else {
        cell[0] = make_float4(Dk, Ik, Mk, c1);
        cell[1] = make_float4(c2, c3, c4, c5);
        }
      }
      // E(i) <- LS(M, LS(D, E)) for k < M and for rows 1..4; rows >= 5 pair M and D first at node M (:392-394)
      const bool pairMD = (k == M) && (row >= 5);
      const float ea = pairMD ? Mk : ech, eb = pairMD ? ech : Mk;
      const float enew = LS(eb, LS(Dk, ea));
      const float dnew = LS(Mk + tb.x, Dk + tb.y);                      // D(i,k+1) (:349-350)
      const float Q = LS(Mk + tb.z, Ik + tb.w);                         // I(i+3,k) (:345-346); -inf at node M by tMI(M) = tII(M) = -inf
      // IVX(i+1,k): the paths leaving row i through node k-1 (:332-335); row 2 takes B(1) only (:150)
      float P = LS(pM + ta.x, LS(pI + ta.y, LS(pD + ta.z, Bcur + ta.w)));
      if (row == 1) P = Bcur + ta.w;
      oP = P; o3 = v3; o2 = v2; o1 = v1; o0 = v0; oQ = Q; oqa = qa; oqb = qb;
      if (lane == 63) {
        const float4 a = make_float4(oP, o0, o1, o2), b = make_float4(o3, oqb, oQ, oqa);
        if (W > 1 && wv < W - 1) { s_mb[((size_t)wv * 2 + (t & 1)) * 2] = a; s_mb[((size_t)wv * 2 + (t & 1)) * 2 + 1] = b; }      // to the wave below, read at the next step
        else if (act) {
          if constexpr (RING_G) { ring_gl[(size_t)k * 2] = a; ring_gl[(size_t)k * 2 + 1] = b; } else { ring_l[(size_t)k * 2] = a; ring_l[(size_t)k * 2 + 1] = b; }
        }
      }
      // ---- end of a row: E(i), C(i) = LS(C(i-3) + tCL, E(i) + tEM) with tEM = 0 (:397-398; rows 1, 2: C(<=0) = -inf gives E(i))
      float cprev = __shfl(cfin, (lane + 61) & 63, 64);
      if (W > 1 && lane < 3) cprev = s_cpub[((wv + W - 1) % W) * 4 + lane];          // rows i-3 of the first three lanes live in the wave above
      if (act && k == M) {
        const float cnew = LS(cprev + tCL, enew + 0.0f);
        cfin = cnew;
        xo[(size_t)row * 5 + 0] = enew; xo[(size_t)row * 5 + 4] = cnew;
      }
      if (W > 1 && lane >= 61) s_cpub[wv * 4 + (lane - 61)] = cfin;
      const bool carry = (k >= 1) && !wrap;                             // a lane that has not started, or starts a new row, has its chains at -inf
      pM = carry ? Mk : -INFINITY; pI = carry ? Ik : -INFINITY; pD = carry ? Dk : -INFINITY;
      dch = carry ? dnew : -INFINITY; ech = carry ? enew : -INFINITY;
      k = kn; row = rown; r1 = q1; r2 = q2; r3 = q3; r4 = q4; r5 = q5; Bcur = Bn;
      if (W > 1) lds_wf_barrier();                                      // mailboxes, ring and C values of this step are in place
     }
     cu1 = nx1; cu2 = nx2; cu3 = nx3; cu4 = nx4; cu5 = nx5;
    }
    if (W == 1) {
      const float cL = __shfl(cfin, (L - 1) & 63, 64), cL1 = __shfl(cfin, (L - 2) & 63, 64), cL2 = __shfl(cfin, (L - 3) & 63, 64);
      if (lane == 0) sc[job] = LS(cL, LS(cL1 + tCL, cL2 + tCL)) + tCM;
    } else {
      s_cfin[gl] = cfin;
      __syncthreads();
      if (gl == 0) sc[job] = LS(s_cfin[(L - 1) % RW], LS(s_cfin[(L - 2) % RW] + tCL, s_cfin[(L - 3) % RW] + tCL)) + tCM;
    }
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward.  bck[(i*(M+1)+k)*3 + {D,I,M}]; lane l owns rows L-l, L-l-64, ... and sweeps the nodes M..1 (the D chain of a row
// runs towards node 1, generic :1296-1316).  Cell (i,k) needs ivx(i,k+1) = logsum_c M(i+c,k+1) + e_c(k+1) and D(i,k+1) --
// the lane's own previous step -- and I(i+3,k); ivx(i,k) needs M(i+1..i+5,k): the shift register M(i..i+4,k) and
// I(i..i+2,k) moves down the lanes as in Forward.  E(i) = C(i) + tEM with C(i) = C(i+3) + tCL does not read the row (unihit).
// B(i) = logsum_k ivx(i,k) + tBM(k-1) runs over k ASCENDING in the reference (:1279-1283) while the sweep descends: the terms
// are left in a scratch array indexed by (step, lane) -- coalesced -- and fs5_bwd_x_kernel adds them up in the reference's order,
// then walks the N and J rows and the score.
// tb[node] = {tMD(k), tMI(k), tMM(k), tDD(k), tDM(k), tII(k), tIM(k), tBM(k-1)}
// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int fs_wf_period(int M, int RW) { return M > RW ? M : RW; }      // RW = 64 W rows in flight
__host__ __device__ inline int64_t fs_bwd_wf_steps(int L, int M, int RW) { return (int64_t)(L / RW) * fs_wf_period(M, RW) + (L % RW) + M; }   // rows L..0

template <bool EXACT, bool RING_G, int W>
__global__ __launch_bounds__(W == 1 ? kWfBlock : 64 * W) void fs5_bwd_wf_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                              float *__restrict__ bck, const int64_t *__restrict__ bck_off, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off,
                                                              float *__restrict__ terms, const int64_t *__restrict__ term_off, float *ring_g, FsJobs jobs, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
#ifdef BATH_WF_PROBES
  const int tbl_n = (dbg & 8) ? 8000 : kLogsumTbl;                     // (the timing probe of fs5_fwd_wf_kernel)
  for (int i = threadIdx.x; i < tbl_n; i += blockDim.x) s_tbl[i] = (i < 15700) ? p.logsum[i] : 0.f;
#else
  dbg = 0;
  constexpr int tbl_n = kLogsumTbl;
  fs_load_logsum_table(s_tbl, p.logsum);
#endif
  float *s_tb = s_tbl + tbl_n;
  const int M = p.M;
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tb[i] = p.tb[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int RW = 64 * W;
  const int gl = (W == 1) ? lane : (int)threadIdx.x;
  const int rstride = (M + 2) * 8;
  const int nrings = (W == 1) ? kWfWaves : 1;
  float4 *ring_l = reinterpret_cast<float4 *>(s_tb + (M + 2) * 8 + (size_t)(W == 1 ? wv : 0) * rstride);
  float4 *ring_gl = RING_G ? reinterpret_cast<float4 *>(ring_g + ((size_t)blockIdx.x * nrings + (W == 1 ? wv : 0)) * rstride) : nullptr;
  float4 *s_mb = reinterpret_cast<float4 *>(s_tb + (M + 2) * 8 + (RING_G ? 0 : (size_t)nrings * rstride));   // W > 1: mailboxes between consecutive waves
  int *s_job = reinterpret_cast<int *>(s_mb + (size_t)W * 4);
  const int Mp = fs_wf_period(M, RW);
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  auto next_job = [&]() -> int64_t {                // W > 1: one envelope per block
    if (W == 1) return fs_next_job(jobs, dna.n, lane);
    if (threadIdx.x == 0) { const unsigned q = atomicAdd(jobs.counter, 1u); s_job[0] = (int64_t)q < dna.n ? (int)jobs.order[q] : -1; }
    __syncthreads();
    const int64_t job = s_job[0];
    __syncthreads();
    return job;
  };
  for (int64_t job = next_job(); job >= 0; job = next_job()) {
    const int L = dna.len[job];
    if (L < 5) continue;                                               // fs5_bwd_x_kernel reports -inf
    const uint8_t *d = dna.data + dna.off[job];
    float *bo = bck + bck_off[job];
    float *xo = xmx + xmx_off[job];
    float *tm = terms + term_off[job];
    const float tCL = loop_tab[L / 3], tCM = move_tab[L / 3];
    // ---- the rows of C and E: C(L) = tCM, C(L-1) = C(L-2) = tCL + tCM, C(i) = C(i+3) + tCL; E(i) = C(i) + tEM, tEM = 0 (:1054-1073, :1290-1294)
    if (gl < 3) {
      float c = (gl == 0) ? tCM : tCL + tCM;
      for (int i = L - gl; i >= 1; i -= 3) {
        if (i <= L - 3) c = c + tCL;
        xo[(size_t)i * 5 + 4] = c; xo[(size_t)i * 5 + 0] = c + 0.0f;
      }
    }
    if (W == 1) __threadfence_block(); else __syncthreads();
    int j = gl, k = 1 - gl;                                             // j = L - row; k = position in the row's sweep (node = M + 1 - k); k <= 0: not started
    float dprev = -INFINITY, ivprev = -INFINITY;
    float oM = -INFINITY, om1 = -INFINITY, om2 = -INFINITY, om3 = -INFINITY, om4 = -INFINITY, oI = -INFINITY, oj1 = -INFINITY, oj2 = -INFINITY;
    int r1, r2, r3, r4, r5;
    float xE;
    auto code = [](int byte, bool inside) -> int { return (inside && byte < 4) ? byte : 1367; };
    // row i emits codons that START at nucleotide i+1: x = x_{i+1}, w = x_{i+2}, ...; the codon's last base is the most significant digit (:1260-1270)
    auto codon_rows = [&](int x, int w, int v, int u, int t, int &q1, int &q2, int &q3, int &q4, int &q5) {
      q1 = imin(x * 341, 1366) * p.pitch;
      q2 = imin(w * 341 + x * 85 + 1, 1365) * p.pitch;
      q3 = imin(v * 341 + w * 85 + x * 21 + 2, 1364) * p.pitch;
      q4 = imin(u * 341 + v * 85 + w * 21 + x * 5 + 3, 1365) * p.pitch;
      q5 = imin(t * 341 + u * 85 + v * 21 + w * 5 + x + 4, 1366) * p.pitch;
    };
    {
      const int i = L - imin(j, L);
      auto at = [&](int q) { return code((int)d[imin(q, L) - 1], q <= L); };     // x_q
      codon_rows(at(i + 1), at(i + 2), at(i + 3), at(i + 4), at(i + 5), r1, r2, r3, r4, r5);
      xE = xo[(size_t)(i > 0 ? i : 1) * 5 + 0];
    }
    int nb0; unsigned nbw; float xEn;                                    // next row's x_{i+1}, x_{i+2..i+5} (one unaligned dword), E(i): fetched a round ahead
    auto prefetch_row = [&](int jn) {                                    // jn >= 64 W: all five nucleotides exist
      const int i = L - imin(jn, L);
      nb0 = d[i]; __builtin_memcpy(&nbw, d + i + 1, 4);
      xEn = xo[(size_t)(i > 0 ? i : 1) * 5 + 0];
    };
    if (L >= RW) prefetch_row(j + RW); else { nb0 = 0; nbw = 0; xEn = 0.f; }
    const int T = (int)fs_bwd_wf_steps(L, M, RW);
    // the emission scores four steps at a time, as in Forward: the sweep descends, so the block's first step is the vector's LAST component
    float4 cu1, cu2, cu3, cu4, cu5, nx1, nx2, nx3, nx4, nx5;
    auto load_block = [&](int kb, float4 &d1, float4 &d2, float4 &d3, float4 &d4, float4 &d5) {
      int b1, b2, b3, b4, b5;                                          // codon rows of the lane's next row
      codon_rows(code(nb0, true), code((int)(nbw & 255u), true), code((int)((nbw >> 8) & 255u), true), code((int)((nbw >> 16) & 255u), true), code((int)(nbw >> 24), true), b1, b2, b3, b4, b5);
      const bool inA = kb >= 1 && kb + 3 <= imin(Mp, M), inB = kb > Mp && kb - Mp + 3 <= M;
      if (inA || inB) {
        const int n0 = M - 2 - (inA ? kb : kb - Mp);                   // the node of the block's last step
        const float *s1 = p.rsc + (size_t)(inA ? r1 : b1) + n0, *s2 = p.rsc + (size_t)(inA ? r2 : b2) + n0, *s3 = p.rsc + (size_t)(inA ? r3 : b3) + n0;
        const float *s4 = p.rsc + (size_t)(inA ? r4 : b4) + n0, *s5 = p.rsc + (size_t)(inA ? r5 : b5) + n0;
        __builtin_memcpy(&d1, s1, 16); __builtin_memcpy(&d2, s2, 16); __builtin_memcpy(&d3, s3, 16); __builtin_memcpy(&d4, s4, 16); __builtin_memcpy(&d5, s5, 16);
      } else {
        float o1[4], o2[4], o3[4], o4[4], o5[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          const int kj = kb + jj;
          const bool nxt = kj > Mp;
          const int ps = nxt ? kj - Mp : kj, nq = M + 1 - (ps < 1 ? 1 : (ps > M ? M : ps));
          o1[3 - jj] = p.rsc[(size_t)(nxt ? b1 : r1) + nq]; o2[3 - jj] = p.rsc[(size_t)(nxt ? b2 : r2) + nq]; o3[3 - jj] = p.rsc[(size_t)(nxt ? b3 : r3) + nq];
          o4[3 - jj] = p.rsc[(size_t)(nxt ? b4 : r4) + nq]; o5[3 - jj] = p.rsc[(size_t)(nxt ? b5 : r5) + nq];
        }
        d1 = make_float4(o1[0], o1[1], o1[2], o1[3]); d2 = make_float4(o2[0], o2[1], o2[2], o2[3]); d3 = make_float4(o3[0], o3[1], o3[2], o3[3]);
        d4 = make_float4(o4[0], o4[1], o4[2], o4[3]); d5 = make_float4(o5[0], o5[1], o5[2], o5[3]);
      }
    };
    load_block(k, cu1, cu2, cu3, cu4, cu5);
    for (int t0 = 0; t0 < T; t0 += 4) {
     load_block(k + 4, nx1, nx2, nx3, nx4, nx5);
#pragma unroll
     for (int ts = 0; ts < 4; ts++) {
      const int t = t0 + ts;
      const float e1 = (ts == 0) ? cu1.w : (ts == 1) ? cu1.z : (ts == 2) ? cu1.y : cu1.x, e2 = (ts == 0) ? cu2.w : (ts == 1) ? cu2.z : (ts == 2) ? cu2.y : cu2.x;
      const float e3 = (ts == 0) ? cu3.w : (ts == 1) ? cu3.z : (ts == 2) ? cu3.y : cu3.x, e4 = (ts == 0) ? cu4.w : (ts == 1) ? cu4.z : (ts == 2) ? cu4.y : cu4.x;
      const float e5 = (ts == 0) ? cu5.w : (ts == 1) ? cu5.z : (ts == 2) ? cu5.y : cu5.x;
      const bool act = (k >= 1) && (k <= M) && (j <= L);
      const int kk = k < 1 ? 1 : (k > M ? M : k);
      const int node = M + 1 - kk;
      const float4 t0 = *reinterpret_cast<const float4 *>(s_tb + node * 8);          // tMD tMI tMM tDD
      const float4 t1 = *reinterpret_cast<const float4 *>(s_tb + node * 8 + 4);      // tDM tII tIM tBM(k-1)
      // ---- from the row below (lane 0: the ring; nothing below row L)
      float m1 = wave_shr1(oM, -INFINITY), m2 = wave_shr1(om1, -INFINITY), m3 = wave_shr1(om2, -INFINITY), m4 = wave_shr1(om3, -INFINITY), m5 = wave_shr1(om4, -INFINITY);
      float j1 = wave_shr1(oI, -INFINITY), j2 = wave_shr1(oj1, -INFINITY), I3 = wave_shr1(oj2, -INFINITY);
      if (lane == 0 && W > 1 && wv > 0) {                                // from the wave above: what its lane 63 left at the previous step
        const float4 a = s_mb[((size_t)(wv - 1) * 2 + ((t + 1) & 1)) * 2], b = s_mb[((size_t)(wv - 1) * 2 + ((t + 1) & 1)) * 2 + 1];
        m1 = a.x; m2 = a.y; m3 = a.z; m4 = a.w; m5 = b.x; j1 = b.y; j2 = b.z; I3 = b.w;
      } else if (lane == 0 && j > 0) {
        float4 a, b;
        if constexpr (RING_G) { a = ring_gl[(size_t)node * 2]; b = ring_gl[(size_t)node * 2 + 1]; } else { a = ring_l[(size_t)node * 2]; b = ring_l[(size_t)node * 2 + 1]; }
        m1 = a.x; m2 = a.y; m3 = a.z; m4 = a.w; m5 = b.x; j1 = b.y; j2 = b.z; I3 = b.w;
      }
      // ---- cell (i, node) from ivx(i, node+1), D(i, node+1), I(i+3, node)
      const float dn = dprev, ivn = ivprev;
      const float base = ivn + t1.x;
      float mv = LS(LS(dn + t0.x, LS(I3 + t0.y, ivn + t0.z)), xE);       // :1303-1306
      if (j < 3) mv = LS(dn + t0.x, LS(ivn + t0.z, xE));                 // rows L, L-1, L-2: no row i+3 (:1083-1085, :1135-1137)
      float dv = LS(LS(xE, dn + t0.w), base);                            // :1313-1315
      float iv_ = LS(I3 + t1.y, ivn + t1.z);                             // :1308-1310
      if (j >= L) { mv = -INFINITY; dv = -INFINITY; iv_ = -INFINITY; }    // row 0 holds no cells (:1376-1380)
      // ---- ivx(i, node) = logsum_c M(i+c, node) + e_c(node); the rows L-1..L-4 add their codons left to right (:1101-1120)
      const float s1 = m1 + e1, s2 = m2 + e2, s3 = m3 + e3, s4 = m4 + e4, s5 = m5 + e5;
      float a = LS(s1, LS(s2, LS(s3, LS(s4, s5))));                       // :1272-1276
      if (j < 5) a = LS(LS(LS(s1, s2), s3), s4);
      // ---- where the lane will be at the next step, and that step's emission scores
      int kn = k + 1, jn = j, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5;
      float xEq = xE;
      const bool wrap = kn > Mp;
      if (wrap) {
        kn = 1; jn = j + RW;
        codon_rows(code(nb0, true), code((int)(nbw & 255u), true), code((int)((nbw >> 8) & 255u), true), code((int)((nbw >> 16) & 255u), true), code((int)(nbw >> 24), true),
                   q1, q2, q3, q4, q5);
        xEq = xEn;
        if (jn + RW <= L) prefetch_row(jn + RW);
      }
      if (act) {
        float *cell = bo + ((size_t)(L - j) * (M + 1) + node) * 3;
        cell[0] = dv; cell[1] = iv_; cell[2] = mv;
        if (node == 1) { cell[-3] = -INFINITY; cell[-2] = -INFINITY; cell[-1] = -INFINITY; }
      }
      tm[(size_t)t * RW + gl] = a + t1.w;                                // B(i)'s term of this node, summed by fs5_bwd_x_kernel
      if (lane == 63) {
        const float4 ra = make_float4(mv, m1, m2, m3), rb = make_float4(m4, iv_, j1, j2);
        if (W > 1 && wv < W - 1) { s_mb[((size_t)wv * 2 + (t & 1)) * 2] = ra; s_mb[((size_t)wv * 2 + (t & 1)) * 2 + 1] = rb; }
        else if (act) {
          if constexpr (RING_G) { ring_gl[(size_t)node * 2] = ra; ring_gl[(size_t)node * 2 + 1] = rb; } else { ring_l[(size_t)node * 2] = ra; ring_l[(size_t)node * 2 + 1] = rb; }
        }
      }
      oM = mv; om1 = m1; om2 = m2; om3 = m3; om4 = m4; oI = iv_; oj1 = j1; oj2 = j2;
      const bool carry = (k >= 1) && !wrap;
      dprev = carry ? dv : -INFINITY; ivprev = carry ? a : -INFINITY;
      k = kn; j = jn; r1 = q1; r2 = q2; r3 = q3; r4 = q4; r5 = q5; xE = xEq;
      if (W > 1) lds_wf_barrier();
     }
     cu1 = nx1; cu2 = nx2; cu3 = nx3; cu4 = nx4; cu5 = nx5;
    }
  }
#undef LS
}

// B(i) = logsum over the nodes 1..M, ascending, of the terms fs5_bwd_wf_kernel left (generic :1279-1283); then the rows of N and J
// from row L down (:1054-1073 for the rows without an emitted codon, :1284-1289) and the score logsum(N(0), N(1), N(2)) (:1383-1385).
// One wave per envelope; lane l adds up the rows it owned in the sweep, reading the scratch array slot by slot (coalesced).
template <bool EXACT>
__global__ __launch_bounds__(kFsBlock) void fs5_bwd_x_kernel(SeqView dna, int M, const float *__restrict__ logsum_g, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                             float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, const float *__restrict__ terms, const int64_t *__restrict__ term_off,
                                                             float *__restrict__ sc, FsJobs jobs, int RW /* rows the sweep had in flight: 64 x its waves per envelope */,
                                                             int team /* 1: the waves of a block share one envelope (few long envelopes: a row group of 64 each in turn) */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  __shared__ int s_job;
  fs_load_logsum_table(s_tbl, logsum_g);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = team ? (int)(blockDim.x >> 6) : 1;
  const int Mp = fs_wf_period(M, RW);
#define LS(a, b) flogsum<EXACT>((a), (b), s_tbl)
  // the next envelope: of the wave, or (team) of the block -- B(i) of different rows are independent sums, so the block's waves take the
  // row groups in turn and wave 0 runs the N / J chain once they are all in place (M = 1024, 257 envelopes of ~3 kb: a wave per envelope
  // is 8.8 ms on 33 CUs)
  auto next_job = [&]() -> int64_t {
    if (!team) return fs_next_job(jobs, dna.n, lane);
    __syncthreads();                                                    // the previous envelope's N / J chain has read what the waves stored
    if (threadIdx.x == 0) { const unsigned q = atomicAdd(jobs.counter, 1u); s_job = (int64_t)q < dna.n ? (int)jobs.order[q] : -1; }
    __syncthreads();
    return (int64_t)s_job;
  };
  for (int64_t job = next_job(); job >= 0; job = next_job()) {
    const int L = dna.len[job];
    if (L < 5) { if (threadIdx.x == 0 || (!team && lane == 0)) sc[job] = -INFINITY; continue; }
    float *xo = xmx + xmx_off[job];
    const float *tm = terms + term_off[job];
    const float tNL = loop_tab[L / 3], tNM = move_tab[L / 3], tJL = tNL, tJM = tNM;
    for (int g0 = team ? 64 * wv : 0; g0 <= L; g0 += 64 * nw) {         // 64 rows at a time: the rows a wave of the sweep owned in one round
      const int j = g0 + lane;                                          // this lane's row: i = L - j
      const int pr = g0 / RW, gl = (g0 % RW) + lane;                    // its round and its place in the sweep's pipeline
      float b = -INFINITY;
      const int t_lo = pr * Mp + (g0 % RW), t_hi = t_lo + 63 + M - 1;
      // the lane's node at step t: M - (t - t_lo - lane); node 1 comes first.  Eight terms are fetched before the eight dependent
      // log-sums that consume them: with the load inside the chain every step waited for global memory (the clusters' batch of a
      // pass, ~250 envelopes: 0.80 -> 0.40 ms; the single-domain batch 1.07 -> 0.50 ms)
      for (int t8 = t_hi; t8 >= t_lo; t8 -= 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t8 - u, node = M - (t - t_lo - lane);
          v[u] = (t >= t_lo && node >= 1 && node <= M && j <= L) ? tm[(size_t)t * RW + gl] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int t = t8 - u, node = M - (t - t_lo - lane);
          if (t >= t_lo && node >= 1 && node <= M && j <= L) b = (node == 1) ? v[u] : LS(b, v[u]);
        }
      }
      if (j <= L) xo[(size_t)(L - j) * 5 + 3] = (j == 0) ? -INFINITY : b;   // row L: no codon starts there, B(L) = -inf
    }
    if (team) { __syncthreads(); if (wv != 0) continue; }               // (the barrier is also the fence: the other waves' B(i) are in place)
    else __threadfence_block();
    float nfin = -INFINITY;                                             // N of the chain's last row (i = 0, 1 or 2)
    if (lane < 3) {
      float n = -INFINITY, jj = -INFINITY;
      for (int i = L - lane; i >= 0; i -= 3) {
        const float B = xo[(size_t)i * 5 + 3];
        if (i == L) { n = -INFINITY; jj = -INFINITY; }
        else if (L - i < 3) { jj = B + tJM; n = B + tNM; }              // :1135-1140
        else { jj = LS(jj + tJL, B + tJM); n = LS(n + tNL, B + tNM); }  // :1284-1289
        if (i > 0) { xo[(size_t)i * 5 + 1] = n; xo[(size_t)i * 5 + 2] = jj; }
        else { xo[0] = -INFINITY; xo[1] = n; xo[2] = -INFINITY; xo[4] = -INFINITY; }      // row 0 as fs_bwd_kernel leaves it
      }
      nfin = n;
    }
    const float n0 = __shfl(nfin, L % 3, 64), n1 = __shfl(nfin, (L - 1) % 3, 64), n2 = __shfl(nfin, (L - 2) % 3, 64);
    if (lane == 0) sc[job] = LS(n0, LS(n1, n2));
  }
#undef LS
}

size_t fs_wf_ring_floats(int M) { return (size_t)(M + 2) * 8; }

// waves per envelope: 1 when there are envelopes for every wave slot; with few envelopes (at most two per CU) as many waves as
// the model keeps busy (64 W rows in flight need M >= 64 W steps per row to be fully used), at least 2
static int fs_wf_waves(bath_hip_ctx *ctx, int64_t n, int M) {
  static const int forced = [] { const char *e = std::getenv("BATH_HIP_WF_WAVES"); return e ? std::atoi(e) : 0; }();
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
  // Thousands of envelopes: a wave each.  Up to a dozen per CU (the single-domain regions of a bench pass: 2.5 k): at a wave each
  // the launch would last as long as its longest envelope with most wave slots idle, and its 1024-thread blocks (the table and
  // 16 rings: 144 KB of LDS) would wait for the CUs the regions' Forward holds beside it; two waves per envelope halve the
  // latency, and a 128-thread block (74 KB) shares a CU with one of those.  (4800 envelopes: equal; 9600: a wave each is 1.25x faster.)
  if (n > (int64_t)ctx->prop.multiProcessorCount * 12 || (n > (int64_t)ctx->prop.multiProcessorCount * 2 && M < 100)) return 1;
  if (n > (int64_t)ctx->prop.multiProcessorCount * 2) return 2;
  int W = 2;
  while (W < 8 && 64 * W * 2 <= M) W *= 2;
  return W;
}

struct WfGeom { int W, grid, block; bool lds_ring; size_t shmem; float *ring_g; };
static int fs_wf_geometry(bath_hip_ctx *ctx, int64_t n, int M, DevBuf &ring_scratch, WfGeom *g) {
  static const bool ring_global = [] { const char *e = std::getenv("BATH_HIP_WF_RING_G"); return e && e[0] == '1'; }();
#ifdef BATH_WF_PROBES
  static const int dbg = [] { const char *e = std::getenv("BATH_HIP_WF_DBG"); return e ? std::atoi(e) : 0; }();   // timing probes, wrong results: probe builds only
#else
  constexpr int dbg = 0;
#endif
  const int W = fs_wf_waves(ctx, n, M);
  const size_t base = (size_t)(((dbg & 8) ? 8000 : kLogsumTbl) + (M + 2) * 8) * sizeof(float);
  const int nrings = (W == 1) ? kWfWaves : 1;
  const size_t ring_b = fs_wf_ring_floats(M) * sizeof(float) * nrings;
  const size_t extra = (W == 1) ? 0 : (size_t)W * 64 + (size_t)W * 16 + (size_t)64 * W * 4 + 64;      // mailboxes, C values, job
  g->W = W;
  g->lds_ring = !ring_global && base + ring_b + extra <= 160 * 1024;
  g->shmem = base + (g->lds_ring ? ring_b : 0) + extra;
  g->block = (W == 1) ? kWfBlock : 64 * W;
  const int per_cu = (W == 1) ? ((kWfBlock <= 512) ? 2 : 1) : (int)std::max<size_t>(1, std::min<size_t>((dbg & 8) ? 4 : 2, (160 * 1024) / g->shmem));
  const int64_t jobs_per_block = (W == 1) ? kWfWaves : 1;
  g->grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + jobs_per_block - 1) / jobs_per_block, (int64_t)ctx->prop.multiProcessorCount * per_cu));
  g->ring_g = nullptr;
  if (!g->lds_ring) {
    BATH_HIP_TRY(ctx, ring_scratch.reserve(ring_b * (size_t)g->grid + 64));
    g->ring_g = ring_scratch.as<float>();
  }
  return BATH_OK;
}

#define BATH_WF_DISPATCH(KERNEL, ...)                                                                                            \
  do {                                                                                                                           \
    auto go = [&](auto kfn) -> int {                                                                                             \
      if (g.shmem > 64 * 1024) BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)kfn)); \
      hipLaunchKernelGGL(kfn, dim3(g.grid), dim3(g.block), g.shmem, stream, __VA_ARGS__);                                         \
      return BATH_OK;                                                                                                            \
    };                                                                                                                           \
    int rc_ = BATH_OK;                                                                                                           \
    const int key_ = (exact ? 1 : 0) | (g.lds_ring ? 0 : 2);                                                                     \
    switch (g.W * 4 + key_) {                                                                                                    \
      case 4 + 0: rc_ = go(KERNEL<false, false, 1>); break;  case 4 + 1: rc_ = go(KERNEL<true, false, 1>); break;                  \
      case 4 + 2: rc_ = go(KERNEL<false, true, 1>); break;   case 4 + 3: rc_ = go(KERNEL<true, true, 1>); break;                   \
      case 8 + 0: rc_ = go(KERNEL<false, false, 2>); break;  case 8 + 1: rc_ = go(KERNEL<true, false, 2>); break;                  \
      case 8 + 2: rc_ = go(KERNEL<false, true, 2>); break;   case 8 + 3: rc_ = go(KERNEL<true, true, 2>); break;                   \
      case 16 + 0: rc_ = go(KERNEL<false, false, 4>); break; case 16 + 1: rc_ = go(KERNEL<true, false, 4>); break;                 \
      case 16 + 2: rc_ = go(KERNEL<false, true, 4>); break;  case 16 + 3: rc_ = go(KERNEL<true, true, 4>); break;                  \
      case 32 + 0: rc_ = go(KERNEL<false, false, 8>); break; case 32 + 1: rc_ = go(KERNEL<true, false, 8>); break;                 \
      case 32 + 2: rc_ = go(KERNEL<false, true, 8>); break;  case 32 + 3: rc_ = go(KERNEL<true, true, 8>); break;                  \
      default: rc_ = BATH_EINVAL;                                                                                                \
    }                                                                                                                            \
    if (rc_ != BATH_OK) return rc_;                                                                                              \
  } while (0)

// Launch the wavefront Forward over all envelopes of <dna> on <stream>.  <ring_scratch>: a context scratch buffer for the ring
// when it does not fit into LDS next to the table.
int launch_fs5_fwd_wf(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int exact, int c5_compat,
                      float *d_sc, float *d_fwd, const int64_t *d_foff, float *d_xmx, const int64_t *d_xoff, DevBuf &ring_scratch, FsJobs jobs) {
  const int M = om->M;
#ifdef BATH_WF_PROBES
  static const int dbg = [] { const char *e = std::getenv("BATH_HIP_WF_DBG"); return e ? std::atoi(e) : 0; }();   // timing probes, wrong results: probe builds only
#else
  constexpr int dbg = 0;
#endif
  WfGeom g{};
  int st = fs_wf_geometry(ctx, dna->n, M, ring_scratch, &g);
  if (st != BATH_OK) return st;
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  BATH_WF_DISPATCH(fs5_fwd_wf_kernel, dna->view(), dev, om->d_loop[1], om->d_move[1], c5_compat, d_sc, d_fwd, d_foff, d_xmx, d_xoff, g.ring_g, jobs, dbg);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

// Backward of all envelopes: the sweep, then the B sums / N, J rows / scores.  The offsets of the envelopes' (steps x rows in
// flight) term arrays in <terms_scratch> are built here.
int launch_fs5_bwd_wf(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int exact,
                      float *d_sc, float *d_bck, const int64_t *d_boff, float *d_xmx, const int64_t *d_xoff, DevBuf &ring_scratch, DevBuf &terms_scratch, DevBuf &toff_scratch,
                      FsJobs jobs_sweep, FsJobs jobs_x) {
  const int M = om->M;
  const int64_t n = dna->n;
#ifdef BATH_WF_PROBES
  static const int dbg = [] { const char *e = std::getenv("BATH_HIP_WF_DBG"); return e ? std::atoi(e) : 0; }();   // timing probes, wrong results: probe builds only
#else
  constexpr int dbg = 0;
#endif
  WfGeom g{};
  int st = fs_wf_geometry(ctx, n, M, ring_scratch, &g);
  if (st != BATH_OK) return st;
  const int RW = 64 * g.W;
  std::vector<int64_t> toff((size_t)n + 1, 0);
  for (int64_t e = 0; e < n; e++) toff[(size_t)e + 1] = toff[(size_t)e] + (dna->h_len[(size_t)e] >= 5 ? ((fs_bwd_wf_steps(dna->h_len[(size_t)e], M, RW) + 3) & ~(int64_t)3) * RW : 0);   // the sweep runs in blocks of four steps
  BATH_HIP_TRY(ctx, terms_scratch.reserve((size_t)toff[(size_t)n] * sizeof(float) + 256));
  BATH_HIP_TRY(ctx, toff_scratch.reserve((size_t)(n + 1) * sizeof(int64_t)));
  if (ctx->stage_upload(3, toff_scratch.p, toff.data(), (size_t)(n + 1), stream) != BATH_OK) return BATH_EFAIL;   // through page-locked staging: no synchronize
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  BATH_WF_DISPATCH(fs5_bwd_wf_kernel, dna->view(), dev, om->d_loop[1], om->d_move[1], d_bck, d_boff, d_xmx, d_xoff, terms_scratch.as<float>(), toff_scratch.as<int64_t>(), g.ring_g, jobs_sweep, dbg);
  BATH_HIP_TRY(ctx, hipGetLastError());
  const int xwaves = kFsBlock / 64;
  // few envelopes (a wave each would leave most CUs without a block): a block per envelope, its waves sharing the rows
  static const int team_env = [] { const char *e = std::getenv("BATH_HIP_FS_BWDX_TEAM"); return e ? std::atoi(e) : -1; }();
  const int team = team_env >= 0 ? (team_env != 0) : (n <= (int64_t)ctx->prop.multiProcessorCount * 2 ? 1 : 0);
  const int xgrid = team ? (int)std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)ctx->prop.multiProcessorCount * 2))
                         : (int)std::max<int64_t>(1, std::min<int64_t>((n + xwaves - 1) / xwaves, (int64_t)ctx->prop.multiProcessorCount * 2));
  const size_t xshmem = (size_t)kLogsumTbl * sizeof(float);
  if (exact) hipLaunchKernelGGL((fs5_bwd_x_kernel<true>), dim3(xgrid), dim3(kFsBlock), xshmem, stream, dna->view(), M, om->d_logsum, om->d_loop[1], om->d_move[1], d_xmx, d_xoff,
                                terms_scratch.as<float>(), toff_scratch.as<int64_t>(), d_sc, jobs_x, RW, team);
  else hipLaunchKernelGGL((fs5_bwd_x_kernel<false>), dim3(xgrid), dim3(kFsBlock), xshmem, stream, dna->view(), M, om->d_logsum, om->d_loop[1], om->d_move[1], d_xmx, d_xoff,
                          terms_scratch.as<float>(), toff_scratch.as<int64_t>(), d_sc, jobs_x, RW, team);
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
