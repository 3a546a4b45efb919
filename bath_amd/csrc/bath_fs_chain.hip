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
#include <algorithm>
#include <cstring>
#include <vector>
#include <type_traits>

#include "bath_fs_device.hpp"

namespace bath {

constexpr int kChainMaxWaves = 16;
// threads per block by nodes per lane: a lane holds both rows of the pair and rows i-1..i-3 for its C nodes, so the register
// budget per wave grows with C while LDS (two staged rows per window) limits the windows per block anyway
constexpr int chain_threads(int C) { return C <= 3 ? 1024 : (C <= 8 ? 512 : 256); }
// Long models (more than 8 nodes per lane: M > 512), Backward parser.  LDS, not registers, limits the windows of a block there: the
// 64 KB table, the transitions (33 KB at 1024 nodes) and 8 KB of staged rows per window left room for FOUR windows -- a quarter of a
// chain wave's lanes carrying a row, and a 1 Gb genome's 2.6 k windows in 658 blocks, 2.6 rounds of the chip.  The chain needs three
// transitions per node (tDD, tDM, tBM): they are kept in LDS compactly (16 B per node), the window waves read theirs from global
// memory (33 KB that every block reads: L2-resident), and a block takes EIGHT windows (512 threads; the kernel fits 256 registers).
constexpr bool chain_compact(int C) { return C >= 12; }
constexpr int bwd_chain_threads(int C) { return chain_compact(C) ? 512 : chain_threads(C); }

// The waves of a block exchange rows through LDS only.  __syncthreads() is also a fence for GLOBAL memory: it would make every
// wave wait, twice per row, until its stores of that row (special-state rows; the regions' matrices, which go to page-locked
// host memory over PCIe) have been acknowledged.  This barrier waits for the LDS operations alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// While the chain wave runs alone, the waves 1..3 of the block -- one on each of the CU's other SIMDs -- do not park at the barrier
// but poll a word of LDS between sleeps (kChainAwakeWaves; the block is launched with at least four waves for this, whatever the
// number of windows it holds).  A CU whose other SIMDs hold only parked waves issues the chain's dependent instructions 5-30 %
// slower, by an amount that differs from block to block and from launch to launch (tools/lsbench7.hip: 198-267 clocks per Forward
// node, median 213, beside parked waves; 199-203 beside pollers that sleep 64 clocks per poll; 226-231 beside waves that spin on
// the VALU).  The word counts row pairs: the chain wave stores the pair's number when its serial part is done.
#ifndef BATH_CHAIN_KEEPALIVE
#define BATH_CHAIN_KEEPALIVE 1
#endif
constexpr int kChainAwakeWaves = BATH_CHAIN_KEEPALIVE ? 4 : 1;   // the chain wave and its pollers
__device__ __forceinline__ void chain_keepalive(const int *flag, int pair) {
#if BATH_CHAIN_KEEPALIVE
  const volatile int *f = flag;
  while (*f != pair) __builtin_amdgcn_s_sleep(1);
#endif
}
__device__ __forceinline__ void chain_done(int *flag, int pair) {
#if BATH_CHAIN_KEEPALIVE
  *reinterpret_cast<volatile int *>(flag) = pair;
#endif
}
__device__ __forceinline__ bool chain_poller(int wv) { return wv >= 1 && wv < kChainAwakeWaves; }

// staged rows: [wave][row of the pair][stride]; stride odd, so that the chain lanes (one row each) read different banks
__host__ __device__ inline int fs_chain_stride(int C) { return C * 64 + 1; }

// ---------------------------------------------------------------------------------------------------------------------------
// The chain loops, scheduled by hand.  A chain lane's node is two DEPENDENT table log-sums (E <- LS(M_k, LS(D_k, E))) with a third
// beside them (D_{k+1}); a wave issues in order, so what the compiler puts between the instructions of the dependent pair is time
// on the chain: it left the quieting v_max of every operand, register moves and the D chain's arithmetic in front of the
// look-ups (145 ns per node at M = 145; the two log-sums alone are 2 x (5 VALU + ds_read + add) ~ 75 ns).  Here every
// instruction that is not on the E path sits in the shadow of a table look-up: the D chain's log-sum and the store of D_k behind
// the first look-up; the loads for node k+1, D_{k+1} and the address arithmetic behind the second.  LDS operations of a wave
// return in order, so the waits count them: lgkmcnt(n) = "all but the youngest n".
//   e, d      E so far, D_k           (in/out)
//   st        LDS byte address of the row's slot k: M_k is read there a node ahead, D_k is left there
//   tp        LDS byte address of {tMD, tDD}(k+1)
//   set A     M_k, tMD(k), tDD(k) on entry; the asm covers TWO nodes and leaves set A for node k+2
// ---------------------------------------------------------------------------------------------------------------------------
// nodes per trip of the chain loops (pairs of nodes repeated)
#ifndef BATH_CHAIN_UNROLL
#define BATH_CHAIN_UNROLL 16
#endif
#if BATH_CHAIN_UNROLL == 8
#define BATH_CHAIN_REPEAT(P) P P P P
#elif BATH_CHAIN_UNROLL == 4
#define BATH_CHAIN_REPEAT(P) P P
#elif BATH_CHAIN_UNROLL == 16
#define BATH_CHAIN_REPEAT(P) P P P P P P P P
#else
#define BATH_CHAIN_REPEAT(P) P
#endif
#define BATH_LS_INDEX(a, x, y)                      \
  "v_sub_f32 " a ", " x ", " y "\n\t"               \
  "v_min_f32_e64 " a ", |" a "|, %[c15]\n\t"        \
  "v_mul_f32 " a ", 0x447a0000, " a "\n\t"          \
  "v_cvt_i32_f32 " a ", " a "\n\t"                  \
  "v_lshl_add_u32 " a ", " a ", 2, %[tbl]\n\t"      \
  "ds_read_b32 " a ", " a "\n\t"

#define BATH_FWD_NODE_TS(MK, TX, TY, MN, UX, UY, TS)                                                          \
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
  "v_add_u32 %[tp], " TS ", %[tp]\n\t"                                                                        \
  "s_waitcnt lgkmcnt(3)\n\t"                                    /* L3 */                                      \
  "v_add_f32 %[e], %[mx1], %[a1]\n\t"

// (TS: bytes per node of the transition table the chain reads: 32 for tf's rows of eight, 8 for the compact {tMD, tDD} pairs)
#define BATH_FWD_NODE(MK, TX, TY, MN, UX, UY) BATH_FWD_NODE_TS(MK, TX, TY, MN, UX, UY, BATH_FWD_TS)

struct FwdChainRegs { float e, d, Mk, tx, ty; unsigned st, tp; };

// <n> nodes of a Forward chain from the state in <r>; on return r.Mk, r.tx, r.ty belong to the node after the last one
#define BATH_FWD_TS "32"
__device__ __forceinline__ void fwd_chain_nodes(FwdChainRegs &r, int n, unsigned tbl, float c15) {
  float Mn, ux, uy, a1, a2, u, w, mx1, mxd, x;
  int k = 0;
  // (sixteen nodes per trip: the chain's dependency runs through the loop's own instructions -- the counter, the branch, the waits
  // either side of the asm block are issued in order between a node's last add and the next node's first subtraction: ~25 clocks
  // per node at two nodes per trip.  M = 1024, clocks per node at 2 / 16 nodes per trip: Forward 224 -> 200, B sum 120 -> 92,
  // D chain 250 -> 212; profiles/r06_chain_loops_ab.txt)
#define BATH_FWD_PAIR BATH_FWD_NODE("%[Mk]", "%[tx]", "%[ty]", "%[Mn]", "%[ux]", "%[uy]") BATH_FWD_NODE("%[Mn]", "%[ux]", "%[uy]", "%[Mk]", "%[tx]", "%[ty]")
  for (; k + BATH_CHAIN_UNROLL <= n; k += BATH_CHAIN_UNROLL)
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_CHAIN_REPEAT(BATH_FWD_PAIR)
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [Mk] "+v"(r.Mk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
  for (; k + 2 <= n; k += 2)
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_FWD_PAIR
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [Mk] "+v"(r.Mk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
  if (k < n) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_FWD_NODE("%[Mk]", "%[tx]", "%[ty]", "%[Mn]", "%[ux]", "%[uy]")
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [Mk] "v"(r.Mk), [tx] "v"(r.tx), [ty] "v"(r.ty), [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
    r.Mk = Mn; r.tx = ux; r.ty = uy;
  }
}

#undef BATH_FWD_TS
#undef BATH_FWD_PAIR
// the same over compact transitions: r.tp walks over pairs {tMD(k), tDD(k)}
#define BATH_FWD_TS "8"
__device__ __forceinline__ void fwd_chain_nodes_compact(FwdChainRegs &r, int n, unsigned tbl, float c15) {
  float Mn, ux, uy, a1, a2, u, w, mx1, mxd, x;
  int k = 0;
  // (sixteen nodes per trip: the chain's dependency runs through the loop's own instructions -- the counter, the branch, the waits
  // either side of the asm block are issued in order between a node's last add and the next node's first subtraction: ~25 clocks
  // per node at two nodes per trip.  M = 1024, clocks per node at 2 / 16 nodes per trip: Forward 224 -> 200, B sum 120 -> 92,
  // D chain 250 -> 212; profiles/r06_chain_loops_ab.txt)
#define BATH_FWD_PAIR BATH_FWD_NODE("%[Mk]", "%[tx]", "%[ty]", "%[Mn]", "%[ux]", "%[uy]") BATH_FWD_NODE("%[Mn]", "%[ux]", "%[uy]", "%[Mk]", "%[tx]", "%[ty]")
  for (; k + BATH_CHAIN_UNROLL <= n; k += BATH_CHAIN_UNROLL)
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_CHAIN_REPEAT(BATH_FWD_PAIR)
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [Mk] "+v"(r.Mk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
  for (; k + 2 <= n; k += 2)
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_FWD_PAIR
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [Mk] "+v"(r.Mk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
  if (k < n) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 BATH_FWD_NODE("%[Mk]", "%[tx]", "%[ty]", "%[Mn]", "%[ux]", "%[uy]")
                 "s_waitcnt lgkmcnt(0)"
                 : [e] "+v"(r.e), [d] "+v"(r.d), [st] "+v"(r.st), [tp] "+v"(r.tp),
                   [Mn] "=&v"(Mn), [ux] "=&v"(ux), [uy] "=&v"(uy), [a1] "=&v"(a1), [a2] "=&v"(a2), [u] "=&v"(u), [w] "=&v"(w),
                   [mx1] "=&v"(mx1), [mxd] "=&v"(mxd), [x] "=&v"(x)
                 : [Mk] "v"(r.Mk), [tx] "v"(r.tx), [ty] "v"(r.ty), [tbl] "s"(tbl), [c15] "s"(c15)
                 : "memory");
    r.Mk = Mn; r.tx = ux; r.ty = uy;
  }
}

#undef BATH_FWD_TS
#undef BATH_FWD_PAIR
// ---- the Backward chains the same way.  B(i) = LS over the nodes, ascending, of ivx(i,k) + tBM(k-1): one log-sum per node,
// the term of node k+1 added up and the raw values of node k+2 loaded behind the look-up.
//   b       B so far (in/out);  v: the term of this node;  sN, tN: ivx and tBM of the NEXT node, loaded by the node before
//   st, tp  LDS byte addresses of the row's slot k and of tBM(k-1) (s_tb[k * 8 + 7])
// (TS, TS2: the transition table's stride per node in bytes and twice that: 32 / 64 for the 8-float rows of tb, 16 / 32 for the
// compact {tDD, tDM, tBM, -} rows that long models keep in LDS)
#define BATH_BSUM_NODE(V, VN, TS, TS2)                                                 \
  BATH_LS_INDEX("%[a1]", "%[b]", V)                                                    \
  "s_waitcnt lgkmcnt(1)\n\t"                                                           \
  "v_add_f32 " VN ", %[sN], %[tN]\n\t"                                                 \
  "ds_read_b32 %[sN], %[st] offset:8\n\t"                                              \
  "ds_read_b32 %[tN], %[tp] offset:" TS2 "\n\t"                                        \
  "v_max_f32 %[mx], %[b], " V "\n\t"                                                   \
  "v_add_u32 %[st], 4, %[st]\n\t"                                                      \
  "v_add_u32 %[tp], " TS ", %[tp]\n\t"                                                 \
  "s_waitcnt lgkmcnt(2)\n\t"                                                           \
  "v_add_f32 %[b], %[mx], %[a1]\n\t"

// b = LS(b, term(k)) for k = k0 .. k0 + n - 1; on entry v = term(k0), sN/tN = the raw values of node k0 + 1, st/tp at node k0
template <bool COMPACT = false>
__device__ __forceinline__ float bwd_bsum_nodes(float b, float v, float sN, float tN, unsigned st, unsigned tp, int n, unsigned tbl, float c15) {
  float vn, a1, mx;
  int k = 0;
#define BATH_BSUM_PAIR(TS, TS2)                                                                                                          \
  for (; k + BATH_CHAIN_UNROLL <= n; k += BATH_CHAIN_UNROLL)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_CHAIN_REPEAT(BATH_BSUM_NODE("%[v]", "%[vn]", TS, TS2) BATH_BSUM_NODE("%[vn]", "%[v]", TS, TS2))                    \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [b] "+v"(b), [v] "+v"(v), [sN] "+v"(sN), [tN] "+v"(tN), [st] "+v"(st), [tp] "+v"(tp), [vn] "=&v"(vn), [a1] "=&v"(a1), [mx] "=&v"(mx) \
                 : [tbl] "s"(tbl), [c15] "s"(c15)                                                                                       \
                 : "memory");                                                                                                           \
  for (; k + 2 <= n; k += 2)                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_BSUM_NODE("%[v]", "%[vn]", TS, TS2)                                                                               \
                 BATH_BSUM_NODE("%[vn]", "%[v]", TS, TS2)                                                                               \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [b] "+v"(b), [v] "+v"(v), [sN] "+v"(sN), [tN] "+v"(tN), [st] "+v"(st), [tp] "+v"(tp), [vn] "=&v"(vn), [a1] "=&v"(a1), [mx] "=&v"(mx) \
                 : [tbl] "s"(tbl), [c15] "s"(c15)                                                                                       \
                 : "memory");                                                                                                           \
  if (k < n)                                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_BSUM_NODE("%[v]", "%[vn]", TS, TS2)                                                                               \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [b] "+v"(b), [sN] "+v"(sN), [tN] "+v"(tN), [st] "+v"(st), [tp] "+v"(tp), [vn] "=&v"(vn), [a1] "=&v"(a1), [mx] "=&v"(mx) \
                 : [v] "v"(v), [tbl] "s"(tbl), [c15] "s"(c15)                                                                           \
                 : "memory");
  if constexpr (COMPACT) { BATH_BSUM_PAIR("16", "32") } else { BATH_BSUM_PAIR("32", "64") }
#undef BATH_BSUM_PAIR
  return b;
}

// D(i,k) = LS(LS(E, p1), p2), nodes descending, with {p1, p2} = {D(i,k+1) + tDD(k), ivx(i,k+1) + tDM(k)} -- the rows L-3, L-4
// (<mid>, a lane mask) take them in the other order (:1524-1526).  Two dependent look-ups per node; the loads of node k-1 sit
// behind the first, the address arithmetic behind the second; D(i,k) is stored over ivx(i,k), which was read a node ahead.
//   dn   D(i,k+1) (in/out);  IVN: ivx(i,k+1);  set A = {ivx(i,k), tDD(k), tDM(k)};  st at slot k-1, tp at s_tb[(k-1) * 8 + 3]
// (H: the node's first instructions -- BATH_BWD_D_HEAD_MID selects the order per lane; BATH_BWD_D_HEAD_PLAIN is the order of every row
// but L-3 and L-4, without the two selects, one of which sits on the dependent path: all pairs of a block but the second and third)
#define BATH_BWD_D_HEAD_MID(IVN, TX, TY)                                               \
  "v_add_f32 %[u], %[dn], " TX "\n\t"                                                  \
  "v_add_f32 %[bs], " IVN ", " TY "\n\t"                                               \
  "v_cndmask_b32_e64 %[p1], %[u], %[bs], %[mid]\n\t"                                   \
  "v_cndmask_b32_e64 %[p2], %[bs], %[u], %[mid]\n\t"
#define BATH_BWD_D_HEAD_PLAIN(IVN, TX, TY)                                             \
  "v_add_f32 %[p1], %[dn], " TX "\n\t"                                                 \
  "v_add_f32 %[p2], " IVN ", " TY "\n\t"
#define BATH_BWD_D_NODE(H, IVN, TX, TY, IVQ, UX, UY, NTS)                              \
  H(IVN, TX, TY)                                                                       \
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
  "v_add_u32 %[tp], " NTS ", %[tp]\n\t"                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "v_add_f32 %[dn], %[mx1], %[a1]\n\t"                                                 \
  "ds_write_b32 %[st], %[dn] offset:8\n\t"

struct BwdChainRegs { float dn, ivn, ivk, tx, ty; unsigned st, tp; };

template <bool COMPACT = false>
__device__ __forceinline__ void bwd_d_nodes(BwdChainRegs &r, float xE, unsigned long long mid, int n, unsigned tbl, float c15) {
  float ivq, ux, uy, u, bs, p1, p2, a1, mx1, x;
  int k = 0;
#define BATH_BWD_D_PAIR(NTS, H)                                                                                                          \
  for (; k + BATH_CHAIN_UNROLL <= n; k += BATH_CHAIN_UNROLL)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_CHAIN_REPEAT(BATH_BWD_D_NODE(H, "%[ivn]", "%[tx]", "%[ty]", "%[ivq]", "%[ux]", "%[uy]", NTS)                          \
                                   BATH_BWD_D_NODE(H, "%[ivk]", "%[ux]", "%[uy]", "%[ivk]", "%[tx]", "%[ty]", NTS)                          \
                                   "v_mov_b32 %[ivn], %[ivq]\n\t")                                                                      \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [dn] "+v"(r.dn), [ivn] "+v"(r.ivn), [ivk] "+v"(r.ivk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp), \
                   [ivq] "=&v"(ivq), [ux] "=&v"(ux), [uy] "=&v"(uy), [u] "=&v"(u), [bs] "=&v"(bs), [p1] "=&v"(p1), [p2] "=&v"(p2),       \
                   [a1] "=&v"(a1), [mx1] "=&v"(mx1), [x] "=&v"(x)                                                                       \
                 : [xE] "v"(xE), [mid] "s"(mid), [tbl] "s"(tbl), [c15] "s"(c15)                                                         \
                 : "memory");                                                                                                           \
  for (; k + 2 <= n; k += 2)                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_BWD_D_NODE(H, "%[ivn]", "%[tx]", "%[ty]", "%[ivq]", "%[ux]", "%[uy]", NTS)                                           \
                 BATH_BWD_D_NODE(H, "%[ivk]", "%[ux]", "%[uy]", "%[ivk]", "%[tx]", "%[ty]", NTS)                                           \
                 "v_mov_b32 %[ivn], %[ivq]\n\t"                                                                                         \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [dn] "+v"(r.dn), [ivn] "+v"(r.ivn), [ivk] "+v"(r.ivk), [tx] "+v"(r.tx), [ty] "+v"(r.ty), [st] "+v"(r.st), [tp] "+v"(r.tp), \
                   [ivq] "=&v"(ivq), [ux] "=&v"(ux), [uy] "=&v"(uy), [u] "=&v"(u), [bs] "=&v"(bs), [p1] "=&v"(p1), [p2] "=&v"(p2),       \
                   [a1] "=&v"(a1), [mx1] "=&v"(mx1), [x] "=&v"(x)                                                                       \
                 : [xE] "v"(xE), [mid] "s"(mid), [tbl] "s"(tbl), [c15] "s"(c15)                                                         \
                 : "memory");                                                                                                           \
  if (k < n) {                                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
                 BATH_BWD_D_NODE(H, "%[ivn]", "%[tx]", "%[ty]", "%[ivq]", "%[ux]", "%[uy]", NTS)                                           \
                 "s_waitcnt lgkmcnt(0)"                                                                                                 \
                 : [dn] "+v"(r.dn), [st] "+v"(r.st), [tp] "+v"(r.tp),                                                                   \
                   [ivq] "=&v"(ivq), [ux] "=&v"(ux), [uy] "=&v"(uy), [u] "=&v"(u), [bs] "=&v"(bs), [p1] "=&v"(p1), [p2] "=&v"(p2),       \
                   [a1] "=&v"(a1), [mx1] "=&v"(mx1), [x] "=&v"(x)                                                                       \
                 : [ivn] "v"(r.ivn), [tx] "v"(r.tx), [ty] "v"(r.ty), [xE] "v"(xE), [mid] "s"(mid), [tbl] "s"(tbl), [c15] "s"(c15)       \
                 : "memory");                                                                                                           \
    r.ivn = r.ivk; r.ivk = ivq; r.tx = ux; r.ty = uy;                                                                                   \
  }
  if (mid == 0) { if constexpr (COMPACT) { BATH_BWD_D_PAIR("-16", BATH_BWD_D_HEAD_PLAIN) } else { BATH_BWD_D_PAIR("-32", BATH_BWD_D_HEAD_PLAIN) } }
  else { if constexpr (COMPACT) { BATH_BWD_D_PAIR("-16", BATH_BWD_D_HEAD_MID) } else { BATH_BWD_D_PAIR("-32", BATH_BWD_D_HEAD_MID) } }
#undef BATH_BWD_D_PAIR
}

// LDS byte address of a pointer into the block's dynamic shared memory (the low half of its flat address)
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)p; }

// ---------------------------------------------------------------------------------------------------------------------------
// 3-codon Forward parser, multihit, strict.  tf[node] = {tMM(k-1), tIM(k-1), tDM(k-1), tBM(k-1), tMD(k), tDD(k), tMI(k), tII(k)}
// xmx (optional): (L+1) x {E,N,J,B,C}
// ---------------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(chain_threads(C)) void fs3_fwd_chain_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                             float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs,
                                                             int W /* windows per block: the block has max(W, kChainAwakeWaves) waves */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  const int M = p.M;
  const int stride = fs_chain_stride(C);
  float *s_stage = s_tf + (M + 2) * 8;                          // [W][2][stride]
  float *s_e = s_stage + (size_t)W * 2 * stride;                // [W][2] E(i) of the pair's rows
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * kChainMaxWaves);   // [0]: first job of the block's batch; [4]: the row pair whose serial part is done
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef BATH_CHAIN_CLOCK
  long long dbg_cyc = 0, dbg_wall = 0, dbg_n = 0;
#endif
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  for (;;) {
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, (unsigned)W);
    __syncthreads();
    const int64_t base = s_ctl[0];
    if (base >= dna.n) break;
    const int64_t job = (wv < W && base + wv < dna.n) ? (int64_t)jobs.order[base + wv] : (int64_t)-1;
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
      ++pair;
      if (wv >= W) { lds_barrier(); chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }   // a poller without a window
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
        // (no branches in this loop: loads and stores are unconditional -- the slots past node M exist -- so that the log-sums of
        // a lane's C nodes, independent of each other, can be interleaved)
        const float la2 = qa2[ne], la3 = qa3[ne], la4 = qa4[ne], lb2 = qb2[ne], lb3 = qb3[ne], lb4 = qb4[ne];
        const float ea2 = in ? la2 : -INFINITY, ea3 = in ? la3 : -INFINITY, ea4 = in ? la4 : -INFINITY;
        const float eb2 = in ? lb2 : -INFINITY, eb3 = in ? lb3 : -INFINITY, eb4 = in ? lb4 : -INFINITY;
        // row A = i: from row i-2 and B(i-2) (:562-569); row 2 takes B(0) only (:503)
        const float mA = (c == 0) ? mInA : M2[c - 1], iA = (c == 0) ? iInA : I2[c - 1], dA = (c == 0) ? dInA : D2[c - 1];
        float a = LS(mA + ta.x, LS(iA + ta.y, LS(dA + ta.z, B2 + ta.w)));
        // (row 2 takes B(0) only (:503): rows 0 and 1 are -inf, and LS(-inf, x) = x exactly)
        ivA[c] = a;
        float mv = a + ea2;
        mv = LS(mv, iv_1[c] + ea3); mv = LS(mv, iv_2[c] + ea4);      // :571-574 (row 2: the IVX of rows 1, 0 are -inf)
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
        s_stage[((size_t)wv * 2 + 0) * stride + node] = mv; s_stage[((size_t)wv * 2 + 1) * stride + node] = mw;
      }
      lds_barrier();
      // ---- 2. the serial part, a lane per row: D(i,k) = LS(M(i,k-1) + tMD, D(i,k-1) + tDD), E(i) = LS(M(i,k), LS(D(i,k), E)) (:577-590)
      if (wv == 0 && lane < 2 * W) {
        float *st = s_stage + (size_t)lane * stride;
        float dch = -INFINITY, ech = -INFINITY;
        // the loop's own loads (M(i,k), the transitions) are a node ahead: what a node waits for is the chain's log-sums only
        // (slot M+1 of the row and of the transitions exists: the loads run a node ahead)
        FwdChainRegs r{ech, dch, st[1], s_tf[1 * 8 + 4], s_tf[1 * 8 + 5], lds_addr(st + 1), lds_addr(s_tf + 2 * 8 + 4)};   // tMD(k), tDD(k)
#ifdef BATH_CHAIN_CLOCK
        const long long cc0 = clock64(), ww0 = wall_clock64();
#endif
        fwd_chain_nodes(r, M, lds_addr(s_tbl), 15.999f);
#ifdef BATH_CHAIN_CLOCK
        dbg_cyc += clock64() - cc0; dbg_wall += wall_clock64() - ww0; dbg_n += M;
#endif
        ech = r.e; dch = r.d;
        s_e[lane] = ech;
      }
      if (wv == 0) chain_done(s_ctl + 4, pair); else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
      lds_barrier();
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
#ifdef BATH_CHAIN_CLOCK
    if (wv == 0 && lane == 0 && blockIdx.x == 0 && dbg_n > 0) printf("fwd chain: %.1f clock64 ticks per node, %.1f ns per node (%lld nodes)\n", (double)dbg_cyc / dbg_n, (double)dbg_wall / dbg_n * 10.0, dbg_n);
#endif
    __syncthreads();                                            // s_ctl is rewritten at the top
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same for LONG models (12 or 16 nodes per lane) with the rows' history in global memory instead of registers.
// fs3_fwd_chain_kernel<16> keeps ten rows of history, six new rows and the emission scores in 460 registers per lane: a CU's
// register file holds four such windows, and a chain wave that could carry 64 rows carries 8.  Here a window's wave keeps
// nothing from one row pair to the next: what a pair leaves (M, I, D, IVX of both rows, <hist>: two records of eight rows per
// window slot, the pair's and the one before) is read back where the next pairs need it, four nodes at a time, so that
// registers and LDS hold EIGHT windows per block of 512 threads.
// A lane owns the nodes in STRIPES of four: its g-th group is the nodes (g * 64 + lane) * 4 + 1 .. + 4, so that everything a wave
// touches per group is contiguous -- 1 KB of a history row per load or store, 256 consecutive emission scores per codon row, the
// stage row's slots -- and the transitions, kept as six arrays by kind (tMM, tIM, tDM, tBM, tMI, tII at node k in slot k - 1),
// are read as one float4 per lane without LDS bank conflicts (with a lane owning C consecutive nodes of tf's rows of eight, the
// kernel above reads them 64 lanes to a bank).  The rows
// i-1, i-2 "at node k-1" are the same vectors moved on by one node: the lane below's last element (DPP), lane 0 taking lane
// 63's of the group before.  Every log-sum has the operands and the order of the kernel above.
// Traffic: 72 KB per window and row pair at M = 1024 (ten rows in, eight out) against a pair's ~100 us: ~5 GB/s per CU.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kChainHistRows = 8;                                // MA IA DA MB IB DB ivA ivB
__host__ __device__ inline int fs_chain_hist_pitch(int C) { return C * 64; }   // floats per row
__host__ __device__ inline size_t fs_chain_mem_fixed_lds(int M, int C) {       // everything but the stage rows
  return (size_t)(kLogsumTbl + 6 * C * 64 + (M + 3) * 2 + 2 * kChainMaxWaves + 16) * sizeof(float);
}
template <int C>
__global__ __launch_bounds__(512) void fs3_fwd_chain_mem_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                                 float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs,
                                                                 int W /* windows per block: the block has max(W, kChainAwakeWaves) waves */,
                                                                 float *__restrict__ hist /* [blocks][W][2][kChainHistRows][fs_chain_hist_pitch(C)] */) {
  static_assert(C % 4 == 0, "groups of four nodes");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int HP = C * 64, G = C / 4;
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_t6 = s_tbl + kLogsumTbl;                             // [6][HP]: tMM(k-1), tIM(k-1), tDM(k-1), tBM(k-1), tMI(k), tII(k) of node k in slot k - 1
  float *s_tc = s_t6 + 6 * HP;                                  // [(M + 3)][2] = {tMD(k), tDD(k)}: the chain's
  const int M = p.M;
  const int stride = fs_chain_stride(C);
  float *s_stage = s_tc + (M + 3) * 2;                          // [W][2][stride]
  float *s_e = s_stage + (size_t)W * 2 * stride;                // [W][2] E(i) of the pair's rows
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * kChainMaxWaves);   // [0]: first job of the block's batch; [4]: the row pair whose serial part is done
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
#ifdef BATH_CHAIN_CLOCK
  long long dbg_t[5] = {0, 0, 0, 0, 0}, dbg_n = 0;
#endif
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int n = threadIdx.x; n < HP; n += blockDim.x) {
    const int nd = imin(n + 1, M + 1);                          // (the nodes past the model's end read row M + 1, as the kernel above does)
    s_t6[0 * HP + n] = p.tf[nd * 8 + 0]; s_t6[1 * HP + n] = p.tf[nd * 8 + 1]; s_t6[2 * HP + n] = p.tf[nd * 8 + 2]; s_t6[3 * HP + n] = p.tf[nd * 8 + 3];
    s_t6[4 * HP + n] = p.tf[nd * 8 + 6]; s_t6[5 * HP + n] = p.tf[nd * 8 + 7];
  }
  for (int k = threadIdx.x; k < M + 3; k += blockDim.x) { const int kk = imin(k, M + 1); s_tc[k * 2] = p.tf[kk * 8 + 4]; s_tc[k * 2 + 1] = p.tf[kk * 8 + 5]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float *const rec0 = hist + ((size_t)blockIdx.x * W + imin(wv, W - 1)) * 2 * kChainHistRows * HP;   // row a of record r: rec0 + (r * 8 + a) * HP, node k in slot k - 1
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  const float4 ninf4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  auto ld4 = [&](const float *row, int g, bool on) -> float4 { float4 v = ninf4; if (on) v = reinterpret_cast<const float4 *>(row)[g * 64 + lane]; return v; };
  auto st4 = [&](float *row, int g, const float4 &v) { reinterpret_cast<float4 *>(row)[g * 64 + lane] = v; };
  auto el = [](const float4 &v, int j) -> float { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };
  // a row's vector moved on by one node: the lane below's last element comes first; lane 0 takes <c63>, lane 63's last element of the group before
  auto moved = [&](const float4 &v, float &c63) -> float4 {
    float first = wave_shr1(v.w, -INFINITY);
    if (lane == 0) first = c63;
    c63 = __shfl(v.w, 63, 64);
    return make_float4(first, v.x, v.y, v.z);
  };
  for (;;) {
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, (unsigned)W);
    __syncthreads();
    const int64_t base = s_ctl[0];
    if (base >= dna.n) break;
    const int64_t job = (wv < W && base + wv < dna.n) ? (int64_t)jobs.order[base + wv] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];                 // the batch's longest window (the list is sorted by length)
    const int L = job >= 0 ? dna.len[job] : 0;
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    float *xo = (xmx && job >= 0) ? xmx + xmx_off[job] : nullptr;
    const int Lc = L / 3;
    const float tNL = loop_tab[Lc], tNM = move_tab[Lc], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    float N1 = 0.f, N2 = 0.f, N3 = 0.f, J1 = -INFINITY, J2 = -INFINITY, J3 = -INFINITY, C1 = -INFINITY, C2 = -INFINITY, C3 = -INFINITY;
    float B1 = tNM, B2 = tNM;                                   // B(i-1), B(i-2)
    if (xo && lane == 0 && L >= 3)
      for (int i = 0; i < 2; i++) { xo[i * 5 + 0] = -INFINITY; xo[i * 5 + 1] = 0.f; xo[i * 5 + 2] = -INFINITY; xo[i * 5 + 3] = tNM; xo[i * 5 + 4] = -INFINITY; }
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 338) : 338; };
    int np = 0;                                                 // pairs this window has stored: record np & 1 takes the next one
    for (int i = 2; i <= Lmax; i += 2) {
      ++pair;
      if (wv >= W) { lds_barrier(); chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }   // a poller without a window
      const bool actA = job >= 0 && L >= 3 && i <= L, actB = job >= 0 && L >= 3 && i + 1 <= L;
      const int xa = nuc(i), wa = nuc(i - 1), va = nuc(i - 2), ua = nuc(i - 3), xb = nuc(i + 1);
      const float *qa2 = p.rsc + (size_t)imin(xa * 84 + wa * 21, 337) * p.pitch;
      const float *qa3 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + 1, 336) * p.pitch;
      const float *qa4 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + ua + 2, 337) * p.pitch;
      const float *qb2 = p.rsc + (size_t)imin(xb * 84 + xa * 21, 337) * p.pitch;
      const float *qb3 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + 1, 336) * p.pitch;
      const float *qb4 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + va + 2, 337) * p.pitch;
      // the records: <prev> holds the rows i-2 (its A rows) and i-1 (its B rows), <cur> -- until this pair overwrites it -- the rows
      // i-4 and i-3.  A window that has stored no pair (one pair) yet reads -inf instead (every row before row 2 is -inf).
      const bool h1 = np >= 1, h2 = np >= 2;
#ifdef BATH_CHAIN_CLOCK
      const long long t0 = clock64();
#endif
      __threadfence_block();                                    // the previous pairs' rows, stored by other lanes of this wave, before they are read
      float *const cur = rec0 + (size_t)(np & 1) * kChainHistRows * HP;
      const float *const prev = rec0 + (size_t)((np + 1) & 1) * kChainHistRows * HP;
      // ---- 1. everything of both rows that is parallel over the nodes, four nodes at a time; the rows and the emission scores of
      //         the next group are loaded while this one is computed
      struct Group { float4 m2, i2, d2, m1, i1, d1, iv2, iv1, m3, i3; float e[4][6]; };
      auto load_group = [&](int g, Group &q) {
        q.m2 = ld4(prev + 0 * HP, g, h1); q.i2 = ld4(prev + 1 * HP, g, h1); q.d2 = ld4(prev + 2 * HP, g, h1);       // row i-2
        q.m1 = ld4(prev + 3 * HP, g, h1); q.i1 = ld4(prev + 4 * HP, g, h1); q.d1 = ld4(prev + 5 * HP, g, h1);       // row i-1
        q.iv2 = ld4(prev + 6 * HP, g, h1); q.iv1 = ld4(prev + 7 * HP, g, h1);                                       // IVX of rows i-2, i-1
        q.m3 = ld4(cur + 3 * HP, g, h2); q.i3 = ld4(cur + 4 * HP, g, h2);                                           // row i-3
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int ne = imin((g * 64 + lane) * 4 + c + 1, M);
          q.e[c][0] = qa2[ne]; q.e[c][1] = qa3[ne]; q.e[c][2] = qa4[ne]; q.e[c][3] = qb2[ne]; q.e[c][4] = qb3[ne]; q.e[c][5] = qb4[ne];
        }
      };
      Group gc, gn;
      load_group(0, gc);
      float km2 = -INFINITY, ki2 = -INFINITY, kd2 = -INFINITY, km1 = -INFINITY, ki1 = -INFINITY, kd1 = -INFINITY;   // lane 63's last elements of the group before (node 0: -inf)
#pragma unroll 1
      for (int g = 0; g < G; g++) {
        const int n0 = (g * 64 + lane) * 4;                     // the group's first node is n0 + 1
        if (g + 1 < G) load_group(g + 1, gn);
        const float4 m2 = gc.m2, i2 = gc.i2, iv2 = gc.iv2, iv1 = gc.iv1, m3 = gc.m3, i3 = gc.i3;
        const float4 m2s = moved(gc.m2, km2), i2s = moved(gc.i2, ki2), d2s = moved(gc.d2, kd2);
        const float4 m1s = moved(gc.m1, km1), i1s = moved(gc.i1, ki1), d1s = moved(gc.d1, kd1);
        const float4 tmm = reinterpret_cast<const float4 *>(s_t6 + 0 * HP)[g * 64 + lane], tim = reinterpret_cast<const float4 *>(s_t6 + 1 * HP)[g * 64 + lane];
        const float4 tdm = reinterpret_cast<const float4 *>(s_t6 + 2 * HP)[g * 64 + lane], tbm = reinterpret_cast<const float4 *>(s_t6 + 3 * HP)[g * 64 + lane];
        const float4 tmi = reinterpret_cast<const float4 *>(s_t6 + 4 * HP)[g * 64 + lane], tii = reinterpret_cast<const float4 *>(s_t6 + 5 * HP)[g * 64 + lane];
        float oMA[4], oIA[4], oVA[4], oMB[4], oIB[4], oVB[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int node = n0 + c + 1;
          const bool in = node <= M;
          const float ea2 = in ? gc.e[c][0] : -INFINITY, ea3 = in ? gc.e[c][1] : -INFINITY, ea4 = in ? gc.e[c][2] : -INFINITY;
          const float eb2 = in ? gc.e[c][3] : -INFINITY, eb3 = in ? gc.e[c][4] : -INFINITY, eb4 = in ? gc.e[c][5] : -INFINITY;
          // row A = i: from row i-2 and B(i-2) (:562-569)
          const float a = LS(el(m2s, c) + el(tmm, c), LS(el(i2s, c) + el(tim, c), LS(el(d2s, c) + el(tdm, c), B2 + el(tbm, c))));
          oVA[c] = a;
          float mv = a + ea2;
          mv = LS(mv, el(iv1, c) + ea3); mv = LS(mv, el(iv2, c) + ea4);      // :571-574
          oMA[c] = mv;
          const float insA = LS(el(m3, c) + el(tmi, c), el(i3, c) + el(tii, c));
          oIA[c] = (i > 2 && node < M) ? insA : -INFINITY;
          // row B = i+1: from row i-1 and B(i-1); its 3- and 4-nucleotide codons start in rows i and i-1
          const float b = LS(el(m1s, c) + el(tmm, c), LS(el(i1s, c) + el(tim, c), LS(el(d1s, c) + el(tdm, c), B1 + el(tbm, c))));
          oVB[c] = b;
          float mw = b + eb2;
          mw = LS(mw, a + eb3); mw = LS(mw, el(iv1, c) + eb4);
          oMB[c] = mw;
          const float insB = LS(el(m2, c) + el(tmi, c), el(i2, c) + el(tii, c));
          oIB[c] = (node < M) ? insB : -INFINITY;
          s_stage[((size_t)wv * 2 + 0) * stride + node] = mv; s_stage[((size_t)wv * 2 + 1) * stride + node] = mw;
        }
        if (actB) {                                             // both rows exist: the pair becomes history (M and I of row i-3 were read above)
          st4(cur + 0 * HP, g, make_float4(oMA[0], oMA[1], oMA[2], oMA[3])); st4(cur + 1 * HP, g, make_float4(oIA[0], oIA[1], oIA[2], oIA[3]));
          st4(cur + 6 * HP, g, make_float4(oVA[0], oVA[1], oVA[2], oVA[3]));
          st4(cur + 3 * HP, g, make_float4(oMB[0], oMB[1], oMB[2], oMB[3])); st4(cur + 4 * HP, g, make_float4(oIB[0], oIB[1], oIB[2], oIB[3]));
          st4(cur + 7 * HP, g, make_float4(oVB[0], oVB[1], oVB[2], oVB[3]));
        }
        gc = gn;
      }
#ifdef BATH_CHAIN_CLOCK
      const long long t1 = clock64();
#endif
      lds_barrier();
#ifdef BATH_CHAIN_CLOCK
      const long long t2 = clock64();
#endif
      // ---- 2. the serial part, a lane per row (as in fs3_fwd_chain_kernel; the transitions are the compact pairs)
      if (wv == 0 && lane < 2 * W) {
        float *st = s_stage + (size_t)lane * stride;
        FwdChainRegs r{-INFINITY, -INFINITY, st[1], s_tc[1 * 2], s_tc[1 * 2 + 1], lds_addr(st + 1), lds_addr(s_tc + 2 * 2)};   // tMD(k), tDD(k)
        fwd_chain_nodes_compact(r, M, lds_addr(s_tbl), 15.999f);
        s_e[lane] = r.e;
      }
      if (wv == 0) chain_done(s_ctl + 4, pair); else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
#ifdef BATH_CHAIN_CLOCK
      const long long t3 = clock64();
#endif
      lds_barrier();
#ifdef BATH_CHAIN_CLOCK
      const long long t4 = clock64();
#endif
      // ---- 3. D of both rows into the pair's record; special states of both rows (:592-603)
      if (actB) {
#pragma unroll 1
        for (int g = 0; g < G; g++) {
          const int n0 = (g * 64 + lane) * 4;
          float da[4], db[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int node = n0 + c + 1, ne = imin(node, M);
            const float x = s_stage[((size_t)wv * 2 + 0) * stride + ne], y = s_stage[((size_t)wv * 2 + 1) * stride + ne];
            da[c] = (node <= M) ? x : -INFINITY; db[c] = (node <= M) ? y : -INFINITY;
          }
          st4(cur + 2 * HP, g, make_float4(da[0], da[1], da[2], da[3])); st4(cur + 5 * HP, g, make_float4(db[0], db[1], db[2], db[3]));
        }
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
        np++;
      } else if (actA) {                                        // the window's last row: only C(L), C(L-1), C(L-2) are still needed
        C3 = C2; C2 = C1; C1 = CA;
      }
#ifdef BATH_CHAIN_CLOCK
      dbg_t[0] += t1 - t0; dbg_t[1] += t2 - t1; dbg_t[2] += t3 - t2; dbg_t[3] += t4 - t3; dbg_t[4] += clock64() - t4; dbg_n++;
#endif
    }
#ifdef BATH_CHAIN_CLOCK
    if (lane == 0 && blockIdx.x == 0 && (wv == 0 || wv == W - 1) && dbg_n > 0)
      printf("fwd mem wave %d: per pair, clocks: parallel part %.0f, wait %.0f, chain (wave 0) or poll %.0f, wait %.0f, D rows + special states %.0f\n", wv,
             (double)dbg_t[0] / dbg_n, (double)dbg_t[1] / dbg_n, (double)dbg_t[2] / dbg_n, (double)dbg_t[3] / dbg_n, (double)dbg_t[4] / dbg_n);
#endif
    if (job >= 0 && lane == 0) sc[job] = (L >= 3) ? LS(C1, LS(C2 + tCL, C3 + tCL)) + tCM : -INFINITY;
    __syncthreads();                                            // s_ctl is rewritten at the top
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same for models of up to 32 x CH nodes with HALF a wave per window: 32 windows per block instead of 16, so the 7.6 k DNA
// windows of a bench block are all resident at once (the parsers are chains of rows: a second round of windows costs the first
// one's duration again) and every lane of the chain wave carries a row.  The special states move into the chain lanes (as in the
// Backward kernel: a lane keeps N, J, C of its slot's last rows and gets row i-3 from its neighbour), so a window lane holds
// nothing per window but its DP rows.
// ---------------------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(1024) void fs3_fwd_chain_half_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                                  float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs,
                                                                  const int32_t *__restrict__ bstart /* [nb + 1]: batch b = windows bstart[b] .. bstart[b+1]-1 of the sorted list */, int nb) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  const int M = p.M;
  constexpr int W = 32;                                         // window slots per block: two per wave
  constexpr int stride = CH * 32 + 1;
  float *s_stage = s_tf + (M + 2) * 8;                          // [W][2][stride]
  float *s_b = s_stage + (size_t)W * 2 * stride;                // [W][2] B(i) of the pair's rows, from the chain lanes
  int *s_ctl = reinterpret_cast<int *>(s_b + 2 * W);         // [0]: the block's batch; [4]: the row pair whose serial part is done
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tf[i] = p.tf[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hl = lane & 31, win = wv * 2 + (lane >> 5);          // lane within its window's half wave; the window's slot in the block
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  auto shr1 = [&](float v) { const float r = wave_shr1(v, -INFINITY); return hl == 0 ? -INFINITY : r; };   // the neighbour move stays inside the half wave
  // The batches come from the host (launch_fs3_fwd_chain): a block lasts as long as its longest window times the duration of a row
  // pair, which grows with the windows it holds -- so the batches of the longest windows are smaller, and all blocks end together.
  // Which batch a block takes next is decided when it asks (one atomic per batch on the launch's job counter): the batches are sorted by
  // decreasing length, so a launch of more batches than blocks is dealt longest-processing-time first instead of round-robin
  for (;;) {
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, 1u);
    __syncthreads();
    const int bb = s_ctl[0];
    if (bb >= nb) break;
    const int64_t base = bstart[bb];
    const int cnt = bstart[bb + 1] - bstart[bb];
    const int64_t job = (win < cnt) ? (int64_t)jobs.order[base + win] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];
    const int L = job >= 0 ? dna.len[job] : 0;
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    const bool wave_idle = wv != 0 && wv * 2 >= cnt;            // neither of the wave's two slots holds a window: it only keeps the barriers
    // ---- the chain wave's lanes: lane c serves row slot (c & 1) of window (c >> 1)
    const bool chain_lane = (wv == 0);
    const int cw = lane >> 1, cs = lane & 1;
    int64_t cjob = -1; int cL = 0; float *cxo = nullptr;
    float ctNL = 0.f, ctNM = 0.f;
    if (chain_lane && cw < cnt) {
      cjob = jobs.order[base + cw]; cL = dna.len[cjob];
      if (xmx) cxo = xmx + xmx_off[cjob];
      ctNL = loop_tab[cL / 3]; ctNM = move_tab[cL / 3];
      if (cxo && cs == 0 && cL >= 3)
        for (int i = 0; i < 2; i++) { cxo[i * 5 + 0] = -INFINITY; cxo[i * 5 + 1] = 0.f; cxo[i * 5 + 2] = -INFINITY; cxo[i * 5 + 3] = ctNM; cxo[i * 5 + 4] = -INFINITY; }
    }
    const bool clive = cjob >= 0 && cL >= 3;
    float hN1 = 0.f, hN2 = 0.f, hJ1 = -INFINITY, hJ2 = -INFINITY, hC1 = -INFINITY, hC2 = -INFINITY;   // N, J, C of the slot's rows one and two pairs ago (rows 0, 1: N = 0)
    float cfin1 = -INFINITY, cfin2 = -INFINITY, cfin3 = -INFINITY;                                      // C of the window's last three rows (slot 0's lane keeps them)
    if (chain_lane) { s_b[lane] = ctNM; }                         // B(0) = B(1) = tNM
    float M1[CH], I1[CH], D1[CH], M2[CH], I2[CH], D2[CH], M3[CH], I3[CH], iv_1[CH], iv_2[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) M1[c] = I1[c] = D1[c] = M2[c] = I2[c] = D2[c] = M3[c] = I3[c] = iv_1[c] = iv_2[c] = -INFINITY;
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 338) : 338; };
    __syncthreads();
    for (int i = 2; i <= Lmax; i += 2) {
      ++pair;
      if (wave_idle) { lds_barrier(); if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }
      const bool actB = job >= 0 && L >= 3 && i + 1 <= L;
      const float B2 = s_b[win * 2 + 0], B1 = s_b[win * 2 + 1];   // B(i-2), B(i-1): the previous pair's rows
      const int xa = nuc(i), wa = nuc(i - 1), va = nuc(i - 2), ua = nuc(i - 3), xb = nuc(i + 1);
      const float *qa2 = p.rsc + (size_t)imin(xa * 84 + wa * 21, 337) * p.pitch;
      const float *qa3 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + 1, 336) * p.pitch;
      const float *qa4 = p.rsc + (size_t)imin(xa * 84 + wa * 21 + va * 5 + ua + 2, 337) * p.pitch;
      const float *qb2 = p.rsc + (size_t)imin(xb * 84 + xa * 21, 337) * p.pitch;
      const float *qb3 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + 1, 336) * p.pitch;
      const float *qb4 = p.rsc + (size_t)imin(xb * 84 + xa * 21 + wa * 5 + va + 2, 337) * p.pitch;
      const float mInA = shr1(M2[CH - 1]), iInA = shr1(I2[CH - 1]), dInA = shr1(D2[CH - 1]);
      const float mInB = shr1(M1[CH - 1]), iInB = shr1(I1[CH - 1]), dInB = shr1(D1[CH - 1]);
      float MA[CH], IA[CH], ivA[CH], MB[CH], IB[CH], ivB[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int node = hl * CH + c + 1, nd = imin(node, M + 1), ne = imin(node, M);
        const float4 ta = *reinterpret_cast<const float4 *>(s_tf + nd * 8);
        const float4 tb = *reinterpret_cast<const float4 *>(s_tf + nd * 8 + 4);
        const bool in = node <= M;
        // (no branches in this loop: loads and stores are unconditional -- the slots past node M exist -- so that the log-sums of
        // a lane's C nodes, independent of each other, can be interleaved)
        const float la2 = qa2[ne], la3 = qa3[ne], la4 = qa4[ne], lb2 = qb2[ne], lb3 = qb3[ne], lb4 = qb4[ne];
        const float ea2 = in ? la2 : -INFINITY, ea3 = in ? la3 : -INFINITY, ea4 = in ? la4 : -INFINITY;
        const float eb2 = in ? lb2 : -INFINITY, eb3 = in ? lb3 : -INFINITY, eb4 = in ? lb4 : -INFINITY;
        const float mA = (c == 0) ? mInA : M2[c - 1], iA = (c == 0) ? iInA : I2[c - 1], dA = (c == 0) ? dInA : D2[c - 1];
        float a = LS(mA + ta.x, LS(iA + ta.y, LS(dA + ta.z, B2 + ta.w)));
        // (row 2 takes B(0) only (:503): rows 0 and 1 are -inf, and LS(-inf, x) = x exactly)
        ivA[c] = a;
        float mv = a + ea2;
        mv = LS(mv, iv_1[c] + ea3); mv = LS(mv, iv_2[c] + ea4);
        MA[c] = mv;
        const float insA = LS(M3[c] + tb.z, I3[c] + tb.w);
        IA[c] = (i > 2 && node < M) ? insA : -INFINITY;
        const float mB = (c == 0) ? mInB : M1[c - 1], iB = (c == 0) ? iInB : I1[c - 1], dB = (c == 0) ? dInB : D1[c - 1];
        const float b = LS(mB + ta.x, LS(iB + ta.y, LS(dB + ta.z, B1 + ta.w)));
        ivB[c] = b;
        float mw = b + eb2;
        mw = LS(mw, a + eb3); mw = LS(mw, iv_1[c] + eb4);
        MB[c] = mw;
        const float insB = LS(M2[c] + tb.z, I2[c] + tb.w);
        IB[c] = (node < M) ? insB : -INFINITY;
        s_stage[((size_t)win * 2 + 0) * stride + node] = mv; s_stage[((size_t)win * 2 + 1) * stride + node] = mw;
      }
      lds_barrier();
      // ---- the serial part: all 64 lanes of wave 0, a row each
      if (chain_lane) {
        float *st = s_stage + (size_t)lane * stride;
        float dch = -INFINITY, ech = -INFINITY;
        FwdChainRegs r{ech, dch, st[1], s_tf[1 * 8 + 4], s_tf[1 * 8 + 5], lds_addr(st + 1), lds_addr(s_tf + 2 * 8 + 4)};   // tMD(k), tDD(k)
        fwd_chain_nodes(r, M, lds_addr(s_tbl), 15.999f);
        ech = r.e; dch = r.d;
        // special states of row i + cs (:592-603); row i-3 is the other slot's row of two pairs ago (slot 0) or of the previous pair (slot 1)
        const float pN1 = __shfl_xor(hN1, 1, 64), pN2 = __shfl_xor(hN2, 1, 64), pJ1 = __shfl_xor(hJ1, 1, 64), pJ2 = __shfl_xor(hJ2, 1, 64);
        const float pC1 = __shfl_xor(hC1, 1, 64), pC2 = __shfl_xor(hC2, 1, 64);
        const float uN = cs ? pN1 : pN2, uJ = cs ? pJ1 : pJ2, uC = cs ? pC1 : pC2;
        const int irow = i + cs;
        float xN, xJ, xC;
        if (irow == 2) { xN = 0.f; xJ = ech + tEL; xC = ech + tEM; }
        else { xN = uN + ctNL; xJ = LS(uJ + ctNL, ech + tEL); xC = LS(uC + ctNL, ech + tEM); }
        const float xB = LS(xN + ctNM, xJ + ctNM);
        s_b[lane] = xB;
        const bool arow = clive && irow <= cL;
        if (arow && cxo) { float *r = cxo + (size_t)irow * 5; r[0] = ech; r[1] = xN; r[2] = xJ; r[3] = xB; r[4] = xC; }
        hN2 = hN1; hN1 = xN; hJ2 = hJ1; hJ1 = xJ; hC2 = hC1; hC1 = xC;
        // C(L), C(L-1), C(L-2) for the score: both slots' C in row order, kept by slot 0's lane
        const float cB = __shfl_xor(xC, 1, 64);                   // the other slot's C of this pair
        if (cs == 0 && clive) {
          if (i <= cL) { cfin3 = cfin2; cfin2 = cfin1; cfin1 = xC; }
          if (i + 1 <= cL) { cfin3 = cfin2; cfin2 = cfin1; cfin1 = cB; }
          if (i == cL || i + 1 == cL) sc[cjob] = LS(cfin1, LS(cfin2 + ctNL, cfin3 + ctNL)) + ctNM;
        }
        chain_done(s_ctl + 4, pair);
      } else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
      lds_barrier();
      float DA[CH], DB[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int node = hl * CH + c + 1, ne = imin(node, M);
        const float da = s_stage[((size_t)win * 2 + 0) * stride + ne], db = s_stage[((size_t)win * 2 + 1) * stride + ne];
        DA[c] = (node <= M) ? da : -INFINITY; DB[c] = (node <= M) ? db : -INFINITY;
      }
      if (actB) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
          M3[c] = M1[c]; I3[c] = I1[c]; M2[c] = MA[c]; I2[c] = IA[c]; D2[c] = DA[c]; M1[c] = MB[c]; I1[c] = IB[c]; D1[c] = DB[c];
          iv_2[c] = ivA[c]; iv_1[c] = ivB[c];
        }
      }
    }
    if (chain_lane && cs == 0 && cjob >= 0 && cL < 3) sc[cjob] = -INFINITY;
    __syncthreads();
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3-codon Backward parser, multihit, strict.  tb[node] = {tMD(k), tMI(k), tMM(k), tDD(k), tDM(k), tII(k), tIM(k), tBM(k-1)}
// Rows come in pairs from the END of each window: pair q holds the rows L-2q and L-2q-1 (they do not read each other: row i
// reads M(i+2..i+4) and I(i+3)), so the row TYPE -- no codon fits yet / fewer than the three codon lengths fit / main
// recursion (generic :1442-1677) -- is the same for every window of the block at a given q.  Per pair:
//   1. window waves: ivx(i,k) = logsum_c M(i+c,k) + e_c(k) for both rows, parallel over the nodes, into LDS;
//   2. the chain wave, a lane per row: B(i) = logsum_k ivx(i,k) + tBM(k-1) over k ascending (:1561-1565), the special states
//      of the row (they need B(i) and the rows i+3 of the other slot: a neighbour lane), then the D chain descending
//      (:1596-1599), leaving D(i,k) in place of ivx(i,k);
//   3. window waves: M(i,k), I(i,k) from D(i,k+1), ivx(i,k+1), I(i+3,k), E(i).
// ---------------------------------------------------------------------------------------------------------------------------
// WPW: waves per window.  Long models take two (C nodes per lane, 128 lanes) when a block holds up to four windows: a window's
// parallel part is a chain of dependent log-sums per node that one wave walks through alone on its SIMD -- 17 us of a 147 us row pair
// at M = 1024 with 16 nodes per lane and 87 spilled registers; two waves of 8 nodes per lane on two SIMDs (193 registers, no spills)
// make it 11 us: 147 -> 141 us per pair (profiles/r06_bwd_wpw_probe.txt).  The one value that crosses the waves (ivx of the first
// node of the wave above) goes through LDS.
template <int C, int WPW = 1, int MAXT = 1024>
__global__ __launch_bounds__(WPW == 1 ? bwd_chain_threads(C) : MAXT) void fs3_bwd_chain_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                             float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs,
                                                             const int32_t *__restrict__ bstart /* batches of the sorted list, chain_batches */, int nb,
                                                             int W /* windows per block: the block has max(W, kChainAwakeWaves) waves */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr bool kCompact = chain_compact(C * WPW);
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tb = s_tbl + kLogsumTbl;                             // kCompact: [(M + 3)][4] = {tDD(k), tDM(k), tBM(k-1), -}; else tb's [(M + 2)][8]
  const int M = p.M;
  const int stride = fs_chain_stride(C * WPW);
  float *s_stage = s_tb + (kCompact ? (M + 3) * 4 : (M + 2) * 8);   // [W][2][stride]: ivx(i,k) in, D(i,k) out
  float *s_e = s_stage + (size_t)W * 2 * stride;                // [W][2] E(i) of the pair's rows
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * kChainMaxWaves);   // [0]: first job of the block's batch; [4]: the row pair whose serial part is done
  float *s_xch = reinterpret_cast<float *>(s_ctl + 16);          // [waves][2]: ivx of both rows at the first (lowest) node of each wave, for the wave above
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
  fs_load_logsum_table(s_tbl, p.logsum);
  if constexpr (kCompact) {
    for (int k = threadIdx.x; k < M + 3; k += blockDim.x) {
      const int kk = imin(k, M + 1);
      s_tb[k * 4 + 0] = p.tb[kk * 8 + 3]; s_tb[k * 4 + 1] = p.tb[kk * 8 + 4]; s_tb[k * 4 + 2] = p.tb[kk * 8 + 7]; s_tb[k * 4 + 3] = 0.f;
    }
  } else {
    for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tb[i] = p.tb[i];
  }
  // the chain's transitions: tDD(k), tDM(k) (consecutive words), tBM(k-1)
  auto TDD = [&](int k) -> float * { return kCompact ? s_tb + k * 4 : s_tb + k * 8 + 3; };
  auto TBM = [&](int k) -> float * { return kCompact ? s_tb + k * 4 + 2 : s_tb + k * 8 + 7; };
  const float *wt = kCompact ? p.tb : s_tb;                    // the window waves' transitions (all eight per node): LDS, or global memory for long models
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Lanes own their nodes in DESCENDING order (logical lane = 63 - physical lane), as in fs_bwd_kernel: "the lane holding the
  // next nodes" is then the physical lane below and the neighbour move is the same wave_shr1
  const int win = wv / WPW, half = wv % WPW;                    // the wave's window slot; its place among the window's waves (0: the highest nodes)
  const int ll = 64 * WPW - 1 - (half * 64 + lane);
#ifdef BATH_CHAIN_CLOCK
  long long dbgb_cyc = 0, dbgd_cyc = 0, dbgb_n = 0;
#endif
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  for (;;) {                                                    // batches dealt longest first, on request (see fs3_fwd_chain_half_kernel)
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, 1u);
    __syncthreads();
    const int bb = s_ctl[0];
    if (bb >= nb) break;
    const int64_t base = bstart[bb];
    const int cnt = bstart[bb + 1] - bstart[bb];                // windows of this batch (<= W): the longest windows come in smaller batches
    const int64_t job = (win < cnt) ? (int64_t)jobs.order[base + win] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];
    const int L = job >= 0 ? dna.len[job] : 0;
    const bool live = job >= 0 && L >= 5;                       // shorter windows report -inf (fs_bwd_kernel does)
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    const bool wave_idle = wv != 0 && win >= cnt;               // no window in this wave's slot: it only keeps the barriers
    // ---- the chain wave's lanes: lane c serves slot (c & 1) of window (c >> 1)
    const int cw = lane >> 1, cs = lane & 1;
    const bool chain_lane = (wv == 0) && (lane < 2 * W);
    int64_t cjob = -1; int cL = 0; float *cxo = nullptr;
    float ctNL = 0.f, ctNM = 0.f;
    if (chain_lane && cw < cnt) {
      cjob = jobs.order[base + cw]; cL = dna.len[cjob];
      if (xmx) cxo = xmx + xmx_off[cjob];
      ctNL = loop_tab[cL / 3]; ctNM = move_tab[cL / 3];
    }
    const bool clive = cjob >= 0 && cL >= 5;
    float hN1 = -INFINITY, hN2 = -INFINITY, hJ1 = -INFINITY, hJ2 = -INFINITY, hC1 = -INFINITY, hC2 = -INFINITY;   // N, J, C of the slot's rows one and two pairs ago
    // ---- the window wave: M(i+1..i+4) and I(i+1..i+3) for the lane's nodes
    float R1[C], R2[C], R3[C], R4[C], J1[C], J2[C], J3[C];
#pragma unroll
    for (int c = 0; c < C; c++) R1[c] = R2[c] = R3[c] = R4[c] = J1[c] = J2[c] = J3[c] = -INFINITY;
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 338) : 338; };
    const int npairs = Lmax / 2 + 1;                            // rows Lmax .. 0 of the longest window
    for (int q = 0; q < npairs; q++) {
      ++pair;
      if (wave_idle) { lds_barrier(); if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }
      const int iA = L - 2 * q, iB = iA - 1;                    // this window's rows of the pair; avail = 2q and 2q + 1 for every window
      const bool mainA = 2 * q >= 5, mainB = 2 * q + 1 >= 5;
      // codons that START at x_{i+1}: x = x_{i+1}, w = x_{i+2}, v = x_{i+3}, u = x_{i+4}; the last base is the most significant digit (:1539-1550)
      const int x0 = nuc(iA), x1 = nuc(iA + 1), x2 = nuc(iA + 2), x3 = nuc(iA + 3), x4 = nuc(iA + 4);
      const float *qa2 = p.rsc + (size_t)imin(x2 * 84 + x1 * 21, 337) * p.pitch;
      const float *qa3 = p.rsc + (size_t)imin(x3 * 84 + x2 * 21 + x1 * 5 + 1, 336) * p.pitch;
      const float *qa4 = p.rsc + (size_t)imin(x4 * 84 + x3 * 21 + x2 * 5 + x1 + 2, 337) * p.pitch;
      const float *qb2 = p.rsc + (size_t)imin(x1 * 84 + x0 * 21, 337) * p.pitch;
      const float *qb3 = p.rsc + (size_t)imin(x2 * 84 + x1 * 21 + x0 * 5 + 1, 336) * p.pitch;
      const float *qb4 = p.rsc + (size_t)imin(x3 * 84 + x2 * 21 + x1 * 5 + x0 + 2, 337) * p.pitch;
      // ---- 1. ivx of both rows
      float ivA[C], ivB[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = ll * C + c + 1, ne = imin(node, M);
        const bool in = node <= M;
        const float sa2 = R2[c] + qa2[ne], sa3 = R3[c] + qa3[ne], sa4 = R4[c] + qa4[ne];
        const float sb2 = R1[c] + qb2[ne], sb3 = R2[c] + qb3[ne], sb4 = R3[c] + qb4[ne];
        const float a = mainA ? LS(sa2, LS(sa3, sa4)) : LS(LS(sa2, sa3), sa4);       // :1552-1553; the rows near the end add their codons left to right (:1483, :1517-1518)
        const float b = mainB ? LS(sb2, LS(sb3, sb4)) : LS(LS(sb2, sb3), sb4);
        ivA[c] = in ? a : -INFINITY; ivB[c] = in ? b : -INFINITY;
        s_stage[((size_t)win * 2 + 0) * stride + node] = ivA[c]; s_stage[((size_t)win * 2 + 1) * stride + node] = ivB[c];
      }
      if (WPW > 1 && half < WPW - 1 && lane == 63) { s_xch[wv * 2 + 0] = ivA[0]; s_xch[wv * 2 + 1] = ivB[0]; }   // for lane 0 of the wave below (read after the barriers)
      lds_barrier();
      // ---- 2. the serial part
      if (chain_lane) {
        float *st = s_stage + (size_t)lane * stride;
        const int avail = 2 * q + cs, irow = cL - avail;
        // (the terms are read two nodes ahead of the chain: the slots M+1, M+2 of the row and of the transitions are inside the block's LDS)
#ifdef BATH_CHAIN_CLOCK
        const long long bc0 = clock64();
#endif
        const float b = bwd_bsum_nodes<kCompact>(st[1] + *TBM(1), st[2] + *TBM(2), st[3], *TBM(3), lds_addr(st + 2), lds_addr(TBM(2)),
                                                 M - 1, lds_addr(s_tbl), 15.999f);
#ifdef BATH_CHAIN_CLOCK
        dbgb_cyc += clock64() - bc0; dbgb_n += M;
#endif
        // N, J, C of row i+3: the other slot's row of two pairs ago (slot 0) or of the previous pair (slot 1)
        const float pN1 = __shfl_xor(hN1, 1, 64), pN2 = __shfl_xor(hN2, 1, 64), pJ1 = __shfl_xor(hJ1, 1, 64), pJ2 = __shfl_xor(hJ2, 1, 64);
        const float pC1 = __shfl_xor(hC1, 1, 64), pC2 = __shfl_xor(hC2, 1, 64);
        const float uN = cs ? pN1 : pN2, uJ = cs ? pJ1 : pJ2, uC = cs ? pC1 : pC2;
        float xJ, xN, xC;
        if (avail < 3) { xJ = b + ctNM; xN = b + ctNM; xC = (avail == 0) ? ctNM : ctNL + ctNM; }      // :1442-1465, :1506-1511 (tJM = tNM = tCM, tCL = tNL)
        else { xJ = LS(uJ + ctNL, b + ctNM); xC = uC + ctNL; xN = LS(uN + ctNL, b + ctNM); }           // :1566-1571
        const float xE = LS(xJ + tEL, xC + tEM);
        const bool mid = (avail == 3) || (avail == 4);            // rows L-3, L-4 associate the D chain differently (:1524-1526)
        // D(i,k) = LS(LS(E, D(i,k+1) + tDD), ivx(i,k+1) + tDM); the rows L-3, L-4 pair E with the ivx term first (:1524-1526).
        // Log-sum is symmetric, so both are LS(LS(E, p1), p2) with the operands swapped: no branch in the loop
        BwdChainRegs r{-INFINITY, -INFINITY, st[M], TDD(M)[0], TDD(M)[1], lds_addr(st + M - 1), lds_addr(TDD(M - 1))};   // ivx(i,M), tDD(M), tDM(M)
#ifdef BATH_CHAIN_CLOCK
        const long long dc0 = clock64();
#endif
        bwd_d_nodes<kCompact>(r, xE, __builtin_amdgcn_ballot_w64(mid), M, lds_addr(s_tbl), 15.999f);
#ifdef BATH_CHAIN_CLOCK
        dbgd_cyc += clock64() - dc0;
#endif
        s_e[lane] = xE;
        const float partnerN = __shfl_xor(xN, 1, 64);
        if (clive && irow >= 0) {
          if (cxo) {
            float *r = cxo + (size_t)irow * 5;
            if (irow > 0) { r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = b; r[4] = xC; }
            else { r[0] = -INFINITY; r[1] = xN; r[2] = -INFINITY; r[3] = b; r[4] = -INFINITY; }          // :1660-1664
          }
          if (irow == 0) {                                        // rows 1 and 2: the other slot's row of this pair (slot 1) or of the previous one, and this slot's previous row
            const float n1 = cs ? partnerN : pN1, n2 = hN1;
            sc[cjob] = LS(xN, LS(n1, n2));
          }
        }
        hN2 = hN1; hN1 = xN; hJ2 = hJ1; hJ1 = xJ; hC2 = hC1; hC1 = xC;
      }
      if (wv == 0) chain_done(s_ctl + 4, pair); else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
      lds_barrier();
      // ---- 3. the cells of both rows (:1574-1600)
      const float EA = s_e[win * 2 + 0], EB = s_e[win * 2 + 1];
      float ivNextA = wave_shr1(ivA[0], -INFINITY), ivNextB = wave_shr1(ivB[0], -INFINITY);
      if (WPW > 1 && half > 0 && lane == 0) { ivNextA = s_xch[(wv - 1) * 2 + 0]; ivNextB = s_xch[(wv - 1) * 2 + 1]; }
      float MA[C], IA[C], MB[C], IB[C];
#pragma unroll
      for (int c = C - 1; c >= 0; c--) {
        const int node = ll * C + c + 1, nd = imin(node, M + 1);
        const bool in = node <= M;
        const float4 t0 = *reinterpret_cast<const float4 *>(wt + nd * 8);              // tMD tMI tMM tDD
        const float tii = wt[nd * 8 + 5], tim = wt[nd * 8 + 6];
        const float dnA = (node < M) ? s_stage[((size_t)win * 2 + 0) * stride + node + 1] : -INFINITY;
        const float dnB = (node < M) ? s_stage[((size_t)win * 2 + 1) * stride + node + 1] : -INFINITY;
        const float inA = (c == C - 1) ? ivNextA : ivA[c + 1], inB = (c == C - 1) ? ivNextB : ivB[c + 1];
        float mvA, ivoA, mvB, ivoB;
        // row A: avail = 2q
        if (2 * q < 3) { mvA = LS(dnA + t0.x, LS(inA + t0.z, EA)); ivoA = inA + tim; }
        else if (!mainA) { mvA = LS(dnA + t0.x, LS(J3[c] + t0.y, LS(inA + t0.z, EA))); ivoA = LS(J3[c] + tii, inA + tim); }
        else { mvA = LS(LS(dnA + t0.x, LS(J3[c] + t0.y, inA + t0.z)), EA); ivoA = LS(J3[c] + tii, inA + tim); }
        // row B: avail = 2q + 1
        if (2 * q + 1 < 3) { mvB = LS(dnB + t0.x, LS(inB + t0.z, EB)); ivoB = inB + tim; }
        else if (!mainB) { mvB = LS(dnB + t0.x, LS(J2[c] + t0.y, LS(inB + t0.z, EB))); ivoB = LS(J2[c] + tii, inB + tim); }
        else { mvB = LS(LS(dnB + t0.x, LS(J2[c] + t0.y, inB + t0.z)), EB); ivoB = LS(J2[c] + tii, inB + tim); }
        MA[c] = in ? mvA : -INFINITY; IA[c] = in ? ivoA : -INFINITY; MB[c] = in ? mvB : -INFINITY; IB[c] = in ? ivoB : -INFINITY;
      }
      if (live && iB >= 0) {
#pragma unroll
        for (int c = 0; c < C; c++) { R4[c] = R2[c]; R3[c] = R1[c]; R2[c] = MA[c]; R1[c] = MB[c]; J3[c] = J1[c]; J2[c] = IA[c]; J1[c] = IB[c]; }
      }
    }
#ifdef BATH_CHAIN_CLOCK
    if (wv == 0 && lane == 0 && blockIdx.x == 0 && dbgb_n > 0) printf("bwd chain: B sum %.1f ticks per node, D chain %.1f ticks per node (%lld nodes)\n", (double)dbgb_cyc / dbgb_n, (double)dbgd_cyc / dbgb_n, dbgb_n);
#endif
    if (job >= 0 && !live && lane == 0 && half == 0) sc[job] = -INFINITY;
    __syncthreads();
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same with HALF a wave per window (models of up to 32 x CH nodes), 32 windows per block: every lane of the chain wave
// carries a row (the full-wave kernel's 16 windows fill half of it), so a batch of windows needs half as many CUs for the same
// duration -- what matters when another worker's kernels are waiting for CUs (a chain block takes a CU's registers and most of
// its LDS whatever it does with them).  Lanes own their nodes in descending order inside each half wave.
// ---------------------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(1024) void fs3_bwd_chain_half_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                                  float tEL, float tEM, float *__restrict__ sc, float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, FsJobs jobs,
                                                                  const int32_t *__restrict__ bstart, int nb) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tb = s_tbl + kLogsumTbl;
  const int M = p.M;
  constexpr int W = 32;                                         // window slots per block: two per wave
  constexpr int stride = CH * 32 + 1;
  float *s_stage = s_tb + (M + 2) * 8;                          // [W][2][stride]: ivx(i,k) in, D(i,k) out
  float *s_e = s_stage + (size_t)W * 2 * stride;                // [W][2] E(i) of the pair's rows
  fs_load_logsum_table(s_tbl, p.logsum);
  for (int i = threadIdx.x; i < (M + 2) * 8; i += blockDim.x) s_tb[i] = p.tb[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hl = lane & 31, win = wv * 2 + (lane >> 5);          // lane within its window's half wave; the window's slot in the block
  const int ll = 31 - hl;                                       // nodes in descending order: the lane holding the next nodes is the physical lane below
#define LS(a, b) flogsum<false>((a), (b), s_tbl)
  auto shr1 = [&](float v) { const float r = wave_shr1(v, -INFINITY); return hl == 0 ? -INFINITY : r; };   // the neighbour move stays inside the half wave
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * W);         // [0]: the block's batch; [4]: the row pair whose serial part is done
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
  for (;;) {                                                    // batches dealt longest first, on request (see fs3_fwd_chain_half_kernel)
    if (threadIdx.x == 0) s_ctl[0] = (int)atomicAdd(jobs.counter, 1u);
    __syncthreads();
    const int bb = s_ctl[0];
    if (bb >= nb) break;
    const int64_t base = bstart[bb];
    const int cnt = bstart[bb + 1] - bstart[bb];                // windows of this batch (<= 32)
    const int64_t job = (win < cnt) ? (int64_t)jobs.order[base + win] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];
    const int L = job >= 0 ? dna.len[job] : 0;
    const bool live = job >= 0 && L >= 5;
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    const bool wave_idle = wv != 0 && wv * 2 >= cnt;            // neither of the wave's two slots holds a window: it only keeps the barriers
    // ---- the chain wave's lanes: lane c serves slot (c & 1) of window (c >> 1)
    const int cw = lane >> 1, cs = lane & 1;
    const bool chain_lane = (wv == 0);
    int64_t cjob = -1; int cL = 0; float *cxo = nullptr;
    float ctNL = 0.f, ctNM = 0.f;
    if (chain_lane && cw < cnt) {
      cjob = jobs.order[base + cw]; cL = dna.len[cjob];
      if (xmx) cxo = xmx + xmx_off[cjob];
      ctNL = loop_tab[cL / 3]; ctNM = move_tab[cL / 3];
    }
    const bool clive = cjob >= 0 && cL >= 5;
    float hN1 = -INFINITY, hN2 = -INFINITY, hJ1 = -INFINITY, hJ2 = -INFINITY, hC1 = -INFINITY, hC2 = -INFINITY;
    float R1[CH], R2[CH], R3[CH], R4[CH], J1[CH], J2[CH], J3[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) R1[c] = R2[c] = R3[c] = R4[c] = J1[c] = J2[c] = J3[c] = -INFINITY;
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 338) : 338; };
    const int npairs = Lmax / 2 + 1;
    for (int q = 0; q < npairs; q++) {
      ++pair;
      if (wave_idle) { lds_barrier(); if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }
      const int iA = L - 2 * q, iB = iA - 1;
      const bool mainA = 2 * q >= 5, mainB = 2 * q + 1 >= 5;
      const int x0 = nuc(iA), x1 = nuc(iA + 1), x2 = nuc(iA + 2), x3 = nuc(iA + 3), x4 = nuc(iA + 4);
      const float *qa2 = p.rsc + (size_t)imin(x2 * 84 + x1 * 21, 337) * p.pitch;
      const float *qa3 = p.rsc + (size_t)imin(x3 * 84 + x2 * 21 + x1 * 5 + 1, 336) * p.pitch;
      const float *qa4 = p.rsc + (size_t)imin(x4 * 84 + x3 * 21 + x2 * 5 + x1 + 2, 337) * p.pitch;
      const float *qb2 = p.rsc + (size_t)imin(x1 * 84 + x0 * 21, 337) * p.pitch;
      const float *qb3 = p.rsc + (size_t)imin(x2 * 84 + x1 * 21 + x0 * 5 + 1, 336) * p.pitch;
      const float *qb4 = p.rsc + (size_t)imin(x3 * 84 + x2 * 21 + x1 * 5 + x0 + 2, 337) * p.pitch;
      // ---- 1. ivx of both rows
      float ivA[CH], ivB[CH];
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const int node = ll * CH + c + 1, ne = imin(node, M);
        const bool in = node <= M;
        const float sa2 = R2[c] + qa2[ne], sa3 = R3[c] + qa3[ne], sa4 = R4[c] + qa4[ne];
        const float sb2 = R1[c] + qb2[ne], sb3 = R2[c] + qb3[ne], sb4 = R3[c] + qb4[ne];
        const float a = mainA ? LS(sa2, LS(sa3, sa4)) : LS(LS(sa2, sa3), sa4);
        const float b = mainB ? LS(sb2, LS(sb3, sb4)) : LS(LS(sb2, sb3), sb4);
        ivA[c] = in ? a : -INFINITY; ivB[c] = in ? b : -INFINITY;
        s_stage[((size_t)win * 2 + 0) * stride + node] = ivA[c]; s_stage[((size_t)win * 2 + 1) * stride + node] = ivB[c];
      }
      lds_barrier();
      // ---- 2. the serial part: all 64 lanes of wave 0, a row each
      if (chain_lane) {
        float *st = s_stage + (size_t)lane * stride;
        const int avail = 2 * q + cs, irow = cL - avail;
        const float b = bwd_bsum_nodes(st[1] + s_tb[1 * 8 + 7], st[2] + s_tb[2 * 8 + 7], st[3], s_tb[3 * 8 + 7], lds_addr(st + 2), lds_addr(s_tb + 2 * 8 + 7),
                                       M - 1, lds_addr(s_tbl), 15.999f);
        const float pN1 = __shfl_xor(hN1, 1, 64), pN2 = __shfl_xor(hN2, 1, 64), pJ1 = __shfl_xor(hJ1, 1, 64), pJ2 = __shfl_xor(hJ2, 1, 64);
        const float pC1 = __shfl_xor(hC1, 1, 64), pC2 = __shfl_xor(hC2, 1, 64);
        const float uN = cs ? pN1 : pN2, uJ = cs ? pJ1 : pJ2, uC = cs ? pC1 : pC2;
        float xJ, xN, xC;
        if (avail < 3) { xJ = b + ctNM; xN = b + ctNM; xC = (avail == 0) ? ctNM : ctNL + ctNM; }
        else { xJ = LS(uJ + ctNL, b + ctNM); xC = uC + ctNL; xN = LS(uN + ctNL, b + ctNM); }
        const float xE = LS(xJ + tEL, xC + tEM);
        const bool mid = (avail == 3) || (avail == 4);
        BwdChainRegs r{-INFINITY, -INFINITY, st[M], s_tb[M * 8 + 3], s_tb[M * 8 + 4], lds_addr(st + M - 1), lds_addr(s_tb + (M - 1) * 8 + 3)};
        bwd_d_nodes(r, xE, __builtin_amdgcn_ballot_w64(mid), M, lds_addr(s_tbl), 15.999f);
        s_e[lane] = xE;
        const float partnerN = __shfl_xor(xN, 1, 64);
        if (clive && irow >= 0) {
          if (cxo) {
            float *r = cxo + (size_t)irow * 5;
            if (irow > 0) { r[0] = xE; r[1] = xN; r[2] = xJ; r[3] = b; r[4] = xC; }
            else { r[0] = -INFINITY; r[1] = xN; r[2] = -INFINITY; r[3] = b; r[4] = -INFINITY; }
          }
          if (irow == 0) {
            const float n1 = cs ? partnerN : pN1, n2 = hN1;
            sc[cjob] = LS(xN, LS(n1, n2));
          }
        }
        hN2 = hN1; hN1 = xN; hJ2 = hJ1; hJ1 = xJ; hC2 = hC1; hC1 = xC;
      }
      if (wv == 0) chain_done(s_ctl + 4, pair); else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
      lds_barrier();
      // ---- 3. the cells of both rows (:1574-1600)
      const float EA = s_e[win * 2 + 0], EB = s_e[win * 2 + 1];
      const float ivNextA = shr1(ivA[0]), ivNextB = shr1(ivB[0]);
      float MA[CH], IA[CH], MB[CH], IB[CH];
#pragma unroll
      for (int c = CH - 1; c >= 0; c--) {
        const int node = ll * CH + c + 1, nd = imin(node, M + 1);
        const bool in = node <= M;
        const float4 t0 = *reinterpret_cast<const float4 *>(s_tb + nd * 8);            // tMD tMI tMM tDD
        const float tii = s_tb[nd * 8 + 5], tim = s_tb[nd * 8 + 6];
        const float dnA = (node < M) ? s_stage[((size_t)win * 2 + 0) * stride + node + 1] : -INFINITY;
        const float dnB = (node < M) ? s_stage[((size_t)win * 2 + 1) * stride + node + 1] : -INFINITY;
        const float inA = (c == CH - 1) ? ivNextA : ivA[c + 1], inB = (c == CH - 1) ? ivNextB : ivB[c + 1];
        float mvA, ivoA, mvB, ivoB;
        if (2 * q < 3) { mvA = LS(dnA + t0.x, LS(inA + t0.z, EA)); ivoA = inA + tim; }
        else if (!mainA) { mvA = LS(dnA + t0.x, LS(J3[c] + t0.y, LS(inA + t0.z, EA))); ivoA = LS(J3[c] + tii, inA + tim); }
        else { mvA = LS(LS(dnA + t0.x, LS(J3[c] + t0.y, inA + t0.z)), EA); ivoA = LS(J3[c] + tii, inA + tim); }
        if (2 * q + 1 < 3) { mvB = LS(dnB + t0.x, LS(inB + t0.z, EB)); ivoB = inB + tim; }
        else if (!mainB) { mvB = LS(dnB + t0.x, LS(J2[c] + t0.y, LS(inB + t0.z, EB))); ivoB = LS(J2[c] + tii, inB + tim); }
        else { mvB = LS(LS(dnB + t0.x, LS(J2[c] + t0.y, inB + t0.z)), EB); ivoB = LS(J2[c] + tii, inB + tim); }
        MA[c] = in ? mvA : -INFINITY; IA[c] = in ? ivoA : -INFINITY; MB[c] = in ? mvB : -INFINITY; IB[c] = in ? ivoB : -INFINITY;
      }
      if (live && iB >= 0) {
#pragma unroll
        for (int c = 0; c < CH; c++) { R4[c] = R2[c]; R3[c] = R1[c]; R2[c] = MA[c]; R1[c] = MB[c]; J3[c] = J1[c]; J2[c] = IA[c]; J1[c] = IB[c]; }
      }
    }
    if (job >= 0 && !live && hl == 0) sc[job] = -INFINITY;
    __syncthreads();
  }
#undef LS
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5-codon Forward, full matrix, MULTIHIT (the regions of p7_domaindef.c:396-455), strict.  IVX(i,k) collects the paths leaving
// row i-1 and B(i-1), so rows cannot be paired: one row per step, W windows per block, the D chain and the E sum of all W rows
// in one wave.  fwd[(i*(M+1)+k)*8 + {D,I,C0..C5}], xmx[i*5 + {E,N,J,B,C}]; done[job] = 1 (system scope) once the matrix has landed.
// ---------------------------------------------------------------------------------------------------------------------------
template <int C, int THREADS>
__global__ __launch_bounds__(THREADS) void fs5_fwd_chain_kernel(SeqView dna, FsDev p, const float *__restrict__ loop_tab, const float *__restrict__ move_tab,
                                                             float tEL, float tEM, int c5_compat, float *__restrict__ sc, float *__restrict__ fwd, const int64_t *__restrict__ fwd_off,
                                                             float *__restrict__ xmx, const int64_t *__restrict__ xmx_off, int cfg_len, FsJobs jobs, int *__restrict__ done,
                                                             int W /* windows per block: the block has max(W, kChainAwakeWaves) waves */) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float *s_tbl = reinterpret_cast<float *>(lds);
  float *s_tf = s_tbl + kLogsumTbl;
  const int M = p.M;
  const int stride = fs_chain_stride(C);
  float *s_stage = s_tf + (M + 2) * 8;                          // [W][2][stride]; row 0 of each pair of slots is used
  float *s_e = s_stage + (size_t)W * 2 * stride;
  int *s_ctl = reinterpret_cast<int *>(s_e + 2 * kChainMaxWaves);   // [0]: first job of the block's batch; [4]: the row whose serial part is done
  if (threadIdx.x == 0) s_ctl[4] = 0;
  int pair = 0;
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
    const int64_t job = (wv < W && base + wv < dna.n) ? (int64_t)jobs.order[base + wv] : (int64_t)-1;
    const int Lmax = dna.len[jobs.order[base]];
    const int L = job >= 0 ? dna.len[job] : 0;
    const bool live = job >= 0 && L >= 5;
    const uint8_t *d = job >= 0 ? dna.data + dna.off[job] : dna.data;
    float *fo = job >= 0 ? fwd + fwd_off[job] : fwd;
    float *xo = job >= 0 ? xmx + xmx_off[job] : xmx;
    const int Lc = cfg_len >= 0 ? cfg_len : L / 3;
    const float tNL = loop_tab[Lc], tNM = move_tab[Lc], tJL = tNL, tJM = tNM, tCL = tNL, tCM = tNM;
    float Mr0[C], Mr1[C], Mr2[C], Ir0[C], Ir1[C], Ir2[C], Dr1[C], iv0[C], iv1[C], iv2[C], iv3[C];   // M, I of rows i-1..i-3; D of row i-1; IVX(i-1..i-4)
#pragma unroll
    for (int c = 0; c < C; c++) Mr0[c] = Mr1[c] = Mr2[c] = Ir0[c] = Ir1[c] = Ir2[c] = Dr1[c] = iv0[c] = iv1[c] = iv2[c] = iv3[c] = -INFINITY;
    if (live) {
      for (int k = lane; k <= M; k += 64)
#pragma unroll
        for (int q = 0; q < 8; q++) fo[(size_t)k * 8 + q] = -INFINITY;
      if (lane == 0) { xo[0] = -INFINITY; xo[1] = 0.f; xo[2] = -INFINITY; xo[3] = tNM; xo[4] = -INFINITY; }
    }
    float xN0 = 0.f, xN1 = 0.f, xN2 = 0.f, xJ0 = -INFINITY, xJ1 = -INFINITY, xJ2 = -INFINITY, xC0 = -INFINITY, xC1 = -INFINITY, xC2 = -INFINITY;   // rows i-1, i-2, i-3
    float xBprev = tNM;
    auto nuc = [&](int i) -> int { return (i >= 1 && i <= L) ? ((d[i - 1] < 4) ? (int)d[i - 1] : 1367) : 1367; };
    // the emission scores of a row are loaded during the chain of the row before: nothing in a row waits for global memory
    float E1[C], E2[C], E3[C], E4[C], E5[C];
    auto load_emissions = [&](int i) {
      const int x = nuc(i), w = nuc(i - 1), v = nuc(i - 2), u = nuc(i - 3), t = nuc(i - 4);
      const float *r1 = p.rsc + (size_t)imin(x * 341, 1366) * p.pitch;
      const float *r2 = p.rsc + (size_t)imin(x * 341 + w * 85 + 1, 1365) * p.pitch;
      const float *r3 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + 2, 1364) * p.pitch;
      const float *r4 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + u * 5 + 3, 1365) * p.pitch;
      const float *r5 = p.rsc + (size_t)imin(x * 341 + w * 85 + v * 21 + u * 5 + t + 4, 1366) * p.pitch;
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int ne = imin(lane * C + c + 1, M);
        E1[c] = r1[ne]; E2[c] = r2[ne]; E3[c] = r3[ne]; E4[c] = r4[ne]; E5[c] = r5[ne];
      }
    };
    load_emissions(1);
    for (int i = 1; i <= Lmax; i++) {
      ++pair;
      if (wv >= W) { lds_barrier(); chain_keepalive(s_ctl + 4, pair); lds_barrier(); continue; }   // a poller without a window
      const bool act = live && i <= L;
      const float mIn = wave_shr1(Mr0[C - 1], -INFINITY), iIn = wave_shr1(Ir0[C - 1], -INFINITY), dIn = wave_shr1(Dr1[C - 1], -INFINITY);
      float Mc[C], Ic[C], ivc[C];
      float *row = fo + (size_t)(act ? i : 0) * (M + 1) * 8;
      if (act && lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) row[q] = -INFINITY;
      }
      // The cells of the row, then their stores.  Rows >= 5 (all but four of a window) go through a loop without branches, so that
      // the log-sums of a lane's C nodes -- independent of each other -- are interleaved; the first rows take the general form.
      float q1[C], q2[C], q3[C], q4[C], q5[C];
      auto cells = [&](auto early_tag) {
        constexpr bool EARLY = decltype(early_tag)::value;
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1, nd = imin(node, M + 1);
          const float4 ta = *reinterpret_cast<const float4 *>(s_tf + nd * 8);
          const float4 tb = *reinterpret_cast<const float4 *>(s_tf + nd * 8 + 4);
          const float m1 = (c == 0) ? mIn : Mr0[c - 1], i1 = (c == 0) ? iIn : Ir0[c - 1], d1 = (c == 0) ? dIn : Dr1[c - 1];
          const float e1 = E1[c], e2 = E2[c], e3 = E3[c], e4 = E4[c], e5 = E5[c];
          float ivn = LS(m1 + ta.x, LS(i1 + ta.y, LS(d1 + ta.z, xBprev + ta.w)));  // :332-335
          if (EARLY && i <= 2) ivn = xBprev + ta.w;                                  // rows 1, 2: only B(i-1) enters (:109, :150)
          ivc[c] = ivn;
          const float c1 = ivn + e1;
          const float c2 = (!EARLY || i >= 2) ? iv0[c] + e2 : -INFINITY;
          const float c3 = (!EARLY || i >= 3) ? iv1[c] + e3 : -INFINITY;
          const float c4 = (!EARLY || i >= 4) ? iv2[c] + e4 : -INFINITY;
          const float c5 = !EARLY ? (c5_compat ? ivn : iv3[c]) + e5 : -INFINITY;
          float c0;
          if (!EARLY) c0 = LS(LS(c1, LS(c2, c3)), LS(c4, c5));
          else if (i == 1) c0 = c1;
          else if (i == 2) c0 = LS(c1, c2);
          else c0 = LS(c1, LS(c2, LS(c3, c4)));
          Mc[c] = c0;
          const float ins = LS(Mr2[c] + tb.z, Ir2[c] + tb.w);
          Ic[c] = ((!EARLY || i >= 3) && node < M) ? ins : -INFINITY;
          q1[c] = c1; q2[c] = c2; q3[c] = c3; q4[c] = c4; q5[c] = c5;
          s_stage[((size_t)wv * 2) * stride + node] = c0;           // (the slots past node M exist)
        }
      };
      if (i >= 5) cells(std::false_type{}); else cells(std::true_type{});
      if (act) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const int node = lane * C + c + 1;
          if (node <= M) {
            float *cell = row + (size_t)node * 8;
            cell[1] = Ic[c]; cell[2] = Mc[c]; cell[3] = q1[c]; cell[4] = q2[c]; cell[5] = q3[c]; cell[6] = q4[c]; cell[7] = q5[c];
          }
        }
      }
      load_emissions(i + 1);
      lds_barrier();
      // ---- the serial part, a lane per window: D(i,k), E(i) in the reference's order; rows >= 5 pair M(i,M) and D(i,M) first (:392-394)
      if (wv == 0 && lane < W) {
        float *st = s_stage + (size_t)lane * 2 * stride;
        float dch = -INFINITY, ech = -INFINITY;
        FwdChainRegs r{ech, dch, st[1], s_tf[1 * 8 + 4], s_tf[1 * 8 + 5], lds_addr(st + 1), lds_addr(s_tf + 2 * 8 + 4)};   // tMD(k), tDD(k)
        fwd_chain_nodes(r, M - 1, lds_addr(s_tbl), 15.999f);
        ech = r.e; dch = r.d;
        const float Mn = r.Mk;                                    // M(i,M)
        st[M] = dch;
        ech = (i >= 5) ? LS(LS(Mn, dch), ech) : LS(Mn, LS(dch, ech));
        s_e[lane] = ech;
      }
      if (wv == 0) chain_done(s_ctl + 4, pair); else if (chain_poller(wv)) chain_keepalive(s_ctl + 4, pair);
      lds_barrier();
      float Dc[C];
#pragma unroll
      for (int c = 0; c < C; c++) {
        const int node = lane * C + c + 1, ne = imin(node, M);
        const float dv = s_stage[((size_t)wv * 2) * stride + ne];
        Dc[c] = (node <= M) ? dv : -INFINITY;
        if (act && node <= M) row[(size_t)node * 8] = Dc[c];
      }
      const float xE = s_e[wv];
      float nN, nJ, nC, nB;
      if (i <= 2) { nN = 0.f; nJ = xE + tEL; nC = xE + tEM; nB = tNM; }           // :126-132, :166-167
      else {
        nN = xN2 + tNL; nJ = LS(xJ2 + tJL, xE + tEL); nC = LS(xC2 + tCL, xE + tEM);
        nB = LS(nN + tNM, nJ + tJM);
      }
      if (act) {
        if (lane == 0) { xo[i * 5 + 0] = xE; xo[i * 5 + 1] = nN; xo[i * 5 + 2] = nJ; xo[i * 5 + 3] = nB; xo[i * 5 + 4] = nC; }
        xN2 = xN1; xN1 = xN0; xN0 = nN; xJ2 = xJ1; xJ1 = xJ0; xJ0 = nJ; xC2 = xC1; xC1 = xC0; xC0 = nC;
        xBprev = nB;
#pragma unroll
        for (int c = 0; c < C; c++) {
          Mr2[c] = Mr1[c]; Mr1[c] = Mr0[c]; Mr0[c] = Mc[c];
          Ir2[c] = Ir1[c]; Ir1[c] = Ir0[c]; Ir0[c] = Ic[c];
          Dr1[c] = Dc[c];
          iv3[c] = iv2[c]; iv2[c] = iv1[c]; iv1[c] = iv0[c]; iv0[c] = ivc[c];
        }
        if (i == L) {                                             // this window is complete: score, then the flag the host waits for
          if (lane == 0) sc[job] = LS(xC0, LS(xC1 + tCL, xC2 + tCL)) + tCM;
          if (done) { __threadfence_system(); if (lane == 0) __hip_atomic_store(done + job, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
      }
    }
    if (job >= 0 && !live) {
      if (lane == 0) sc[job] = -INFINITY;
      if (done) { __threadfence_system(); if (lane == 0) __hip_atomic_store(done + job, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    __syncthreads();
  }
#undef LS
}

// Batches of windows for the blocks of a chain kernel.  A row pair of a block with w windows takes t(w) = t0 + w * dt (the chain
// and the chain wave's own parallel part, plus every other window's parallel part) and a block lasts L_longest / 2 pairs: the
// smallest T such that batches of w(L) = (2T / L - t0) / dt windows (at most <wmax>), longest windows first, need no more blocks than
// there are CUs -- the batches of the longest windows are smaller, and all blocks end together.  <bst>: batch b = windows
// bst[b] .. bst[b+1]-1 of the list sorted by decreasing length.  BATH_HIP_FS_BATCH=w: uniform batches of w (A/B runs).
static int chain_batches(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_seqs *dna, double t0, double dt, int wmax, DevBuf &buf, int stage_slot, int *nbat_out, int cu_share = 1) {
  const int64_t n = dna->n;
  std::vector<int32_t> ls;
  fs_order_by_length_desc(dna->h_len.data(), n, nullptr, &ls);
  const int cus = std::max(1, (int)ctx->prop.multiProcessorCount / cu_share);      // <cu_share>: another chain kernel runs beside this one (blocks of the two do not share a CU's LDS)
  std::vector<int32_t> bst;
  auto batches = [&](double T, std::vector<int32_t> *out) -> int64_t {
    int64_t q = 0, nbat = 0;
    if (out) out->clear();
    while (q < n) {
      const double w = (2.0 * T / std::max(ls[(size_t)q], 2) - t0) / dt;
      const int take = w < 1.0 ? 1 : (w > (double)wmax ? wmax : (int)w);
      if (out) out->push_back((int32_t)q);
      q += take; nbat++;
    }
    if (out) out->push_back((int32_t)n);
    return nbat;
  };
  double T_lo = 0.5 * ls[0] * (t0 + dt), T_hi = 0.5 * ls[0] * (t0 + wmax * dt);
  static const int fixed_batch = [] { const char *e = std::getenv("BATH_HIP_FS_BATCH"); return e ? std::atoi(e) : 0; }();
  if (fixed_batch >= 1) {
    const int fb = std::min(fixed_batch, wmax);
    for (int64_t q = 0; q < n; q += fb) bst.push_back((int32_t)q);
    bst.push_back((int32_t)n);
  } else if (batches(T_hi, nullptr) > cus) batches(1e30, &bst);       // more than a round of full blocks: uniform batches of <wmax>
  else {
    for (int it = 0; it < 24; it++) { const double T = 0.5 * (T_lo + T_hi); if (batches(T, nullptr) <= cus) T_hi = T; else T_lo = T; }
    batches(T_hi, &bst);
  }
  *nbat_out = (int)bst.size() - 1;
  BATH_HIP_TRY(ctx, buf.reserve(bst.size() * sizeof(int32_t) + 64));
  return ctx->stage_upload(stage_slot, buf.p, bst.data(), bst.size(), stream);   // through page-locked staging: the host does not wait for what the stream already holds
}

static int chain_waves(bath_hip_ctx *ctx, int64_t n, int M, int C, size_t *shmem_out, int cu_share = 1, int threads = 0, int trans_floats = 0) {
  // as many windows per block as LDS holds next to the table and the transitions ...
  const size_t fixed = (size_t)(kLogsumTbl + (trans_floats ? trans_floats : (M + 2) * 8) + 2 * kChainMaxWaves + 16 + 2 * kChainMaxWaves) * sizeof(float);   // ... E values, control words, the waves' hand-over
  int W = (threads ? threads : chain_threads(C)) / 64;
  while (W > 1 && fixed + (size_t)W * 2 * fs_chain_stride(C) * sizeof(float) > 160 * 1024) W >>= 1;
  // ... but no more than it takes to give every block a CU of its own: the waves of a block go through the parallel part of a
  // row one after the other on the CU's four SIMDs, which is time on top of the chain's, so with few windows (the regions: a
  // couple of hundred) a block holds one or two.  Two blocks on one CU is worse than one twice the size: the chain wave of one
  // then shares its SIMD with the other's parallel part and every dependent log-sum waits for an issue slot (measured: the
  // Backward parser of 2.5 k windows 14.8 ms as 157 blocks of 16, 17.8 ms as 313 blocks of 8).
  // <cu_share>: the launch is to leave the rest of the chip to kernels running beside it (the regions' Forward: the envelope
  // kernels of the single-domain regions run on another stream, and their blocks -- a 64 KB table and the rings -- do not fit
  // into a CU's LDS next to a block of this kernel: spread over every CU it would hold them back until it ends)
  const int64_t slots = std::max<int64_t>(1, (int64_t)ctx->prop.multiProcessorCount / cu_share);
  while (W > 1 && (int64_t)(W / 2) * slots >= n) W >>= 1;
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
    case 20: { constexpr int CC = 20; BODY } break;       \
    default: ctx->set_error("frameshift kernels support models up to 1280 nodes"); return BATH_EINVAL; \
  }

int launch_fs3_fwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM,
                         float *d_sc, float *d_xmx, const int64_t *d_xoff, FsJobs jobs, int cu_share) {
  const int M = om->M;
  size_t shmem = 0;
  const int64_t n = dna->n;
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  // half a wave per window when the model fits 32 lanes x 6 nodes and there are more windows than a wave each would keep resident
  static const bool no_half = [] { const char *e = std::getenv("BATH_HIP_FS_FULLWAVE"); return e && e[0] == '1'; }();
  const int CH = (M + 31) / 32;
  static const bool force_half = [] { const char *e = std::getenv("BATH_HIP_FS_HALFWAVE"); return e && e[0] == '1'; }();     // tests: also for a handful of windows
  if (!no_half && CH <= 6 && (force_half || n > (int64_t)ctx->prop.multiProcessorCount * 8)) {
    const size_t hs = (size_t)(kLogsumTbl + (M + 2) * 8 + 32 * 2 * (CH * 32 + 1) + 64 + 16) * sizeof(float);
    // batches by length (chain_batches); t(w) measured at M = 145 (CH = 5): 18.0 us at 1-2 windows, 19.3 at 8, 19.8 at 16, 21.3 at 32.
    // Bench block (7.6 k windows of 300-1000 nt, the longest 500 within 860-1000): a handful of windows in the first batches,
    // 32 for the bulk: 10.9 -> 9.9 ms
    int nbat = 0;
    DevBuf &b_bst = ctx->scratch[47];
    int stb = chain_batches(ctx, stream, dna, 0.085 * M + 3.5 + 0.4 * CH, 0.021 * CH, 32, b_bst, 1, &nbat);
    if (stb != BATH_OK) return stb;
    const int cus = ctx->prop.multiProcessorCount;
    const int hgrid = std::max(1, std::min(nbat, cus));
#define BATH_HALF(C_)                                                                                                              \
    case C_: BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_fwd_chain_half_kernel<C_>)); \
             hipLaunchKernelGGL((fs3_fwd_chain_half_kernel<C_>), dim3(hgrid), dim3(1024), hs, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, d_sc, d_xmx, d_xoff, jobs, \
                                b_bst.as<int32_t>(), nbat); break;
    switch (CH) { BATH_HALF(1) BATH_HALF(2) BATH_HALF(3) BATH_HALF(4) BATH_HALF(5) BATH_HALF(6) }
#undef BATH_HALF
    BATH_HIP_TRY(ctx, hipGetLastError());
    return BATH_OK;
  }
  // <cu_share> = 2: the Backward parser of the same windows runs beside this launch (fs3_regions before the branch decision): the
  // two kernels' blocks do not fit into one CU's LDS together, so each takes half of the CUs with blocks of twice the windows --
  // with a CU each per kernel, 2 x 164 blocks for the 327 windows of configs[4]'s slice queued on 256 CUs
  const int W = chain_waves(ctx, dna->n, M, Cv, &shmem, cu_share);
  // Long models with more windows than a round and a half of such blocks (configs[4] at 500 Mb and more): the kernel that keeps the
  // rows' history in global memory carries twice the windows per CU (BATH_HIP_FS_FWD_MEM=1 / =0: always, with full blocks / never).
  // M = 1024, windows of 8000 nt, ms: 1308 windows 811 -> 476, 2616 windows 1218 -> 945; configs[4] at 1 Gb 2.61 -> 2.12 s
  static const int mem_env = [] { const char *e = std::getenv("BATH_HIP_FS_FWD_MEM"); return e ? std::atoi(e) : -1; }();
  if ((Cv == 12 || Cv == 16) && mem_env != 0) {
    const int64_t slots = std::max<int64_t>(1, (int64_t)ctx->prop.multiProcessorCount / std::max(1, cu_share));
    const size_t fixed = fs_chain_mem_fixed_lds(M, Cv);
    int Wm = 8;
    while (Wm > 1 && fixed + (size_t)Wm * 2 * fs_chain_stride(Cv) * sizeof(float) > 160 * 1024) Wm >>= 1;
    if (mem_env != 1) while (Wm > 1 && (int64_t)(Wm / 2) * slots >= n) Wm >>= 1;
    if (Wm > W && (mem_env == 1 || 2 * n > 3 * slots * W)) {     // (a pair takes 119 us here against 101: not before the other kernel needs a round and a half)
      const size_t ms = fixed + (size_t)Wm * 2 * fs_chain_stride(Cv) * sizeof(float);
      static const int grid_env = [] { const char *e = std::getenv("BATH_HIP_FS_FWD_MEM_GRID"); return e ? std::atoi(e) : 0; }();   // tests: few blocks, several batches each
      const int mgrid = (int)std::max<int64_t>(1, std::min<int64_t>((n + Wm - 1) / Wm, grid_env > 0 ? (int64_t)grid_env : (int64_t)ctx->prop.multiProcessorCount));
      DevBuf &b_hist = ctx->scratch[58];
      BATH_HIP_TRY(ctx, b_hist.reserve((size_t)mgrid * Wm * 2 * kChainHistRows * fs_chain_hist_pitch(Cv) * sizeof(float)));
#define BATH_MEM(C_)                                                                                                               \
      case C_: BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_fwd_chain_mem_kernel<C_>));                                  \
               hipLaunchKernelGGL((fs3_fwd_chain_mem_kernel<C_>), dim3(mgrid), dim3(64 * std::max(Wm, kChainAwakeWaves)), ms, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, \
                                  d_sc, d_xmx, d_xoff, jobs, Wm, b_hist.as<float>()); break;
      switch (Cv) { BATH_MEM(12) BATH_MEM(16) }
#undef BATH_MEM
      BATH_HIP_TRY(ctx, hipGetLastError());
      return BATH_OK;
    }
  }
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + W - 1) / W, (int64_t)ctx->prop.multiProcessorCount));
  BATH_CHAIN_SWITCH(Cv, {
    BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_fwd_chain_kernel<CC>));
    hipLaunchKernelGGL((fs3_fwd_chain_kernel<CC>), dim3(grid), dim3(64 * std::max(W, kChainAwakeWaves)), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, d_sc, d_xmx, d_xoff, jobs, W);
  })
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_fs3_bwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM,
                         float *d_sc, float *d_xmx, const int64_t *d_xoff, FsJobs jobs, int cu_share, int bst_slot, int stage_slot) {
  // <bst_slot>, <stage_slot>: where the launch keeps its batch starts (device scratch, page-locked staging).  A second launch that
  // may run while the first is still pulling batches (the speculative Backward of the longest windows) brings its own.
  const int M = om->M;
  size_t shmem = 0;
  // half a wave per window when the model fits 32 lanes x 6 nodes and the windows would otherwise take more than half the chip:
  // the same duration on half the CUs (BATH_HIP_FS_BWD_HALFWAVE=1 / =0: always / never, for tests and A/B runs)
  {
    static const int half_env = [] { const char *e = std::getenv("BATH_HIP_FS_BWD_HALFWAVE"); return e ? std::atoi(e) : -1; }();
    const int CH = (M + 31) / 32;
    const int64_t n = dna->n;
    if (half_env != 0 && CH <= 6 && (half_env == 1 || n > (int64_t)ctx->prop.multiProcessorCount * 8)) {
      const size_t hs = (size_t)(kLogsumTbl + (M + 2) * 8 + 32 * 2 * (CH * 32 + 1) + 64 + 16) * sizeof(float);
      int nbat = 0;
      DevBuf &b_bst = ctx->scratch[bst_slot];
      const int stb = chain_batches(ctx, stream, dna, 0.128 * M + 6.6 + 0.5 * CH, 0.03 * CH, 32, b_bst, stage_slot, &nbat, cu_share);
      if (stb != BATH_OK) return stb;
      const int hgrid = std::max(1, std::min(nbat, (int)ctx->prop.multiProcessorCount));
      FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
#define BATH_BHALF(C_)                                                                                                              \
      case C_: BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_bwd_chain_half_kernel<C_>)); \
               hipLaunchKernelGGL((fs3_bwd_chain_half_kernel<C_>), dim3(hgrid), dim3(1024), hs, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, d_sc, d_xmx, d_xoff, jobs, \
                                  b_bst.as<int32_t>(), nbat); break;
      switch (CH) { BATH_BHALF(1) BATH_BHALF(2) BATH_BHALF(3) BATH_BHALF(4) BATH_BHALF(5) BATH_BHALF(6) }
#undef BATH_BHALF
      BATH_HIP_TRY(ctx, hipGetLastError());
      return BATH_OK;
    }
  }
  const int W = chain_waves(ctx, dna->n, M, Cv, &shmem, cu_share, bwd_chain_threads(Cv), chain_compact(Cv) ? (M + 3) * 4 : 0);
  // batches by length (chain_batches); t(w) measured at M = 145 (C = 3): 25.4 us per row pair at one window, 28.5 at 16
  int nbat = 0;
  DevBuf &b_bst = ctx->scratch[bst_slot];                          // (its own buffer: Forward's launch may be running on another stream)
  const int stb = chain_batches(ctx, stream, dna, 0.128 * M + 6.6, 0.067 * Cv, W, b_bst, stage_slot, &nbat, cu_share);
  if (stb != BATH_OK) return stb;
  const int grid = std::max(1, std::min(nbat, (int)ctx->prop.multiProcessorCount));
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  // long models: two waves per window (BATH_HIP_FS_BWD_WPW=1: one, for A/B runs)
  static const int wpw_env = [] { const char *e = std::getenv("BATH_HIP_FS_BWD_WPW"); return e ? std::atoi(e) : 2; }();
  if (chain_compact(Cv) && wpw_env == 2 && W <= 4 && (Cv == 12 || Cv == 16 || Cv == 20)) {
    // (up to four windows per block: eight waves of 256 registers.  Eight windows would be sixteen waves of 128 registers: 86 spilled at
    // 8 nodes per lane, and no faster than one wave per window -- 157 us per row pair either way at M = 1024)
#define BATH_BWD2(C_, T_)                                                                                                          \
    case 2 * C_: BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_bwd_chain_kernel<C_, 2, T_>));                              \
                 hipLaunchKernelGGL((fs3_bwd_chain_kernel<C_, 2, T_>), dim3(grid), dim3(64 * std::max(2 * W, kChainAwakeWaves)), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, \
                                    d_sc, d_xmx, d_xoff, jobs, b_bst.as<int32_t>(), nbat, W); break;
    switch (Cv) { BATH_BWD2(6, 512) BATH_BWD2(8, 512) BATH_BWD2(10, 512) }
#undef BATH_BWD2
    BATH_HIP_TRY(ctx, hipGetLastError());
    return BATH_OK;
  }
  BATH_CHAIN_SWITCH(Cv, {
    BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs3_bwd_chain_kernel<CC>));
    hipLaunchKernelGGL((fs3_bwd_chain_kernel<CC>), dim3(grid), dim3(64 * std::max(W, kChainAwakeWaves)), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, d_sc, d_xmx, d_xoff, jobs,
                       b_bst.as<int32_t>(), nbat, W);
  })
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

int launch_fs5_fwd_chain(bath_hip_ctx *ctx, hipStream_t stream, const bath_hip_fsprofile *om, const bath_hip_seqs *dna, int Cv, float tEL, float tEM, int c5_compat,
                         float *d_sc, float *d_fwd, const int64_t *d_foff, float *d_xmx, const int64_t *d_xoff, int cfg_len, FsJobs jobs, int *d_done) {
  const int M = om->M;
  size_t shmem = 0;
  static const int share = [] { const char *e = std::getenv("BATH_HIP_FS_REGION_CU_SHARE"); return e ? std::max(1, std::atoi(e)) : 1; }();
  const int W = chain_waves(ctx, dna->n, M, Cv, &shmem, share);
  const int64_t n = dna->n;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + W - 1) / W, (int64_t)ctx->prop.multiProcessorCount));
  // A block of one or two regions needs ~75 KB of LDS: two of them, or one and a 74 KB block of the envelope wavefronts running beside
  // this kernel, fit into one CU -- and a chain wave that shares its SIMD with another block's waves waits for issue slots at every
  // dependent log-sum (the kernel then lasts 18 ms instead of 16, in the passes where the dispatcher happens to pair blocks up).  With
  // fewer blocks than CUs every block asks for enough LDS to have its CU to itself (BATH_HIP_FS_REGION_LDS_KB, 0: only what it needs).
  static const int lds_kb = [] { const char *e = std::getenv("BATH_HIP_FS_REGION_LDS_KB"); return e ? std::atoi(e) : 100; }();
  if (grid < ctx->prop.multiProcessorCount && 64 * W <= 256) shmem = std::max(shmem, std::min<size_t>((size_t)lds_kb * 1024, (size_t)160 * 1024));
  FsDev dev{om->M, om->pitch, om->maxcodons, om->d_rsc, om->d_tf, om->d_tb, om->d_logsum};
  // a couple of hundred regions are one or two per block: then the kernel built for 256 threads, whose lanes have registers for
  // the row's cells and the next row's emission scores without spilling (a 1024-thread block leaves a lane 128)
  BATH_CHAIN_SWITCH(Cv, {
    if (64 * W <= 256) {
      BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs5_fwd_chain_kernel<CC, 256>));
      hipLaunchKernelGGL((fs5_fwd_chain_kernel<CC, 256>), dim3(grid), dim3(64 * std::max(W, kChainAwakeWaves)), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, c5_compat, d_sc, d_fwd, d_foff,
                         d_xmx, d_xoff, cfg_len, jobs, d_done, W);
    } else {
      BATH_HIP_TRY(ctx, bath::allow_max_lds((const void *)fs5_fwd_chain_kernel<CC, chain_threads(CC)>));
      hipLaunchKernelGGL((fs5_fwd_chain_kernel<CC, chain_threads(CC)>), dim3(grid), dim3(64 * std::max(W, kChainAwakeWaves)), shmem, stream, dna->view(), dev, om->d_loop[0], om->d_move[0], tEL, tEM, c5_compat, d_sc, d_fwd, d_foff,
                         d_xmx, d_xoff, cfg_len, jobs, d_done, W);
    }
  })
  BATH_HIP_TRY(ctx, hipGetLastError());
  return BATH_OK;
}

}  // namespace bath
