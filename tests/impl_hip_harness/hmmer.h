/* TEST HARNESS ONLY -- not part of the product and not a build of the reference.
 *
 * impl_hip/ is written against the reference's own hmmer.h and easel headers, which this repository does not contain (easel is
 * an un-vendored dependency of the reference).  To compile and exercise impl_hip/ without them, this file declares just the
 * generic types, constants and helper prototypes that impl_hip/ *.c name -- the fields impl_hip reads, in declarations of our
 * own -- and harness.c supplies small definitions of the helpers (trace append, window list, random numbers, table log-sum).
 * tests/test_impl_hip_cpu.py compiles impl_hip/ against it; tests/test_impl_hip_gpu.py calls the shims through it on the GPU
 * and compares with the batched C ABI.  Nothing here is used by the library.
 */
#ifndef IMPL_HIP_TEST_HMMER_H
#define IMPL_HIP_TEST_HMMER_H

#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <sys/types.h>

/* ---- the bits of easel impl_hip names */
#define eslOK 0
#define eslFAIL 1
#define eslEMEM 5
#define eslEINVAL 11
#define eslERANGE 16
#define eslENORESULT 19
#define eslINFINITY INFINITY
#define eslCONST_LOG2 0.69314718055994529
#define TRUE 1
#define FALSE 0
#define ESL_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define ESL_MIN(a, b) (((a) < (b)) ? (a) : (b))
extern void esl_fatal(const char *fmt, ...);
extern void hs_exception(int code, const char *file, int line, const char *fmt, ...);
#define ESL_EXCEPTION(code, ...) do { hs_exception(code, __FILE__, __LINE__, __VA_ARGS__); return code; } while (0)
typedef uint8_t ESL_DSQ;
typedef struct { int type, K, Kp; } ESL_ALPHABET;
typedef struct { uint32_t x; } ESL_RANDOMNESS;
extern double esl_random(ESL_RANDOMNESS *r);
extern int    esl_rnd_FChoose(ESL_RANDOMNESS *r, const float *p, int N);
extern void   esl_vec_FNorm(float *v, int n);
extern void   esl_vec_FLogNorm(float *v, int n);
extern int    esl_abc_FAvgScVec(const ESL_ALPHABET *abc, float *sc);

/* ---- the bits of hmmer.h impl_hip names */
#define p7_NEVPARAM 8
#define p7_NCUTOFFS 6
#define p7_NOFFSETS 3
#define p7_MAXABET  20
#define p7_MAXCODE  29
#define p7_EVPARAM_UNSET -99999.0f
#define p7_CUTOFF_UNSET  -99999.0f
#define p7_COMPO_UNSET   -1.0f
enum { p7_NO_MODE = 0, p7_LOCAL = 1, p7_GLOCAL = 2, p7_UNILOCAL = 3, p7_UNIGLOCAL = 4 };
enum { p7_NOCOMPLEMENT = 0, p7_COMPLEMENT = 1 };
#define p7P_NTRANS 8
enum { p7P_MM = 0, p7P_IM = 1, p7P_DM = 2, p7P_BM = 3, p7P_MD = 4, p7P_DD = 5, p7P_MI = 6, p7P_II = 7 };
#define p7P_NXSTATES 4
#define p7P_NXTRANS 2
enum { p7P_E = 0, p7P_N = 1, p7P_J = 2, p7P_C = 3 };
enum { p7P_LOOP = 0, p7P_MOVE = 1 };
#define p7P_NR 2
#define p7P_MAXCODONS1 65
#define p7P_MAXCODONS3 338
#define p7P_MAXCODONS5 1367
enum { p7G_M = 0, p7G_I = 1, p7G_D = 2 };
#define p7G_NSCELLS 3
enum { p7G_E = 0, p7G_N = 1, p7G_J = 2, p7G_B = 3, p7G_C = 4 };
#define p7G_NXCELLS 5
enum p7t_statetype_e { p7T_BOGUS = 0, p7T_M = 1, p7T_D = 2, p7T_I = 3, p7T_S = 4, p7T_N = 5, p7T_B = 6, p7T_E = 7, p7T_C = 8, p7T_T = 9, p7T_J = 10, p7T_X = 11 };

typedef struct p7_profile_s {
  float  *tsc; float **rsc; float xsc[p7P_NXSTATES][p7P_NXTRANS];
  int mode, L, allocM, M, max_length; float nj;
  char *name, *acc, *desc, *rf, *mm, *cs, *consensus;
  float evparam[p7_NEVPARAM], cutoff[p7_NCUTOFFS], compo[p7_MAXABET];
  const ESL_ALPHABET *abc;
} P7_PROFILE;
typedef struct p7_fs_profile_s {
  float  *tsc; float **rsc; float xsc[p7P_NXSTATES][p7P_NXTRANS];
  int mode, codon_lengths, L, allocM, M, max_length; float nj; int fs; float fsprob;
  char *name, *acc, *desc, *rf, *mm, *cs, *consensus;
  float evparam[p7_NEVPARAM], cutoff[p7_NCUTOFFS], compo[p7_MAXABET];
  ESL_DSQ **codons, **indel_pos;
  const ESL_ALPHABET *abc;
} P7_FS_PROFILE;
typedef struct { int dummy; } P7_BG;
extern int p7_bg_SetLength(P7_BG *bg, int L);
typedef struct { int M, L; float **dp; float *xmx; } P7_GMX;
typedef struct { int N, nalloc; char *st; int *k, *i, *c; float *pp; int M, L; } P7_TRACE;
extern int p7_trace_Append(P7_TRACE *tr, char st, int k, int i);
extern int p7_trace_AppendWithPP(P7_TRACE *tr, char st, int k, int i, float pp);
extern int p7_trace_fs_Append(P7_TRACE *tr, char st, int k, int i, int c);
extern int p7_trace_fs_AppendWithPP(P7_TRACE *tr, char st, int k, int i, int c, float pp);
extern int p7_trace_Reverse(P7_TRACE *tr);
extern int p7_trace_fs_Reverse(P7_TRACE *tr);
typedef struct { float *mocc, *btot, *etot; int L, Lalloc; float *n2sc; } P7_DOMAINDEF;
typedef struct { int type, M; uint8_t *ssv_scores; } P7_SCOREDATA;
typedef struct { float score; int64_t id, n; int32_t length, k; int64_t target_len; int8_t complementarity; } P7_HMM_WINDOW;
typedef struct { P7_HMM_WINDOW *windows; int count, size; } P7_HMM_WINDOWLIST;
extern P7_HMM_WINDOW *p7_hmmwindow_new(P7_HMM_WINDOWLIST *list, uint32_t id, uint32_t pos, uint32_t k, uint32_t length, float score, uint8_t complementarity, uint32_t target_len);
extern float p7_FLogsum(float a, float b);

#include "impl_hip.h"          /* where the reference's hmmer.h:1044-1052 includes the implementation header */
#endif
