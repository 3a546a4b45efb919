/* multi_gpu_surface.c -- the C surface of the multi-GPU job, used the way a C bathsearch would use it (CPU tier: no GPU call).
 *   1. a hit list with CIGAR strings and traces -> bath_hits_serialize -> bath_hits_deserialize -> bath_hits_serialize again:
 *      the two streams are equal byte for byte, every field survives, the stream is big-endian and self-delimiting;
 *   2. streams back to back in one message are walked with bath_hits_stream_size; a truncated / corrupted stream is refused;
 *   3. bath_tophits_add_serialized merges a remote rank's hits (window indices shifted to the search's own) into a P7_TOPHITS;
 *   4. bath_dist_items / bath_dist_deal / bath_dist_shard_range print their division of a 12-query job for the Python test to
 *      compare with bath_amd.dist (which calls the same functions) and with the rule written out.
 * Built and run by tests/test_multi_gpu_c_cpu.py:  gcc -std=c99 -Iinclude tests/c/multi_gpu_surface.c -Lbath_amd -lbathhip */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bath_hip.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

int main(void)
{
  enum { NH = 5 };
  bath_fs_domain dom[NH];
  bath_domain_trace tr[NH];
  char cigars[256];
  int8_t st[64], c[64];
  int32_t k[64], i[64];
  float pp[64];
  int ncol = 0, pool = 0;
  memset(dom, 0, sizeof dom);
  for (int h = 0; h < NH; h++) {
    bath_fs_domain *d = &dom[h];
    d->window = 1000000007LL * (h + 1); d->strand = h & 1; d->fs_window = h - 1;
    d->ienv = 10 + h; d->jenv = 400 + h; d->iali = 13 + h; d->jali = 390 + h; d->ihmm = 1 + h; d->jhmm = 120 + h;
    d->envsc = 57.25f + h; d->oasc = 100.5f; d->domcorrection = 0.125f * h; d->dombias = 0.0625f; d->bitscore = 82.8f - h; d->pre_score = 83.0f;
    d->lnP = -61.5 - 1e-3 * h; d->reported = 1; d->n_shifted_codons = h; d->n_stops = h / 2; d->pid = 33.3f; d->ali_columns = 3 + h;
    d->cigar_off = pool;
    pool += sprintf(cigars + pool, "%dM1F%dM", 30 + h, 9) + 1;
    tr[h].off = ncol; tr[h].N = 3 + h; tr[h].win_start = 1 + 7 * h; tr[h].orf_start = h & 1 ? 22 : 0; tr[h].frameshift = !(h & 1);
    for (int z = 0; z < tr[h].N; z++, ncol++) { st[ncol] = (int8_t)(1 + z % 3); c[ncol] = (int8_t)(z % 3 == 0 ? 3 + (z & 1) : 0); k[ncol] = 5 + z; i[ncol] = 100 + 3 * z; pp[ncol] = 1.0f / (1 + z); }
  }
  dom[3].cigar_off = -1;                                            /* a hit that came without a CIGAR string */

  /* ---- 1. round trip */
  const int64_t n1 = bath_hits_serialize(dom, NH, cigars, tr, st, k, i, c, pp, NULL, 0);
  CHECK(n1 > 16);
  uint8_t *b1 = malloc((size_t) n1), *b2 = malloc((size_t) n1);
  CHECK(bath_hits_serialize(dom, NH, cigars, tr, st, k, i, c, pp, b1, n1) == n1);
  CHECK(bath_hits_serialize(dom, NH, cigars, tr, st, k, i, c, pp, b1, n1 - 1) == -1);              /* too small a buffer is refused */
  CHECK(b1[0] == 'B' && b1[1] == 'H' && b1[2] == 'I' && b1[3] == 'T' && b1[7] == 1 && b1[15] == NH);   /* network byte order */
  bath_hits *H = NULL;
  CHECK(bath_hits_deserialize(b1, n1, &H) == BATH_OK && bath_hits_count(H) == NH);
  const bath_fs_domain *d2 = bath_hits_domains(H);
  int64_t cbytes = 0;
  const char *cig2 = bath_hits_cigars(H, &cbytes);
  const bath_domain_trace *tr2; const int8_t *st2, *c2; const int32_t *k2, *i2; const float *pp2;
  CHECK(bath_hits_traces(H, &tr2, &st2, &k2, &i2, &c2, &pp2) == BATH_OK);
  for (int h = 0; h < NH; h++) {
    bath_fs_domain a = dom[h], b = d2[h];
    CHECK((a.cigar_off < 0) == (b.cigar_off < 0));
    if (a.cigar_off >= 0) CHECK(strcmp(cigars + a.cigar_off, cig2 + b.cigar_off) == 0);
    a.cigar_off = b.cigar_off = 0;
    CHECK(memcmp(&a, &b, sizeof a) == 0);                            /* every field, bit for bit */
    CHECK(tr2[h].N == tr[h].N && tr2[h].win_start == tr[h].win_start && tr2[h].orf_start == tr[h].orf_start && tr2[h].frameshift == tr[h].frameshift);
    for (int z = 0; z < tr[h].N; z++) {
      const int p = (int) tr[h].off + z, q = (int) tr2[h].off + z;
      CHECK(st2[q] == st[p] && c2[q] == c[p] && k2[q] == k[p] && i2[q] == i[p] && memcmp(&pp2[q], &pp[p], 4) == 0);
    }
  }
  CHECK(bath_hits_serialize(d2, NH, cig2, tr2, st2, k2, i2, c2, pp2, b2, n1) == n1 && memcmp(b1, b2, (size_t) n1) == 0);   /* byte for byte */
  bath_hits_destroy(H);

  /* ---- 2. streams back to back; damaged streams */
  const int64_t n0 = bath_hits_serialize(dom, 2, cigars, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0);   /* without traces */
  uint8_t *msg = malloc((size_t)(n0 + n1));
  CHECK(bath_hits_serialize(dom, 2, cigars, NULL, NULL, NULL, NULL, NULL, NULL, msg, n0) == n0);
  memcpy(msg + n0, b1, (size_t) n1);
  CHECK(bath_hits_stream_size(msg, n0 + n1) == n0 && bath_hits_stream_size(msg + n0, n1) == n1);
  CHECK(bath_hits_deserialize(msg, n0, &H) == BATH_OK && bath_hits_count(H) == 2 && bath_hits_traces(H, NULL, NULL, NULL, NULL, NULL, NULL) == BATH_EINVAL);
  bath_hits_destroy(H);
  CHECK(bath_hits_deserialize(b1, n1 - 3, &H) == BATH_EFORMAT && H == NULL);                        /* truncated */
  b2[18] ^= 0x40;                                                                                    /* a record's size field damaged */
  CHECK(bath_hits_deserialize(b2, n1, &H) == BATH_EFORMAT && bath_hits_stream_size(b2, n1) == -1);
  b2[0] = 'X';
  CHECK(bath_hits_deserialize(b2, n1, &H) == BATH_EFORMAT);
  { /* a header that claims more hits than the bytes behind it could hold is refused before anything is reserved */
    uint8_t hdr[24]; memcpy(hdr, b1, 16); memset(hdr + 16, 0, 8);
    hdr[8] = 0x7f; hdr[15] = 24;
    CHECK(bath_hits_deserialize(hdr, 24, &H) == BATH_EFORMAT && H == NULL);
    memset(hdr + 8, 0, 8); hdr[15] = 24;                                                             /* n = 24 <= nbytes, the old bound */
    CHECK(bath_hits_deserialize(hdr, 24, &H) == BATH_EFORMAT && H == NULL);
  }
  { int64_t lo = 7, hi = 7; bath_dist_shard_range(10, 0, 0, &lo, &hi); CHECK(lo == 0 && hi == 0); } /* world 0: an empty share, no SIGFPE */

  /* ---- 3. a remote rank's hits join the hit list */
  const char *names[4] = { "chr1", "chr2", "chr3", "chr4" };
  const int64_t lens[4] = { 5000, 5000, 5000, 5000 };
  bath_fs_domain rem[2];
  memcpy(rem, dom, sizeof rem);
  rem[0].window = 0; rem[1].window = 1;                              /* the rank numbered its shard's windows from 0 ... */
  const int64_t nr = bath_hits_serialize(rem, 2, cigars, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0);
  uint8_t *br = malloc((size_t) nr);
  bath_hits_serialize(rem, 2, cigars, NULL, NULL, NULL, NULL, NULL, NULL, br, nr);
  bath_tophits *th = bath_tophits_create();
  /* a stream whose windows fall outside the search's sequences (a wrong shift, a mismatched stream) is refused whole */
  CHECK(bath_tophits_add_serialized(th, br, nr, 3, 4, 0, names, NULL, NULL, lens) == BATH_EFORMAT && bath_tophits_count(th) == 0);
  CHECK(bath_tophits_add_serialized(th, br, nr, -1, 4, 0, names, NULL, NULL, lens) == BATH_EFORMAT && bath_tophits_count(th) == 0);
  CHECK(bath_tophits_add_serialized(th, br, nr, 2 /* ... and owns windows [2, 4) of the search */, 4, 0, names, NULL, NULL, lens) == BATH_OK);
  CHECK(bath_tophits_count(th) == 2 && bath_tophits_finalize(th, 20000, 100, 10.0) == BATH_OK && bath_tophits_reported(th) == 2);
  char tbl[4096];
  const int64_t nt = bath_tophits_tabular_targets(th, "query", "-", 134, 1, 1, 0, tbl, sizeof tbl - 1);
  CHECK(nt > 0 && nt < (int64_t) sizeof tbl);
  tbl[nt] = 0;
  CHECK(strstr(tbl, "chr3") && strstr(tbl, "chr4") && !strstr(tbl, "chr1") && strstr(tbl, "30M1F9M"));
  bath_tophits_destroy(th);

  /* ---- 4. the division of a 12-query job (the models of tutorial/tRNA-proteins.bhmm: nodes; 382 windows each) over 8 ranks */
  const int M[12] = { 78, 185, 56, 209, 218, 109, 153, 247, 220, 226, 90, 459 };
  int64_t nwin[12]; double cost[12];
  for (int q = 0; q < 12; q++) { nwin[q] = 382; cost[q] = 382.0 * (M[q] + 150); }
  bath_dist_item items[64];
  const int64_t ni = bath_dist_items(nwin, cost, 12, 8, 3, items, 64);
  CHECK(ni > 12 && ni <= 64);
  double ic[64]; int32_t owner[64];
  for (int x = 0; x < ni; x++) ic[x] = (double)(items[x].hi - items[x].lo) * (M[items[x].query] + 150);
  CHECK(bath_dist_deal(ic, ni, 8, owner) == BATH_OK);
  printf("items %d\n", (int) ni);
  for (int x = 0; x < ni; x++) printf("item %d %lld %lld %d\n", items[x].query, (long long) items[x].lo, (long long) items[x].hi, owner[x]);
  for (int r = 0; r < 3; r++) { int64_t lo, hi; bath_dist_shard_range(10, r, 3, &lo, &hi); printf("shard %d %lld %lld\n", r, (long long) lo, (long long) hi); }
  printf("multi-GPU C surface ok\n");
  free(b1); free(b2); free(msg); free(br);
  return 0;
}
