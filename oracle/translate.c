/* translate.c -- six-frame ORF finder.  ORACLE (test infra only).
 *
 * Restates what bathsearch gets from easel's esl_gencode_ProcessStart/Piece/End (bathsearch.c:384-392;
 * easel is absent from the reference tree, so this follows the published esl-translate algorithm):
 * per reading frame, an ORF is a maximal run of non-stop codons ("any codon may start an ORF",
 * esl_gencode_SetInitiatorAny, bathsearch.c:718-719); runs shorter than <minlen> aa (-l 20,
 * bathsearch.c:104) are dropped; a codon holding a degenerate nucleotide translates to the amino
 * acid all its expansions agree on, else X, and does not end the run.  ORFs are emitted in the
 * order they close (position of the terminating stop, then the three still-open frames at the end).
 * Coordinates are 1-based nt positions on the strand passed in; the caller reverse-complements for
 * the bottom strand exactly as bathsearch.c:1086 does.
 * Parity: pinned only through the tutorial pipeline counters (tests/golden).
 */
#include <stdlib.h>
#include <string.h>
#include "bath_oracle.h"

void bo_orfblock_init(bo_orfblock *b) { memset(b, 0, sizeof *b); }
void bo_orfblock_reuse(bo_orfblock *b) { b->count = 0; b->aa_n = 0; }
void bo_orfblock_free(bo_orfblock *b) { free(b->orf); free(b->aa); memset(b, 0, sizeof *b); }

static void emit(bo_orfblock *b, const uint8_t *buf, int n, int start, int end, int frame)
{
  if (b->count == b->size) { b->size = b->size ? b->size * 2 : 64; b->orf = realloc(b->orf, sizeof(bo_orf) * (size_t) b->size); }
  if (b->aa_n + n + 2 > b->aa_size) {
    b->aa_size = (b->aa_n + n + 2) * 2 + 1024;
    b->aa = realloc(b->aa, (size_t) b->aa_size);
  }
  bo_orf *o = &b->orf[b->count++];
  o->start = start; o->end = end; o->n = n; o->frame = frame; o->off = b->aa_n;
  b->aa[b->aa_n] = BO_DSQ_SENTINEL;
  memcpy(b->aa + b->aa_n + 1, buf, (size_t) n);
  b->aa[b->aa_n + n + 1] = BO_DSQ_SENTINEL;
  b->aa_n += n + 2;
}

int bo_translate_orfs(const uint8_t *dsq, int n, const uint8_t basic[64], int minlen, bo_orfblock *out)
{
  return bo_translate_orfs_init(dsq, n, basic, NULL, 0, minlen, out);
}

/* With initiation codons (esl_gencode_ProcessPiece, restated from the published esl-translate behaviour): a stop closes the
 * frame's ORF whether or not one is open; any other codon extends an open ORF; with no ORF open it starts one only if it is
 * an initiator, and is then translated as M when -m / -M is in force (using_initiators).  minlen counts that M. */
int bo_translate_orfs_init(const uint8_t *dsq, int n, const uint8_t basic[64], const uint8_t *is_init, int using_initiators,
                           int minlen, bo_orfblock *out)
{
  uint8_t *buf[3];
  int len[3] = { 0, 0, 0 }, start[3] = { 0, 0, 0 };
  for (int f = 0; f < 3; f++) buf[f] = malloc((size_t) n / 3 + 2);
  for (int p = 1; p + 2 <= n; p++) {
    int f = (p - 1) % 3;
    uint8_t aa = bo_gencode_translate(basic, dsq + p);
    if (aa == BO_KP_AMINO - 2) {                 /* stop: close this frame's run */
      if (len[f] >= minlen) emit(out, buf[f], len[f], start[f], p - 1, f);
      len[f] = 0;
    } else {
      if (len[f] == 0) {
        if (is_init && !bo_gencode_is_initiator(is_init, dsq + p)) continue;
        start[f] = p;
        if (is_init && using_initiators) aa = 10;   /* 'M' in "ACDEFGHIKLMNPQRSTVWY" */
      }
      buf[f][len[f]++] = aa;
    }
  }
  for (int f = 0; f < 3; f++) {
    if (len[f] >= minlen) emit(out, buf[f], len[f], start[f], start[f] + 3 * len[f] - 1, f);
    free(buf[f]);
  }
  return BO_OK;
}
