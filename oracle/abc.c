/* abc.c -- digital alphabets and genetic codes.  ORACLE (test infrastructure only).
 *
 * The reference takes these from easel (esl_alphabet.c, esl_gencode.c), which is NOT in
 * /root/reference (INSTALL:7-8). They are restated here from easel's published conventions as
 * the reference uses them:
 *   amino  "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~"  K=20 Kp=29  (modelconfig.c:139-141 uses K, Kp-2, Kp-1;
 *                                                          modelconfig.c:406,522 use Kp-2='*', Kp-3='X')
 *   dna    "ACGT-RYMKSWHBVDN*~"             K=4  Kp=18  (hmmer.h:281 p7P_MAXNUC=4)
 *   gcode->basic[16*n1+4*n2+n3]                          (modelconfig.c:364-365)
 * Parity for this file is pinned only end-to-end (tutorial pipeline counters).
 */
#include <string.h>
#include "bath_oracle.h"

const char bo_amino_syms[] = "ACDEFGHIKLMNPQRSTVWY-BJZOUX*~";
const char bo_dna_syms[]   = "ACGT-RYMKSWHBVDN*~";

int bo_amino_digitize(char c)
{
  if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
  const char *p = strchr(bo_amino_syms, c);
  if (!p || !c) return -1;
  return (int)(p - bo_amino_syms);
}

int bo_dna_digitize(char c)
{
  if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');
  if (c == 'U') c = 'T';
  if (c == 'X') c = 'N';     /* easel synonym for DNA */
  const char *p = strchr(bo_dna_syms, c);
  if (!p || !c) return -1;
  return (int)(p - bo_dna_syms);
}

/* easel amino degeneracies: B=ND J=IL Z=QE O=K U=C X=any */
int bo_amino_degen(int x, int y)
{
  if (y < 0 || y >= 20) return 0;
  if (x < 20) return x == y;
  switch (x) {
  case 21: return (y == 11 || y == 2);   /* B: N D */
  case 22: return (y == 7  || y == 9);   /* J: I L */
  case 23: return (y == 13 || y == 3);   /* Z: Q E */
  case 24: return (y == 8);              /* O: K   */
  case 25: return (y == 1);              /* U: C   */
  case 26: return 1;                     /* X      */
  default: return 0;
  }
}

static const char *dna_degen_str(int x)
{
  switch (x) {
  case 0: return "A"; case 1: return "C"; case 2: return "G"; case 3: return "T";
  case 5: return "AG";  case 6: return "CT";  case 7: return "AC";  case 8: return "GT";
  case 9: return "CG";  case 10: return "AT"; case 11: return "ACT"; case 12: return "CGT";
  case 13: return "ACG"; case 14: return "AGT"; case 15: return "ACGT";
  default: return "";
  }
}

int bo_dna_degen(int x, int y)
{
  static const char nt[] = "ACGT";
  if (y < 0 || y > 3) return 0;
  return strchr(dna_degen_str(x), nt[y]) != NULL;
}

uint8_t bo_dna_complement(uint8_t x)
{
  /*                       A  C  G  T  -  R  Y  M  K  S  W   H   B   V   D   N   *   ~ */
  static const uint8_t c[18] = { 3, 2, 1, 0, 4, 6, 5, 8, 7, 9, 10, 14, 13, 12, 11, 15, 16, 17 };
  return x < 18 ? c[x] : x;
}

/* out[1..n] = reverse complement of dsq[1..n]; sentinels at 0 and n+1 (esl_sq_ReverseComplement, bathsearch.c:1086) */
void bo_revcomp(const uint8_t *dsq, int n, uint8_t *out)
{
  out[0] = out[n + 1] = BO_DSQ_SENTINEL;
  for (int i = 1; i <= n; i++) out[i] = bo_dna_complement(dsq[n + 1 - i]);
}

/* NCBI translation tables, in NCBI's TCAG order. */
static const char *ncbi_table(int ct)
{
  switch (ct) {
  case 1:  case 11:
           return "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 2:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG";
  case 3:  return "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 4:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 5:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG";
  case 6:  return "FFLLSSSSYYQQCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 9:  return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
  case 10: return "FFLLSSSSYY**CCCWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 12: return "FFLLSSSSYY**CC*WLLLSPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 13: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG";
  case 14: return "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
  case 16: return "FFLLSSSSYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 21: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG";
  case 22: return "FFLLSS*SYY*LCC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 23: return "FF*LSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  case 24: return "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSSKVVVVAAAADDEEGGGG";
  case 25: return "FFLLSSSSYY**CCGWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
  default: return NULL;
  }
}

/* Start codons of the NCBI translation tables (the "sncbieaa" line of NCBI's gc.prt, 'M' = may initiate), TCAG order like the
 * tables above.  esl_gencode_Set copies them into gcode->is_initiator; bathsearch keeps them only under -M (bathsearch.c:718-719).
 * Easel is not in the reference tree, so its copy of these lines cannot be compared: PARITY UNPINNED (NCBI has added start
 * codons to some tables over the years; these are the lines of the gc.prt generation easel's tables date from). */
static const char *ncbi_starts(int ct)
{
  switch (ct) {
  case 1:  return "---M---------------M---------------M----------------------------";
  case 2:  return "--------------------------------MMMM---------------M------------";
  case 3:  return "----------------------------------MM----------------------------";
  case 4:  return "--MM---------------M------------MMMM---------------M------------";
  case 5:  return "---M----------------------------MMMM---------------M------------";
  case 6:  return "-----------------------------------M----------------------------";
  case 9:  return "-----------------------------------M---------------M------------";
  case 10: return "-----------------------------------M----------------------------";
  case 11: return "---M---------------M------------MMMM---------------M------------";
  case 12: return "-------------------M---------------M----------------------------";
  case 13: return "---M------------------------------MM---------------M------------";
  case 14: return "-----------------------------------M----------------------------";
  case 16: return "-----------------------------------M----------------------------";
  case 21: return "-----------------------------------M---------------M------------";
  case 22: return "-----------------------------------M----------------------------";
  case 23: return "--------------------------------M--M---------------M------------";
  case 24: return "---M---------------M---------------M---------------M------------";
  case 25: return "---M-------------------------------M---------------M------------";
  default: return NULL;
  }
}

int bo_gencode_initiators(int ct, int mode, uint8_t is_init[64])
{
  static const int tcag2acgt[4] = { 3, 1, 0, 2 };
  if (mode == 0) { memset(is_init, 1, 64); return BO_OK; }           /* esl_gencode_SetInitiatorAny: every codon, stops included (a stop still ends the ORF) */
  memset(is_init, 0, 64);
  if (mode == 2) { is_init[16 * 0 + 4 * 3 + 2] = 1; return BO_OK; }  /* esl_gencode_SetInitiatorOnlyAUG */
  const char *st = ncbi_starts(ct);
  if (!st) return BO_EINVAL;
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int c = 0; c < 4; c++)
        is_init[16 * tcag2acgt[a] + 4 * tcag2acgt[b] + tcag2acgt[c]] = (uint8_t)(st[16 * a + 4 * b + c] == 'M');
  return BO_OK;
}

/* esl_gencode_IsInitiator: a canonical codon by the table; a degenerate one only when every codon it stands for is an initiator */
int bo_gencode_is_initiator(const uint8_t is_init[64], const uint8_t *d)
{
  if (d[0] < 4 && d[1] < 4 && d[2] < 4) return is_init[16 * d[0] + 4 * d[1] + d[2]];
  int n = 0;
  for (int x = 0; x < 4; x++) {
    if (!bo_dna_degen(d[0], x)) continue;
    for (int y = 0; y < 4; y++) {
      if (!bo_dna_degen(d[1], y)) continue;
      for (int z = 0; z < 4; z++) {
        if (!bo_dna_degen(d[2], z)) continue;
        if (!is_init[16 * x + 4 * y + z]) return 0;
        n++;
      }
    }
  }
  return n > 0;
}

/* basic[16*n1+4*n2+n3] with n in easel order A,C,G,T (modelconfig.c:364) */
int bo_gencode_basic(int ct, uint8_t basic[64])
{
  static const int tcag2acgt[4] = { 3, 1, 0, 2 };   /* T C A G -> easel codes */
  const char *tbl = ncbi_table(ct);
  if (!tbl) return BO_EINVAL;
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int c = 0; c < 4; c++) {
        char aa = tbl[16 * a + 4 * b + c];
        basic[16 * tcag2acgt[a] + 4 * tcag2acgt[b] + tcag2acgt[c]] = (uint8_t) bo_amino_digitize(aa);
      }
  return BO_OK;
}

/* esl_gencode_GetTranslation (easel esl_gencode.c; used at p7_bg.c:548): canonical codon -> basic[];
 * degenerate codon -> the common amino acid of all its expansions, else X. */
uint8_t bo_gencode_translate(const uint8_t basic[64], const uint8_t *d)
{
  if (d[0] < 4 && d[1] < 4 && d[2] < 4) return basic[16 * d[0] + 4 * d[1] + d[2]];
  int aa = -1;
  for (int x = 0; x < 4; x++) {
    if (!bo_dna_degen(d[0], x)) continue;
    for (int y = 0; y < 4; y++) {
      if (!bo_dna_degen(d[1], y)) continue;
      for (int z = 0; z < 4; z++) {
        if (!bo_dna_degen(d[2], z)) continue;
        int a = basic[16 * x + 4 * y + z];
        if (aa == -1) aa = a;
        else if (aa != a) return 26;   /* X */
      }
    }
  }
  return (uint8_t)(aa == -1 ? 26 : aa);
}
