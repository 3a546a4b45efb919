/* hmmfile.c -- reader for the BATH3/f (and HMMER3/f) ASCII profile format.  ORACLE (test infra only).
 * Follows read_asc30hmm(), p7_hmmfile.c:1342-1697: probabilities are stored as -ln p, '*' = 0;
 * value = expf(-1.0 * atof(tok)) (p7_hmmfile.c:1600,1617,1630); STATS/FRAMESHIFT/CODON tags at
 * p7_hmmfile.c:1497-1543.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "bath_oracle.h"

static float tok2p(const char *tok) { return (*tok == '*') ? 0.0f : expf((float)(-1.0 * atof(tok))); }

static char *next_tok(char **save) { return strtok_r(NULL, " \t\r\n", save); }

int bo_hmmfile_count(const char *path)
{
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  char line[8192]; int n = 0;
  while (fgets(line, sizeof line, fp)) if (!strncmp(line, "//", 2)) n++;
  fclose(fp);
  return n;
}

void bo_hmm_free(bo_hmm *h)
{
  if (!h) return;
  free(h->t); free(h->mat); free(h->ins); free(h->consensus); free(h);
}

int bo_hmmfile_read(const char *path, int index, bo_hmm **ret)
{
  FILE *fp = fopen(path, "r");
  if (!fp) return BO_FAIL;
  size_t cap = 1 << 16;
  char *line = malloc(cap);
  bo_hmm *h = NULL;
  int status = BO_EFORMAT;
  int cur = 0;

  /* skip to the index-th record */
  while (cur < index) {
    if (!fgets(line, (int)cap, fp)) goto DONE;
    if (!strncmp(line, "//", 2)) cur++;
  }
  if (!fgets(line, (int)cap, fp)) goto DONE;
  if (strncmp(line, "BATH3/f", 7) && strncmp(line, "HMMER3/f", 8)) goto DONE;

  h = calloc(1, sizeof *h);
  h->ct = 1; h->fsprob = 0.01f;   /* defaults when tags are absent (HMMER3/f files) */
  for (int z = 0; z < BO_NEVPARAM; z++) h->evparam[z] = -99999.0f;

  /* header */
  int have_hmm = 0;
  while (fgets(line, (int)cap, fp)) {
    char *save, *tag = strtok_r(line, " \t\r\n", &save);
    if (!tag) continue;
    if      (!strcmp(tag, "NAME")) { char *t = next_tok(&save); if (t) strncpy(h->name, t, sizeof h->name - 1); }
    else if (!strcmp(tag, "ACC"))  { char *t = next_tok(&save); if (t) strncpy(h->acc, t, sizeof h->acc - 1); }
    else if (!strcmp(tag, "LENG")) { char *t = next_tok(&save); h->M = t ? atoi(t) : 0; }
    else if (!strcmp(tag, "MAXL")) { char *t = next_tok(&save); h->max_length = t ? atoi(t) : 0; }
    else if (!strcmp(tag, "ALPH")) { char *t = next_tok(&save); if (!t || strcasecmp(t, "amino")) goto DONE; }
    else if (!strcmp(tag, "STATS")) {
      char *t1 = next_tok(&save), *t2 = next_tok(&save), *t3 = next_tok(&save), *t4 = next_tok(&save);
      if (!t1 || !t2 || !t3 || !t4) goto DONE;
      if      (!strcasecmp(t2, "MSV"))     { h->evparam[BO_MMU]  = (float)atof(t3); h->evparam[BO_MLAMBDA] = (float)atof(t4); }
      else if (!strcasecmp(t2, "VITERBI")) { h->evparam[BO_VMU]  = (float)atof(t3); h->evparam[BO_VLAMBDA] = (float)atof(t4); }
      else if (!strcasecmp(t2, "FORWARD")) { h->evparam[BO_FTAU] = (float)atof(t3); h->evparam[BO_FLAMBDA] = (float)atof(t4); }
      else if (!strcasecmp(t2, "FS3"))     { h->evparam[BO_FTAUFS3] = (float)atof(t4); }   /* p7_hmmfile.c:1509: tok4 is tau */
      else if (!strcasecmp(t2, "FS5"))     { h->evparam[BO_FTAUFS5] = (float)atof(t4); }
    }
    else if (!strcmp(tag, "FRAMESHIFT")) { next_tok(&save); char *t = next_tok(&save); if (t) h->fsprob = (float)atof(t); }
    else if (!strcmp(tag, "CODON"))      { next_tok(&save); char *t = next_tok(&save); if (t) h->ct = atoi(t); }
    else if (!strcmp(tag, "HMM"))        { have_hmm = 1; break; }
  }
  if (!have_hmm || h->M <= 0) goto DONE;
  if (!fgets(line, (int)cap, fp)) goto DONE;           /* the "m->m m->i ..." header line */

  int M = h->M;
  h->t   = calloc((size_t)(M + 1) * BO_H_NTRANS, sizeof(float));
  h->mat = calloc((size_t)(M + 1) * BO_K_AMINO, sizeof(float));
  h->ins = calloc((size_t)(M + 1) * BO_K_AMINO, sizeof(float));
  h->consensus = calloc((size_t)M + 2, 1);
  h->consensus[0] = ' ';

  if (!fgets(line, (int)cap, fp)) goto DONE;
  {
    char *save, *tok = strtok_r(line, " \t\r\n", &save);
    if (tok && !strcmp(tok, "COMPO")) {
      for (int x = 0; x < BO_K_AMINO; x++) { tok = next_tok(&save); if (!tok) goto DONE; h->compo[x] = tok2p(tok); }
      if (!fgets(line, (int)cap, fp)) goto DONE;
      tok = strtok_r(line, " \t\r\n", &save);
    }
    /* node 0 insert emissions */
    for (int x = 0; x < BO_K_AMINO; x++) { if (!tok) goto DONE; h->ins[x] = tok2p(tok); tok = next_tok(&save); }
  }
  if (!fgets(line, (int)cap, fp)) goto DONE;
  {
    char *save, *tok = strtok_r(line, " \t\r\n", &save);
    for (int x = 0; x < BO_H_NTRANS; x++) { if (!tok) goto DONE; h->t[x] = tok2p(tok); tok = next_tok(&save); }
  }
  for (int k = 1; k <= M; k++) {
    char *save, *tok;
    if (!fgets(line, (int)cap, fp)) goto DONE;
    tok = strtok_r(line, " \t\r\n", &save);
    if (!tok || atoi(tok) != k) goto DONE;
    for (int x = 0; x < BO_K_AMINO; x++) { tok = next_tok(&save); if (!tok) goto DONE; h->mat[k * BO_K_AMINO + x] = tok2p(tok); }
    tok = next_tok(&save);                       /* MAP  */
    tok = next_tok(&save);                       /* CONS */
    h->consensus[k] = tok ? *tok : '-';
    if (!fgets(line, (int)cap, fp)) goto DONE;
    tok = strtok_r(line, " \t\r\n", &save);
    for (int x = 0; x < BO_K_AMINO; x++) { if (!tok) goto DONE; h->ins[k * BO_K_AMINO + x] = tok2p(tok); tok = next_tok(&save); }
    if (!fgets(line, (int)cap, fp)) goto DONE;
    tok = strtok_r(line, " \t\r\n", &save);
    for (int x = 0; x < BO_H_NTRANS; x++) { if (!tok) goto DONE; h->t[k * BO_H_NTRANS + x] = tok2p(tok); tok = next_tok(&save); }
  }
  if (!fgets(line, (int)cap, fp) || strncmp(line, "//", 2)) goto DONE;
  status = BO_OK;

DONE:
  free(line);
  fclose(fp);
  if (status != BO_OK) { bo_hmm_free(h); h = NULL; }
  *ret = h;
  return status;
}
