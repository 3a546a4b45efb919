/* stand-in for easel's esl_alphabet.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_ALPHABET_H
#define STUB_ESL_ALPHABET_H
#include "easel.h"
typedef struct {        /* the public members (esl_alphabet.h) */
  int      type;
  int      K;
  int      Kp;
  char    *sym;
  ESL_DSQ  inmap[128];
  char   **degen;
  int     *ndegen;
  ESL_DSQ *complement;
} ESL_ALPHABET;
extern int esl_abc_FAvgScVec(const ESL_ALPHABET *a, float *sc);
#endif
