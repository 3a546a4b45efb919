/* stand-in for easel's esl_random.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_RANDOM_H
#define STUB_ESL_RANDOM_H
#include "easel.h"
typedef struct stub_ESL_RANDOMNESS ESL_RANDOMNESS;
extern double esl_random(ESL_RANDOMNESS *r);
extern int    esl_rnd_FChoose(ESL_RANDOMNESS *r, const float *p, int N);
#endif
