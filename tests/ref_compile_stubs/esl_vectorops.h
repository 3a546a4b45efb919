/* stand-in for easel's esl_vectorops.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_VECTOROPS_H
#define STUB_ESL_VECTOROPS_H
#include "easel.h"
extern void esl_vec_FNorm(float *vec, int n);
extern void esl_vec_FLogNorm(float *vec, int n);
#endif
