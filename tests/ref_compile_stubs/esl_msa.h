/* stand-in for easel's esl_msa.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_MSA_H
#define STUB_ESL_MSA_H
#include "easel.h"
typedef struct stub_ESL_MSA ESL_MSA;
typedef struct stub_ESL_SSI ESL_SSI;                 /* (esl_msa.h brings esl_ssi.h) */
#endif
