/* stand-in for easel's esl_hmm.h: opaque types only */
#ifndef STUB_ESL_HMM_H
#define STUB_ESL_HMM_H
#include "easel.h"
typedef struct stub_ESL_HMM ESL_HMM;
#endif
