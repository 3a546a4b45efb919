/* stand-in for easel's esl_rand64.h: opaque types only */
#ifndef STUB_ESL_RAND64_H
#define STUB_ESL_RAND64_H
#include "easel.h"
typedef struct stub_ESL_RAND64 ESL_RAND64;
#endif
