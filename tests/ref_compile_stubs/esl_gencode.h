/* stand-in for easel's esl_gencode.h: opaque types only */
#ifndef STUB_ESL_GENCODE_H
#define STUB_ESL_GENCODE_H
#include "easel.h"
typedef struct stub_ESL_GENCODE ESL_GENCODE;
typedef struct stub_ESL_GENCODE_WORKSTATE ESL_GENCODE_WORKSTATE;
#endif
