/* stand-in for easel's esl_dmatrix.h: opaque types only */
#ifndef STUB_ESL_DMATRIX_H
#define STUB_ESL_DMATRIX_H
#include "easel.h"
typedef struct stub_ESL_DMATRIX ESL_DMATRIX;
#endif
