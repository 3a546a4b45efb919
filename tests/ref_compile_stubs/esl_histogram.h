/* stand-in for easel's esl_histogram.h: opaque types only */
#ifndef STUB_ESL_HISTOGRAM_H
#define STUB_ESL_HISTOGRAM_H
#include "easel.h"
typedef struct stub_ESL_HISTOGRAM ESL_HISTOGRAM;
#endif
