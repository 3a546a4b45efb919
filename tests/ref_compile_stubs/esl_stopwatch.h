/* stand-in for easel's esl_stopwatch.h: opaque types only */
#ifndef STUB_ESL_STOPWATCH_H
#define STUB_ESL_STOPWATCH_H
#include "easel.h"
typedef struct stub_ESL_STOPWATCH ESL_STOPWATCH;
#endif
