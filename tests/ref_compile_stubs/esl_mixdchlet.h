/* stand-in for easel's esl_mixdchlet.h: opaque types only */
#ifndef STUB_ESL_MIXDCHLET_H
#define STUB_ESL_MIXDCHLET_H
#include "easel.h"
typedef struct stub_ESL_MIXDCHLET ESL_MIXDCHLET;
#endif
