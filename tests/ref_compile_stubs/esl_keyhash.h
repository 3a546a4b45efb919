/* stand-in for easel's esl_keyhash.h: opaque types only */
#ifndef STUB_ESL_KEYHASH_H
#define STUB_ESL_KEYHASH_H
#include "easel.h"
typedef struct stub_ESL_KEYHASH ESL_KEYHASH;
#endif
