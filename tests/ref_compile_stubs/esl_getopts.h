/* stand-in for easel's esl_getopts.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_GETOPTS_H
#define STUB_ESL_GETOPTS_H
#include "easel.h"
typedef struct stub_ESL_GETOPTS ESL_GETOPTS;
typedef struct stub_ESL_OPTIONS ESL_OPTIONS;
#endif
