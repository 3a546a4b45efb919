/* stand-in for easel's esl_scorematrix.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_SCOREMATRIX_H
#define STUB_ESL_SCOREMATRIX_H
#include "easel.h"
typedef struct stub_ESL_SCOREMATRIX ESL_SCOREMATRIX;
typedef struct stub_ESL_FILEPARSER ESL_FILEPARSER;   /* (esl_scorematrix.h brings esl_fileparser.h) */
#endif
