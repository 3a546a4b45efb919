/* stand-in for easel's esl_sq.h (test infrastructure, see README): declarations by their published names, nothing more */
#ifndef STUB_ESL_SQ_H
#define STUB_ESL_SQ_H
#include "easel.h"
typedef struct stub_ESL_SQ ESL_SQ;
typedef struct stub_ESL_SQ_BLOCK ESL_SQ_BLOCK;
typedef struct stub_ESL_SQFILE ESL_SQFILE;           /* (named by hmmer.h:1551 without its header) */
#endif
