/* minimal stand-in for easel.h: status codes, constants and macros by their published names */
#ifndef eslEASEL_INCLUDED
#define eslEASEL_INCLUDED
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <sys/types.h>
#define eslOK 0
#define eslFAIL 1
#define eslEOL 2
#define eslEOF 3
#define eslEOD 4
#define eslEMEM 5
#define eslENOTFOUND 6
#define eslEFORMAT 7
#define eslEAMBIGUOUS 8
#define eslEDIVZERO 9
#define eslEINCOMPAT 10
#define eslEINVAL 11
#define eslESYS 12
#define eslECORRUPT 13
#define eslEINCONCEIVABLE 14
#define eslESYNTAX 15
#define eslERANGE 16
#define eslEDUP 17
#define eslENOHALT 18
#define eslENORESULT 19
#define eslERRBUFSIZE 128
#define eslINFINITY INFINITY
#define eslCONST_LOG2 0.69314718055994529
#define eslCONST_LOG2R 1.44269504088896341
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define ESL_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define ESL_MIN(a, b) (((a) < (b)) ? (a) : (b))
extern void esl_exception(int errcode, int use_errno, char *sourcefile, int sourceline, char *format, ...);
extern void esl_fatal(const char *format, ...);
#define ESL_EXCEPTION(code, ...) do { esl_exception(code, FALSE, __FILE__, __LINE__, __VA_ARGS__); return code; } while (0)
#define ESL_XEXCEPTION(code, ...) do { status = code; esl_exception(code, FALSE, __FILE__, __LINE__, __VA_ARGS__); goto ERROR; } while (0)
#define ESL_ALLOC(p, size) do { if (((p) = malloc(size)) == NULL) { status = eslEMEM; goto ERROR; } } while (0)
#define ESL_RALLOC(p, tmp, newsize) do { (tmp) = realloc((p), (newsize)); if ((tmp) != NULL) (p) = (tmp); else { status = eslEMEM; goto ERROR; } } while (0)
typedef uint8_t ESL_DSQ;
typedef int64_t esl_pos_t;
#define eslDSQ_SENTINEL 255
#endif
