/* stand-in for the generated p7_config.h (src/p7_config.h.in): the constants hmmer.h needs, SSE selected so that hmmer.h:1048 includes "impl_sse/impl_sse.h" (redirected to impl_hip.h) */
#ifndef P7_CONFIGH_INCLUDED
#define P7_CONFIGH_INCLUDED
#define eslENABLE_SSE 1
#define p7_RAMLIMIT 32
#define p7_NCPU "2"
#define p7_ETARGET_AMINO 0.43
#define p7_ETARGET_DNA 0.62
#define p7_ETARGET_OTHER 1.0
#define p7_SEQDBENV "BLASTDB"
#define p7_HMMDBENV "PFAMDB"
#define p7_MAX_RESIDUE_COUNT (1024 * 256)
#define p7_MAXABET 20
#define p7_MAXCODE 29
#define p7_MAX_SC_TXTLEN 11
#define p7_MAXDCHLET 20
#define p7_SEQDBENV "BLASTDB"
#define HMMER_VERSION "stub"
#define HMMER_DATE "stub"
#define HMMER_COPYRIGHT "stub"
#define HMMER_LICENSE "stub"
#define HMMER_URL "stub"
#define BATH_VERSION "stub"
#define BATH_DATE "stub"
#define BATH_COPYRIGHT "stub"
#define BATH_LICENSE "stub"
#define BATH_URL "stub"
#endif
