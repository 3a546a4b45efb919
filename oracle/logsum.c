/* logsum.c -- table-driven log-sum-exp.  ORACLE (test infra only).
 * Follows p7_FLogsumInit / p7_FLogsum, logsum.c:80-111: 16000-entry table of
 * float(log(1+exp(-i/1000.))) computed in double, truncating index, cutoff 15.7 nats.
 */
#include <math.h>
#include "bath_oracle.h"

#define TBL   16000
#define SCALE 1000.f

static float tbl[TBL];
static int   inited = 0;
static int   use_exact = 0;

void bo_flogsum_init(void)
{
  if (inited) return;
  for (int i = 0; i < TBL; i++) tbl[i] = (float) log(1. + exp((double) -i / SCALE));
  inited = 1;
}

void bo_flogsum_set_exact(int exact) { use_exact = exact; }

const float *bo_flogsum_table(void) { bo_flogsum_init(); return tbl; }

float bo_flogsum(float a, float b)
{
  const float max = (a > b) ? a : b;
  const float min = (a > b) ? b : a;
  if (!inited) bo_flogsum_init();
  if (min == -INFINITY || (max - min) >= 15.7f) return max;
  if (use_exact) return (float)(max + log(1.0 + exp(min - max)));     /* logsum.c:109 */
  return max + tbl[(int)((max - min) * SCALE)];
}
