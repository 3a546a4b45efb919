/* stats.c -- tail probabilities used by the filter thresholds.  ORACLE (test infra only).
 * Restated from easel's esl_gumbel.c / esl_exponential.c closed forms (easel is absent from the
 * reference tree); call sites: p7_pipeline.c:1651,1661,1673,1737,1782; msvfilter.c:302; vitfilter.c:314.
 */
#include <math.h>
#include "bath_oracle.h"

#define SMALLX1 5e-9   /* eslSMALLX1 */

double bo_gumbel_surv(double x, double mu, double lambda)
{
  double y  = lambda * (x - mu);
  double ey = -exp(-y);
  if (fabs(ey) < SMALLX1) return -ey;       /* 1 - e^x ~ -x for small |x| */
  return 1.0 - exp(ey);
}

double bo_gumbel_invsurv(double p, double mu, double lambda)
{
  double log_part;
  if (p < SMALLX1) log_part = (pow(p, p) - 1) / p;
  else             log_part = log(-1. * log(1 - p));
  return mu - (log_part / lambda);
}

double bo_exp_surv(double x, double mu, double lambda)
{
  if (x < mu) return 1.0;
  return exp(-lambda * (x - mu));
}

double bo_exp_logsurv(double x, double mu, double lambda)   /* easel esl_exp_logsurv */
{
  if (x < mu) return 0.0;
  return -lambda * (x - mu);
}
