/* carrier_params.c -- extended-precision parameters of pmdemod's carrier recurrence (gcc, quadmath).
 *
 * pmdemod.c:326-335 spins a block down with a SEQUENTIAL complex recurrence in double:
 *     cpstep = cos(cstep) - j sin(cstep);  carrier = 1;  per sample: buf *= carrier; carrier *= cpstep;
 * The rounded cpstep = (c, -s) is not exactly on the unit circle and not exactly at angle cstep, so
 * carrier_i = rho^i * exp(-j i theta) with rho = |(c,s)|, theta = atan2(s, c) -- a drift of up to
 * i * 1e-16 ~ 1e-9 over a 2^23-sample block.  A GPU cannot run the recurrence sequentially, so it
 * evaluates the closed form of what the recurrence actually computes, which needs rho and theta far
 * beyond double precision:
 *     u       = theta / 2pi as a 0.128 fixed-point fraction (u_hi:u_lo), so frac(i*u) is exact integer math
 *     logrho  = log(rho) (~1e-17), accurate to ~1e-18 relative
 * Measured against the true recurrence the closed form agrees to < 1e-11 at i = 2^23 (the remainder is
 * the recurrence's own random rounding walk, ~sqrt(i) * 1e-16).
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include "../../include/isee3_dsp_hip.h"

typedef struct { uint64_t u_hi, u_lo; double logrho; } pmd_carrier_t;

void pmd_carrier_params(double cstep, uint64_t *u_hi, uint64_t *u_lo, double *logrho) {
  double c = cos(cstep), s = sin(cstep);            /* exactly the doubles pmdemod.c:327 forms */
  __float128 qc = c, qs = s;
  __float128 x = qc * qc + qs * qs - 1;             /* rho^2 - 1, exact products, ~1e-34 abs */
  *logrho = (double)(0.5Q * log1pq(x));
  __float128 u = atan2q(qs, qc) / (2 * M_PIq);      /* turns, in (-0.5, 0.5] */
  u -= floorq(u);                                   /* [0, 1) */
  __float128 hi = floorq(ldexpq(u, 64));
  __float128 lo = floorq(ldexpq(ldexpq(u, 64) - hi, 64));
  *u_hi = (uint64_t)hi;
  *u_lo = (uint64_t)lo;
}
