/* recip_div_cpu.c -- test infrastructure for DESIGN.md section 9, item 0: the quotient x / y from the reciprocal the
 * host rounded once, inv = RN(1 / y), by residual corrections with fused multiply-adds (Markstein 1990):
 *     q0 = x * inv;  q1 = fma(fma(-y, q0, x), inv, q0);  q2 = fma(fma(-y, q1, x), inv, q1)
 * q0 may be two ulp off, q1 is faithful, and for a faithful q1 and a correctly rounded reciprocal q2 is the correctly
 * rounded quotient -- the IEEE division's result -- unless the significand of y is all ones.  The reference-rounding
 * build does NOT use this yet (it divides); this file counts, for the divisors that build meets (interval widths of the
 * axes, 1000, pi RE, the meteo interval), how often each stage differs from x / y. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static int same(double a, double b) {
  uint64_t ua, ub;
  memcpy(&ua, &a, 8);
  memcpy(&ub, &b, 8);
  return ua == ub || (a != a && b != b);
}

/* counts[0..2]: numerators for which q0 / q1 / q2 is not x / y */
void recip_div_count(const double *x, size_t n, double y, size_t *counts) {
  const double inv = 1.0 / y;
  size_t c0 = 0, c1 = 0, c2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : c0, c1, c2)
  for (size_t i = 0; i < n; i++) {
    const double want = x[i] / y;
    const double q0 = x[i] * inv;
    const double q1 = fma(fma(-y, q0, x[i]), inv, q0);
    const double q2 = fma(fma(-y, q1, x[i]), inv, q1);
    c0 += !same(q0, want);
    c1 += !same(q1, want);
    c2 += !same(q2, want);
  }
  counts[0] = c0;
  counts[1] = c1;
  counts[2] = c2;
}
