/* libm_cpu.c -- test infrastructure: the arithmetic of mptrac_amd/csrc/mphip_libm.h (the device's exp / log / pow)
 * compiled for the CPU, compared bit by bit with the running C library's exp / log / pow.  Built by
 * tests/test_libm_bits.py as a shared object (gcc -O2 -ffp-contract=off [-mfma]); the device runs the same header
 * under hipcc and is compared with the library through the oracle's orc_libm_* in the GPU suite. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "mphip_libm.h"
#include "mphip_libmtab.h"

static const mphip_libm_tabs tabs = { mphip_libm_exp_tab, mphip_libm_log_tab, mphip_libm_pow_tab };

void rst_exp(const double *x, size_t n, double *out) {
  for (size_t i = 0; i < n; i++)
    out[i] = mphip_libm_exp(tabs.exp_tab, x[i]);
}

void rst_log(const double *x, size_t n, double *out) {
  for (size_t i = 0; i < n; i++)
    out[i] = mphip_libm_log(tabs.log_tab, x[i]);
}

void rst_pow(const double *x, const double *y, size_t n, double *out) {
  for (size_t i = 0; i < n; i++)
    out[i] = mphip_libm_pow(&tabs, x[i], y[i]);
}

static int same(double a, double b) {
  uint64_t ua, ub;
  memcpy(&ua, &a, 8);
  memcpy(&ub, &b, 8);
  return ua == ub || (a != a && b != b);     /* any NaN equals any NaN */
}

/* number of arguments whose restated value differs from the library's; first_bad = index of the first one */
size_t cmp_exp(const double *x, size_t n, size_t *first_bad) {
  size_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t i = 0; i < n; i++)
    if (!same(mphip_libm_exp(tabs.exp_tab, x[i]), exp(x[i]))) {
      bad++;
#pragma omp critical
      if (i < *first_bad)
        *first_bad = i;
    }
  return bad;
}

size_t cmp_log(const double *x, size_t n, size_t *first_bad) {
  size_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t i = 0; i < n; i++)
    if (!same(mphip_libm_log(tabs.log_tab, x[i]), log(x[i]))) {
      bad++;
#pragma omp critical
      if (i < *first_bad)
        *first_bad = i;
    }
  return bad;
}

size_t cmp_pow(const double *x, const double *y, size_t n, size_t *first_bad) {
  size_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t i = 0; i < n; i++)
    if (!same(mphip_libm_pow(&tabs, x[i], y[i]), pow(x[i], y[i]))) {
      bad++;
#pragma omp critical
      if (i < *first_bad)
        *first_bad = i;
    }
  return bad;
}

/* sin / cos inside the restated range |x| < 2.426265 (arguments outside are skipped by the caller's sets) */
size_t cmp_cos(const double *x, size_t n, size_t *first_bad) {
  size_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t i = 0; i < n; i++) {
    int handled;
    const double v = mphip_libm_cos(mphip_libm_sincos_tab, x[i], &handled);
    if (!handled || !same(v, cos(x[i]))) {
      bad++;
#pragma omp critical
      if (i < *first_bad)
        *first_bad = i;
    }
  }
  return bad;
}

size_t cmp_sin(const double *x, size_t n, size_t *first_bad) {
  size_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t i = 0; i < n; i++) {
    int handled;
    const double v = mphip_libm_sin(mphip_libm_sincos_tab, x[i], &handled);
    if (!handled || !same(v, sin(x[i]))) {
      bad++;
#pragma omp critical
      if (i < *first_bad)
        *first_bad = i;
    }
  }
  return bad;
}

/* the table header against the constants the algorithms carry as literals */
int check_constants(void) {
  const double k[8] = { MPHIP_EXP_INVLN2N, MPHIP_EXP_SHIFT, MPHIP_EXP_NEGLN2HIN, MPHIP_EXP_NEGLN2LON,
                        MPHIP_EXP_C2, MPHIP_EXP_C3, MPHIP_EXP_C4, MPHIP_EXP_C5 };
  for (int i = 0; i < 8; i++)
    if (k[i] != mphip_libm_exp_k[i])
      return 1 + i;
  if (mphip_libm_ln2[0] != MPHIP_LN2HI || mphip_libm_ln2[1] != MPHIP_LN2LO)
    return 20;
  return 0;
}
