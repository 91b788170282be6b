/* mphip_libm.h -- double-precision exp, log and pow with the bits of the C library the reference's CPU build
 * links (glibc >= 2.28 on x86-64 with FMA: the `__exp_fma` / `__log_fma` / `__pow_fma` variants its ifunc resolvers
 * select on every CPU with FMA + AVX2, i.e. on the hosts of this image and of the GPU boxes).
 *
 * The algorithms are the published ARM optimized-routines ones (math/exp.c, log.c, pow.c; S. Nagy 2018), restated
 * here: 128-entry table reductions and short polynomials.  What decides the LAST bit is which multiply-add pairs
 * of the C source the library's compiler contracted into fused operations; that was read off the machine code of
 * the three variants in libm.so.6 of glibc 2.35 (the operation lists in the comments below), and every fused
 * operation here is an explicit fma, every other one is kept unfused (contraction is switched off for this file's
 * functions).  Tables: mphip_libmtab.h (tools/gen_libm_tables.py).
 *
 * Plain C99 / C++: the device code includes it under hipcc (functions become __device__), and tests/c/libm_cpu.c
 * includes it under gcc to check the same arithmetic against the running libm on the CPU -- the functions take
 * their tables as pointers (LDS on the device).
 *
 * Domain: every finite or infinite argument returns the library's value (errno / exception flags are not modelled):
 * exp over- and underflow incl. the subnormal results, log of zero / negatives / subnormals, pow of negative bases with
 * integer exponents, zeros, infinities, NaNs.
 */
#ifndef MPHIP_LIBM_H
#define MPHIP_LIBM_H

#include <stdint.h>

#ifdef __HIPCC__
#define MPHIP_LIBM_FN __device__ __forceinline__
#else
#define MPHIP_LIBM_FN static inline
#endif

/* the tables a call site hands over (global memory, or a copy in LDS) */
typedef struct {
  const uint64_t *exp_tab;     /* 2 x 128: {tail bits, scale bits} */
  const double *log_tab;       /* 2 x 128: {invc, logc} */
  const double *pow_tab;       /* 3 x 128: {invc, logc, logctail} */
} mphip_libm_tabs;

MPHIP_LIBM_FN uint64_t mphip_libm_bits(double x) {
#ifdef __HIPCC__
  return (uint64_t) __double_as_longlong(x);
#else
  union { double d; uint64_t u; } c;
  c.d = x;
  return c.u;
#endif
}

MPHIP_LIBM_FN double mphip_libm_from_bits(uint64_t u) {
#ifdef __HIPCC__
  return __longlong_as_double((long long) u);
#else
  union { double d; uint64_t u; } c;
  c.u = u;
  return c.d;
#endif
}

MPHIP_LIBM_FN double mphip_libm_from_words(uint32_t hi, uint32_t lo) {
  return mphip_libm_from_bits(((uint64_t) hi << 32) | lo);
}

#define MPHIP_LIBM_FMA(a, b, c) __builtin_fma((a), (b), (c))

/* constants (mphip_libmtab.h holds the same values for the tables' generator check; literals here so that the
 * device compiler can place them in scalar registers / instruction literals) */
#define MPHIP_EXP_INVLN2N 0x1.71547652b82fep+7
#define MPHIP_EXP_SHIFT 0x1.8p+52
#define MPHIP_EXP_NEGLN2HIN (-0x1.62e42fefa0000p-8)
#define MPHIP_EXP_NEGLN2LON (-0x1.cf79abc9e3b3ap-47)
#define MPHIP_EXP_C2 0x1.ffffffffffdbdp-2
#define MPHIP_EXP_C3 0x1.555555555543cp-3
#define MPHIP_EXP_C4 0x1.55555cf172b91p-5
#define MPHIP_EXP_C5 0x1.1111167a4d017p-7
#define MPHIP_LN2HI 0x1.62e42fefa3800p-1
#define MPHIP_LN2LO 0x1.ef35793c76730p-45

/* ---- exp ------------------------------------------------------------------------------------------------------
 * The tail of exp and of pow: 2^(k/128) exp(r) from the reduced argument; `tmp` = exp(r) - 1 + tail, `sbits` the
 * bits of the scale.  Results whose scale's exponent field over- or underflowed (|x| >= 512) take the library's
 * two-step scaling; its k < 0 side forms scale + scale * tmp from a separate product (the product is used twice
 * there), unlike the fused operation of the main path. */
MPHIP_LIBM_FN double mphip_libm_exp_special(double tmp, uint64_t sbits, uint64_t ki) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  if ((ki & 0x80000000u) == 0) {
    /* k > 0: the exponent of the scale may have overflowed by up to 460 */
    sbits -= 1009ull << 52;
    const double scale = mphip_libm_from_bits(sbits);
    return 0x1p1009 * MPHIP_LIBM_FMA(scale, tmp, scale);
  }
  /* k < 0: care in the subnormal range */
  sbits += 1022ull << 52;
  const double scale = mphip_libm_from_bits(sbits);
  const double st = scale * tmp;
  double y = scale + st;
  const double ay = y < 0.0 ? -y : y;
  if (ay < 1.0) {
    /* round to the subnormal grid once: add and remove 1 (sign-matched in pow) with a compensated low part */
    const double one = y < 0.0 ? -1.0 : 1.0;
    double lo = scale - y + st;
    const double hi = one + y;
    lo = one - hi + y + lo;
    y = (hi + lo) - one;
    if (y == 0.0)
      y = mphip_libm_from_bits(sbits & 0x8000000000000000ull);
  }
  return 0x1p-1022 * y;
}

/* exp(x + xtail) with the sign of pow's result (sign_bias = 0 or 0x800 << 7) */
MPHIP_LIBM_FN double mphip_libm_exp_core(const uint64_t *tab, double x, double xtail, int with_tail, uint32_t sign_bias) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  uint32_t abstop = (uint32_t) (mphip_libm_bits(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {
    if (abstop - 0x3c9u >= 0x80000000u) {
      /* |x| < 2^-54 */
      const double one = 1.0 + x;
      return sign_bias ? -one : one;
    }
    if (abstop >= 0x409u) {
      /* |x| >= 1024 (pow: NaN and infinities were dealt with before) */
      if (!with_tail) {
        if (mphip_libm_bits(x) == 0xfff0000000000000ull)
          return 0.0;
        if (abstop >= 0x7ffu)
          return 1.0 + x;
      }
      const double big = (mphip_libm_bits(x) >> 63) ? 0x1p-767 : 0x1p769;     /* the library's under- / overflow products */
      return (sign_bias ? -big : big) * big;
    }
    abstop = 0;
  }
  /* x = k ln2 / 128 + r */
  const double z = MPHIP_LIBM_FMA(MPHIP_EXP_INVLN2N, x, MPHIP_EXP_SHIFT);
  const uint64_t ki = mphip_libm_bits(z);
  const double kd = z - MPHIP_EXP_SHIFT;
  double r = MPHIP_LIBM_FMA(kd, MPHIP_EXP_NEGLN2HIN, x);
  r = MPHIP_LIBM_FMA(kd, MPHIP_EXP_NEGLN2LON, r);
  if (with_tail)
    r = r + xtail;
  const uint32_t idx = 2 * ((uint32_t) ki & 127u);
  const double tail = mphip_libm_from_bits(tab[idx]);
  /* sbits = tab[idx + 1] + ((ki + sign_bias) << 45): the shifted term has a zero low word */
  const uint64_t sbits = tab[idx + 1] + ((uint64_t) (((uint32_t) ki + sign_bias) << 13) << 32);
  const double r2 = r * r;
  const double p23 = MPHIP_LIBM_FMA(r, MPHIP_EXP_C3, MPHIP_EXP_C2);
  const double p45 = MPHIP_LIBM_FMA(r, MPHIP_EXP_C5, MPHIP_EXP_C4);
  const double lowp = MPHIP_LIBM_FMA(p23, r2, r + tail);
  const double tmp = MPHIP_LIBM_FMA(r2 * r2, p45, lowp);
  if (abstop == 0)
    return mphip_libm_exp_special(tmp, sbits, ki);
  const double scale = mphip_libm_from_bits(sbits);
  return MPHIP_LIBM_FMA(scale, tmp, scale);
}

MPHIP_LIBM_FN double mphip_libm_exp(const uint64_t *exp_tab, double x) {
  return mphip_libm_exp_core(exp_tab, x, 0.0, 0, 0);
}

/* ---- log ------------------------------------------------------------------------------------------------------ */

/* log x for 1 - 2^-4 <= x < 1 + 0x1.09p-4: a degree-11 polynomial in r = x - 1 whose quadratic term is split */
MPHIP_LIBM_FN double mphip_libm_log_near_one(double x) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double B0 = -0x1.0000000000000p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3,
               B4 = -0x1.55555556745a7p-3, B5 = 0x1.24924a344de30p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4,
               B8 = -0x1.999eb43b068ffp-4, B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
  const double r = x - 1.0;
  const double r2 = r * r;
  const double r3 = r * r2;
  /* (written innermost group first: the order that keeps the fewest values alive) */
  double q = MPHIP_LIBM_FMA(r2, B9, MPHIP_LIBM_FMA(r, B8, B7));
  q = MPHIP_LIBM_FMA(r3, B10, q);
  q = MPHIP_LIBM_FMA(q, r3, MPHIP_LIBM_FMA(r2, B6, MPHIP_LIBM_FMA(r, B5, B4)));
  q = MPHIP_LIBM_FMA(q, r3, MPHIP_LIBM_FMA(r2, B3, MPHIP_LIBM_FMA(r, B2, B1)));
  /* r = rhi + rlo with rhi * rhi exact */
  const double w = MPHIP_LIBM_FMA(r, 0x1p27, r);
  const double rhi = MPHIP_LIBM_FMA(-0x1p27, r, w);
  const double rlo = r - rhi;
  const double rhi2 = rhi * rhi;
  const double hi = MPHIP_LIBM_FMA(rhi2, B0, r);
  double lo = MPHIP_LIBM_FMA(rhi2, B0, r - hi);
  lo = MPHIP_LIBM_FMA(B0 * rlo, r + rhi, lo);
  return hi + MPHIP_LIBM_FMA(q, r3, lo);
}

/* 1 - 2^-4 <= x < 1 + 0x1.09p-4 (bits of x): the arguments of the polynomial above */
MPHIP_LIBM_FN int mphip_libm_log_is_near_one(uint64_t ix) {
  return ix - 0x3fee000000000000ull < 0x3090000000000ull;
}

/* log x away from 1 for the bits ix of a positive normal x = 2^k z, z in [0.6875, 1.375): 1 / c and log c of the
 * subinterval of z from the table, a degree-5 polynomial in r = z / c - 1 */
MPHIP_LIBM_FN double mphip_libm_log_away(const double *log_tab, uint64_t ix) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  /* tmp = ix - OFF, i = (tmp >> 45) % 128, k = (int64) tmp >> 52, z = ix - (tmp & 0xfff << 52) of the algorithm: OFF =
   * 0x3fe6000000000000 has a zero low word, so all of it happens in the high word */
  const uint32_t hx = (uint32_t) (ix >> 32);
  const int32_t tmp = (int32_t) (hx - 0x3fe60000u);
  const uint32_t i = ((uint32_t) tmp >> 13) & 127u;
  const int32_t k = tmp >> 20;
  const double z = mphip_libm_from_words(hx - ((uint32_t) tmp & 0xfff00000u), (uint32_t) ix);
  const double invc = log_tab[2 * i], logc = log_tab[2 * i + 1];
  const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb4590p-3, A3 = 0x1.999b324f10111p-3,
               A4 = -0x1.55575e506c89fp-3;
  const double r = MPHIP_LIBM_FMA(z, invc, -1.0);
  const double kd = (double) k;
  const double w = MPHIP_LIBM_FMA(kd, MPHIP_LN2HI, logc);
  const double hi = w + r;
  const double lo = MPHIP_LIBM_FMA(kd, MPHIP_LN2LO, w - hi + r);
  const double r2 = r * r;
  const double r3 = r * r2;
  const double p12 = MPHIP_LIBM_FMA(r, A2, A1);
  const double p34 = MPHIP_LIBM_FMA(r, A4, A3);
  const double low = MPHIP_LIBM_FMA(r2, A0, lo);
  return MPHIP_LIBM_FMA(r3, MPHIP_LIBM_FMA(p34, r2, p12), low) + hi;
}

MPHIP_LIBM_FN double mphip_libm_log(const double *log_tab, double x) {
  uint64_t ix = mphip_libm_bits(x);
  if (mphip_libm_log_is_near_one(ix))
    return ix == 0x3ff0000000000000ull ? 0.0 : mphip_libm_log_near_one(x);
  const uint32_t top = (uint32_t) (ix >> 48);
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    /* zero, subnormal, negative, infinite, NaN */
    if (ix * 2 == 0)
      return -__builtin_inf();
    if (ix == 0x7ff0000000000000ull)
      return x;
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u)
      return (top & 0x7ff0u) == 0x7ff0u && (ix << 12) != 0 ? x + x : __builtin_nan("");
    ix = mphip_libm_bits(x * 0x1p52) - (52ull << 52);
  }
  return mphip_libm_log_away(log_tab, ix);
}

/* ---- pow ------------------------------------------------------------------------------------------------------ */

/* 0: y is not an integer, 1: odd, 2: even */
MPHIP_LIBM_FN int mphip_libm_checkint(uint64_t iy) {
  const int e = (int) (iy >> 52) & 0x7ff;
  if (e < 0x3ff)
    return 0;
  if (e > 0x3ff + 52)
    return 2;
  if (iy & ((1ull << (0x3ff + 52 - e)) - 1))
    return 0;
  if (iy & (1ull << (0x3ff + 52 - e)))
    return 1;
  return 2;
}

MPHIP_LIBM_FN int mphip_libm_zeroinfnan(uint64_t i) {
  return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1;
}

MPHIP_LIBM_FN double mphip_libm_pow(const mphip_libm_tabs *T, double x, double y) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  uint32_t sign_bias = 0;
  uint64_t ix = mphip_libm_bits(x);
  const uint64_t iy = mphip_libm_bits(y);
  uint32_t topx = (uint32_t) (ix >> 52);
  const uint32_t topy = (uint32_t) (iy >> 52);
  if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
    /* x is zero, subnormal, negative, infinite or NaN, or y is tiny, huge, infinite or NaN (C Annex F cases) */
    if (mphip_libm_zeroinfnan(iy)) {
      if (2 * iy == 0)
        return 1.0;                                   /* (signalling NaNs are not told apart) */
      if (ix == 0x3ff0000000000000ull)
        return 1.0;
      if (2 * ix > 2 * 0x7ff0000000000000ull || 2 * iy > 2 * 0x7ff0000000000000ull)
        return x + y;
      if (2 * ix == 2 * 0x3ff0000000000000ull)
        return 1.0;
      if ((2 * ix < 2 * 0x3ff0000000000000ull) == !(iy >> 63))
        return 0.0;                                   /* |x| < 1 && y == inf  or  |x| > 1 && y == -inf */
      return y * y;
    }
    if (mphip_libm_zeroinfnan(ix)) {
      double x2 = x * x;
      if ((ix >> 63) && mphip_libm_checkint(iy) == 1)
        x2 = -x2;
      return (iy >> 63) ? 1.0 / x2 : x2;
    }
    /* here x and y are non-zero finite */
    if (ix >> 63) {
      const int yint = mphip_libm_checkint(iy);
      if (yint == 0)
        return __builtin_nan("");
      if (yint == 1)
        sign_bias = 0x800u << 7;
      ix &= 0x7fffffffffffffffull;
      topx &= 0x7ff;
    }
    if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
      /* (sign_bias is 0 here: such a y is not odd) */
      if (ix == 0x3ff0000000000000ull)
        return 1.0;
      if ((topy & 0x7ff) < 0x3beu)
        return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;     /* |y| < 2^-65 */
      return (ix > 0x3ff0000000000000ull) == (topy < 0x800u) ? __builtin_inf() : 0.0;
    }
    if (topx == 0) {
      ix = mphip_libm_bits(mphip_libm_from_bits(ix) * 0x1p52);
      ix &= 0x7fffffffffffffffull;
      ix -= 52ull << 52;
    }
  }
  /* log x = hi + lo: k ln2 + log c + log1p(z / c - 1) in double-double */
  const double A0 = -0x1.0000000000000p-1, A1 = -0x1.5555555555560p-1, A2 = 0x1.0000000000006p-1, A3 = 0x1.999999959554ep-1,
               A4 = -0x1.555555529a47ap-1, A5 = -0x1.2495b9b4845e9p+0, A6 = 0x1.0002b8b263fc3p+0;
  const uint32_t hx = (uint32_t) (ix >> 32);                  /* (OFF = 0x3fe6955500000000: zero low word, as in log) */
  const int32_t tmp = (int32_t) (hx - 0x3fe69555u);
  const uint32_t i = ((uint32_t) tmp >> 13) & 127u;
  const int32_t k = tmp >> 20;
  const double z = mphip_libm_from_words(hx - ((uint32_t) tmp & 0xfff00000u), (uint32_t) ix);
  const double kd = (double) k;
  const double invc = T->pow_tab[3 * i], logc = T->pow_tab[3 * i + 1], logctail = T->pow_tab[3 * i + 2];
  const double r = MPHIP_LIBM_FMA(z, invc, -1.0);
  const double t1 = MPHIP_LIBM_FMA(kd, MPHIP_LN2HI, logc);
  const double t2 = t1 + r;
  const double lo1 = MPHIP_LIBM_FMA(kd, MPHIP_LN2LO, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A0 * r;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = MPHIP_LIBM_FMA(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p12 = MPHIP_LIBM_FMA(r, A2, A1);
  const double p34 = MPHIP_LIBM_FMA(r, A4, A3);
  const double p56 = MPHIP_LIBM_FMA(r, A6, A5);
  const double p = MPHIP_LIBM_FMA(ar2, MPHIP_LIBM_FMA(p56, ar2, p34), p12);
  const double lo = MPHIP_LIBM_FMA(ar3, p, lo1 + lo2 + lo3 + lo4);
  const double lhi = hi + lo;
  const double llo = hi - lhi + lo;
  /* y log x in double-double, then exp */
  const double ehi = y * lhi;
  const double elo = MPHIP_LIBM_FMA(y, llo, MPHIP_LIBM_FMA(y, lhi, -ehi));
  return mphip_libm_exp_core(T->exp_tab, ehi, elo, 1, sign_bias);
}

/* ---- sin and cos for |x| < 2.426265 (high word below 0x400368fd: every latitude in radians, ZETA's argument) ----
 * glibc's sysdeps/ieee754/dbl-64/s_sin.c (`__sin_fma` / `__cos_fma`; IBM Accurate Mathematical Library): the argument
 * is split at a multiple of 1 / 128 by adding 1.5 x 2^45, sin and cos of that multiple come from a table with their
 * tails, the rest from two short polynomials; next to pi / 2 the co-function of (pi / 2 - |x|) is taken, and below
 * 0.126 a Taylor polynomial.  Fused operations as the machine code of the two variants has them (glibc 2.35):
 *   do_cos(x, dx):  t = fma(xx, sn5, sn3); s = fma(x xx, t, x); c = xx fma(xx, fma(xx, cs6, cs4), cs2);
 *                   cor = fnma(s, sn, fnma(c, cs, fnma(s, ssn, ccs))); cs + cor
 *   do_sin(x, dx):  s = x + fma(x xx, t, dx); c = fma(x, dx, xx fma(...)); cor = fma(s, cs, fnma(c, sn, fma(s, ccs, ssn))); sn + cor
 *   Taylor:         p = fma chain s5..s1; a + fma(xx, fms(p, a, 0.5 da), da)
 * Outside the range the functions return NaN with *handled = 0 (the caller takes another path). */
#define MPHIP_SC_SN3 (-0x1.5555555555515p-3)
#define MPHIP_SC_SN5 0x1.11110e829872fp-7
#define MPHIP_SC_CS2 0x1.0000000000000p-1
#define MPHIP_SC_CS4 (-0x1.5555555555535p-5)
#define MPHIP_SC_CS6 0x1.6c16bedd9e239p-10
#define MPHIP_SC_S1 (-0x1.5555555555555p-3)
#define MPHIP_SC_S2 0x1.1111111110ecep-7
#define MPHIP_SC_S3 (-0x1.a01a019db08b8p-13)
#define MPHIP_SC_S4 0x1.71de27b9a7ed9p-19
#define MPHIP_SC_S5 (-0x1.addffc2fcdf59p-26)
#define MPHIP_SC_HP0 0x1.921fb54442d18p+0
#define MPHIP_SC_HP1 0x1.1a62633145c07p-54
#define MPHIP_SC_BIG 0x1.8000000000000p+45
#define MPHIP_SC_TAYLOR_BELOW 0x1.020c49ba5e354p-3

MPHIP_LIBM_FN double mphip_libm_fabs(double x) {
  return mphip_libm_from_bits(mphip_libm_bits(x) & 0x7fffffffffffffffULL);
}

MPHIP_LIBM_FN double mphip_libm_copysign(double mag, double sgn) {
  return mphip_libm_from_bits((mphip_libm_bits(mag) & 0x7fffffffffffffffULL) | (mphip_libm_bits(sgn) & 0x8000000000000000ULL));
}

/* cos(ax + dx) for ax >= 0 (the caller has folded the sign of its argument into dx) */
MPHIP_LIBM_FN double mphip_libm_do_cos(const double *tab, double ax, double dx) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double u = MPHIP_SC_BIG + ax;
  const int k = (int) ((uint32_t) mphip_libm_bits(u) << 2);
  const double x = ax - (u - MPHIP_SC_BIG) + dx;
  const double xx = x * x;
  const double t = MPHIP_LIBM_FMA(xx, MPHIP_SC_SN5, MPHIP_SC_SN3);
  const double s = MPHIP_LIBM_FMA(x * xx, t, x);
  const double c = xx * MPHIP_LIBM_FMA(xx, MPHIP_LIBM_FMA(xx, MPHIP_SC_CS6, MPHIP_SC_CS4), MPHIP_SC_CS2);
  const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
  const double cor = MPHIP_LIBM_FMA(-s, sn, MPHIP_LIBM_FMA(-c, cs, MPHIP_LIBM_FMA(-s, ssn, ccs)));
  return cs + cor;
}

MPHIP_LIBM_FN double mphip_libm_taylor_sin(double a, double da) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double xx = a * a;
  double p = MPHIP_LIBM_FMA(xx, MPHIP_SC_S5, MPHIP_SC_S4);
  p = MPHIP_LIBM_FMA(xx, p, MPHIP_SC_S3);
  p = MPHIP_LIBM_FMA(xx, p, MPHIP_SC_S2);
  p = MPHIP_LIBM_FMA(xx, p, MPHIP_SC_S1);
  const double half_da = da * 0.5;
  return a + MPHIP_LIBM_FMA(xx, MPHIP_LIBM_FMA(p, a, -half_da), da);
}

/* sin(a + da), any sign of a */
MPHIP_LIBM_FN double mphip_libm_do_sin(const double *tab, double a, double da) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const double aa = mphip_libm_fabs(a);
  if (aa < MPHIP_SC_TAYLOR_BELOW)
    return mphip_libm_taylor_sin(a, da);
  const double dx = a <= 0.0 ? -da : da;
  const double u = MPHIP_SC_BIG + aa;
  const int k = (int) ((uint32_t) mphip_libm_bits(u) << 2);
  const double x = aa - (u - MPHIP_SC_BIG);
  const double xx = x * x;
  const double t = MPHIP_LIBM_FMA(xx, MPHIP_SC_SN5, MPHIP_SC_SN3);
  const double s = x + MPHIP_LIBM_FMA(x * xx, t, dx);
  const double c = MPHIP_LIBM_FMA(x, dx, xx * MPHIP_LIBM_FMA(xx, MPHIP_LIBM_FMA(xx, MPHIP_SC_CS6, MPHIP_SC_CS4), MPHIP_SC_CS2));
  const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
  const double cor = MPHIP_LIBM_FMA(s, cs, MPHIP_LIBM_FMA(-c, sn, MPHIP_LIBM_FMA(s, ccs, ssn)));
  return mphip_libm_copysign(sn + cor, a);
}

MPHIP_LIBM_FN double mphip_libm_cos(const double *sincos_tab, double x, int *handled) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const uint32_t k = (uint32_t) (mphip_libm_bits(x) >> 32) & 0x7fffffffu;
  *handled = 1;
  if (k < 0x3e400000u)      /* |x| < 2^-27 */
    return 1.0;
  if (k < 0x3feb6000u)      /* |x| < 0.855469 */
    return mphip_libm_do_cos(sincos_tab, mphip_libm_fabs(x), 0.0);
  if (k < 0x400368fdu) {    /* |x| < 2.426265: sin(pi / 2 - |x|) */
    const double y = MPHIP_SC_HP0 - mphip_libm_fabs(x);
    const double a = y + MPHIP_SC_HP1;
    const double da = (y - a) + MPHIP_SC_HP1;
    return mphip_libm_do_sin(sincos_tab, a, da);
  }
  *handled = 0;
  return mphip_libm_from_bits(0x7ff8000000000000ULL);
}

MPHIP_LIBM_FN double mphip_libm_sin(const double *sincos_tab, double x, int *handled) {
#ifdef __clang__
#pragma clang fp contract(off)
#endif
  const uint32_t k = (uint32_t) (mphip_libm_bits(x) >> 32) & 0x7fffffffu;
  *handled = 1;
  if (k < 0x3e500000u)      /* |x| < 2^-26 */
    return x;
  if (k < 0x3feb6000u)
    return mphip_libm_do_sin(sincos_tab, x, 0.0);
  if (k < 0x400368fdu) {    /* cos(pi / 2 - |x|) with the sign of x */
    const double t = MPHIP_SC_HP0 - mphip_libm_fabs(x);
    const double r = mphip_libm_do_cos(sincos_tab, mphip_libm_fabs(t), t >= 0.0 ? MPHIP_SC_HP1 : -MPHIP_SC_HP1);
    return mphip_libm_copysign(r, x);
  }
  *handled = 0;
  return mphip_libm_from_bits(0x7ff8000000000000ULL);
}

#endif
