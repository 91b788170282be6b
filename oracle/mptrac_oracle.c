/*
 * mptrac_oracle.c -- CPU oracle for the MPTRAC per-particle time-step loop.
 *
 * TEST INFRASTRUCTURE ONLY (see mptrac_oracle.h).  Plain-C restatement of the
 * reference algorithm; every function cites the reference lines it follows
 * (paths relative to the reference repo, mptrac.c = src/mptrac.c,
 * mptrac.h = src/mptrac.h).
 *
 * Pinned (tests/test_oracle_pins.py) against the reference's own golden files: sedi.tab, the dd_test
 * trajectories, the met_sample humidity table, the survey's known-answer values, and a whole stochastic
 * run -- tests/coord_test, reproduced in every printed digit.  oracle/README.md lists what each pin fixes
 * and which modules no reference artefact reaches.
 *
 * Build: gcc -O3 -ffp-contract=off -fopenmp (oracle/Makefile).  Contraction is
 * switched off so that the arithmetic is the reference's gcc/x86-64 arithmetic
 * (no FMA), operation by operation.
 */
#include "mptrac_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants (mptrac.h:255-345, 430-460, 535) ------------------------- */
#define C_G0 9.80665
#define C_H0 7.0
#define C_KB 1.3806504e-23
#define C_MA 28.9644
#define C_P0 1013.25
#define C_RI 8.3144598
#define C_RA (1e3 * C_RI / C_MA)
#define C_RE 6367.421
#define C_M_AIR 4.8096e-26
#define C_TREF 298.15
#define C_T0 273.15
#define C_SO2_K1_REF 1.23e-2
#define C_SO2_K1_TEMP 2.01e3
#define C_SO2_K2_REF 6e-8
#define C_SO2_K2_TEMP 1.12e3
#define C_WD_T_LIQUID C_T0
#define C_WD_T_ICE 238.15
#define C_WD_T_LIQUID_BC 270.
#define C_MH2O 18.01528
#define C_EPS (C_MH2O / C_MA)
#define C_KAPPA 0.286
#define C_CPD 1003.5
#define C_KARMAN 0.40

#define SQ(x) ((x) * (x))

static void die(const char *msg) {
  /* the reference's ERRMSG prints and exits (mptrac.h:2406-2410) */
  printf("\nOracle error: %s\n\n", msg);
  exit(EXIT_FAILURE);
}

size_t orc_sizeof_ctl(void) {
  return sizeof(orc_ctl_t);
}

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0)
    omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void) n;
  return 1;
#endif
}

/* ---- arithmetic conventions (mptrac.h) ---------------------------------- */

/* FMOD, mptrac.h:1121-1122: truncation through (int), not fmod() */
static inline double fmod_trunc(double x, double y) {
  return x - (int) (x / y) * y;
}

/* DEG2RAD, mptrac.h:857 */
static inline double deg2rad(double deg) {
  return deg * (M_PI / 180.0);
}

/* DX2DEG, mptrac.h:904-906 (dx in km) */
static inline double dx2deg(double dx, double lat) {
  if (lat < -89.999 || lat > 89.999)
    return 0;
  return dx * 180. / (M_PI * C_RE * cos(deg2rad(lat)));
}

/* DY2DEG, mptrac.h:922 (dy in km) */
static inline double dy2deg(double dy) {
  return dy * 180. / (M_PI * C_RE);
}

/* DX2COORD / DY2COORD, mptrac.h:966, 989 (distance in m) */
static inline double dx2coord(int coord_type, double dx, double lat) {
  return coord_type == 0 ? dx2deg(dx / 1000.0, lat) : dx;
}

static inline double dy2coord(int coord_type, double dy) {
  return coord_type == 0 ? dy2deg(dy / 1000.0) : dy;
}

/* DZ2DP, mptrac.h:941 */
static inline double dz2dp(double dz, double p) {
  return -dz * p / C_H0;
}

/* Z, mptrac.h:2243 */
static inline double zfromp(double p) {
  return C_H0 * log(C_P0 / p);
}

/* LIN, mptrac.h:1351 */
static inline double lin(double x0, double y0, double x1, double y1, double x) {
  return y0 + (y1 - y0) / (x1 - x0) * (x - x0);
}

/* RHO, mptrac.h:1961 */
static inline double rho_air(double p, double t) {
  return 100. * p / (C_RA * t);
}

static inline double dmin(double a, double b) {
  return a < b ? a : b;
}

static inline double dmax(double a, double b) {
  return a > b ? a : b;
}

/* ---- locate (mptrac.c:3495-3574) ---------------------------------------- */

int orc_locate_irr(const double *xx, int n, double x) {
  /* mptrac.c:3495-3521: bisection; direction decided at the first midpoint */
  int lo = 0, hi = n - 1;
  int mid = (hi + lo) >> 1;
  if (xx[mid] < xx[mid + 1]) {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] > x)
        hi = mid;
      else
        lo = mid;
    }
  } else {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] <= x)
        hi = mid;
      else
        lo = mid;
    }
  }
  return lo;
}

int orc_locate_reg(const double *xx, int n, double x) {
  /* mptrac.c:3559-3574 */
  const int i = (int) ((x - xx[0]) / (xx[1] - xx[0]));
  if (i < 0)
    return 0;
  if (i > n - 2)
    return n - 2;
  return i;
}

/* ---- interpolation (mptrac.c:2755-3170) --------------------------------- */

/* indices/weights carried between variables: ci[3], cw[4] (mptrac.h:1174) */
typedef struct {
  int ip, ix, iy;
  double wp, wx, wy;
} stencil_t;

static const stencil_t STENCIL_ZERO = { 0, 0, 0, 0.0, 0.0, 0.0 };

/* mptrac.c:2755-2778 */
static void check_lon_lat(const orc_met_t *m, double lon, double lat, double *lon2, double *lat2) {
  double l = fmod_trunc(lon, 360.);
  if (l < m->lon[0])
    l += 360;
  else if (l > m->lon[m->nx - 1])
    l -= 360;
  *lon2 = l;
  if (m->lat[0] < m->lat[m->ny - 1])
    *lat2 = dmin(dmax(lat, m->lat[0]), m->lat[m->ny - 1]);
  else
    *lat2 = dmin(dmax(lat, m->lat[m->ny - 1]), m->lat[0]);
}

/* mptrac.c:2782-2803 */
static void check_cartesian(const orc_met_t *m, double lon, double lat, double *lon2, double *lat2) {
  if (m->lon[0] < m->lon[m->nx - 1])
    *lon2 = dmin(dmax(lon, m->lon[0]), m->lon[m->nx - 1]);
  else
    *lon2 = dmin(dmax(lon, m->lon[m->nx - 1]), m->lon[0]);
  if (m->lat[0] < m->lat[m->ny - 1])
    *lat2 = dmin(dmax(lat, m->lat[0]), m->lat[m->ny - 1]);
  else
    *lat2 = dmin(dmax(lat, m->lat[m->ny - 1]), m->lat[0]);
}

static void check_horizontal(const orc_met_t *m, double lon, double lat, double *lon2, double *lat2) {
  if (m->coord_type == 0)
    check_lon_lat(m, lon, lat, lon2, lat2);
  else
    check_cartesian(m, lon, lat, lon2, lat2);
}

#define A3(a, m, i, j, k) ((a)[((size_t) (i) * (size_t) (m)->ny + (size_t) (j)) * (size_t) (m)->np + (size_t) (k)])
#define A2(a, m, i, j) ((a)[(size_t) (i) * (size_t) (m)->ny + (size_t) (j)])

/* index / weight set-up of intpol_met_space_3d (init != 0), mptrac.c:2997-3021 */
static void stencil_init_3d(const orc_met_t *m, double p, double lon, double lat, stencil_t *s) {
  double lon2, lat2;
  check_horizontal(m, lon, lat, &lon2, &lat2);
  s->ip = orc_locate_irr(m->p, m->np, p);
  s->ix = orc_locate_reg(m->lon, m->nx, lon2);
  s->iy = orc_locate_irr(m->lat, m->ny, lat2);
  s->wp = (m->p[s->ip + 1] - p) / (m->p[s->ip + 1] - m->p[s->ip]);
  s->wx = (m->lon[s->ix + 1] - lon2) / (m->lon[s->ix + 1] - m->lon[s->ix]);
  s->wy = (m->lat[s->iy + 1] - lat2) / (m->lat[s->iy + 1] - m->lat[s->iy]);
}

/* intpol_met_space_3d, mptrac.c:2985-3044 */
static double space_3d(const orc_met_t *m, const float *a, double p, double lon, double lat,
                       stencil_t *s, int init) {
  if (init)
    stencil_init_3d(m, p, lon, lat, s);
  const int ix = s->ix, iy = s->iy, ip = s->ip;
  /* vertical first, then latitude, then longitude */
  const double c00 = s->wp * (A3(a, m, ix, iy, ip) - A3(a, m, ix, iy, ip + 1)) + A3(a, m, ix, iy, ip + 1);
  const double c01 = s->wp * (A3(a, m, ix, iy + 1, ip) - A3(a, m, ix, iy + 1, ip + 1))
    + A3(a, m, ix, iy + 1, ip + 1);
  const double c10 = s->wp * (A3(a, m, ix + 1, iy, ip) - A3(a, m, ix + 1, iy, ip + 1))
    + A3(a, m, ix + 1, iy, ip + 1);
  const double c11 = s->wp * (A3(a, m, ix + 1, iy + 1, ip) - A3(a, m, ix + 1, iy + 1, ip + 1))
    + A3(a, m, ix + 1, iy + 1, ip + 1);
  const double r0 = s->wy * (c00 - c01) + c01;
  const double r1 = s->wy * (c10 - c11) + c11;
  return s->wx * (r0 - r1) + r1;
}

/* intpol_met_space_2d, mptrac.c:3048-3108 (NaN-aware nearest neighbour) */
static double space_2d(const orc_met_t *m, const float *a, double lon, double lat,
                       stencil_t *s, int init) {
  if (init) {
    double lon2, lat2;
    check_horizontal(m, lon, lat, &lon2, &lat2);
    s->ix = orc_locate_reg(m->lon, m->nx, lon2);
    s->iy = orc_locate_irr(m->lat, m->ny, lat2);
    s->wx = (m->lon[s->ix + 1] - lon2) / (m->lon[s->ix + 1] - m->lon[s->ix]);
    s->wy = (m->lat[s->iy + 1] - lat2) / (m->lat[s->iy + 1] - m->lat[s->iy]);
  }
  const double c00 = A2(a, m, s->ix, s->iy);
  const double c01 = A2(a, m, s->ix, s->iy + 1);
  const double c10 = A2(a, m, s->ix + 1, s->iy);
  const double c11 = A2(a, m, s->ix + 1, s->iy + 1);
  if (isfinite(c00) && isfinite(c01) && isfinite(c10) && isfinite(c11)) {
    const double r0 = s->wy * (c00 - c01) + c01;
    const double r1 = s->wy * (c10 - c11) + c11;
    return s->wx * (r0 - r1) + r1;
  }
  if (s->wy < 0.5)
    return s->wx < 0.5 ? c11 : c01;
  return s->wx < 0.5 ? c10 : c00;
}

/* intpol_met_time_3d, mptrac.c:3112-3137: met0 with init, met1 re-using indices */
static double time_3d(const orc_met_t *m0, const orc_met_t *m1, int f, double ts, double p,
                      double lon, double lat, stencil_t *s, int init) {
  const double v0 = space_3d(m0, m0->f3[f], p, lon, lat, s, init);
  const double v1 = space_3d(m1, m1->f3[f], p, lon, lat, s, 0);
  const double wt = (m1->time - ts) / (m1->time - m0->time);
  return wt * (v0 - v1) + v1;
}

/* intpol_met_time_2d, mptrac.c:3141-3170 */
static double time_2d(const orc_met_t *m0, const orc_met_t *m1, int f, double ts, double lon,
                      double lat, stencil_t *s, int init) {
  const double v0 = space_2d(m0, m0->f2[f], lon, lat, s, init);
  const double v1 = space_2d(m1, m1->f2[f], lon, lat, s, 0);
  const double wt = (m1->time - ts) / (m1->time - m0->time);
  if (isfinite(v0) && isfinite(v1))
    return wt * (v0 - v1) + v1;
  return wt < 0.5 ? v1 : v0;
}

void orc_intpol_met_time_3d(const orc_met_t *met0, const orc_met_t *met1, int field, double ts,
                            double p, double lon, double lat, double *var) {
  stencil_t s = STENCIL_ZERO;
  *var = time_3d(met0, met1, field, ts, p, lon, lat, &s, 1);
}

void orc_intpol_met_time_2d(const orc_met_t *met0, const orc_met_t *met1, int field, double ts,
                            double lon, double lat, double *var) {
  stencil_t s = STENCIL_ZERO;
  *var = time_2d(met0, met1, field, ts, lon, lat, &s, 1);
}

/* ---- climatological tropopause and weights ------------------------------ */

/* clim_tropo, mptrac.c:213-237 */
double orc_clim_tropo(const orc_clim_t *clim, double t, double lat) {
  double sec = fmod_trunc(t, 365.25 * 86400.);
  while (sec < 0)
    sec += 365.25 * 86400.;
  const int it = orc_locate_irr(clim->tropo_time, clim->tropo_ntime, sec);
  const int il = orc_locate_reg(clim->tropo_lat, clim->tropo_nlat, lat);
  const double pa = lin(clim->tropo_lat[il], clim->tropo[it][il],
                        clim->tropo_lat[il + 1], clim->tropo[it][il + 1], lat);
  const double pb = lin(clim->tropo_lat[il], clim->tropo[it + 1][il],
                        clim->tropo_lat[il + 1], clim->tropo[it + 1][il + 1], lat);
  return lin(clim->tropo_time[it], pa, clim->tropo_time[it + 1], pb, sec);
}

/* tropo_weight, mptrac.c:12748-12770 */
double orc_tropo_weight(const orc_ctl_t *ctl, const orc_clim_t *clim, double time, double lat,
                        double p) {
  const double pt = orc_clim_tropo(clim, time, ctl->met_coord_type == 0 ? lat : ctl->met_utm_ref_lat);
  const double p1 = pt * 0.866877899;
  const double p0 = pt / 0.866877899;
  if (p > p0)
    return 1;
  if (p < p1)
    return 0;
  return lin(p0, 1.0, p1, 0.0, p);
}

/* pbl_weight, mptrac.c:8358-8376 */
double orc_pbl_weight(const orc_ctl_t *ctl, double p, double pbl, double ps) {
  const double p1 = pbl - ctl->turb_pbl_trans * (ps - pbl);
  const double p0 = pbl;
  if (p > p0)
    return 1;
  if (p < p1)
    return 0;
  return lin(p0, 1.0, p1, 0.0, p);
}

/* sedi, mptrac.c:12506-12535 */
double orc_sedi(double p, double T, double rp, double rhop) {
  const double r = rp * 1e-6;
  const double rho = rho_air(p, T);
  const double eta = 1.8325e-5 * (416.16 / (T + 120.)) * pow(T / 296.16, 1.5);
  const double v = sqrt(8. * C_KB * T / (M_PI * C_M_AIR));
  const double lambda = 2. * eta / (rho * v);
  const double K = lambda / r;
  const double G = 1. + K * (1.249 + 0.42 * exp(-0.87 / K));
  return 2. * SQ(r) * (rhop - rho) * C_G0 / (9. * eta) * G;
}

/* ---- random numbers (mptrac.c:5784-5828) -------------------------------- */

/* Squares counter-based RNG (Widynski 2022), five rounds, fixed key;
 * mptrac.c:5788, 5798-5809 */
uint64_t orc_squares(uint64_t ctr) {
  const uint64_t key = 0xc8e4fd154ce32f6dULL;
  uint64_t x, y, z, t;
  y = x = ctr * key;
  z = y + key;
  x = x * x + y;
  x = (x >> 32) | (x << 32);
  x = x * x + z;
  x = (x >> 32) | (x << 32);
  x = x * x + y;
  x = (x >> 32) | (x << 32);
  t = x = x * x + z;
  x = (x >> 32) | (x << 32);
  return t ^ ((x * x + y) >> 32);
}

void orc_libm_sincosf(const float *x, size_t n, float *cos_out, float *sin_out) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    cos_out[i] = cosf(x[i]);
    sin_out[i] = sinf(x[i]);
  }
}

/* the C library's double functions over arrays: op 0 exp(x), 1 log(x), 2 pow(x, y), 3 sqrt(x) -- the checker of
 * the device's restatements (csrc/mphip_libm.h) */
void orc_libm_f64(int op, const double *x, const double *y, size_t n, double *out) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++)
    out[i] = op == 0 ? exp(x[i]) : op == 1 ? log(x[i]) : op == 2 ? pow(x[i], y[i]) : op == 4 ? cos(x[i]) : op == 5 ? sin(x[i]) : sqrt(x[i]);
}

/* module_rng, RNG_TYPE=1 branch: n+1 uniforms, counter advanced by n+1, then
 * Box-Muller over flat pairs with single-precision trig (mptrac.c:5797-5827) */
void orc_module_rng(const orc_ctl_t *ctl, orc_cache_t *cache, size_t n, int method) {
  if (ctl->rng_type != 1)
    die("oracle restates RNG_TYPE=1 (Squares) only");
  double *rs = cache->rs;
  const uint64_t base = cache->rng_ctr;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n + 1; ++i)
    rs[i] = (double) orc_squares(base + i) / (double) UINT64_MAX;
  cache->rng_ctr += n + 1;
  if (method == 1) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i += 2) {
      const double r = sqrt(-2.0 * log(rs[i]));
      const double phi = 2.0 * M_PI * rs[i + 1];
      rs[i] = r * cosf((float) phi);
      rs[i + 1] = r * sinf((float) phi);
    }
  }
}

/* The random numbers of one module for the particles of `atm`: n_per values per particle, slot
 * rs[n_per * ip + k] as the modules read them (mptrac.c:4645-4647, 4322-4331, 4143).  Normally this is
 * module_rng over n_per * np numbers.  With cache->ip_global set, `atm` holds a SUBSAMPLE of a run with
 * cache->np_global particles and particle ip is that run's particle ip_global[ip]: it gets the numbers the
 * full run's module_rng call puts into rs[n_per * ip_global[ip] + k] (the uniform at flat index f is Squares of
 * counter + f, mptrac.c:5797-5810; the normal at f comes from the uniforms of the flat pair (f & ~1, f | 1),
 * mptrac.c:5820-5826), and the counter advances by the full run's n + 1 (mptrac.c:5812). */
static void module_random_numbers(const orc_ctl_t *ctl, orc_cache_t *cache, int np, int n_per, int method) {
  if (!cache->ip_global) {
    orc_module_rng(ctl, cache, (size_t) n_per * (size_t) np, method);
    return;
  }
  if (ctl->rng_type != 1)
    die("oracle restates RNG_TYPE=1 (Squares) only");
  const uint64_t base = cache->rng_ctr;
  double *rs = cache->rs;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < np; ip++)
    for (int k = 0; k < n_per; k++) {
      const uint64_t f = (uint64_t) n_per * (uint64_t) cache->ip_global[ip] + (uint64_t) k;
      double *out = &rs[(size_t) n_per * (size_t) ip + (size_t) k];
      if (method != 1) {
        *out = (double) orc_squares(base + f) / (double) UINT64_MAX;
        continue;
      }
      const uint64_t e = f & ~(uint64_t) 1;
      const double u0 = (double) orc_squares(base + e) / (double) UINT64_MAX;
      const double u1 = (double) orc_squares(base + e + 1) / (double) UINT64_MAX;
      const double r = sqrt(-2.0 * log(u0));
      const double phi = 2.0 * M_PI * u1;
      *out = (f & 1) ? r * sinf((float) phi) : r * cosf((float) phi);
    }
  cache->rng_ctr += (uint64_t) n_per * (uint64_t) cache->np_global + 1;
}

/* ---- module_timesteps (mptrac.c:5999-6073) ------------------------------ */

void orc_module_timesteps(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                          const orc_atm_t *atm, double t) {
  /* gsl_stats_min/max over the latitude axis, mptrac.c:6009-6010 */
  double latmin = met0->lat[0], latmax = met0->lat[0];
  for (int j = 1; j < met0->ny; j++) {
    latmin = dmin(latmin, met0->lat[j]);
    latmax = dmax(latmax, met0->lat[j]);
  }
  const int local = (fabs(met0->lon[met0->nx - 1] - met0->lon[0] - 360.0) >= 0.01);
  const double dir = ctl->direction;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    const double tp = atm->time[ip];
    if (dir * (tp - ctl->t_start) >= 0 && dir * (tp - ctl->t_stop) <= 0 && dir * (tp - t) < 0)
      cache->dt[ip] = t - tp;
    else
      cache->dt[ip] = 0.0;
    if (local && (atm->lon[ip] <= met0->lon[0] || atm->lon[ip] >= met0->lon[met0->nx - 1]
                  || atm->lat[ip] <= latmin || atm->lat[ip] >= latmax))
      cache->dt[ip] = 0.0;
  }
}

void orc_module_timesteps_init(orc_ctl_t *ctl, const orc_atm_t *atm) {
  /* mptrac.c:6046-6073 */
  double tmin = atm->time[0], tmax = atm->time[0];
  for (int ip = 1; ip < atm->np; ip++) {
    tmin = dmin(tmin, atm->time[ip]);
    tmax = dmax(tmax, atm->time[ip]);
  }
  if (ctl->direction == 1) {
    ctl->t_start = tmin;
    if (ctl->t_stop > 1e99)
      ctl->t_stop = tmax;
  } else {
    ctl->t_start = tmax;
    if (ctl->t_stop > 1e99)
      ctl->t_stop = tmin;
  }
  if (ctl->direction * (ctl->t_stop - ctl->t_start) <= 0)
    die("Nothing to do! Check T_STOP and DIRECTION!");
  if (ctl->direction == 1)
    ctl->t_start = floor(ctl->t_start / ctl->dt_mod) * ctl->dt_mod;
  else
    ctl->t_start = ceil(ctl->t_start / ctl->dt_mod) * ctl->dt_mod;
}

/* ---- module_position (mptrac.c:5435-5489) ------------------------------- */

void orc_module_position(const orc_cache_t *cache, const orc_met_t *met0, const orc_met_t *met1,
                         orc_atm_t *atm) {
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    double lon = atm->lon[ip], lat = atm->lat[ip], p = atm->p[ip];
    if (met0->coord_type == 0) {
      lon = fmod_trunc(lon, 360.);
      lat = fmod_trunc(lat, 360.);
      while (lat < -90 || lat > 90) {
        if (lat > 90) {
          lat = 180 - lat;
          lon += 180;
        }
        if (lat < -90) {
          lat = -180 - lat;
          lon += 180;
        }
      }
      while (lon < -180)
        lon += 360;
      while (lon >= 180)
        lon -= 360;
    } else {
      double lon2, lat2;
      check_cartesian(met0, lon, lat, &lon2, &lat2);
      lon = lon2;
      lat = lat2;
    }
    const double ptop = met0->p[met0->np - 1];
    if (p < ptop) {
      p = ptop * ptop / p;
    } else if (p > 300.) {
      /* INTPOL_2D(ps, 0) on a freshly zeroed stencil (mptrac.c:5449, 5484):
       * indices 0 and weights 0 select grid node [1][1] -- reference quirk,
       * reproduced as is. */
      stencil_t s = STENCIL_ZERO;
      const double ps = time_2d(met0, met1, ORC_PS, atm->time[ip], lon, lat, &s, 0);
      if (p > ps)
        p = ps * ps / p;
    }
    atm->lon[ip] = lon;
    atm->lat[ip] = lat;
    atm->p[ip] = p;
  }
}

/* ---- module_advect, pressure-level branch (mptrac.c:3609-3678) ---------- */

/* indices/weights of intpol_met_4d_zeta: ci[3], cw[4] */
typedef struct {
  int ix, iy, iz;
  double wx, wy, wz, wt;
} stencil4_t;

static void advect_model_levels(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                                const orc_met_t *met1, orc_atm_t *atm);
static double intpol_4d_zeta(const orc_met_t *m0, int fh, int fa, const orc_met_t *m1, double ts, double height,
                             double lon, double lat, stencil4_t *s, int init);

void orc_module_advect(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                       const orc_met_t *met1, orc_atm_t *atm) {
  if (ctl->advect_vert_coord == 1 || ctl->advect_vert_coord == 3) {
    advect_model_levels(ctl, cache, met0, met1, atm);
    return;
  }
  const int ct = met0->coord_type;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    const double dt = cache->dt[ip];
    stencil_t s = STENCIL_ZERO;
    stencil4_t s4 = { 0, 0, 0, 0, 0, 0, 0 };
    double u[4], v[4], w[4], um = 0, vm = 0, wm = 0;
    double x0 = 0, x1 = 0, x2 = 0;
    for (int i = 0; i < ctl->advect; i++) {
      double dts;
      if (i == 0) {
        dts = 0.0;
        x0 = atm->lon[ip];
        x1 = atm->lat[ip];
        x2 = atm->p[ip];
      } else {
        dts = (i == 3 ? 1.0 : 0.5) * dt;
        x0 = atm->lon[ip] + dx2coord(ct, dts * u[i - 1], atm->lat[ip]);
        x1 = atm->lat[ip] + dy2coord(ct, dts * v[i - 1]);
        x2 = atm->p[ip] + dts * w[i - 1];
      }
      const double tm = atm->time[ip] + dts;
      if (ctl->advect_vert_coord == 0) {
        u[i] = time_3d(met0, met1, ORC_U, tm, x2, x0, x1, &s, 1);
        v[i] = time_3d(met0, met1, ORC_V, tm, x2, x0, x1, &s, 0);
        w[i] = time_3d(met0, met1, ORC_W, tm, x2, x0, x1, &s, 0);
      } else {   /* ADVECT_VERT_COORD 2: winds from the model levels, pressure as the height variable (mptrac.c:3649-3659) */
        u[i] = intpol_4d_zeta(met0, ORC_PL, ORC_UL, met1, tm, x2, x0, x1, &s4, 1);
        v[i] = intpol_4d_zeta(met0, ORC_PL, ORC_VL, met1, tm, x2, x0, x1, &s4, 0);
        w[i] = intpol_4d_zeta(met0, ORC_PL, ORC_WL, met1, tm, x2, x0, x1, &s4, 0);
      }
      double k = 1.0;
      if (ctl->advect == 2)
        k = (i == 0 ? 0.0 : 1.0);
      else if (ctl->advect == 4)
        k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
      um += k * u[i];
      vm += k * v[i];
      wm += k * w[i];
    }
    atm->time[ip] += dt;
    atm->lon[ip] += dx2coord(ct, dt * um, (ctl->advect == 2 ? x1 : atm->lat[ip]));
    atm->lat[ip] += dy2coord(ct, dt * vm);
    atm->p[ip] += dt * wm;
  }
}

/* ---- model-level interpolation (mptrac.c:2808-2981, 3525-3594) ----------- */

#define AL(a, m, i, j, k) ((a)[((size_t) (i) * (size_t) (m)->ny + (size_t) (j)) * (size_t) (m)->npl + (size_t) (k)])

/* locate_irr_float, mptrac.c:3525-3555: bisection on a float profile with a
 * first guess */
static int locate_irr_float(const float *xx, int n, double x, int ig) {
  int lo = 0, hi = n - 1;
  int mid = (hi + lo) >> 1;
  if ((xx[ig] <= x && x < xx[ig + 1]) || (xx[ig] >= x && x > xx[ig + 1]))
    return ig;
  if (xx[mid] < xx[mid + 1]) {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] > x)
        hi = mid;
      else
        lo = mid;
    }
  } else {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] <= x)
        hi = mid;
      else
        lo = mid;
    }
  }
  return lo;
}

/* time-then-horizontal interpolation of the height field at level k */
static double height_at(const orc_met_t *m0, const float *h0, const float *h1, const stencil4_t *s, int k) {
  const int ix = s->ix, iy = s->iy;
  const double h00 = s->wt * (AL(h1, m0, ix, iy, k) - AL(h0, m0, ix, iy, k)) + AL(h0, m0, ix, iy, k);
  const double h01 = s->wt * (AL(h1, m0, ix, iy + 1, k) - AL(h0, m0, ix, iy + 1, k)) + AL(h0, m0, ix, iy + 1, k);
  const double h10 = s->wt * (AL(h1, m0, ix + 1, iy, k) - AL(h0, m0, ix + 1, iy, k)) + AL(h0, m0, ix + 1, iy, k);
  const double h11 = s->wt * (AL(h1, m0, ix + 1, iy + 1, k) - AL(h0, m0, ix + 1, iy + 1, k))
    + AL(h0, m0, ix + 1, iy + 1, k);
  const double a = s->wy * (h01 - h00) + h00;
  const double b = s->wy * (h11 - h10) + h10;
  return s->wx * (b - a) + a;
}

/* intpol_met_4d_zeta, mptrac.c:2808-2981 */
static double intpol_4d_zeta(const orc_met_t *m0, int fh, int fa, const orc_met_t *m1, double ts, double height,
                             double lon, double lat, stencil4_t *s, int init) {
  const float *h0 = m0->f3[fh], *h1 = m1->f3[fh];
  const float *a0 = m0->f3[fa], *a1 = m1->f3[fa];
  if (init) {
    double lon2, lat2;
    check_horizontal(m0, lon, lat, &lon2, &lat2);
    s->ix = orc_locate_reg(m0->lon, m0->nx, lon2);
    s->iy = orc_locate_irr(m0->lat, m0->ny, lat2);
    /* locate_vert on both snapshots, mptrac.c:3578-3594 */
    int ind[2][4];
    for (int t = 0; t < 2; t++) {
      const orc_met_t *m = t ? m1 : m0;
      const float *h = t ? h1 : h0;
      ind[t][0] = locate_irr_float(&AL(h, m0, s->ix, s->iy, 0), m->npl, height, 0);
      ind[t][1] = locate_irr_float(&AL(h, m0, s->ix + 1, s->iy, 0), m->npl, height, ind[t][0]);
      ind[t][2] = locate_irr_float(&AL(h, m0, s->ix, s->iy + 1, 0), m->npl, height, ind[t][1]);
      ind[t][3] = locate_irr_float(&AL(h, m0, s->ix + 1, s->iy + 1, 0), m->npl, height, ind[t][2]);
    }
    s->iz = ind[0][0];
    int k_max = ind[0][0];
    for (int t = 0; t < 2; t++)
      for (int j = 0; j < 4; j++) {
        if (s->iz > ind[t][j])
          s->iz = ind[t][j];
        if (k_max < ind[t][j])
          k_max = ind[t][j];
      }
    s->wt = (ts - m0->time) / (m1->time - m0->time);
    s->wx = (lon2 - m0->lon[s->ix]) / (m0->lon[s->ix + 1] - m0->lon[s->ix]);
    s->wy = (lat2 - m0->lat[s->iy]) / (m0->lat[s->iy + 1] - m0->lat[s->iy]);
    double height_bot = height_at(m0, h0, h1, s, s->iz);
    double height_top = height_at(m0, h0, h1, s, s->iz + 1);
    /* search upward until the height is inside the box, mptrac.c:2905-2938 */
    const float g0 = h0[0], g1 = h0[1];     /* heights0[0][0][0], heights0[0][0][1] */
    while (((g0 > g1) && ((height_bot <= height) || (height_top > height)) && (height_bot >= height)
            && (s->iz < k_max))
           || ((g0 < g1) && ((height_bot >= height) || (height_top < height)) && (height_bot <= height)
               && (s->iz < k_max))) {
      s->iz++;
      height_bot = height_top;
      height_top = height_at(m0, h0, h1, s, s->iz + 1);
    }
    s->wz = (height - height_bot) / (height_top - height_bot);
  }
  /* time first, then longitude, latitude, vertical (mptrac.c:2945-2980) */
  const int ix = s->ix, iy = s->iy, iz = s->iz;
#define TI(i, j, k) (s->wt * (AL(a1, m0, i, j, k) - AL(a0, m0, i, j, k)) + AL(a0, m0, i, j, k))
  const double a000 = TI(ix, iy, iz), a100 = TI(ix + 1, iy, iz), a010 = TI(ix, iy + 1, iz), a110 = TI(ix + 1, iy + 1, iz);
  const double a001 = TI(ix, iy, iz + 1), a101 = TI(ix + 1, iy, iz + 1), a011 = TI(ix, iy + 1, iz + 1),
    a111 = TI(ix + 1, iy + 1, iz + 1);
#undef TI
  const double a00 = s->wx * (a100 - a000) + a000;
  const double a10 = s->wx * (a110 - a010) + a010;
  const double a01 = s->wx * (a101 - a001) + a001;
  const double a11 = s->wx * (a111 - a011) + a011;
  const double lo = s->wy * (a10 - a00) + a00;
  const double hi = s->wy * (a11 - a01) + a01;
  return s->wz * (hi - lo) + lo;
}

/* module_advect, zeta / eta branch (mptrac.c:3681-3757) */
static void advect_model_levels(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                                const orc_met_t *met1, orc_atm_t *atm) {
  const int qnt = (ctl->advect_vert_coord == 1 ? ctl->qnt_zeta : ctl->qnt_eta);
  const int ct = met0->coord_type;
  if (qnt < 0)
    die("model-level advection needs quantity zeta (or eta)");
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    const double dt = cache->dt[ip];
    stencil4_t s = { 0, 0, 0, 0, 0, 0, 0 };
    /* pressure -> vertical coordinate */
    atm->q[qnt][ip] = intpol_4d_zeta(met0, ORC_PL, ORC_ZETAL, met1, atm->time[ip], atm->p[ip], atm->lon[ip],
                                     atm->lat[ip], &s, 1);
    double u[4], v[4], wdot[4], um = 0, vm = 0, wdotm = 0, x0 = 0, x1 = 0, x2 = 0;
    for (int i = 0; i < ctl->advect; i++) {
      double dts;
      if (i == 0) {
        dts = 0.0;
        x0 = atm->lon[ip];
        x1 = atm->lat[ip];
        x2 = atm->q[qnt][ip];
      } else {
        dts = (i == 3 ? 1.0 : 0.5) * dt;
        x0 = atm->lon[ip] + dx2coord(ct, dts * u[i - 1], atm->lat[ip]);
        x1 = atm->lat[ip] + dy2coord(ct, dts * v[i - 1]);
        x2 = atm->q[qnt][ip] + dts * wdot[i - 1];
      }
      const double tm = atm->time[ip] + dts;
      u[i] = intpol_4d_zeta(met0, ORC_ZETAL, ORC_UL, met1, tm, x2, x0, x1, &s, 1);
      v[i] = intpol_4d_zeta(met0, ORC_ZETAL, ORC_VL, met1, tm, x2, x0, x1, &s, 0);
      wdot[i] = intpol_4d_zeta(met0, ORC_ZETAL, ORC_ZETA_DOTL, met1, tm, x2, x0, x1, &s, 0);
      double k = 1.0;
      if (ctl->advect == 2)
        k = (i == 0 ? 0.0 : 1.0);
      else if (ctl->advect == 4)
        k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
      um += k * u[i];
      vm += k * v[i];
      wdotm += k * wdot[i];
    }
    atm->time[ip] += dt;
    atm->lon[ip] += dx2coord(ct, dt * um, (ctl->advect == 2 ? x1 : atm->lat[ip]));
    atm->lat[ip] += dy2coord(ct, dt * vm);
    atm->q[qnt][ip] += dt * wdotm;
    /* vertical coordinate -> pressure */
    atm->p[ip] = intpol_4d_zeta(met0, ORC_ZETAL, ORC_PL, met1, atm->time[ip], atm->q[qnt][ip], atm->lon[ip],
                                atm->lat[ip], &s, 1);
  }
}

/* module_advect_init, mptrac.c:3762-3785: pressure consistent with zeta */
void orc_module_advect_init(const orc_ctl_t *ctl, const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm) {
  if (ctl->advect_vert_coord != 1)
    return;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    stencil4_t s = { 0, 0, 0, 0, 0, 0, 0 };
    atm->p[ip] = intpol_4d_zeta(met0, ORC_ZETAL, ORC_PL, met1, atm->time[ip], atm->q[ctl->qnt_zeta][ip],
                                atm->lon[ip], atm->lat[ip], &s, 1);
  }
}

/* ---- module_diff_turb (mptrac.c:4588-4734) ------------------------------ */

static double kz_blend(const orc_ctl_t *ctl, const orc_clim_t *clim, double time, double lat,
                       double p, double pbl, double ps) {
  /* the Kz expression evaluated at a displaced pressure (mptrac.c:4669-4688);
   * the reference overwrites atm->p[ip] temporarily, the value is the same */
  const double wpbl = orc_pbl_weight(ctl, p, pbl, ps);
  const double wtrop = orc_tropo_weight(ctl, clim, time, lat, p) * (1.0 - wpbl);
  const double wstrat = 1.0 - wpbl - wtrop;
  return wpbl * ctl->turb_dz_pbl + wtrop * ctl->turb_dz_trop + wstrat * ctl->turb_dz_strat;
}

void orc_module_diff_turb(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_clim_t *clim,
                          const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm) {
  module_random_numbers(ctl, cache, atm->np, 3, 1);
  const int ct = met0->coord_type;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    stencil_t s = STENCIL_ZERO;
    const double pbl = time_2d(met0, met1, ORC_PBL, atm->time[ip], atm->lon[ip], atm->lat[ip], &s, 1);
    if (ctl->turb_pbl_scheme > 0 && atm->p[ip] >= pbl)
      continue;
    const double ps = time_2d(met0, met1, ORC_PS, atm->time[ip], atm->lon[ip], atm->lat[ip], &s, 0);
    const double ptop = met0->p[met0->np - 1];

    const double wpbl = orc_pbl_weight(ctl, atm->p[ip], pbl, ps);
    const double wtrop = orc_tropo_weight(ctl, clim, atm->time[ip], atm->lat[ip], atm->p[ip]) * (1.0 - wpbl);
    const double wstrat = 1.0 - wpbl - wtrop;
    const double Kx = wpbl * ctl->turb_dx_pbl + wtrop * ctl->turb_dx_trop + wstrat * ctl->turb_dx_strat;
    const double Kz = wpbl * ctl->turb_dz_pbl + wtrop * ctl->turb_dz_trop + wstrat * ctl->turb_dz_strat;
    const double dt_abs = fabs(cache->dt[ip]);

    if (Kx > 0) {
      const double sigma_h = sqrt(2.0 * Kx * dt_abs);
      atm->lon[ip] += dx2coord(ct, cache->rs[3 * (size_t) ip] * sigma_h, atm->lat[ip]);
      atm->lat[ip] += dy2coord(ct, cache->rs[3 * (size_t) ip + 1] * sigma_h);
    }

    if (Kz > 0) {
      const double sigma_z = sqrt(2.0 * Kz * dt_abs) * 1e-3;
      const double p_save = atm->p[ip];
      const double eps_km = 0.01;
      const double p_up = p_save + dz2dp(eps_km, p_save);
      const double p_dn = p_save + dz2dp(-eps_km, p_save);
      /* note: the latitude used here is the one already displaced above */
      const double Kz_up = kz_blend(ctl, clim, atm->time[ip], atm->lat[ip],
                                    dmax(ptop, dmin(ps, p_up)), pbl, ps);
      const double Kz_dn = kz_blend(ctl, clim, atm->time[ip], atm->lat[ip],
                                    dmax(ptop, dmin(ps, p_dn)), pbl, ps);
      const double dKz_dz = (Kz_up - Kz_dn) / (2.0 * eps_km * 1e3);
      const double dlnrho_dz = -1.0 / (1e3 * C_H0);
      const double w_drift = dKz_dz + Kz * dlnrho_dz;
      const double dz_drift = w_drift * dt_abs * 1e-3;
      const double dz_tot = cache->rs[3 * (size_t) ip + 2] * sigma_z + dz_drift;
      double ptrial = p_save + dz2dp(dz_tot, p_save);
      for (int iter = 0; iter < 10; iter++) {
        if (ptrial > ps)
          ptrial = ps * ps / ptrial;
        else if (ptrial < ptop)
          ptrial = ptop * ptop / ptrial;
        else
          break;
      }
      atm->p[ip] = dmax(ptop, dmin(ps, ptrial));
    }
  }
}

/* ---- module_diff_pbl (mptrac.c:4343-4584) ------------------------------- */

static inline double clampd(double v, double lo, double hi) {   /* CLAMP, mptrac.h:756 */
  return v < lo ? lo : (v > hi ? hi : v);
}

/* TVIRT, mptrac.h:2199 */
static inline double tvirt(double t, double h2o) {
  return t * (1. + (1. - C_EPS) * dmax(h2o, 0.1e-6));
}

void orc_module_diff_pbl(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm) {
  module_random_numbers(ctl, cache, atm->np, 3, 1);
  const int ct = met0->coord_type;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    double dsigw_dz = 0.0, sig_u = 0.0, sig_v = 0.0, sig_w = 0.0, tau_u = 0.0, tau_v = 0.0, tau_w = 0.0;
    const double tp = atm->time[ip], lon = atm->lon[ip], lat = atm->lat[ip];
    stencil_t s = STENCIL_ZERO;
    const double pbl = time_2d(met0, met1, ORC_PBL, tp, lon, lat, &s, 1);
    if (atm->p[ip] < pbl)
      continue;
    const double ps = time_2d(met0, met1, ORC_PS, tp, lon, lat, &s, 0);
    if (!(ps > 0.0 && pbl > 0.0 && ps > pbl))
      continue;
    const double p = dmin(atm->p[ip], ps);
    const double zs = zfromp(ps);
    const double z_raw = 1e3 * (zfromp(p) - zs);
    const double zi = 1e3 * (zfromp(pbl) - zs);
    if (!(zi > 1.0))
      continue;
    const double z = clampd(z_raw, 0.0, zi);
    const double zeta = clampd(z / zi, 1e-6, 1.0 - 1e-6);
    const double z_m = dmax(z, 1.0);

    const double ess = time_2d(met0, met1, ORC_ESS, tp, lon, lat, &s, 0);
    const double nss = time_2d(met0, met1, ORC_NSS, tp, lon, lat, &s, 0);
    /* INTPOL_3D at the clamped pressure (the reference overwrites atm->p temporarily) */
    const double t = time_3d(met0, met1, ORC_T, tp, p, lon, lat, &s, 1);
    const double h2o = time_3d(met0, met1, ORC_H2O, tp, p, lon, lat, &s, 0);

    const double tv = tvirt(t, h2o);
    const double thetav = tvirt(t * pow(1000. / p, C_KAPPA), dmax(h2o, 0.1e-6));   /* THETAVIRT */
    const double rho = rho_air(p, tv);
    const double tau = sqrt(SQ(ess) + SQ(nss));
    if (!(rho > 0.0))
      continue;
    const double ustar = sqrt(dmax(tau / rho, 0.0));
    const double ust = dmax(1e-4, ustar);
    const double shf = time_2d(met0, met1, ORC_SHF, tp, lon, lat, &s, 1);
    double ol = 1e12;
    if (fabs(shf) > 1e-6)
      ol = thetav * rho * C_CPD * SQ(ust) * ust / (C_KARMAN * C_G0 * shf);

    if (zi / fabs(ol) < 1.0) {          /* neutral */
      const double corr = z_m / ust;
      const double sigw0 = 1.3 * ust * exp(-2e-4 * corr);
      sig_u = dmax(2.0 * ust * exp(-3e-4 * corr), 1e-5);
      sig_v = dmax(sigw0, 1e-5);
      sig_w = dmax(sigw0, 1e-5);
      dsigw_dz = -2e-4 * sigw0 / ust;
      tau_u = 0.5 * z_m / sig_w / (1.0 + 1.5e-3 * corr);
      tau_v = tau_u;
      tau_w = tau_u;
    } else if (ol < 0.0) {              /* unstable */
      const double wstar_arg = -C_G0 / thetav * shf / (rho * C_CPD) * zi;
      const double wstar = pow(dmax(wstar_arg, 0.0), 1.0 / 3.0);
      double dsigw2_dz = 0.0;
      sig_u = dmax(ust * pow(dmax(12.0 - 0.5 * zi / ol, 0.0), 1.0 / 3.0), 1e-6);
      sig_v = sig_u;
      if (zeta < 0.03) {
        const double arg = dmax(3.0 * zeta - ol / zi, 1e-12);
        sig_w = 0.96 * wstar * pow(arg, 1.0 / 3.0);
        dsigw2_dz = 1.8432 * SQ(wstar) / zi * pow(arg, -1.0 / 3.0);
      } else if (zeta < 0.4) {
        const double arg = dmax(3.0 * zeta - ol / zi, 1e-12);
        const double s1 = 0.96 * pow(arg, 1.0 / 3.0);
        const double s2 = 0.763 * pow(zeta, 0.175);
        if (s1 < s2) {
          sig_w = wstar * s1;
          dsigw2_dz = 1.8432 * SQ(wstar) / zi * pow(arg, -1.0 / 3.0);
        } else {
          sig_w = wstar * s2;
          dsigw2_dz = 0.203759 * SQ(wstar) / zi * pow(zeta, -0.65);
        }
      } else if (zeta < 0.96) {
        sig_w = 0.722 * wstar * pow(1.0 - zeta, 0.207);
        dsigw2_dz = -0.215812 * SQ(wstar) / zi * pow(1.0 - zeta, -0.586);
      } else {
        sig_w = 0.37 * wstar;
        dsigw2_dz = 0.0;
      }
      sig_w = dmax(sig_w, 1e-6);
      dsigw_dz = sig_w > 1e-12 ? 0.5 * dsigw2_dz / sig_w : 0.0;
      tau_u = 0.15 * zi / dmax(sig_u, 1e-12);
      tau_v = tau_u;
      if (z_m < fabs(ol)) {
        const double denom = 0.55 - 0.38 * fabs(z_m / ol);
        tau_w = 0.1 * z_m / (sig_w * dmax(denom, 0.05));
      } else if (zeta < 0.1)
        tau_w = 0.59 * z_m / sig_w;
      else
        tau_w = 0.15 * zi / sig_w * (1.0 - exp(-5.0 * zeta));
    } else {                            /* stable */
      sig_u = dmax(2.0 * ust * (1.0 - zeta), 1e-6);
      sig_v = dmax(1.3 * ust * (1.0 - zeta), 1e-6);
      sig_w = dmax(1.3 * ust * (1.0 - zeta), 1e-6);
      dsigw_dz = -1.3 * ust / zi;
      tau_u = 0.15 * zi / sig_u * sqrt(zeta);
      tau_v = 0.467 * tau_u;
      tau_w = 0.1 * zi / sig_w * pow(zeta, 0.8);
    }
    tau_u = dmax(tau_u, 10.0);
    tau_v = dmax(tau_v, 10.0);
    tau_w = dmax(tau_w, 30.0);
    if (!(sig_u > 0.0 && sig_v > 0.0 && sig_w > 0.0 && tau_u > 0.0 && tau_v > 0.0 && tau_w > 0.0))
      continue;

    const double dt = cache->dt[ip];
    const double dt_abs = fabs(dt);
    float *uvwp = &cache->uvwp[3 * (size_t) ip];
    const double ru = exp(-dt_abs / tau_u);
    const double ru2 = sqrt(dmax(0.0, 1.0 - SQ(ru)));
    const double rv = exp(-dt_abs / tau_v);
    const double rv2 = sqrt(dmax(0.0, 1.0 - SQ(rv)));
    uvwp[0] = (float) (uvwp[0] * ru + sig_u * ru2 * cache->rs[3 * (size_t) ip]);
    uvwp[1] = (float) (uvwp[1] * rv + sig_v * rv2 * cache->rs[3 * (size_t) ip + 1]);
    const double rw = exp(-dt_abs / tau_w);
    const double rw2 = sqrt(dmax(0.0, 1.0 - SQ(rw)));
    const double rhoaux = -1.0 / (1e3 * C_H0);
    uvwp[2] = (float) (uvwp[2] * rw + sig_w * rw2 * cache->rs[3 * (size_t) ip + 2]
                       + tau_w * (1.0 - rw) * (2.0 * sig_w * dsigw_dz + rhoaux * SQ(sig_w)));
    atm->lon[ip] += dx2coord(ct, uvwp[0] * dt, atm->lat[ip]);
    atm->lat[ip] += dy2coord(ct, uvwp[1] * dt);
    double znew = z + uvwp[2] * dt;
    while (znew < 0.0 || znew > zi) {
      if (znew < 0.0) {
        znew = -znew;
        uvwp[2] = -uvwp[2];
      }
      if (znew > zi) {
        znew = 2.0 * zi - znew;
        uvwp[2] = -uvwp[2];
      }
    }
    atm->p[ip] = C_P0 * exp(-(zs + znew / 1000.0) / C_H0);     /* P(z), mptrac.h:1784 */
    atm->p[ip] = clampd(atm->p[ip], pbl, ps);
  }
}

/* ---- module_diff_meso (mptrac.c:4266-4339) ------------------------------ */

void orc_module_diff_meso(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                          const orc_met_t *met1, orc_atm_t *atm) {
  module_random_numbers(ctl, cache, atm->np, 3, 1);
  const int ct = met0->coord_type;
  const float *u0 = met0->f3[ORC_U], *v0 = met0->f3[ORC_V], *w0 = met0->f3[ORC_W];
  const float *u1 = met1->f3[ORC_U], *v1 = met1->f3[ORC_V], *w1 = met1->f3[ORC_W];
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    /* raw (un-wrapped) coordinates, mptrac.c:4283-4285 */
    const int ix = orc_locate_reg(met0->lon, met0->nx, atm->lon[ip]);
    const int iy = orc_locate_irr(met0->lat, met0->ny, atm->lat[ip]);
    const int iz = orc_locate_irr(met0->p, met0->np, atm->p[ip]);

    /* single-precision statistics over the 2x2x2x2 neighbourhood, loop order
     * i (lon), j (lat), k (level), met0 before met1 (mptrac.c:4288-4311) */
    float umean = 0, usig = 0, vmean = 0, vsig = 0, wmean = 0, wsig = 0;
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++)
        for (int k = 0; k < 2; k++) {
          const float a = A3(u0, met0, ix + i, iy + j, iz + k);
          const float b = A3(v0, met0, ix + i, iy + j, iz + k);
          const float c = A3(w0, met0, ix + i, iy + j, iz + k);
          umean += a;
          usig += a * a;
          vmean += b;
          vsig += b * b;
          wmean += c;
          wsig += c * c;
          const float d = A3(u1, met0, ix + i, iy + j, iz + k);
          const float e = A3(v1, met0, ix + i, iy + j, iz + k);
          const float f = A3(w1, met0, ix + i, iy + j, iz + k);
          umean += d;
          usig += d * d;
          vmean += e;
          vsig += e * e;
          wmean += f;
          wsig += f * f;
        }
    usig = usig / 16.f - SQ(umean / 16.f);
    usig = (usig > 0 ? sqrtf(usig) : 0);
    vsig = vsig / 16.f - SQ(vmean / 16.f);
    vsig = (vsig > 0 ? sqrtf(vsig) : 0);
    wsig = wsig / 16.f - SQ(wmean / 16.f);
    wsig = (wsig > 0 ? sqrtf(wsig) : 0);

    const double r = 1 - 2 * fabs(cache->dt[ip]) / ctl->dt_met;
    const double r2 = sqrt(1 - r * r);
    float *uvwp = &cache->uvwp[3 * (size_t) ip];

    if (ctl->turb_mesox > 0) {
      uvwp[0] = (float) (r * uvwp[0] + r2 * cache->rs[3 * (size_t) ip] * ctl->turb_mesox * usig);
      atm->lon[ip] += dx2coord(ct, uvwp[0] * cache->dt[ip], atm->lat[ip]);
      uvwp[1] = (float) (r * uvwp[1] + r2 * cache->rs[3 * (size_t) ip + 1] * ctl->turb_mesox * vsig);
      atm->lat[ip] += dy2coord(ct, uvwp[1] * cache->dt[ip]);
    }
    if (ctl->turb_mesoz > 0) {
      uvwp[2] = (float) (r * uvwp[2] + r2 * cache->rs[3 * (size_t) ip + 2] * ctl->turb_mesoz * wsig);
      atm->p[ip] += uvwp[2] * cache->dt[ip];
    }
  }
}

/* ---- module_convection (mptrac.c:4102-4171) ----------------------------- */

void orc_module_convection(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                           const orc_met_t *met1, orc_atm_t *atm) {
  module_random_numbers(ctl, cache, atm->np, 1, 0);
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    const double tp = atm->time[ip], lon = atm->lon[ip], lat = atm->lat[ip];
    stencil_t s = STENCIL_ZERO;
    const double ps = time_2d(met0, met1, ORC_PS, tp, lon, lat, &s, 1);
    double pbot = ps, ptop = ps;
    if (ctl->conv_mix_pbl) {
      const double pbl = time_2d(met0, met1, ORC_PBL, tp, lon, lat, &s, 0);
      ptop = pbl - ctl->conv_pbl_trans * (ps - pbl);
    }
    if (ctl->conv_cape >= 0) {
      const double cape = time_2d(met0, met1, ORC_CAPE, tp, lon, lat, &s, 0);
      const double cin = time_2d(met0, met1, ORC_CIN, tp, lon, lat, &s, 0);
      const double pel = time_2d(met0, met1, ORC_PEL, tp, lon, lat, &s, 0);
      if (isfinite(cape) && cape >= ctl->conv_cape
          && (ctl->conv_cin <= 0 || (isfinite(cin) && cin >= ctl->conv_cin)))
        ptop = dmin(ptop, pel);     /* GSL_MIN */
    }
    if (ptop != pbot && atm->p[ip] >= ptop) {
      const double tbot = time_3d(met0, met1, ORC_T, tp, pbot, lon, lat, &s, 1);
      const double ttop = time_3d(met0, met1, ORC_T, tp, ptop, lon, lat, &s, 1);
      const double rhobot = pbot / tbot;
      const double rhotop = ptop / ttop;
      const double rho = rhobot + (rhotop - rhobot) * cache->rs[ip];
      atm->p[ip] = lin(rhobot, pbot, rhotop, ptop, rho);
    }
  }
}

/* ---- module_sedi (mptrac.c:5859-5883) ----------------------------------- */

void orc_module_sedi(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                     const orc_met_t *met1, orc_atm_t *atm) {
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    stencil_t s = STENCIL_ZERO;
    const double t = time_3d(met0, met1, ORC_T, atm->time[ip], atm->p[ip], atm->lon[ip],
                             atm->lat[ip], &s, 1);
    const double v_s = orc_sedi(atm->p[ip], t, atm->q[ctl->qnt_rp][ip], atm->q[ctl->qnt_rhop][ip]);
    atm->p[ip] += dz2dp(v_s * cache->dt[ip] / 1000., atm->p[ip]);
  }
}

/* ---- module_decay (mptrac.c:4227-4262) ---------------------------------- */

void orc_module_decay(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_clim_t *clim,
                      orc_atm_t *atm) {
  if (ctl->qnt_m < 0 && ctl->qnt_vmr < 0)
    die("Module needs quantity mass or volume mixing ratio!");
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    const double w = orc_tropo_weight(ctl, clim, atm->time[ip], atm->lat[ip], atm->p[ip]);
    const double tdec = w * ctl->tdec_trop + (1 - w) * ctl->tdec_strat;
    const double aux = exp(-cache->dt[ip] / tdec);
    if (ctl->qnt_m >= 0) {
      if (ctl->qnt_mloss_decay >= 0)
        atm->q[ctl->qnt_mloss_decay][ip] += atm->q[ctl->qnt_m][ip] * (1 - aux);
      atm->q[ctl->qnt_m][ip] *= aux;
      if (ctl->qnt_loss_rate >= 0)
        atm->q[ctl->qnt_loss_rate][ip] += 1. / tdec;
    }
    if (ctl->qnt_vmr >= 0)
      atm->q[ctl->qnt_vmr][ip] *= aux;
  }
}

/* ---- module_mixing (mptrac.c:5169-5347) --------------------------------- */

static void mixing_one(const orc_ctl_t *ctl, const orc_clim_t *clim, orc_atm_t *atm,
                       const int *ixs, const int *iys, const int *izs, int qnt) {
  /* module_mixing_help, mptrac.c:5249-5347; the accumulation loop is serial
   * in the CPU build of the reference (no pragma at l.5289) */
  const int np = atm->np;
  const int ngrid = ctl->mixing_nx * ctl->mixing_ny * ctl->mixing_nz;
  const int use_ens = (ctl->nens > 0);
  const int nens = use_ens ? ctl->nens : 1;
  const size_t total = (size_t) ngrid * (size_t) nens;
  double *cmean = calloc(total, sizeof(double));
  int *count = calloc(total, sizeof(int));
  if (!cmean || !count)
    die("Out of memory!");
  for (int ip = 0; ip < np; ip++)
    if (izs[ip] >= 0) {
      const int ens = use_ens ? (int) atm->q[ctl->qnt_ens][ip] : 0;
      const int idx = ens * ngrid + (ixs[ip] * ctl->mixing_ny + iys[ip]) * ctl->mixing_nz + izs[ip];
      cmean[idx] += atm->q[qnt][ip];
      count[idx]++;
    }
  for (size_t i = 0; i < total; i++)
    if (count[i] > 0)
      cmean[i] /= count[i];
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < np; ip++)
    if (izs[ip] >= 0) {
      const int ens = use_ens ? (int) atm->q[ctl->qnt_ens][ip] : 0;
      double mixparam = 1.0;
      if (ctl->mixing_trop < 1 || ctl->mixing_strat < 1) {
        const double w = orc_tropo_weight(ctl, clim, atm->time[ip], atm->lat[ip], atm->p[ip]);
        mixparam = w * ctl->mixing_trop + (1.0 - w) * ctl->mixing_strat;
      }
      const int idx = ens * ngrid + (ixs[ip] * ctl->mixing_ny + iys[ip]) * ctl->mixing_nz + izs[ip];
      atm->q[qnt][ip] += (cmean[idx] - atm->q[qnt][ip]) * mixparam;
    }
  free(cmean);
  free(count);
}

void orc_module_mixing(const orc_ctl_t *ctl, const orc_clim_t *clim, orc_atm_t *atm, double t) {
  const int np = atm->np;
  int *ixs = malloc((size_t) np * sizeof(int));
  int *iys = malloc((size_t) np * sizeof(int));
  int *izs = malloc((size_t) np * sizeof(int));
  if (!ixs || !iys || !izs)
    die("Out of memory!");
  const double dz = (ctl->mixing_z1 - ctl->mixing_z0) / ctl->mixing_nz;
  const double dlon = (ctl->mixing_lon1 - ctl->mixing_lon0) / ctl->mixing_nx;
  const double dlat = (ctl->mixing_lat1 - ctl->mixing_lat0) / ctl->mixing_ny;
  const double t0 = t - 0.5 * ctl->dt_mod;
  const double t1 = t + 0.5 * ctl->dt_mod;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < np; ip++) {
    const double zpart = zfromp(atm->p[ip]);
    ixs[ip] = iys[ip] = 0;
    if (atm->time[ip] < t0 || atm->time[ip] > t1
        || atm->lon[ip] < ctl->mixing_lon0 || atm->lon[ip] >= ctl->mixing_lon1
        || atm->lat[ip] < ctl->mixing_lat0 || atm->lat[ip] >= ctl->mixing_lat1
        || zpart < ctl->mixing_z0 || zpart >= ctl->mixing_z1) {
      izs[ip] = -1;
      continue;
    }
    ixs[ip] = (int) ((atm->lon[ip] - ctl->mixing_lon0) / dlon);
    iys[ip] = (int) ((atm->lat[ip] - ctl->mixing_lat0) / dlat);
    izs[ip] = (int) ((zpart - ctl->mixing_z0) / dz);
    if (ixs[ip] >= ctl->mixing_nx || iys[ip] >= ctl->mixing_ny || izs[ip] >= ctl->mixing_nz)
      izs[ip] = -1;
  }
  /* of the reference's quantity list (mptrac.c:5223-5230): mass, volume mixing ratio, the five trace gases and
   * age of air -- the chemistry species and radionuclides of the list are not carried (every quantity is
   * mixed on its own, so the order does not matter) */
  const int quantities[8] = { ctl->qnt_m, ctl->qnt_vmr, ctl->qnt_tracer[ORC_TR_CCL4], ctl->qnt_tracer[ORC_TR_CCL3F],
    ctl->qnt_tracer[ORC_TR_CCL2F2], ctl->qnt_tracer[ORC_TR_N2O], ctl->qnt_tracer[ORC_TR_SF6], ctl->qnt_aoa };
  for (int i = 0; i < 8; i++)
    if (quantities[i] >= 0)
      mixing_one(ctl, clim, atm, ixs, iys, izs, quantities[i]);
  free(ixs);
  free(iys);
  free(izs);
}

/* ---- deposition (mptrac.c:6155-6290, 4738-4797) ------------------------- */

static void apply_loss(const orc_ctl_t *ctl, orc_atm_t *atm, int ip, double aux, int qnt_mloss,
                       double rate) {
  /* common tail of decay / wet / dry deposition (e.g. mptrac.c:6279-6288) */
  if (ctl->qnt_m >= 0) {
    if (qnt_mloss >= 0)
      atm->q[qnt_mloss][ip] += atm->q[ctl->qnt_m][ip] * (1 - aux);
    atm->q[ctl->qnt_m][ip] *= aux;
    if (ctl->qnt_loss_rate >= 0)
      atm->q[ctl->qnt_loss_rate][ip] += rate;
  }
  if (ctl->qnt_vmr >= 0)
    atm->q[ctl->qnt_vmr][ip] *= aux;
}

void orc_module_wet_depo(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm) {
  if (ctl->qnt_m < 0 && ctl->qnt_vmr < 0)
    die("Module needs quantity mass or volume mixing ratio!");
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    const double tp = atm->time[ip], lon = atm->lon[ip], lat = atm->lat[ip], p = atm->p[ip];
    stencil_t s = STENCIL_ZERO;
    const double pct = time_2d(met0, met1, ORC_PCT, tp, lon, lat, &s, 1);
    if (!isfinite(pct) || p <= pct)
      continue;
    const double pcb = time_2d(met0, met1, ORC_PCB, tp, lon, lat, &s, 0);
    const double cl = time_2d(met0, met1, ORC_CL, tp, lon, lat, &s, 0);
    const double Is = pow(1. / ctl->wet_depo_pre[0] * cl, 1. / ctl->wet_depo_pre[1]);
    if (Is < 0.01)
      continue;
    const double lwc = time_3d(met0, met1, ORC_LWC, tp, p, lon, lat, &s, 1);
    const double rwc = time_3d(met0, met1, ORC_RWC, tp, p, lon, lat, &s, 0);
    const double iwc = time_3d(met0, met1, ORC_IWC, tp, p, lon, lat, &s, 0);
    const double swc = time_3d(met0, met1, ORC_SWC, tp, p, lon, lat, &s, 0);
    const int inside = (lwc > 0 || rwc > 0 || iwc > 0 || swc > 0);
    const double t = time_3d(met0, met1, ORC_T, tp, p, lon, lat, &s, 0);

    double lambda = 0;
    if (inside) {
      double eta;
      if (t > C_WD_T_LIQUID)
        eta = 1;
      else if (t <= C_WD_T_ICE)
        eta = ctl->wet_depo_ic_ret_ratio;
      else
        eta = lin(C_WD_T_LIQUID, 1, C_WD_T_ICE, ctl->wet_depo_ic_ret_ratio, t);
      if (ctl->wet_depo_ic_a > 0)
        lambda = ctl->wet_depo_ic_a * pow(Is, ctl->wet_depo_ic_b) * eta;
      else if (ctl->wet_depo_ic_h[0] > 0) {
        double h = ctl->wet_depo_ic_h[0] * exp(ctl->wet_depo_ic_h[1] * (1. / t - 1. / C_TREF));
        if (ctl->wet_depo_so2_ph > 0) {
          const double H_ion = pow(10., -ctl->wet_depo_so2_ph);
          const double K_1 = C_SO2_K1_REF * exp(C_SO2_K1_TEMP * (1. / t - 1. / C_TREF));
          const double K_2 = C_SO2_K2_REF * exp(C_SO2_K2_TEMP * (1. / t - 1. / C_TREF));
          h *= (1. + K_1 / H_ion + K_1 * K_2 / SQ(H_ion));
        }
        const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
        lambda = h * C_RI * t * Is / 3.6e6 / dz * eta;
      }
    } else {
      const double eta = (t > C_WD_T_LIQUID_BC) ? 1 : ctl->wet_depo_bc_ret_ratio;
      if (ctl->wet_depo_bc_a > 0)
        lambda = ctl->wet_depo_bc_a * pow(Is, ctl->wet_depo_bc_b) * eta;
      else if (ctl->wet_depo_bc_h[0] > 0) {
        const double h = ctl->wet_depo_bc_h[0] * exp(ctl->wet_depo_bc_h[1] * (1. / t - 1. / C_TREF));
        const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
        lambda = h * C_RI * t * Is / 3.6e6 / dz * eta;
      }
    }
    const double aux = exp(-cache->dt[ip] * lambda);
    apply_loss(ctl, atm, ip, aux, ctl->qnt_mloss_wet, lambda);
  }
}

void orc_module_dry_depo(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm) {
  if (ctl->qnt_m < 0 && ctl->qnt_vmr < 0)
    die("Module needs quantity mass or volume mixing ratio!");
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    stencil_t s = STENCIL_ZERO;
    const double ps = time_2d(met0, met1, ORC_PS, atm->time[ip], atm->lon[ip], atm->lat[ip], &s, 1);
    if (atm->p[ip] < ps - ctl->dry_depo_dp)
      continue;
    const double dz = 1000. * (zfromp(ps - ctl->dry_depo_dp) - zfromp(ps));
    double v_dep;
    /* the reference tests the quantity indices with "> 0" here (mptrac.c:4769) */
    if (ctl->qnt_rp > 0 && ctl->qnt_rhop > 0) {
      const double t = time_3d(met0, met1, ORC_T, atm->time[ip], atm->p[ip], atm->lon[ip],
                               atm->lat[ip], &s, 1);
      v_dep = orc_sedi(atm->p[ip], t, atm->q[ctl->qnt_rp][ip], atm->q[ctl->qnt_rhop][ip]);
    } else
      v_dep = ctl->dry_depo_vdep;
    const double aux = exp(-cache->dt[ip] * v_dep / dz);
    apply_loss(ctl, atm, ip, aux, ctl->qnt_mloss_dry, v_dep / dz);
  }
}

/* ---- module_sort (mptrac.c:5887-5995) ----------------------------------- */

typedef struct {
  long long key;
  int idx;
} sort_item_t;

static int sort_cmp(const void *a, const void *b) {
  const sort_item_t *x = a, *y = b;
  if (x->key != y->key)
    return x->key < y->key ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);     /* ties: original index */
}

void orc_module_sort(const orc_ctl_t *ctl, const orc_met_t *met0, orc_atm_t *atm, double *keys,
                     int *perm) {
  const int np = atm->np;
  sort_item_t *items = malloc((size_t) np * sizeof(sort_item_t));
  double *help = malloc((size_t) np * sizeof(double));
  if (!items || !help)
    die("Out of memory!");
  /* key on raw coordinates, mptrac.c:5913-5919 */
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < np; ip++) {
    const int k = (orc_locate_reg(met0->lon, met0->nx, atm->lon[ip]) * met0->ny
                   + orc_locate_irr(met0->lat, met0->ny, atm->lat[ip])) * met0->np
      + orc_locate_irr(met0->p, met0->np, atm->p[ip]);
    items[ip].key = k;
    items[ip].idx = ip;
    if (keys)
      keys[ip] = (double) k;
  }
  qsort(items, (size_t) np, sizeof(sort_item_t), sort_cmp);
  if (perm)
    for (int ip = 0; ip < np; ip++)
      perm[ip] = items[ip].idx;
  /* gather every array through the permutation, mptrac.c:5944-5949, 5980-5988 */
  double *arrays[4 + ORC_NQ_MAX];
  int na = 0;
  arrays[na++] = atm->time;
  arrays[na++] = atm->p;
  arrays[na++] = atm->lon;
  arrays[na++] = atm->lat;
  for (int iq = 0; iq < ctl->nq; iq++)
    arrays[na++] = atm->q[iq];
  for (int a = 0; a < na; a++) {
#pragma omp parallel for schedule(static)
    for (int ip = 0; ip < np; ip++)
      help[ip] = arrays[a][items[ip].idx];
    memcpy(arrays[a], help, (size_t) np * sizeof(double));
  }
  free(items);
  free(help);
}

/* ---- module_meteo (mptrac.c:5062-5165) ---------------------------------- */

#define C_LV 2501000.   /* mptrac.h:275 */

static inline double pw_of(double p, double h2o) {   /* PW, mptrac.h:1859 */
  return p * dmax(h2o, 0.1e-6) / (1. + (1. - C_EPS) * dmax(h2o, 0.1e-6));
}

static inline double psat_of(double t) {   /* PSAT, mptrac.h:1808 */
  return 6.112 * exp(17.62 * (t - C_T0) / (243.12 + t - C_T0));
}

static inline double psice_of(double t) {   /* PSICE, mptrac.h:1832 */
  return 6.112 * exp(22.46 * (t - C_T0) / (272.62 + t - C_T0));
}

static inline double sh_of(double h2o) {   /* SH, mptrac.h:2024 */
  return C_EPS * dmax(h2o, 0.1e-6);
}

double orc_rh(double p, double t, double h2o) {   /* RH, mptrac.h:1906 */
  return pw_of(p, h2o) / psat_of(t) * 100.;
}

double orc_rhice(double p, double t, double h2o) {   /* RHICE, mptrac.h:1936 */
  return pw_of(p, h2o) / psice_of(t) * 100.;
}

double orc_tdew(double p, double h2o) {   /* TDEW, mptrac.h:2075 */
  return C_T0 + 243.12 * log(pw_of(p, h2o) / 6.112) / (17.62 - log(pw_of(p, h2o) / 6.112));
}

double orc_tice(double p, double h2o) {   /* TICE, mptrac.h:2100 */
  return C_T0 + 272.62 * log(pw_of(p, h2o) / 6.112) / (22.46 - log(pw_of(p, h2o) / 6.112));
}

double orc_theta(double p, double t) {   /* THETA, mptrac.h:2124 */
  return t * pow(1000. / p, C_KAPPA);
}

double orc_zeta(double ps, double p, double t) {   /* ZETA, mptrac.h:2293 */
  return (p / ps <= 0.3 ? 1. : sin(M_PI / 2. * (1. - p / ps) / (1. - 0.3))) * orc_theta(p, t);
}

double orc_lapse_rate(double t, double h2o) {   /* lapse_rate, mptrac.c:3324-3338 */
  const double a = C_RA * SQ(t), r = sh_of(h2o) / (1. - sh_of(h2o));
  return 1e3 * C_G0 * (a + C_LV * r * t) / (C_CPD * a + SQ(C_LV) * r * C_EPS);
}

/* clim_zm, mptrac.c:414-466: a zonal-mean climatology at (time of year, latitude, pressure), clamped to the
 * table in pressure and latitude, linear in pressure, latitude and time, never negative */
double orc_clim_zm(const orc_zm_t *zm, double t, double lat, double p) {
  double sec = fmod_trunc(t, 365.25 * 86400.);
  while (sec < 0)
    sec += 365.25 * 86400.;
  double p_help = p;
  if (p < zm->p[zm->np - 1])
    p_help = zm->p[zm->np - 1];
  else if (p > zm->p[0])
    p_help = zm->p[0];
  double lat_help = lat;
  if (lat < zm->lat[0])
    lat_help = zm->lat[0];
  else if (lat > zm->lat[zm->nlat - 1])
    lat_help = zm->lat[zm->nlat - 1];
  const int isec = orc_locate_irr(zm->time, zm->ntime, sec);
  const int ilat = orc_locate_reg(zm->lat, zm->nlat, lat_help);
  const int ip = orc_locate_irr(zm->p, zm->np, p_help);
#define VMR(it, iz, iy) zm->vmr[((size_t) (it) * (size_t) zm->np + (size_t) (iz)) * (size_t) zm->nlat + (size_t) (iy)]
  const double aux00 = lin(zm->p[ip], VMR(isec, ip, ilat), zm->p[ip + 1], VMR(isec, ip + 1, ilat), p_help);
  const double aux01 = lin(zm->p[ip], VMR(isec, ip, ilat + 1), zm->p[ip + 1], VMR(isec, ip + 1, ilat + 1), p_help);
  const double aux10 = lin(zm->p[ip], VMR(isec + 1, ip, ilat), zm->p[ip + 1], VMR(isec + 1, ip + 1, ilat), p_help);
  const double aux11 =
    lin(zm->p[ip], VMR(isec + 1, ip, ilat + 1), zm->p[ip + 1], VMR(isec + 1, ip + 1, ilat + 1), p_help);
#undef VMR
  const double aux0 = lin(zm->lat[ilat], aux00, zm->lat[ilat + 1], aux01, lat_help);
  const double aux1 = lin(zm->lat[ilat], aux10, zm->lat[ilat + 1], aux11, lat_help);
  const double aux = lin(zm->time[isec], aux0, zm->time[isec + 1], aux1, sec);
  return aux > 0.0 ? aux : 0.0;
}

/* cos_sza, mptrac.c:1857-1897: cosine of the solar zenith angle (low-precision almanac formulas) */
double orc_cos_sza(double sec, double lon, double lat) {
  const double D = sec / 86400 - 0.5;
  const double g = deg2rad(357.529 + 0.98560028 * D);
  const double q = 280.459 + 0.98564736 * D;
  const double L = deg2rad(q + 1.915 * sin(g) + 0.020 * sin(2 * g));
  const double e = deg2rad(23.439 - 0.00000036 * D);
  const double sindec = sin(e) * sin(L);
  const double ra = atan2(cos(e) * sin(L), cos(L));
  const double GMST = 18.697374558 + 24.06570982441908 * D;
  const double LST = GMST + lon / 15;
  const double h = LST / 12 * M_PI - ra;
  const double lat_help = deg2rad(lat);
  return sin(lat_help) * sindec + cos(lat_help) * sqrt(1 - SQ(sindec)) * cos(h);
}

/* clim_oh, mptrac.c:89-120: the OH climatology with the optional diurnal scaling exp(-beta / cos(sza)) */
double orc_clim_oh(const orc_ctl_t *ctl, const orc_clim_t *clim, double t, double lon, double lat, double p) {
  const double csza_thresh = cos(deg2rad(85.));
  const double lat_ref = ctl->met_coord_type == 0 ? lat : ctl->met_utm_ref_lat;
  double lon_ref = ctl->met_coord_type == 0 ? lon : ctl->met_utm_ref_lon;
  while (lon_ref < -180.0)
    lon_ref += 360.0;
  while (lon_ref >= 180.0)
    lon_ref -= 360.0;
  const double oh = orc_clim_zm(&clim->zm[ORC_ZM_OH], t, lat_ref, p);
  if (ctl->oh_chem_beta <= 0)
    return oh;
  const double csza = orc_cos_sza(t, lon_ref, lat_ref);
  const double denom = (csza >= csza_thresh) ? csza : csza_thresh;
  return oh * exp(-ctl->oh_chem_beta / denom);
}

/* nat_temperature, mptrac.c:8334-8355: existence temperature of nitric acid trihydrate (Hanson and Mauersberger) */
double orc_nat_temperature(double p, double h2o, double hno3) {
  const double h2o_help = h2o > 0.1e-6 ? h2o : 0.1e-6;
  const double p_hno3 = hno3 * p / 1.333224;
  const double p_h2o = h2o_help * p / 1.333224;
  const double a = 0.009179 - 0.00088 * log10(p_h2o);
  const double b = (38.9855 - log10(p_hno3) - 2.7836 * log10(p_h2o)) / a;
  const double c = -11397.0 / a;
  double tnat = (-b + sqrt(b * b - 4. * c)) / 2.;
  const double x2 = (-b - sqrt(b * b - 4. * c)) / 2.;
  if (x2 > 0)
    tnat = x2;
  return tnat;
}

/* A field the caller did not provide reads as the zero-initialised met_t array
 * of the reference (mptrac_alloc uses calloc) and interpolates to exactly 0. */
#define M3(f, init) ((met0->f3[f] && met1->f3[f]) ? time_3d(met0, met1, f, tm, p, lon, lat, &s, init) : 0.0)
#define M2(f) ((met0->f2[f] && met1->f2[f]) ? time_2d(met0, met1, f, tm, lon, lat, &s, 0) : 0.0)
#define SETQ(k, val) if (ctl->qnt_met[k] >= 0) atm->q[ctl->qnt_met[k]][ip] = (val)

void orc_module_meteo(const orc_ctl_t *ctl, const orc_clim_t *clim, const orc_met_t *met0, const orc_met_t *met1,
                      orc_atm_t *atm) {
  /* mptrac.c:5074-5076 ("Need T_ice and T_NAT to calculate T_STS!"): the caller's business here */
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {   /* PARTICLE_LOOP with check_dt = 0 */
    const double tm = atm->time[ip], p = atm->p[ip], lon = atm->lon[ip], lat = atm->lat[ip];
    stencil_t s = STENCIL_ZERO;
    /* INTPOL_TIME_ALL (mptrac.h:1278-1318): the first call sets indices and weights, every other one
     * (3-D and 2-D alike) re-uses them */
    stencil_init_3d(met0, p, lon, lat, &s);
    const double z = M3(ORC_Z, 0);
    const double t = M3(ORC_T, 0), u = M3(ORC_U, 0), v = M3(ORC_V, 0), w = M3(ORC_W, 0);
    const double pv = M3(ORC_PV, 0), h2o = M3(ORC_H2O, 0), o3 = M3(ORC_O3, 0);
    const double lwc = M3(ORC_LWC, 0), rwc = M3(ORC_RWC, 0), iwc = M3(ORC_IWC, 0), swc = M3(ORC_SWC, 0);
    const double cc = M3(ORC_CC, 0);
    const double ps = M2(ORC_PS), ts = M2(ORC_TS), zs = M2(ORC_ZS), us = M2(ORC_US), vs = M2(ORC_VS);
    const double ess = M2(ORC_ESS), nss = M2(ORC_NSS), shf = M2(ORC_SHF), lsm = M2(ORC_LSM);
    const double sst = M2(ORC_SST), pbl = M2(ORC_PBL), pt = M2(ORC_PT), tt = M2(ORC_TT), zt = M2(ORC_ZT);
    const double h2ot = M2(ORC_H2OT), pct = M2(ORC_PCT), pcb = M2(ORC_PCB), cl = M2(ORC_CL);
    const double plcl = M2(ORC_PLCL), plfc = M2(ORC_PLFC), pel = M2(ORC_PEL), cape = M2(ORC_CAPE);
    const double cin = M2(ORC_CIN), o3c = M2(ORC_O3C);

    SETQ(ORC_MQ_PS, ps);
    SETQ(ORC_MQ_TS, ts);
    SETQ(ORC_MQ_ZS, zs);
    SETQ(ORC_MQ_US, us);
    SETQ(ORC_MQ_VS, vs);
    SETQ(ORC_MQ_ESS, ess);
    SETQ(ORC_MQ_NSS, nss);
    SETQ(ORC_MQ_SHF, shf);
    SETQ(ORC_MQ_LSM, lsm);
    SETQ(ORC_MQ_SST, sst);
    SETQ(ORC_MQ_PBL, pbl);
    SETQ(ORC_MQ_PT, pt);
    SETQ(ORC_MQ_TT, tt);
    SETQ(ORC_MQ_ZT, zt);
    SETQ(ORC_MQ_H2OT, h2ot);
    SETQ(ORC_MQ_ZG, z);
    SETQ(ORC_MQ_P, p);
    SETQ(ORC_MQ_T, t);
    SETQ(ORC_MQ_RHO, rho_air(p, t));
    SETQ(ORC_MQ_U, u);
    SETQ(ORC_MQ_V, v);
    SETQ(ORC_MQ_W, w);
    SETQ(ORC_MQ_H2O, h2o);
    SETQ(ORC_MQ_O3, o3);
    SETQ(ORC_MQ_LWC, lwc);
    SETQ(ORC_MQ_RWC, rwc);
    SETQ(ORC_MQ_IWC, iwc);
    SETQ(ORC_MQ_SWC, swc);
    SETQ(ORC_MQ_CC, cc);
    SETQ(ORC_MQ_PCT, pct);
    SETQ(ORC_MQ_PCB, pcb);
    SETQ(ORC_MQ_CL, cl);
    SETQ(ORC_MQ_PLCL, plcl);
    SETQ(ORC_MQ_PLFC, plfc);
    SETQ(ORC_MQ_PEL, pel);
    SETQ(ORC_MQ_CAPE, cape);
    SETQ(ORC_MQ_CIN, cin);
    SETQ(ORC_MQ_O3C, o3c);
    SETQ(ORC_MQ_VH, sqrt(u * u + v * v));
    SETQ(ORC_MQ_VZ, -1e3 * C_H0 / p * w);
    SETQ(ORC_MQ_PSAT, psat_of(t));
    SETQ(ORC_MQ_PSICE, psice_of(t));
    SETQ(ORC_MQ_PW, pw_of(p, h2o));
    SETQ(ORC_MQ_SH, sh_of(h2o));
    SETQ(ORC_MQ_RH, orc_rh(p, t, h2o));
    SETQ(ORC_MQ_RHICE, orc_rhice(p, t, h2o));
    SETQ(ORC_MQ_THETA, orc_theta(p, t));
    SETQ(ORC_MQ_ZETA_D, orc_zeta(ps, p, t));
    SETQ(ORC_MQ_TVIRT, tvirt(t, h2o));
    SETQ(ORC_MQ_LAPSE, orc_lapse_rate(t, h2o));
    SETQ(ORC_MQ_PV, pv);
    SETQ(ORC_MQ_TDEW, orc_tdew(p, h2o));
    SETQ(ORC_MQ_TICE, orc_tice(p, h2o));
    /* the climatology part of the list (mptrac.c:5129-5141, 5158-5163); a table is only touched for a
     * requested quantity */
    const double lat_ref = ctl->met_coord_type == 0 ? lat : ctl->met_utm_ref_lat;
    SETQ(ORC_MQ_HNO3, orc_clim_zm(&clim->zm[ORC_ZM_HNO3], tm, lat_ref, p));
    SETQ(ORC_MQ_OH, orc_clim_oh(ctl, clim, tm, lon, lat, p));
    SETQ(ORC_MQ_H2O2, orc_clim_zm(&clim->zm[ORC_ZM_H2O2], tm, lat_ref, p));
    SETQ(ORC_MQ_HO2, orc_clim_zm(&clim->zm[ORC_ZM_HO2], tm, lat_ref, p));
    SETQ(ORC_MQ_O1D, orc_clim_zm(&clim->zm[ORC_ZM_O1D], tm, lat_ref, p));
    SETQ(ORC_MQ_TNAT, orc_nat_temperature(p, h2o, orc_clim_zm(&clim->zm[ORC_ZM_HNO3], tm, lat, p)));
    if (ctl->qnt_met[ORC_MQ_TSTS] >= 0)   /* from the two quantities as stored */
      atm->q[ctl->qnt_met[ORC_MQ_TSTS]][ip] =
        0.5 * (atm->q[ctl->qnt_met[ORC_MQ_TICE]][ip] + atm->q[ctl->qnt_met[ORC_MQ_TNAT]][ip]);
  }
}

#undef M3
#undef M2
#undef SETQ

/* ---- module_isosurf (mptrac.c:4886-5005) -------------------------------- */

/* module_isosurf_init, modes 1-3 (mode 4 reads the balloon file on the host: the
 * caller fills cache->iso_ts / iso_ps / iso_n) */
void orc_module_isosurf_init(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                             const orc_met_t *met1, const orc_atm_t *atm) {
  if (ctl->isosurf < 1 || ctl->isosurf > 3)
    return;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (ctl->isosurf == 1)
      cache->iso_var[ip] = atm->p[ip];
    else {
      stencil_t s = STENCIL_ZERO;
      const double t = time_3d(met0, met1, ORC_T, atm->time[ip], atm->p[ip], atm->lon[ip], atm->lat[ip], &s, 1);
      cache->iso_var[ip] = ctl->isosurf == 2 ? atm->p[ip] / t : orc_theta(atm->p[ip], t);
    }
  }
}

void orc_module_isosurf(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                        const orc_met_t *met1, orc_atm_t *atm) {
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {   /* check_dt = 0 */
    if (ctl->isosurf == 1)
      atm->p[ip] = cache->iso_var[ip];
    else if (ctl->isosurf == 2 || ctl->isosurf == 3) {
      stencil_t s = STENCIL_ZERO;
      const double t = time_3d(met0, met1, ORC_T, atm->time[ip], atm->p[ip], atm->lon[ip], atm->lat[ip], &s, 1);
      if (ctl->isosurf == 2)
        atm->p[ip] = cache->iso_var[ip] * t;
      else
        atm->p[ip] = 1000. * pow(cache->iso_var[ip] / t, -1. / C_KAPPA);
    } else if (ctl->isosurf == 4) {
      if (atm->time[ip] <= cache->iso_ts[0])
        atm->p[ip] = cache->iso_ps[0];
      else if (atm->time[ip] >= cache->iso_ts[cache->iso_n - 1])
        atm->p[ip] = cache->iso_ps[cache->iso_n - 1];
      else {
        const int idx = orc_locate_irr(cache->iso_ts, cache->iso_n, atm->time[ip]);
        atm->p[ip] = lin(cache->iso_ts[idx], cache->iso_ps[idx], cache->iso_ts[idx + 1], cache->iso_ps[idx + 1],
                         atm->time[ip]);
      }
    }
  }
}

/* clim_ts, mptrac.c:394-410: a time series at time t, constant beyond its ends */
double orc_clim_ts(const orc_ts_t *ts, double t) {
  if (t <= ts->time[0])
    return ts->vmr[0];
  else if (t >= ts->time[ts->ntime - 1])
    return ts->vmr[ts->ntime - 1];
  const int idx = orc_locate_irr(ts->time, ts->ntime, t);
  return lin(ts->time[idx], ts->vmr[idx], ts->time[idx + 1], ts->vmr[idx + 1], t);
}

/* ---- module_bound_cond (mptrac.c:3789-3881): mass, volume mixing ratio, the trace gases with a surface time
 * series and age of air ------------------------------------------------------------------------------------- */

void orc_module_bound_cond(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_clim_t *clim,
                           const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm) {
  /* mptrac.c:3800-3804, as written there: the CFC-10 index is tested for "non-zero", not for "absent" -- with
   * Cccl4 as the only quantity of the list the module runs if and only if it is quantity 0 */
  const int *tr = ctl->qnt_tracer;
  if (ctl->qnt_m < 0 && ctl->qnt_vmr < 0 && tr[ORC_TR_CCL4] && tr[ORC_TR_CCL3F] < 0 && tr[ORC_TR_CCL2F2] < 0
      && tr[ORC_TR_N2O] < 0 && tr[ORC_TR_SF6] < 0 && ctl->qnt_aoa < 0)
    return;
#pragma omp parallel for schedule(static)
  for (int ip = 0; ip < atm->np; ip++) {
    if (cache->dt[ip] == 0)
      continue;
    if (atm->lat[ip] < ctl->bound_lat0 || atm->lat[ip] > ctl->bound_lat1
        || atm->p[ip] > ctl->bound_p0 || atm->p[ip] < ctl->bound_p1)
      continue;
    if (ctl->bound_dps > 0 || ctl->bound_dzs > 0 || ctl->bound_zetas > 0 || ctl->bound_pbl) {
      stencil_t s = STENCIL_ZERO;
      const double ps = time_2d(met0, met1, ORC_PS, atm->time[ip], atm->lon[ip], atm->lat[ip], &s, 1);
      if (ctl->bound_dps > 0 && atm->p[ip] < ps - ctl->bound_dps)
        continue;
      if (ctl->bound_dzs > 0 && zfromp(atm->p[ip]) > zfromp(ps) + ctl->bound_dzs)
        continue;
      if (ctl->bound_zetas > 0) {
        const double t = time_3d(met0, met1, ORC_T, atm->time[ip], atm->p[ip], atm->lon[ip], atm->lat[ip], &s, 1);
        if (orc_zeta(ps, atm->p[ip], t) > ctl->bound_zetas)
          continue;
      }
      if (ctl->bound_pbl) {
        const double pbl = time_2d(met0, met1, ORC_PBL, atm->time[ip], atm->lon[ip], atm->lat[ip], &s, 0);
        if (atm->p[ip] < pbl)
          continue;
      }
    }
    if (ctl->qnt_m >= 0 && ctl->bound_mass >= 0)
      atm->q[ctl->qnt_m][ip] = ctl->bound_mass + ctl->bound_mass_trend * atm->time[ip];
    if (ctl->qnt_vmr >= 0 && ctl->bound_vmr >= 0)
      atm->q[ctl->qnt_vmr][ip] = ctl->bound_vmr + ctl->bound_vmr_trend * atm->time[ip];
    for (int k = 0; k < ORC_NTR; k++)   /* mptrac.c:3857-3875 */
      if (tr[k] >= 0 && clim->ts[k].ntime > 0)
        atm->q[tr[k]][ip] = orc_clim_ts(&clim->ts[k], atm->time[ip]);
    if (ctl->qnt_aoa >= 0)
      atm->q[ctl->qnt_aoa][ip] = atm->time[ip];
  }
}

/* ---- scheduler: mptrac_run_timestep (mptrac.c:7851-8001) ---------------- */

void orc_run_timestep(orc_ctl_t *ctl, orc_cache_t *cache, const orc_clim_t *clim,
                      const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm, double t) {
  if (t == ctl->t_start) {   /* mptrac.c:7863-7874 */
    orc_module_isosurf_init(ctl, cache, met0, met1, atm);
    orc_module_advect_init(ctl, met0, met1, atm);
  }
  orc_module_timesteps(ctl, cache, met0, atm, t);
  if (cache->ip_global && (ctl->sort_dt > 0 || (ctl->mixing_trop >= 0 && ctl->mixing_strat >= 0)))
    die("a subsample (ip_global) cannot follow module_sort (slots are rebound) or module_mixing (cell means)");
  if (ctl->sort_dt > 0 && fmod(t, ctl->sort_dt) == 0)
    orc_module_sort(ctl, met0, atm, NULL, NULL);
  orc_module_position(cache, met0, met1, atm);
  if (ctl->advect > 0)
    orc_module_advect(ctl, cache, met0, met1, atm);
  if (ctl->diffusion
      && (ctl->turb_dx_pbl > 0 || ctl->turb_dz_pbl > 0 || ctl->turb_dx_trop > 0
          || ctl->turb_dz_trop > 0 || ctl->turb_dx_strat > 0 || ctl->turb_dz_strat > 0))
    orc_module_diff_turb(ctl, cache, clim, met0, met1, atm);
  if (ctl->diffusion && ctl->turb_pbl_scheme == 1)
    orc_module_diff_pbl(ctl, cache, met0, met1, atm);
  if (ctl->diffusion && (ctl->turb_mesox > 0 || ctl->turb_mesoz > 0))
    orc_module_diff_meso(ctl, cache, met0, met1, atm);
  if ((ctl->conv_mix_pbl || ctl->conv_cape >= 0)
      && (ctl->conv_dt <= 0 || fmod(t, ctl->conv_dt) == 0))
    orc_module_convection(ctl, cache, met0, met1, atm);
  if (ctl->qnt_rp >= 0 && ctl->qnt_rhop >= 0)
    orc_module_sedi(ctl, cache, met0, met1, atm);
  if (ctl->isosurf >= 1 && ctl->isosurf <= 4)
    orc_module_isosurf(ctl, cache, met0, met1, atm);   /* mptrac.c:7914-7916 */
  orc_module_position(cache, met0, met1, atm);
  if (ctl->met_dt_out > 0 && (ctl->met_dt_out < ctl->dt_mod || fmod(t, ctl->met_dt_out) == 0))
    orc_module_meteo(ctl, clim, met0, met1, atm);   /* mptrac.c:7921-7924 */
  if (ctl->bound_lat0 < ctl->bound_lat1 && ctl->bound_p0 > ctl->bound_p1)
    orc_module_bound_cond(ctl, cache, clim, met0, met1, atm);   /* mptrac.c:7926-7929 */
  /* zero the total loss rate, mptrac.c:7932-7936 */
  if (ctl->qnt_loss_rate >= 0)
    for (int ip = 0; ip < atm->np; ip++)
      if (cache->dt[ip] != 0)
        atm->q[ctl->qnt_loss_rate][ip] = 0;
  if (ctl->tdec_trop > 0 && ctl->tdec_strat > 0)
    orc_module_decay(ctl, cache, clim, atm);
  if (ctl->mixing_trop >= 0 && ctl->mixing_strat >= 0
      && (ctl->mixing_dt <= 0 || fmod(t, ctl->mixing_dt) == 0))
    orc_module_mixing(ctl, clim, atm, t);
  if ((ctl->wet_depo_ic_a > 0 || ctl->wet_depo_ic_h[0] > 0)
      && (ctl->wet_depo_bc_a > 0 || ctl->wet_depo_bc_h[0] > 0))
    orc_module_wet_depo(ctl, cache, met0, met1, atm);
  if (ctl->dry_depo_vdep > 0)
    orc_module_dry_depo(ctl, cache, met0, met1, atm);
  if (ctl->bound_lat0 < ctl->bound_lat1 && ctl->bound_p0 > ctl->bound_p1)
    orc_module_bound_cond(ctl, cache, clim, met0, met1, atm);   /* mptrac.c:7997-8000 */
}

/* ---- write_grid binning (mptrac.c:13815-13872) -------------------------- */

/* kernel_weight, mptrac.c:3298-3320 */
static double kernel_weight(const double *kz, const double *kw, int nk, double p) {
  if (nk < 2)
    return 1.0;
  const double z = zfromp(p);
  if (z < kz[0])
    return kw[0];
  else if (z > kz[nk - 1])
    return kw[nk - 1];
  else {
    const int idx = orc_locate_irr(kz, nk, z);
    return lin(kz[idx], kw[idx], kz[idx + 1], kw[idx + 1], z);
  }
}

void orc_grid_sums(const orc_ctl_t *ctl, const orc_atm_t *atm, double t, int *cnt, double *mean,
                   double *sigma) {
  orc_grid_sums_kernel(ctl, atm, t, 0, NULL, NULL, cnt, mean, sigma);
}

void orc_grid_sums_kernel(const orc_ctl_t *ctl, const orc_atm_t *atm, double t, int nk, const double *kz,
                          const double *kw, int *cnt, double *mean, double *sigma) {
  const size_t ncell = (size_t) ctl->grid_nx * (size_t) ctl->grid_ny * (size_t) ctl->grid_nz;
  memset(cnt, 0, ncell * sizeof(int));
  memset(mean, 0, ncell * (size_t) ctl->nq * sizeof(double));
  memset(sigma, 0, ncell * (size_t) ctl->nq * sizeof(double));
  const double dz = (ctl->grid_z1 - ctl->grid_z0) / ctl->grid_nz;
  const double dlon = (ctl->grid_lon1 - ctl->grid_lon0) / ctl->grid_nx;
  const double dlat = (ctl->grid_lat1 - ctl->grid_lat0) / ctl->grid_ny;
  const double t0 = t - 0.5 * ctl->dt_mod;
  const double t1 = t + 0.5 * ctl->dt_mod;
  for (int ip = 0; ip < atm->np; ip++) {
    const double zpart = zfromp(atm->p[ip]);
    if (atm->time[ip] < t0 || atm->time[ip] > t1
        || atm->lon[ip] < ctl->grid_lon0 || atm->lon[ip] >= ctl->grid_lon1
        || atm->lat[ip] < ctl->grid_lat0 || atm->lat[ip] >= ctl->grid_lat1
        || zpart < ctl->grid_z0 || zpart >= ctl->grid_z1)
      continue;
    const int ix = (int) ((atm->lon[ip] - ctl->grid_lon0) / dlon);
    const int iy = (int) ((atm->lat[ip] - ctl->grid_lat0) / dlat);
    const int iz = (int) ((zpart - ctl->grid_z0) / dz);
    if (ix >= ctl->grid_nx || iy >= ctl->grid_ny || iz >= ctl->grid_nz)
      continue;
    const size_t idx = ((size_t) ix * (size_t) ctl->grid_ny + (size_t) iy) * (size_t) ctl->grid_nz + (size_t) iz;
    const double kernel = kernel_weight(kz, kw, nk, atm->p[ip]);   /* mptrac.c:13866 */
    cnt[idx]++;
    for (int iq = 0; iq < ctl->nq; iq++) {
      mean[(size_t) iq * ncell + idx] += kernel * atm->q[iq][ip];
      sigma[(size_t) iq * ncell + idx] += SQ(kernel * atm->q[iq][ip]);
    }
  }
}
