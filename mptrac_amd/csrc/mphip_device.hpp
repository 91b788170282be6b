// mphip_device.hpp -- device-side building blocks of the MPTRAC time-step loop
// for gfx950 (MI355X): arithmetic conventions, axis search, packed-grid
// interpolation, counter-based random numbers, and one function per reference
// module.  All state of one particle lives in registers; the only memory
// traffic is the SoA particle arrays (coalesced) and the gathers from the
// packed meteo grids.
//
// Reference citations are relative to the reference repository
// (mptrac.c = src/mptrac.c, mptrac.h = src/mptrac.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The libraries are compiled with -fvisibility=hidden: only the C ABI is exported.  (Two builds of these sources can then
// live in one process -- LD_PRELOAD=libmptrac_hip_exact.so in front of a program linked against libmptrac_hip.so --
// without the kernels' host stubs of one interposing on the other's.)
#pragma GCC visibility push(default)
#include "../../include/mptrac_hip.h"
#pragma GCC visibility pop
#include "mphip_libm.h"
#include "mphip_libmtab.h"

namespace mphip {

// ---- exp / log / pow with the C library's bits (mphip_libm.h) ---------------
// The three 128-entry tables as one object in device memory: {invc, logc} of log, {tail, scale} of exp, {invc, logc,
// logctail} of pow -- 7 kB.  Kernels that call them per particle copy the part they use to LDS (the step kernel: log
// and exp, 4 kB; with the boundary-layer closure all of it); everything else reads the object through the caches.
// A table pointer (`lt` below, `ltab` in the modules) is the address of the object or of such a copy.
struct LibmBlob {
  double log_tab[2 * MPHIP_LIBM_N];
  uint64_t exp_tab[2 * MPHIP_LIBM_N];
  double pow_tab[3 * MPHIP_LIBM_N];
};
static __device__ const LibmBlob g_libm = { { MPHIP_LIBM_LOG_TAB_INIT }, { MPHIP_LIBM_EXP_TAB_INIT }, { MPHIP_LIBM_POW_TAB_INIT } };
constexpr int kLibmLogExpDoubles = 4 * MPHIP_LIBM_N;     // log + exp, what the lean kernels copy
constexpr int kLibmDoubles = 7 * MPHIP_LIBM_N;           // all three

__device__ __forceinline__ const double *libm_tables() {
  return g_libm.log_tab;
}

__device__ __forceinline__ double libm_log(const double *lt, double x) {
  return mphip_libm_log(lt, x);
}

__device__ __forceinline__ double libm_exp(const double *lt, double x) {
  return mphip_libm_exp((const uint64_t *) (lt + 2 * MPHIP_LIBM_N), x);
}

__device__ __forceinline__ double libm_pow(const double *lt, double x, double y) {
  const mphip_libm_tabs T = { (const uint64_t *) (lt + 2 * MPHIP_LIBM_N), lt, lt + 4 * MPHIP_LIBM_N };
  return mphip_libm_pow(&T, x, y);
}

// the same from the object in device memory (call sites outside the per-particle hot loops: calls, so that the
// rarely taken branches of the library's algorithms do not weigh on the registers of the kernels around them)
__device__ __noinline__ double libm_log(double x) {
  return libm_log(libm_tables(), x);
}

__device__ __noinline__ double libm_exp(double x) {
  return libm_exp(libm_tables(), x);
}

__device__ __noinline__ double libm_pow(double x, double y) {
  return libm_pow(libm_tables(), x, y);
}

// sin and cos of the C library for |x| < 2.426 (mphip_libm.h; every latitude in radians, ZETA's argument); the device
// library's beyond.  Calls: they serve the reference-rounding build (DX2DEG) and module_meteo, not the headline path.
static __device__ const double g_sincos_tab[440] = { MPHIP_LIBM_SINCOS_TAB_INIT };

__device__ __noinline__ double libm_cos(double x) {
  int handled;
  const double r = mphip_libm_cos(g_sincos_tab, x, &handled);
  return handled ? r : cos(x);
}

__device__ __noinline__ double libm_sin(double x) {
  int handled;
  const double r = mphip_libm_sin(g_sincos_tab, x, &handled);
  return handled ? r : sin(x);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- constants (mptrac.h:255-345, 430-460, 535) ---------------------------
constexpr double kG0 = 9.80665;
constexpr double kH0 = 7.0;
constexpr double kKB = 1.3806504e-23;
constexpr double kMA = 28.9644;
constexpr double kP0 = 1013.25;
constexpr double kRI = 8.3144598;
constexpr double kRA = 1e3 * kRI / kMA;
constexpr double kRE = 6367.421;
constexpr double kMAir = 4.8096e-26;
constexpr double kTRef = 298.15;
constexpr double kT0 = 273.15;
constexpr double kPi = 3.14159265358979323846;
constexpr double kSO2K1Ref = 1.23e-2, kSO2K1Temp = 2.01e3, kSO2K2Ref = 6e-8, kSO2K2Temp = 1.12e3;
constexpr double kWdTLiquid = kT0, kWdTIce = 238.15, kWdTLiquidBC = 270.;
constexpr double kEps = 18.01528 / kMA, kKappa = 0.286, kCpd = 1003.5, kKarman = 0.40;

// ---- device views ---------------------------------------------------------

// Both bracketing snapshots packed per grid point (record layouts: see the
// interpolation section)
struct DevMet {
  const float *wind;     // [cell][6] {u0,v0,u1,v1,w0,w1}
  const float *temp;     // [cell][2]
  const f32x4 *cloud;    // [cell][2] (optional)
  // level / surface fields only module_meteo reads, two fields x two snapshots per record {a0,b0,a1,b1}
  // (so that a request touching one pair gathers 16 useful bytes per lane and instruction):
  const f32x4 *mx;       // [2][cell]: {z,pv}, {o3,cc} (optional)
  const f32x4 *mx2;      // [7][col]: {ts,zs} {us,vs} {lsm,sst} {pt,tt} {zt,h2ot} {plcl,plfc} {o3c,-} (optional)
  const f32x4 *sfa;      // [col]
  const f32x4 *sfb;      // [col][2]
  const f32x4 *cp2;      // [col] {cape,pel}0 {cape,pel}1: what module_convection reads when CONV_CIN is off
  const f32x4 *sfc;      // [col][2]
  const f32x4 *sfd;      // [col][2] {ess,nss,shf,-}0 {..}1 (optional)
  const float *h2o;      // [cell][2] {h2o}0 {h2o}1 (optional)
  const float *mlw;      // [cellL][6] {ul,vl,zeta_dot}0 {..}1 on model levels (optional)
  const float *zl[2];    // zetal of met0 / met1, [nx][ny][npl]
  const float *pll[2];   // pl of met0 / met1
  const float *zl2, *pl2;   // [cellL][2] {zetal0,zetal1}, {pl0,pl1}: both snapshots of a level in one load (optional)
  int ml_monotonic;      // every zetal / pl column of both snapshots is strictly monotonic (checked when packing)
  int npl;
  // axes blob in global memory, copied to LDS by every workgroup:
  //   double lon[nx], lat[ny], p[np], 1/dlon[nx], 1/dlat[ny], 1/dp[np]; int16 p_lut[lut_size]
  const double *axes;
  int nx, ny, np, coord_type;
  int lut_base, lut_size;        // pressure look-up table (0 entries = bisection)
  double lat_x0, lat_inv_dx;     // first guess of the latitude index
  double time0, time1;
  double inv_dtime;              // 1 / (time1 - time0)
  double latmin, latmax;         // module_timesteps, mptrac.c:6009-6010
  int local;                     // mptrac.c:6012-6013
  int lat_ascending, p_ascending;
  // wave-uniform scalars of the lean stencil set-up (specialised kernels, lat/lon grids with a pressure table)
  double lon_first, lon_last;    // lon[0], lon[nx - 1]
  double inv_dlon0;              // 1 / (lon[1] - lon[0])
  double lat_search_max;         // largest double below latmax: the search value of a latitude on the last node
  double p_min, p_search_max;    // smallest node of the pressure axis, largest double below its largest node
  int p_cmp_off, p_step;         // ascending axis: 1, +1; descending: 0, -1 (the node that decides table index vs neighbour)
  float ps11[2];                 // ps of met0 / met1 at grid node [1][1] (module_position, quirk Q1)
  // lower bounds of the surface pressure / the finite cloud-top pressures of both snapshots (minus a guard band;
  // -inf = unknown): a particle between the snapshots with p < ps_skip - dry_depo_dp lies above the surface layer
  // of module_dry_depo wherever it is, one with p <= pct_skip above every cloud top of module_wet_depo -- both
  // modules return for it before they gather anything, with the result the gathers would have led to
  double ps_skip, pct_skip;
  // module_diff_turb: a particle between the snapshots with 1.01 p < turb_skip lies above the boundary layer and
  // its transition zone wherever it is (weight 0) and more than its +-10 m probes below the surface;
  // module_convection: one with p < conv_skip lies above every top the convective column can have (-inf = unknown)
  double turb_skip, conv_skip;
  // module_diff_meso: r = 1 - 2 |dt| / DT_MET and sqrt(1 - r^2) for |dt| = meso_dt = |DT_MOD| (every particle of a
  // step but the last, shortened one), from the host -- IEEE division and square root as the reference's
  double meso_dt, meso_r, meso_r2;
};

struct DevAtm {
  double *time, *p, *lon, *lat;
  double *q[MPHIP_NQ_MAX];
  float *up, *vp, *wp;           // cache->uvwp, kept SoA on the device
  double *dt;                    // cache->dt (only used across separate launches)
  const int *ext;                // external slot of each stored particle (NULL = identity)
  double *iso;                   // cache->iso_var (module_isosurf; NULL = not allocated)
  int *kz;                       // model levels: vertical index of the last step (search hint only; NULL = none)
  const double *iso_ts, *iso_ps; // balloon time series of ISOSURF 4
  int iso_n;
  // module_sort fused into the following step launch: particle i of the new order is read from slot perm[i]
  // of the arrays below and written to slot i of time / p / lon / lat (NULL = read where it is written)
  const int *perm;
  const double *s_time, *s_p, *s_lon, *s_lat;
  const double *s_q[MPHIP_NQ_MAX];   // ... and the quantity arrays (nq_perm of them: 0 = they were moved by a pass of their own)
  int nq_perm;
  long long np;                  // particles owned by this context
  long long ip0;                 // global index of the first one
  long long np_total;            // particles of the whole simulation
};

struct DevClim {
  int ntime, nlat;
  double time[12];
  double lat[73];
  double tropo[12][73];
  double inv_dtime[12];   // 1 / (time[i+1] - time[i])
  double inv_dlat[73];    // 1 / (lat[i+1] - lat[i])
};

// per-block LDS copy of the three axes, the reciprocal interval widths and
// the pressure look-up table
struct Axes {
  const double *lon, *lat, *p;
  const double *inv_lon, *inv_lat, *inv_p;
  const short *p_lut;
};

__device__ __forceinline__ size_t axes_doubles(const DevMet &M) {
  return 2 * (size_t) (M.nx + M.ny + M.np);
}

// cooperative copy of the axes blob into LDS (call from every thread, then
// __syncthreads())
__device__ __forceinline__ Axes load_axes(const DevMet &M, double *smem) {
  const size_t nd = axes_doubles(M);
  for (size_t i = threadIdx.x; i < nd; i += blockDim.x)
    smem[i] = M.axes[i];
  short *lut = (short *) (smem + nd);
  const short *glut = (const short *) (M.axes + nd);
  for (int i = threadIdx.x; i < M.lut_size; i += blockDim.x)
    lut[i] = glut[i];
  Axes A;
  A.lon = smem;
  A.lat = A.lon + M.nx;
  A.p = A.lat + M.ny;
  A.inv_lon = A.p + M.np;
  A.inv_lat = A.inv_lon + M.nx;
  A.inv_p = A.inv_lat + M.ny;
  A.p_lut = lut;
  return A;
}

// Two builds of these sources (mptrac_amd/build.py):
//   libmptrac_hip.so        MPHIP_EXACT_DIV = 0.  Where the divisor is a grid constant (interval widths, 1000, pi * RE,
//                           time1 - time0) the kernels multiply by a reciprocal the host rounded once; quotients of
//                           variables are rcp + Newton (fdiv), cos(latitude) a polynomial, and the compiler contracts
//                           a * b + c.  A weight moves by at most an ulp: positions within 1e-15 of the oracle's after
//                           20 steps, 97 % of them its bits.  Indices that are observable (sort key, mixing / output
//                           cells) always use the exact form.
//   libmptrac_hip_exact.so  MPHIP_EXACT_DIV = 1 and -ffp-contract=off: every division of the reference an IEEE
//                           division, its sqrt / cos / sin calls the C library's (sqrt_rn, libm_cos, libm_sin), no
//                           contraction -- the lean kernels included.  Positions, quantities and perturbations are the
//                           oracle's bits (tests/test_gpu_exact_library.py); the step costs ~40 % more on workload C3.
#ifndef MPHIP_EXACT_DIV
#define MPHIP_EXACT_DIV 0
#endif
// 1: keep the last wind corners of a particle and reload only where its grid cell changed
#ifndef MPHIP_WIND_CACHE
#define MPHIP_WIND_CACHE 1
#endif
// > 0: wave priority (s_setprio) while a wave computes the addresses of a gather round of the Runge-Kutta
// stages and issues the loads -- the sooner a round is in flight, the more of its latency other waves cover
#ifndef MPHIP_SETPRIO
#define MPHIP_SETPRIO 0
#endif


__device__ __forceinline__ double div_const(double x, double y, double inv_y) {
#if MPHIP_EXACT_DIV
  (void) inv_y;
  return x / y;
#else
  (void) y;
  return x * inv_y;
#endif
}

// ---- arithmetic conventions ------------------------------------------------

// Division, square root and log of the default build (MPHIP_EXACT_DIV = 0).  On gfx950 an IEEE fp64
// division costs ~68 cycles per wave, the library sqrt ~104 and the library log ~420
// (tools/micro/valu_latency.hip); these take ~45, ~60 and ~150 and stay within 1-2 ulp -- far inside
// the 1e-10 parity bar.  MPHIP_EXACT_DIV = 1 restores the IEEE / library versions everywhere.
__device__ __forceinline__ double fdiv(double a, double b) {
#if MPHIP_EXACT_DIV
  return a / b;
#else
  // v_rcp_f64 is good to ~2^-26; one Newton step squares that, and the residual correction of the quotient
  // multiplies the two errors: the result is the correctly rounded quotient or its neighbour
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#endif
}

__device__ __forceinline__ double fsqrt(double x) {
#if MPHIP_EXACT_DIV
  return sqrt(x);
#else
  const double y = __builtin_amdgcn_rsq(x);       // coupled Newton iteration for sqrt(x) and 1/(2 sqrt(x))
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? x : g;                        // (also keeps -0 and avoids 0 * inf)
#endif
}

// ---- one-instruction forms the compiler does not pick by itself ---------------

__device__ __forceinline__ double vmin(double a, double b) {   // v_min_f64 without the canonicalising copies
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the same with a wave-uniform second operand (kernel argument): stays in its SGPR pair
__device__ __forceinline__ double vmin_s(double a, double s) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s));
  return r;
}

__device__ __forceinline__ double vmax_s(double a, double s) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s));
  return r;
}

// min(max(x, 0), hi) with a wave-uniform hi >= 0
__device__ __forceinline__ int clamp0_s(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}

// x * y + c with a literal constant c: the constant goes into an SGPR pair (two scalar moves beside the
// vector pipe) -- left to itself the compiler often builds it in a VGPR pair with two v_mov per Horner step
__device__ __forceinline__ double fma_k(double x, double y, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "s"(c));
  return r;
}

// The IEEE square root (correctly rounded) for 0 <= x < 2^1000, x not subnormal: the hardware's reciprocal square
// root estimate, one coupled Newton step for sqrt(x) and 1 / (2 sqrt(x)), and two residual corrections -- the sequence
// the compiler emits for sqrt() without its range scaling (the arguments here are -2 log u <= 89 and 1 - r^2 <= 1).
__device__ __forceinline__ double sqrt_rn(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return x == 0.0 || x == __builtin_inf() ? x : g;
}

// 1 / b to an ulp or two (rcp + two Newton steps)
__device__ __forceinline__ double frcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}

// cos_latitude with the polynomial constants in SGPRs
__device__ __forceinline__ double cos_latitude_k(double x) {
  const double ax = fabs(x);
  const bool hi = ax > 0.78539816339744830962;
  const double y = hi ? (1.57079632679489655800e+00 - ax) + 6.12323399573676603587e-17 : ax;
  const double z = y * y;
  double a = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  a = fma_k(z, a, 2.75573137070700676789e-06);
  a = fma_k(z, a, -1.98412698298579493134e-04);
  a = fma_k(z, a, 8.33333333332248946124e-03);
  a = fma_k(z, a, -1.66666666666666324348e-01);
  const double ps = y + y * z * a;
  double b = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  b = fma_k(z, b, -2.75573143513906633035e-07);
  b = fma_k(z, b, 2.48015872894767294178e-05);
  b = fma_k(z, b, -1.38888888888741095749e-03);
  b = fma_k(z, b, 4.16666666666666019037e-02);
  const double pc = 1.0 - (0.5 * z - z * z * b);
  return hi ? ps : pc;
}

__device__ __forceinline__ double cos_latitude(double x) {
  return cos_latitude_k(x);
}

// FMOD, mptrac.h:1121-1122.  (int)(x / y) is 0 whenever |x| < y, so the
// division is only executed outside that range; the result is identical.
__device__ __forceinline__ double fmod_trunc(double x, double y) {
  if (fabs(x) < y)
    return x - 0 * y;
  return x - (int) (x / y) * y;
}

__device__ __forceinline__ double deg2rad(double deg) {   // mptrac.h:857
  return deg * (kPi / 180.0);
}

// cos(x) for |x| <= pi/2 (a latitude in radians): the fdlibm kernels -- cosine
// polynomial on [0, pi/4], sine polynomial of pi/2 - |x| above, both evaluated
// and selected (no divergent branch, no argument-reduction code for the general
// case).  Below one ulp, like the C library's cos() the reference calls.
// (The products of the conversion functions are rounded before the caller adds them to a coordinate, as in
// the reference's `lon += DX2COORD(...)`: contraction is switched off inside them, so that no call site
// fuses the last multiply into its addition while another -- behind a select -- cannot.)
__device__ __forceinline__ double dx2deg(double dx, double lat) {   // mptrac.h:904-906
#pragma clang fp contract(off)
  if (lat < -89.999 || lat > 89.999)
    return 0;
#if MPHIP_EXACT_DIV
  return dx * 180. / (kPi * kRE * libm_cos(deg2rad(lat)));
#else
  return (dx * 180.) * frcp(kPi * kRE * cos_latitude(deg2rad(lat)));
#endif
}

__device__ __forceinline__ double dy2deg(double dy) {   // mptrac.h:922
#pragma clang fp contract(off)
#if MPHIP_EXACT_DIV
  return dy * 180. / (kPi * kRE);
#else
  return (dy * 180.) * (1.0 / (kPi * kRE));
#endif
}

// DX2COORD / DY2COORD (mptrac.h:966, 989) of a distance in metres.  The default build folds the constant factors of
// dx / 1000 * 180 / (pi RE cos(lat)) into one (kMetresToDeg = 180e-3, kDegPerMetreY = 180e-3 / (pi RE)): one
// rounding instead of three on a displacement of metres -- the lean kernels (DegPerMetre) form the same products
constexpr double kMetresToDeg = 1e-3 * 180.;
constexpr double kDegPerMetreY = 1e-3 * 180. / (kPi * kRE);

__device__ __forceinline__ double dx2coord(int coord_type, double dx, double lat) {
#pragma clang fp contract(off)
#if MPHIP_EXACT_DIV
  return coord_type == 0 ? dx2deg(div_const(dx, 1000.0, 1e-3), lat) : dx;
#else
  if (coord_type != 0)
    return dx;
  if (lat < -89.999 || lat > 89.999)
    return 0;
  return dx * (kMetresToDeg * frcp(kPi * kRE * cos_latitude(deg2rad(lat))));
#endif
}

__device__ __forceinline__ double dy2coord(int coord_type, double dy) {
#pragma clang fp contract(off)
#if MPHIP_EXACT_DIV
  return coord_type == 0 ? dy2deg(div_const(dy, 1000.0, 1e-3)) : dy;
#else
  return coord_type == 0 ? dy * kDegPerMetreY : dy;
#endif
}

__device__ __forceinline__ double dz2dp(double dz, double p) {   // mptrac.h:941
  return div_const(-dz * p, kH0, 1.0 / kH0);
}

__device__ __forceinline__ double zfromp(double p) {   // mptrac.h:2243
  return kH0 * libm_log(kP0 / p);
}

__device__ __forceinline__ double lin(double x0, double y0, double x1, double y1, double x) {   // mptrac.h:1351
  return y0 + fdiv(y1 - y0, x1 - x0) * (x - x0);
}

__device__ __forceinline__ double rho_air(double p, double t) {   // mptrac.h:1961
  return fdiv(100. * p, kRA * t);
}

__device__ __forceinline__ double dmin(double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }

// ---- axis search (mptrac.c:3495-3574) --------------------------------------

// locate_irr: the bisection of the reference only ever compares interior
// nodes 1..n-2, so on a monotonic axis its result is the unique i in [0, n-2]
// with (i == 0 or node i on the near side of x) and (i == n-2 or node i+1 on
// the far side).  locate_irr() is that bisection; locate_from() reaches the
// same index from a first guess with a short linear correction.
__device__ __forceinline__ int locate_irr(const double *xx, int n, double x, int ascending) {
  int lo = 0, hi = n - 1;
  if (ascending) {
    while (hi > lo + 1) {
      const int mid = (hi + lo) >> 1;
      if (xx[mid] > x)
        hi = mid;
      else
        lo = mid;
    }
  } else {
    while (hi > lo + 1) {
      const int mid = (hi + lo) >> 1;
      if (xx[mid] <= x)
        hi = mid;
      else
        lo = mid;
    }
  }
  return lo;
}

__device__ __forceinline__ int locate_from(const double *xx, int n, double x, int ascending, int guess) {
  if (!(x == x))
    return n - 2;   // every comparison of the bisection fails for NaN
  int g = guess < 0 ? 0 : (guess > n - 2 ? n - 2 : guess);
  if (ascending) {
    while (g > 0 && xx[g] > x)
      g--;
    while (g < n - 2 && xx[g + 1] <= x)
      g++;
  } else {
    while (g > 0 && xx[g] <= x)
      g--;
    while (g < n - 2 && xx[g + 1] > x)
      g++;
  }
  return g;
}

// Index plus the interval it selects.  The three LDS reads are issued together
// on the first guess; the linear correction only runs when the guess is off.
struct AxisHit {
  int i;
  double x0, x1, inv;   // xx[i], xx[i+1], 1 / (xx[i+1] - xx[i])
};

__device__ __forceinline__ AxisHit locate_hit(const double *xx, const double *inv, int n, double x, int ascending,
                                              int guess) {
  AxisHit h;
  int g = guess < 0 ? 0 : (guess > n - 2 ? n - 2 : guess);
  h.x0 = xx[g];
  h.x1 = xx[g + 1];
  h.inv = inv[g];
  bool ok;
  if (ascending)
    ok = (g == 0 || h.x0 <= x) && (g == n - 2 || h.x1 > x);
  else
    ok = (g == 0 || h.x0 > x) && (g == n - 2 || h.x1 <= x);
  if (!ok || !(x == x)) {
    g = locate_from(xx, n, x, ascending, g);
    h.x0 = xx[g];
    h.x1 = xx[g + 1];
    h.inv = inv[g];
  }
  h.i = g;
  return h;
}

// latitude: first guess from the mean spacing
__device__ __forceinline__ int lat_guess(const DevMet &M, double lat) {
  return (int) ((lat - M.lat_x0) * M.lat_inv_dx);
}

__device__ __forceinline__ int locate_lat(const DevMet &M, const Axes &A, double lat) {
  return locate_from(A.lat, M.ny, lat, M.lat_ascending, lat_guess(M, lat));
}

// pressure: first guess from a table indexed by the exponent and the top seven
// mantissa bits of p (128 bins per octave, at most a node or two per bin)
__device__ __forceinline__ int p_guess(const DevMet &M, const Axes &A, double p) {
  int j = (int) (__double_as_longlong(p) >> 45) - M.lut_base;
  j = j < 0 ? 0 : (j >= M.lut_size ? M.lut_size - 1 : j);
  return (int) A.p_lut[j];
}

__device__ __forceinline__ int locate_p(const DevMet &M, const Axes &A, double p) {
  if (M.lut_size == 0)
    return locate_irr(A.p, M.np, p, M.p_ascending);
  return locate_from(A.p, M.np, p, M.p_ascending, p_guess(M, A, p));
}

__device__ __forceinline__ AxisHit hit_lat(const DevMet &M, const Axes &A, double lat) {
  return locate_hit(A.lat, A.inv_lat, M.ny, lat, M.lat_ascending, lat_guess(M, lat));
}

__device__ __forceinline__ AxisHit hit_p(const DevMet &M, const Axes &A, double p) {
  const int g = M.lut_size ? p_guess(M, A, p) : locate_irr(A.p, M.np, p, M.p_ascending);
  return locate_hit(A.p, A.inv_p, M.np, p, M.p_ascending, g);
}

__device__ __forceinline__ int locate_reg(const double *xx, int n, double x) {   // mptrac.c:3559-3574
  const int i = (int) ((x - xx[0]) / (xx[1] - xx[0]));
  return i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
}

// the same with the reciprocal of the (regular) spacing; may differ from
// locate_reg when x is within an ulp of a grid line -- used for interpolation
// stencils only, where both neighbours give the same value
__device__ __forceinline__ int locate_lon(const DevMet &M, const Axes &A, double x) {
#if MPHIP_EXACT_DIV
  return locate_reg(A.lon, M.nx, x);
#else
  const int i = (int) ((x - A.lon[0]) * A.inv_lon[0]);
  return i < 0 ? 0 : (i > M.nx - 2 ? M.nx - 2 : i);
#endif
}

// ---- interpolation (mptrac.c:2755-3170) ------------------------------------

struct Stencil {   // ci[3], cw[3] of INTPOL_INIT (mptrac.h:1174)
  int ip, ix, iy;
  double wp, wx, wy;
};

__device__ __forceinline__ Stencil stencil_zero() {
  Stencil s;
  s.ip = s.ix = s.iy = 0;
  s.wp = s.wx = s.wy = 0.0;
  return s;
}

// intpol_check_lon_lat / intpol_check_cartesian, mptrac.c:2755-2803
__device__ __forceinline__ void check_horizontal(const DevMet &M, const Axes &A, double lon, double lat,
                                                 double &lon2, double &lat2) {
  const double x0 = A.lon[0], x1 = A.lon[M.nx - 1];
  const double y0 = A.lat[0], y1 = A.lat[M.ny - 1];
  if (M.coord_type == 0) {
    lon2 = fmod_trunc(lon, 360.);
    if (lon2 < x0)
      lon2 += 360;
    else if (lon2 > x1)
      lon2 -= 360;
  } else {
    if (x0 < x1)
      lon2 = dmin(dmax(lon, x0), x1);
    else
      lon2 = dmin(dmax(lon, x1), x0);
  }
  if (y0 < y1)
    lat2 = dmin(dmax(lat, y0), y1);
  else
    lat2 = dmin(dmax(lat, y1), y0);
}

// index/weight set-up of intpol_met_space_3d, mptrac.c:2997-3021
__device__ __forceinline__ void stencil_3d(const DevMet &M, const Axes &A, double p, double lon, double lat,
                                           Stencil &s) {
  double lon2, lat2;
  check_horizontal(M, A, lon, lat, lon2, lat2);
  const AxisHit hp = hit_p(M, A, p);
  const AxisHit hy = hit_lat(M, A, lat2);
  s.ix = locate_lon(M, A, lon2);
  const double lx0 = A.lon[s.ix], lx1 = A.lon[s.ix + 1], linv = A.inv_lon[s.ix];
  s.ip = hp.i;
  s.iy = hy.i;
  s.wp = div_const(hp.x1 - p, hp.x1 - hp.x0, hp.inv);
  s.wx = div_const(lx1 - lon2, lx1 - lx0, linv);
  s.wy = div_const(hy.x1 - lat2, hy.x1 - hy.x0, hy.inv);
}

// index/weight set-up of intpol_met_space_2d, mptrac.c:3059-3081
__device__ __forceinline__ void stencil_2d(const DevMet &M, const Axes &A, double lon, double lat, Stencil &s) {
  double lon2, lat2;
  check_horizontal(M, A, lon, lat, lon2, lat2);
  const AxisHit hy = hit_lat(M, A, lat2);
  s.ix = locate_lon(M, A, lon2);
  const double lx0 = A.lon[s.ix], lx1 = A.lon[s.ix + 1], linv = A.inv_lon[s.ix];
  s.iy = hy.i;
  s.wx = div_const(lx1 - lon2, lx1 - lx0, linv);
  s.wy = div_const(hy.x1 - lat2, hy.x1 - hy.x0, hy.inv);
}

// 16-byte vectors that are only 4-byte aligned in memory (the packed records
// below start on 8- or 24-byte multiples); gfx950 global loads need dword
// alignment only.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// Packed grids (both snapshots in one record, level index fastest):
//   wind [cell][6]  {u0,v0,u1,v1,w0,w1}    a column's level pair = 48 B = 3 x 16 B; the (u,v) of one snapshot are an
//                                          aligned register pair after the loads (packed two-float arithmetic)
//   temp [cell][2]  {t}0 {t}1              a column's level pair = 16 B = 1 load
//   cloud[cell][8]  {lwc,rwc,iwc,swc}0 {..}1   level pair = 64 B (wet deposition only)
//   sfa  [col][4]   {ps,pbl}0 {ps,pbl}1    one load per corner
//   sfb  [col][8]   {cape,cin,pel,-}0 {..}1
//   sfc  [col][8]   {pct,pcb,cl,-}0 {..}1
// so a wind stencil is 12 loads, a temperature stencil 4, a {ps,pbl} stencil 4.

struct WindCorners {
  f32x4u r[2][2][3];   // [di][dj][piece]: 12 floats = level ip {u0,v0,u1,v1,w0,w1}, level ip+1 {..}
};

// Index arithmetic in 24 x 24 -> 32-bit multiplies (full rate; a 64-bit integer multiply is four
// quarter-rate ones): nx * ny < 2^24 and nx * ny * np < 2^31 are checked when the grid is uploaded.  The
// four corners of a stencil share one base index.
__device__ __forceinline__ unsigned col_of(const DevMet &M, const Stencil &s, int di, int dj) {
  return __umul24((unsigned) s.ix, (unsigned) M.ny) + (unsigned) s.iy + (unsigned) di * (unsigned) M.ny + (unsigned) dj;
}

__device__ __forceinline__ size_t cell_of(const DevMet &M, const Stencil &s, int di, int dj) {
  const unsigned base = __umul24(__umul24((unsigned) s.ix, (unsigned) M.ny) + (unsigned) s.iy, (unsigned) M.np)
    + (unsigned) s.ip;
  return (size_t) (base + (unsigned) di * ((unsigned) M.ny * (unsigned) M.np) + (unsigned) dj * (unsigned) M.np);
}

__device__ __forceinline__ void load_wind(const DevMet &M, const Stencil &s, WindCorners &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4u *q = (const f32x4u *) (M.wind + 6 * cell_of(M, s, di, dj));
      c.r[di][dj][0] = q[0];
      c.r[di][dj][1] = q[1];
      c.r[di][dj][2] = q[2];
    }
}

// The 2 x 2 x 2 x 2 wind corners a particle used last, with the cell they belong to.  A gather
// instruction costs the vector-memory path ~40-60 cycles per wave whatever its width, and costs per
// active lane; consecutive Runge-Kutta stages (and module_diff_meso afterwards) mostly stay in the same
// grid cell, so only the lanes whose cell changed reload (the loads run under their exec mask).
struct WindCache {
  WindCorners c;
  int ix, iy, ip;
  bool enabled;   // compile-time constant per kernel instantiation (off in the generic one: register budget)
};

__device__ __forceinline__ void wind_cache_reset(WindCache &w, bool enabled) {
  w.enabled = enabled && MPHIP_WIND_CACHE;
  w.ix = w.iy = w.ip = -1;
  // the corners have no value yet (every lane loads them at the first stencil: no cell is -1); an empty
  // asm gives the registers a definition without the 48 moves a zero fill costs per particle
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
#pragma unroll
      for (int k = 0; k < 3; k++)
        asm volatile("" : "=v"(w.c.r[di][dj][k]));
}

// The reload is written with the load instructions as inline assembly whose destination is a
// read-write ("+v") operand: a lane that keeps its corners keeps its registers, and the compiler sees
// one value per register instead of a 48-register merge of "old" and "reloaded" corners (which it
// resolved by holding both and spilling).  wind_cache_wait() closes the sequence: it waits for the
// loads and is the point after which the corners may be read.
__device__ __forceinline__ void load_wind_cached(const DevMet &M, const Stencil &s, WindCache &w) {
  if (s.ix != w.ix || s.iy != w.iy || s.ip != w.ip) {
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
      for (int dj = 0; dj < 2; dj++) {
        const float *q = M.wind + 6 * cell_of(M, s, di, dj);
        asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                     "global_load_dwordx4 %1, %3, off offset:16\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32"
                     : "+v"(w.c.r[di][dj][0]), "+v"(w.c.r[di][dj][1]), "+v"(w.c.r[di][dj][2])
                     : "v"(q)
                     : "memory");
      }
    w.ix = s.ix;
    w.iy = s.iy;
    w.ip = s.ip;
  }
}

__device__ __forceinline__ void wind_cache_wait(WindCache &w) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(w.c.r[0][0][0]), "+v"(w.c.r[0][0][1]), "+v"(w.c.r[0][0][2]), "+v"(w.c.r[0][1][0]),
                 "+v"(w.c.r[0][1][1]), "+v"(w.c.r[0][1][2]), "+v"(w.c.r[1][0][0]), "+v"(w.c.r[1][0][1]),
                 "+v"(w.c.r[1][0][2]), "+v"(w.c.r[1][1][0]), "+v"(w.c.r[1][1][1]), "+v"(w.c.r[1][1][2])
               :
               : "memory");
}

// value of component k (0 u, 1 v, 2 w) of snapshot t at level ip + lvl
__device__ __forceinline__ float wind_elem(const WindCorners &c, int di, int dj, int lvl, int t, int k) {
  const int e = 6 * lvl + (k < 2 ? 2 * t + k : 4 + t);   // record layout {u0,v0,u1,v1,w0,w1}
  return c.r[di][dj][e >> 2][e & 3];
}

// intpol_met_space_3d, mptrac.c:3023-3043.  The difference of the two float
// corners is taken in single precision, as the reference's C expression does
// (float - float), and only then widened.
__device__ __forceinline__ double lerp3(const Stencil &s, float l00, float h00, float l01, float h01, float l10,
                                        float h10, float l11, float h11) {
  const double c00 = s.wp * (double) (l00 - h00) + (double) h00;
  const double c01 = s.wp * (double) (l01 - h01) + (double) h01;
  const double c10 = s.wp * (double) (l10 - h10) + (double) h10;
  const double c11 = s.wp * (double) (l11 - h11) + (double) h11;
  const double r0 = s.wy * (c00 - c01) + c01;
  const double r1 = s.wy * (c10 - c11) + c11;
  return s.wx * (r0 - r1) + r1;
}

__device__ __forceinline__ double wind_space_3d(const WindCorners &c, const Stencil &s, int t, int k) {
  return lerp3(s, wind_elem(c, 0, 0, 0, t, k), wind_elem(c, 0, 0, 1, t, k), wind_elem(c, 0, 1, 0, t, k),
               wind_elem(c, 0, 1, 1, t, k), wind_elem(c, 1, 0, 0, t, k), wind_elem(c, 1, 0, 1, t, k),
               wind_elem(c, 1, 1, 0, t, k), wind_elem(c, 1, 1, 1, t, k));
}

// intpol_met_time_3d, mptrac.c:3112-3137
__device__ __forceinline__ double wind_time_3d(const WindCorners &c, const Stencil &s, double wt, int k) {
  const double v0 = wind_space_3d(c, s, 0, k);
  const double v1 = wind_space_3d(c, s, 1, k);
  return wt * (v0 - v1) + v1;
}

__device__ __forceinline__ double time_weight(const DevMet &M, double ts) {   // mptrac.c:3133
  return div_const(M.time1 - ts, M.time1 - M.time0, M.inv_dtime);
}

// temperature stencil: one 16-byte load per column = {t0,t1} at ip, {t0,t1} at ip+1
__device__ __forceinline__ double pair_time_3d(const float *__restrict__ g, const DevMet &M, const Stencil &s,
                                               double wt) {
  f32x4u v[2][2];
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
      v[di][dj] = *(const f32x4u *) (g + 2 * cell_of(M, s, di, dj));
  const double v0 = lerp3(s, v[0][0][0], v[0][0][2], v[0][1][0], v[0][1][2], v[1][0][0], v[1][0][2], v[1][1][0],
                          v[1][1][2]);
  const double v1 = lerp3(s, v[0][0][1], v[0][0][3], v[0][1][1], v[0][1][3], v[1][0][1], v[1][0][3], v[1][1][1],
                          v[1][1][3]);
  return wt * (v0 - v1) + v1;
}

__device__ __forceinline__ double temp_time_3d(const DevMet &M, const Stencil &s, double wt) {
  return pair_time_3d(M.temp, M, s, wt);
}

// cloud water stencil (wet deposition): {lwc,rwc,iwc,swc}0 {..}1 per level
struct CloudCorners {
  f32x4 lo[2][2][2];   // [di][dj][snapshot] at level ip
  f32x4 hi[2][2][2];
};

__device__ __forceinline__ void load_quad(const f32x4 *__restrict__ g, const DevMet &M, const Stencil &s,
                                          CloudCorners &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4 *q = g + 2 * cell_of(M, s, di, dj);
      c.lo[di][dj][0] = q[0];
      c.lo[di][dj][1] = q[1];
      c.hi[di][dj][0] = q[2];
      c.hi[di][dj][1] = q[3];
    }
}

__device__ __forceinline__ void load_cloud(const DevMet &M, const Stencil &s, CloudCorners &c) {
  load_quad(M.cloud, M, s, c);
}

__device__ __forceinline__ double cloud_time_3d(const CloudCorners &c, const Stencil &s, double wt, int k) {
  const double v0 = lerp3(s, c.lo[0][0][0][k], c.hi[0][0][0][k], c.lo[0][1][0][k], c.hi[0][1][0][k], c.lo[1][0][0][k],
                          c.hi[1][0][0][k], c.lo[1][1][0][k], c.hi[1][1][0][k]);
  const double v1 = lerp3(s, c.lo[0][0][1][k], c.hi[0][0][1][k], c.lo[0][1][1][k], c.hi[0][1][1][k], c.lo[1][0][1][k],
                          c.hi[1][0][1][k], c.lo[1][1][1][k], c.hi[1][1][1][k]);
  return wt * (v0 - v1) + v1;
}

// intpol_met_space_2d value part (mptrac.c:3083-3107): NaN-aware nearest
// neighbour when a corner is not finite
__device__ __forceinline__ double bilin_2d(const Stencil &s, double c00, double c01, double c10, double c11) {
  if (isfinite(c00) && isfinite(c01) && isfinite(c10) && isfinite(c11)) {
    const double r0 = s.wy * (c00 - c01) + c01;
    const double r1 = s.wy * (c10 - c11) + c11;
    return s.wx * (r0 - r1) + r1;
  }
  if (s.wy < 0.5)
    return s.wx < 0.5 ? c11 : c01;
  return s.wx < 0.5 ? c10 : c00;
}

// intpol_met_time_2d, mptrac.c:3141-3170
__device__ __forceinline__ double blend_time_2d(double v0, double v1, double wt) {
  if (isfinite(v0) && isfinite(v1))
    return wt * (v0 - v1) + v1;
  return wt < 0.5 ? v1 : v0;
}

// {ps,pbl} of both snapshots at the four corners: one load each
struct SurfA {
  f32x4 v[2][2];   // {ps0, pbl0, ps1, pbl1}
};

__device__ __forceinline__ void load_pair_2d(const f32x4 *__restrict__ g, const DevMet &M, const Stencil &s,
                                             SurfA &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
      c.v[di][dj] = g[col_of(M, s, di, dj)];
}

__device__ __forceinline__ void load_sfa(const DevMet &M, const Stencil &s, SurfA &c) {
  load_pair_2d(M.sfa, M, s, c);
}

// f = 0: ps, f = 1: pbl
__device__ __forceinline__ double sfa_time_2d(const SurfA &c, const Stencil &s, double wt, int f) {
  const double v0 = bilin_2d(s, c.v[0][0][f], c.v[0][1][f], c.v[1][0][f], c.v[1][1][f]);
  const double v1 = bilin_2d(s, c.v[0][0][2 + f], c.v[0][1][2 + f], c.v[1][0][2 + f], c.v[1][1][2 + f]);
  return blend_time_2d(v0, v1, wt);
}

// two level fields x two snapshots per record {a0,b0,a1,b1}: a column's level pair is two loads
struct PairCorners {
  f32x4 lo[2][2], hi[2][2];   // level ip, level ip + 1
};

__device__ __forceinline__ void load_pair_3d(const f32x4 *__restrict__ g, const DevMet &M, const Stencil &s,
                                             PairCorners &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4 *q = g + cell_of(M, s, di, dj);
      c.lo[di][dj] = q[0];
      c.hi[di][dj] = q[1];
    }
}

// field f (0 / 1) of the record
__device__ __forceinline__ double pair_field_time_3d(const PairCorners &c, const Stencil &s, double wt, int f) {
  const double v0 = lerp3(s, c.lo[0][0][f], c.hi[0][0][f], c.lo[0][1][f], c.hi[0][1][f], c.lo[1][0][f], c.hi[1][0][f],
                          c.lo[1][1][f], c.hi[1][1][f]);
  const double v1 = lerp3(s, c.lo[0][0][2 + f], c.hi[0][0][2 + f], c.lo[0][1][2 + f], c.hi[0][1][2 + f],
                          c.lo[1][0][2 + f], c.hi[1][0][2 + f], c.lo[1][1][2 + f], c.hi[1][1][2 + f]);
  return wt * (v0 - v1) + v1;
}

// three-field surface records: {a,b,c,-}0 {a,b,c,-}1 (sfb: cape,cin,pel; sfc: pct,pcb,cl)
struct SurfB {
  f32x4 v[2][2][2];   // [di][dj][snapshot]
};

__device__ __forceinline__ void load_sfb(const f32x4 *__restrict__ g, const DevMet &M, const Stencil &s, SurfB &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4 *q = g + 2 * (size_t) col_of(M, s, di, dj);
      c.v[di][dj][0] = q[0];
      c.v[di][dj][1] = q[1];
    }
}

__device__ __forceinline__ double sfb_time_2d(const SurfB &c, const Stencil &s, double wt, int f) {
  const double v0 = bilin_2d(s, c.v[0][0][0][f], c.v[0][1][0][f], c.v[1][0][0][f], c.v[1][1][0][f]);
  const double v1 = bilin_2d(s, c.v[0][0][1][f], c.v[0][1][1][f], c.v[1][0][1][f], c.v[1][1][1][f]);
  return blend_time_2d(v0, v1, wt);
}

// ---- model-level interpolation (mptrac.c:2808-2981, 3525-3594) -------------

struct Stencil4 {   // ci[3], cw[4] of intpol_met_4d_zeta
  int ix, iy, iz;
  double wx, wy, wz, wt;
};

// locate_irr_float, mptrac.c:3525-3555
__device__ __forceinline__ int locate_irr_float(const float *__restrict__ xx, int n, double x, int ig) {
  if ((xx[ig] <= x && x < xx[ig + 1]) || (xx[ig] >= x && x > xx[ig + 1]))
    return ig;
  int lo = 0, hi = n - 1;
  const int mid0 = (hi + lo) >> 1;
  if (xx[mid0] < xx[mid0 + 1]) {
    while (hi > lo + 1) {
      const int mid = (hi + lo) >> 1;
      if (xx[mid] > x)
        hi = mid;
      else
        lo = mid;
    }
  } else {
    while (hi > lo + 1) {
      const int mid = (hi + lo) >> 1;
      if (xx[mid] <= x)
        hi = mid;
      else
        lo = mid;
    }
  }
  return lo;
}

__device__ __forceinline__ size_t col_ml(const DevMet &M, int ix, int iy) {
  return ((size_t) ix * (size_t) M.ny + (size_t) iy) * (size_t) M.npl;
}

// time-then-horizontal interpolation of a model-level field pair at level k
__device__ __forceinline__ double level_value(const DevMet &M, const float *__restrict__ h0,
                                              const float *__restrict__ h1, const Stencil4 &s, int k) {
  const size_t c00 = col_ml(M, s.ix, s.iy) + k, c01 = col_ml(M, s.ix, s.iy + 1) + k;
  const size_t c10 = col_ml(M, s.ix + 1, s.iy) + k, c11 = col_ml(M, s.ix + 1, s.iy + 1) + k;
  const double v00 = s.wt * (double) (h1[c00] - h0[c00]) + (double) h0[c00];
  const double v01 = s.wt * (double) (h1[c01] - h0[c01]) + (double) h0[c01];
  const double v10 = s.wt * (double) (h1[c10] - h0[c10]) + (double) h0[c10];
  const double v11 = s.wt * (double) (h1[c11] - h0[c11]) + (double) h0[c11];
  const double a = s.wy * (v01 - v00) + v00;
  const double b = s.wy * (v11 - v10) + v10;
  return s.wx * (b - a) + a;
}

// index/weight set-up of intpol_met_4d_zeta (mptrac.c:2824-2943) with the
// height field (h0, h1) = zetal or pl of the two snapshots
__device__ __forceinline__ void stencil_4d(const DevMet &M, const Axes &A, const float *__restrict__ h0,
                                           const float *__restrict__ h1, double ts, double height, double lon,
                                           double lat, Stencil4 &s) {
  double lon2, lat2;
  check_horizontal(M, A, lon, lat, lon2, lat2);
  const AxisHit hy = hit_lat(M, A, lat2);
  s.ix = locate_lon(M, A, lon2);
  s.iy = hy.i;
  // locate_vert on both snapshots (mptrac.c:3578-3594), guesses chained as in the reference
  int kmin = 0, kmax = 0;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const float *h = t ? h1 : h0;
    const int i0 = locate_irr_float(h + col_ml(M, s.ix, s.iy), M.npl, height, 0);
    const int i1 = locate_irr_float(h + col_ml(M, s.ix + 1, s.iy), M.npl, height, i0);
    const int i2 = locate_irr_float(h + col_ml(M, s.ix, s.iy + 1), M.npl, height, i1);
    const int i3 = locate_irr_float(h + col_ml(M, s.ix + 1, s.iy + 1), M.npl, height, i2);
    const int lo = min(min(i0, i1), min(i2, i3)), hi = max(max(i0, i1), max(i2, i3));
    kmin = t ? min(kmin, lo) : lo;
    kmax = t ? max(kmax, hi) : hi;
  }
  s.iz = kmin;
  s.wt = div_const(ts - M.time0, M.time1 - M.time0, M.inv_dtime);
  s.wx = div_const(lon2 - A.lon[s.ix], A.lon[s.ix + 1] - A.lon[s.ix], A.inv_lon[s.ix]);
  s.wy = div_const(lat2 - hy.x0, hy.x1 - hy.x0, hy.inv);
  double bot = level_value(M, h0, h1, s, s.iz);
  double top = level_value(M, h0, h1, s, s.iz + 1);
  const float g0 = h0[0], g1 = h0[1];   // heights0[0][0][0], heights0[0][0][1]
  while (((g0 > g1) && ((bot <= height) || (top > height)) && (bot >= height) && (s.iz < kmax))
         || ((g0 < g1) && ((bot >= height) || (top < height)) && (bot <= height) && (s.iz < kmax))) {
    s.iz++;
    bot = top;
    top = level_value(M, h0, h1, s, s.iz + 1);
  }
  s.wz = (height - bot) / (top - bot);
}

// value part of intpol_met_4d_zeta (mptrac.c:2945-2980): time, longitude,
// latitude, vertical -- for one of the packed components {ul, vl, zeta_dot}
__device__ __forceinline__ double ml_combine(const Stencil4 &s, double a000, double a100, double a010, double a110,
                                             double a001, double a101, double a011, double a111) {
  const double a00 = s.wx * (a100 - a000) + a000;
  const double a10 = s.wx * (a110 - a010) + a010;
  const double a01 = s.wx * (a101 - a001) + a001;
  const double a11 = s.wx * (a111 - a011) + a011;
  const double lo = s.wy * (a10 - a00) + a00;
  const double hi = s.wy * (a11 - a01) + a01;
  return s.wz * (hi - lo) + lo;
}

struct MlCorners {
  f32x4u r[2][2][3];   // as WindCorners: level iz {ul,vl,zd}0{..}1, level iz+1 {..}
};

__device__ __forceinline__ void load_ml(const DevMet &M, const Stencil4 &s, MlCorners &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4u *q = (const f32x4u *) (M.mlw + 6 * (col_ml(M, s.ix + di, s.iy + dj) + s.iz));
      c.r[di][dj][0] = q[0];
      c.r[di][dj][1] = q[1];
      c.r[di][dj][2] = q[2];
    }
}

__device__ __forceinline__ double ml_packed(const MlCorners &c, const Stencil4 &s, int k) {
  double v[2][2][2];   // [di][dj][level], time-interpolated
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
#pragma unroll
      for (int l = 0; l < 2; l++) {
        const int e0 = 6 * l + k, e1 = 6 * l + 3 + k;
        const float a0 = c.r[di][dj][e0 >> 2][e0 & 3], a1 = c.r[di][dj][e1 >> 2][e1 & 3];
        v[di][dj][l] = s.wt * (double) (a1 - a0) + (double) a0;
      }
  return ml_combine(s, v[0][0][0], v[1][0][0], v[0][1][0], v[1][1][0], v[0][0][1], v[1][0][1], v[0][1][1],
                    v[1][1][1]);
}

// the same for a plain model-level field pair (zetal or pl used as the array)
__device__ __forceinline__ double ml_field(const DevMet &M, const float *__restrict__ a0,
                                           const float *__restrict__ a1, const Stencil4 &s) {
  double v[2][2][2];
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
#pragma unroll
      for (int l = 0; l < 2; l++) {
        const size_t c = col_ml(M, s.ix + di, s.iy + dj) + s.iz + l;
        v[di][dj][l] = s.wt * (double) (a1[c] - a0[c]) + (double) a0[c];
      }
  return ml_combine(s, v[0][0][0], v[1][0][0], v[0][1][0], v[1][1][0], v[0][0][1], v[1][0][1], v[0][1][1],
                    v[1][1][1]);
}

// ---- climatological tropopause and weights --------------------------------

// clim_tropo, mptrac.c:213-237
__device__ __forceinline__ double clim_tropo(const DevClim &C, double t, double lat) {
  double sec = fmod_trunc(t, 365.25 * 86400.);
  while (sec < 0)
    sec += 365.25 * 86400.;
  const int it = locate_irr(C.time, C.ntime, sec, 1);
  const int il = locate_reg(C.lat, C.nlat, lat);
  // LIN (mptrac.h:1351) with the reciprocal of the node spacing
  const double dlat = lat - C.lat[il];
  const double pa = C.tropo[it][il] + div_const(C.tropo[it][il + 1] - C.tropo[it][il], C.lat[il + 1] - C.lat[il],
                                                C.inv_dlat[il]) * dlat;
  const double pb = C.tropo[it + 1][il] + div_const(C.tropo[it + 1][il + 1] - C.tropo[it + 1][il],
                                                    C.lat[il + 1] - C.lat[il], C.inv_dlat[il]) * dlat;
  return pa + div_const(pb - pa, C.time[it + 1] - C.time[it], C.inv_dtime[it]) * (sec - C.time[it]);
}

// ---- zonal-mean climatologies of clim_t (module_meteo only) ---------------------------------

// view of a clim_zm_t (mptrac.h:3745-3776): vmr[ntime][np][nlat], compact, in device memory
struct DevZm {
  const double *time, *p, *lat, *vmr;
  int ntime, np, nlat, pad;
};

__device__ __forceinline__ double lin_nodes(double x0, double y0, double x1, double y1, double x) {   // LIN, mptrac.h:1351
  return y0 + (y1 - y0) / (x1 - x0) * (x - x0);
}

// clim_zm, mptrac.c:414-466
__device__ inline double clim_zm(const DevZm &Z, double t, double lat, double p) {
  double sec = fmod_trunc(t, 365.25 * 86400.);
  while (sec < 0)
    sec += 365.25 * 86400.;
  double p_help = p;
  if (p < Z.p[Z.np - 1])
    p_help = Z.p[Z.np - 1];
  else if (p > Z.p[0])
    p_help = Z.p[0];
  double lat_help = lat;
  if (lat < Z.lat[0])
    lat_help = Z.lat[0];
  else if (lat > Z.lat[Z.nlat - 1])
    lat_help = Z.lat[Z.nlat - 1];
  const int isec = locate_irr(Z.time, Z.ntime, sec, 1);
  const int ilat = locate_reg(Z.lat, Z.nlat, lat_help);
  const int ip = locate_irr(Z.p, Z.np, p_help, 0);
  auto vmr = [&](int it, int iz, int iy) {
    return Z.vmr[((size_t) it * (size_t) Z.np + (size_t) iz) * (size_t) Z.nlat + (size_t) iy];
  };
  const double aux00 = lin_nodes(Z.p[ip], vmr(isec, ip, ilat), Z.p[ip + 1], vmr(isec, ip + 1, ilat), p_help);
  const double aux01 = lin_nodes(Z.p[ip], vmr(isec, ip, ilat + 1), Z.p[ip + 1], vmr(isec, ip + 1, ilat + 1), p_help);
  const double aux10 = lin_nodes(Z.p[ip], vmr(isec + 1, ip, ilat), Z.p[ip + 1], vmr(isec + 1, ip + 1, ilat), p_help);
  const double aux11 =
    lin_nodes(Z.p[ip], vmr(isec + 1, ip, ilat + 1), Z.p[ip + 1], vmr(isec + 1, ip + 1, ilat + 1), p_help);
  const double aux0 = lin_nodes(Z.lat[ilat], aux00, Z.lat[ilat + 1], aux01, lat_help);
  const double aux1 = lin_nodes(Z.lat[ilat], aux10, Z.lat[ilat + 1], aux11, lat_help);
  const double aux = lin_nodes(Z.time[isec], aux0, Z.time[isec + 1], aux1, sec);
  return aux > 0.0 ? aux : 0.0;
}

// the time series of module_bound_cond's trace gases (clim_ts_t members of clim_t), in device memory
struct DevTracerSeries {
  const double *time[MPHIP_NTR], *vmr[MPHIP_NTR];
  int ntime[MPHIP_NTR];   // 0: not present
};

// clim_ts, mptrac.c:394-410
__device__ inline double clim_ts(const DevTracerSeries &T, int k, double t) {
  const double *time = T.time[k], *vmr = T.vmr[k];
  const int n = T.ntime[k];
  if (t <= time[0])
    return vmr[0];
  if (t >= time[n - 1])
    return vmr[n - 1];
  const int idx = locate_irr(time, n, t, 1);
  return lin_nodes(time[idx], vmr[idx], time[idx + 1], vmr[idx + 1], t);
}

// cos_sza, mptrac.c:1857-1897
__device__ inline double cos_sza(double sec, double lon, double lat) {
  // (the hour angle is ~1e4 rad years after 2000: no contraction, so that its roundings are the host's)
#pragma clang fp contract(off)
  const double d2r = kPi / 180.0;
  const double D = sec / 86400 - 0.5;
  const double g = (357.529 + 0.98560028 * D) * d2r;
  const double q = 280.459 + 0.98564736 * D;
  const double L = (q + 1.915 * sin(g) + 0.020 * sin(2 * g)) * d2r;
  const double e = (23.439 - 0.00000036 * D) * d2r;
  const double sindec = sin(e) * sin(L);
  const double ra = atan2(cos(e) * sin(L), cos(L));
  const double GMST = 18.697374558 + 24.06570982441908 * D;
  const double LST = GMST + lon / 15;
  const double h = LST / 12 * kPi - ra;
  const double lat_help = lat * d2r;
  return sin(lat_help) * sindec + cos(lat_help) * sqrt(1 - sindec * sindec) * cos(h);
}

// clim_oh, mptrac.c:89-120
__device__ inline double clim_oh(const mphip_ctl_t &ctl, const DevZm &Z, double t, double lon, double lat, double p) {
  const double csza_thresh = cos(85. * (kPi / 180.0));
  const double lat_ref = ctl.met_coord_type == 0 ? lat : ctl.met_utm_ref_lat;
  double lon_ref = ctl.met_coord_type == 0 ? lon : ctl.met_utm_ref_lon;
  while (lon_ref < -180.0)
    lon_ref += 360.0;
  while (lon_ref >= 180.0)
    lon_ref -= 360.0;
  const double oh = clim_zm(Z, t, lat_ref, p);
  if (ctl.oh_chem_beta <= 0)
    return oh;
  const double csza = cos_sza(t, lon_ref, lat_ref);
  const double denom = (csza >= csza_thresh) ? csza : csza_thresh;
  return oh * libm_exp(-ctl.oh_chem_beta / denom);
}

// nat_temperature, mptrac.c:8334-8355
__device__ inline double nat_temperature(double p, double h2o, double hno3) {
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

// tropo_weight, mptrac.c:12748-12770, split so that the climatological
// tropopause pressure (a function of time and latitude only) is looked up once
// for several pressures
__device__ __forceinline__ double tropo_pressure(const mphip_ctl_t &ctl, const DevClim &C, double time, double lat) {
  return clim_tropo(C, time, ctl.met_coord_type == 0 ? lat : ctl.met_utm_ref_lat);
}

// clim_tropo with its time part (second of the year, month interval) evaluated once: module_diff_turb asks for the
// tropopause at two latitudes of the same time.  Same operations as clim_tropo, same bits.
struct TropoTime {
  double sec;
  int it;
};

__device__ __forceinline__ TropoTime tropo_time(const DevClim &C, double t) {
  TropoTime tt;
  tt.sec = fmod_trunc(t, 365.25 * 86400.);
  while (tt.sec < 0)
    tt.sec += 365.25 * 86400.;
  tt.it = locate_irr(C.time, C.ntime, tt.sec, 1);
  return tt;
}

__device__ __forceinline__ double clim_tropo_at(const DevClim &C, const TropoTime &tt, double lat) {
  const int it = tt.it;
  const int il = locate_reg(C.lat, C.nlat, lat);
  const double dlat = lat - C.lat[il];
  const double pa = C.tropo[it][il] + div_const(C.tropo[it][il + 1] - C.tropo[it][il], C.lat[il + 1] - C.lat[il],
                                                C.inv_dlat[il]) * dlat;
  const double pb = C.tropo[it + 1][il] + div_const(C.tropo[it + 1][il + 1] - C.tropo[it + 1][il],
                                                    C.lat[il + 1] - C.lat[il], C.inv_dlat[il]) * dlat;
  return pa + div_const(pb - pa, C.time[it + 1] - C.time[it], C.inv_dtime[it]) * (tt.sec - C.time[it]);
}

__device__ __forceinline__ double tropo_pressure_at(const mphip_ctl_t &ctl, const DevClim &C, const TropoTime &tt,
                                                    double lat) {
  return clim_tropo_at(C, tt, ctl.met_coord_type == 0 ? lat : ctl.met_utm_ref_lat);
}

__device__ __forceinline__ double tropo_weight_pt(double pt, double p) {
  const double p1 = pt * 0.866877899;
  const double p0 = div_const(pt, 0.866877899, 1.0 / 0.866877899);
  if (p > p0)
    return 1;
  if (p < p1)
    return 0;
  return lin(p0, 1.0, p1, 0.0, p);
}

__device__ __forceinline__ double tropo_weight(const mphip_ctl_t &ctl, const DevClim &C, double time, double lat,
                                               double p) {
  return tropo_weight_pt(tropo_pressure(ctl, C, time, lat), p);
}

// pbl_weight, mptrac.c:8358-8376
__device__ __forceinline__ double pbl_weight(const mphip_ctl_t &ctl, double p, double pbl, double ps) {
  const double p1 = pbl - ctl.turb_pbl_trans * (ps - pbl);
  const double p0 = pbl;
  if (p > p0)
    return 1;
  if (p < p1)
    return 0;
  return lin(p0, 1.0, p1, 0.0, p);
}

// sedi, mptrac.c:12506-12535.  The default build forms the six quotients of the formula from two reciprocals,
// 1 / (s T) and 1 / ((T + 120) rho v r) with s = (T / 296.16)^1.5: eta = c s / (T + 120), rho = 100 p / (RA T),
// K = 2 eta / (rho v r), and 1 / K, 1 / eta follow by products -- a few ulp from the reference's operation order
// on a velocity that moves the pressure by parts per million.
__device__ __forceinline__ double sedi(double p, double T, double rp, double rhop, const double *lt) {
  const double r = rp * 1e-6;
#if MPHIP_EXACT_DIV
  const double rho = rho_air(p, T);
  const double eta = 1.8325e-5 * (416.16 / (T + 120.)) * libm_pow(T / 296.16, 1.5);
  const double v = fsqrt(div_const(8. * kKB * T, kPi * kMAir, 1.0 / (kPi * kMAir)));
  const double lambda = fdiv(2. * eta, rho * v);
  const double K = fdiv(lambda, r);
  const double G = 1. + K * (1.249 + 0.42 * libm_exp(lt, fdiv(-0.87, K)));
  return fdiv(2. * (r * r) * (rhop - rho) * kG0, 9. * eta) * G;
#else
  const double tr = T * (1.0 / 296.16);   // x^1.5 = x sqrt(x): within 2 ulp of pow()
  const double s = tr * fsqrt(tr);
  const double a = T + 120.;
  const double d = frcp(s * T);
  const double inv_t = d * s, inv_s = d * T;
  const double rho = (100. / kRA) * p * inv_t;
  const double v = fsqrt((8. * kKB / (kPi * kMAir)) * T);
  const double c = 1.8325e-5 * 416.16;    // eta = c s / a
  const double arvr = a * rho * v * r;
  const double K = (2. * c) * s * frcp(arvr);
  const double inv_K = (0.5 / c) * arvr * inv_s;
  const double G = 1. + K * (1.249 + 0.42 * libm_exp(lt, -0.87 * inv_K));
  // 2 r^2 (rhop - rho) g / (9 eta) with 1 / eta = a / (c s)
  return (2. * kG0 / (9. * c)) * (r * r) * (rhop - rho) * (a * inv_s) * G;
#endif
}

// ---- random numbers (mptrac.c:5784-5828) -----------------------------------

// Squares (Widynski 2022), five rounds, the reference's key (mptrac.c:5788).
// squares_from() takes the product ctr * key: consecutive counters differ by
// one key there, so the four draws behind a normal triple share one 64-bit
// multiply (3 of the 14 quarter-rate multiplies of a draw) instead of four.
constexpr uint64_t kSquaresKey = 0xc8e4fd154ce32f6dULL;

__device__ __forceinline__ uint64_t squares_from(uint64_t y) {
  uint64_t x = y, t;
  const uint64_t z = y + kSquaresKey;
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

__device__ __forceinline__ uint64_t squares(uint64_t ctr) {
  return squares_from(ctr * kSquaresKey);
}

// (double) r / (double) UINT64_MAX, mptrac.c:5810; the divisor is 2^64
__device__ __forceinline__ double uniform01_from(uint64_t y) {
  return (double) squares_from(y) * 0x1p-64;
}

__device__ __forceinline__ double uniform01(uint64_t ctr) {
  return uniform01_from(ctr * kSquaresKey);
}

// Single-precision sine / cosine exactly as the C library the reference's CPU
// build links against computes them (glibc >= 2.28 sinf/cosf, taken from the
// ARM optimized-routines "sincosf": reduction by multiples of pi/2 and two
// short polynomials, all in double precision, result rounded to float).
// Restated from the published algorithm; valid for |x| < 120 (the Box-Muller
// angle is in [0, 2 pi]).  Bit-identical to glibc 2.35 for every float in
// [0, 2 pi] (sampled on the device by the GPU suite, exhaustively on the CPU at design time).
//
// glibc branches to shorter paths for |y| < pi/4 and |y| < 2^-12; the general path below returns the same
// bits there (n = 0, x - 0 * hpi = x, and the polynomial rounds to y resp. 1.0f), so it is used for every
// lane -- one eighth of the Box-Muller angles would otherwise diverge.  The library evaluates one
// polynomial per function, chosen by the parity of the quadrant n: sine -> (n even ? sine : cosine
// polynomial), cosine the other way round, same reduced argument and signs for both; here each polynomial
// is evaluated once and the two results are assigned by that parity (a branch on n would run both sides
// in every wavefront, twice).  The sign of the cosine polynomial (the library's second coefficient table)
// is applied to its rounded result instead of to its five coefficients -- rounding commutes with
// negation: same bits, five multiplies fewer.
__device__ __forceinline__ void libm_sincosf_both(float y, float &sinv, float &cosv) {
  double x = (double) y;
  const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
  const int n = ((int32_t) (x * hpi_inv) + 0x800000) >> 24;
  x = x - n * hpi;
  const double xs = __hiloint2double(__double2hiint(x) ^ (((n + 1) & 2) << 30), __double2loint(x));   // quadrants 1, 2: -x
  const double x2 = x * x;
  float sp, cp;
  {
    const double x3 = xs * x2;
    const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    const double x7 = x3 * x2;
    const double sn = __builtin_fma(x3, -0x1.555545995a603p-3, xs);
    sp = (float) __builtin_fma(x7, s1, sn);
  }
  {
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 0x1p0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
    cp = __uint_as_float(__float_as_uint((float) __builtin_fma(x6, c2, c)) ^ ((uint32_t) (n & 2) << 30));
  }
  const bool even = (n & 1) == 0;
  sinv = even ? sp : cp;
  cosv = even ? cp : sp;
}

// which = 0: sinf(y), which = 1: cosf(y)
__device__ __forceinline__ float libm_sincosf(float y, int which) {
  float sv, cv;
  libm_sincosf_both(y, sv, cv);
  return which ? cv : sv;
}

// (double) r of a 64-bit integer in three instructions (both halves convert exactly, the fma rounds once)
__device__ __forceinline__ double u64_to_double(uint64_t r) {
  // (the conversion of the high word as an instruction of its own: written as a cast, the optimiser widens it back
  // into a 64-bit conversion of r >> 32 and adds a zero high part, one fp64 addition per call)
  double hi;
  asm("v_cvt_f64_u32_e32 %0, %1" : "=v"(hi) : "v"((uint32_t) (r >> 32)));
  return __builtin_fma(hi, 0x1p32, (double) (uint32_t) r);
}

// log u of the two uniforms u = X 2^-64 (X = (double) of a 64-bit draw; u = 0 -> -inf) behind two Box-Muller radii,
// the C library's bits.  Its algorithm has a second polynomial for arguments next to 1, and one uniform in sixteen
// is there: evaluated where the call stands, every wavefront would run both polynomials for both arguments.  Here
// both take the table path first (harmless where it does not apply), and the lanes with an argument next to 1 then
// share ONE copy of that polynomial in a loop that runs once per wavefront, seldom twice.
__device__ __forceinline__ void libm_log_unit_pair(const double *__restrict__ lt, double xa, double xb, double &la, double &lb) {
  const double ua = xa * 0x1p-64, ub = xb * 0x1p-64;     // exact
  const uint64_t ia = mphip_libm_bits(ua), ib = mphip_libm_bits(ub);
  la = mphip_libm_log_away(lt, ia);
  lb = mphip_libm_log_away(lt, ib);
  // (a zero draw, one in 2^64, joins the lanes that are sent round again)
  bool pa = mphip_libm_log_is_near_one(ia) | (xa == 0.0), pb = mphip_libm_log_is_near_one(ib) | (xb == 0.0);
  while (pa | pb) {
    const double x = pa ? ua : ub;
    const double v = x == 1.0 ? 0.0 : (x == 0.0 ? -__builtin_inf() : mphip_libm_log_near_one(x));
    if (pa) {
      la = v;
      pa = false;
    } else {
      lb = v;
      pb = false;
    }
  }
}

// Element i of the array module_rng(..., method = 1) would have produced for
// base counter c0 (mptrac.c:5821-5826): Box-Muller over the flat pairs
// (2j, 2j+1) of the uniform stream.
// y = (c0 + 2j) * key
__device__ __forceinline__ void box_muller(double log_u, uint64_t rb, double &even, double &odd) {
  // sqrt(-2 log u) with the IEEE square root: the reference's bits (u = 0 -> inf as there)
  const double r = sqrt_rn(-2.0 * log_u);
  const float phif = (float) (u64_to_double(rb) * (2.0 * kPi * 0x1p-64));   // 2 pi u, u = r 2^-64 (exact scaling)
  float sv, cv;
  libm_sincosf_both(phif, sv, cv);
  even = r * cv;
  odd = r * sv;
}

__device__ __forceinline__ void normal_pair_from(const double *__restrict__ ltab, uint64_t y, double &even, double &odd) {
  const uint64_t ra = squares_from(y), rb = squares_from(y + kSquaresKey);
  box_muller(libm_log(ltab, u64_to_double(ra) * 0x1p-64), rb, even, odd);
}

// two consecutive pairs (the four draws behind a triple of normals)
__device__ __forceinline__ void normal_two_pairs_from(const double *__restrict__ ltab, uint64_t y, double &ea, double &oa,
                                                      double &eb, double &ob) {
  // (the draws of the two angles are taken where they are used, not ahead of the logarithms: two registers each that
  // would be live through them, in a kernel that has none to spare)
  double la, lb;
  libm_log_unit_pair(ltab, u64_to_double(squares_from(y)), u64_to_double(squares_from(y + 2 * kSquaresKey)), la, lb);
  asm volatile("" : "+v"(la), "+v"(lb));
  box_muller(la, squares_from(y + kSquaresKey), ea, oa);
  box_muller(lb, squares_from(y + 3 * kSquaresKey), eb, ob);
}

__device__ __forceinline__ void normal_pair(const double *__restrict__ ltab, uint64_t c0, uint64_t j2, double &even,
                                            double &odd) {
  normal_pair_from(ltab, (c0 + j2) * kSquaresKey, even, odd);
}

// the three normals rs[3g], rs[3g+1], rs[3g+2] of global particle g.  They
// always come from two consecutive Box-Muller pairs; which three of the four
// outputs depends on the parity of 3g, i.e. alternates between neighbouring
// lanes -- so both pairs are evaluated by every lane and the outputs selected,
// instead of branching on the parity (a branch would run both sides per wave).
__device__ __forceinline__ void normal_triple(const double *__restrict__ ltab, uint64_t c0, uint64_t g, double &r0,
                                              double &r1, double &r2) {
  const uint64_t i0 = 3 * g;
  const bool odd = (i0 & 1) != 0;
  const uint64_t y = (c0 + (i0 & ~1ull)) * kSquaresKey;
  double ea, oa, eb, ob;
  normal_two_pairs_from(ltab, y, ea, oa, eb, ob);
  r0 = odd ? oa : ea;
  r1 = odd ? eb : oa;
  r2 = odd ? ob : eb;
}

// 1: module_diff_turb evaluates only the Box-Muller pair of rs[3g + 2] where there is no horizontal diffusion
// (Kx = 0: 57 % of the C3 particles, 46 % of the waves entirely).  Measured on C3: 0.766-0.771 ms per step against
// 0.758-0.762 without, and no fewer instructions issued (profiles/r04_variants.txt) -- off
#ifndef MPHIP_TURB_SHORTCUT
#define MPHIP_TURB_SHORTCUT 0
#endif
// rs[3g + 2] alone: the second Box-Muller pair of the triple
__device__ __forceinline__ double normal_third(const double *__restrict__ ltab, uint64_t c0, uint64_t g) {
  const uint64_t i0 = 3 * g;
  const uint64_t y = (c0 + (i0 & ~1ull)) * kSquaresKey;
  double eb, ob;
  normal_pair_from(ltab, y + 2 * kSquaresKey, eb, ob);
  return (i0 & 1) ? ob : eb;
}

// ---- per-particle state -----------------------------------------------------

struct Particle {
  double time, lon, lat, p, dt;
};

// module_timesteps, mptrac.c:6016-6041
__device__ __forceinline__ double timestep_of(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, double time,
                                              double lon, double lat, double t) {
  const double dir = ctl.direction;
  double dt = 0.0;
  if (dir * (time - ctl.t_start) >= 0 && dir * (time - ctl.t_stop) <= 0 && dir * (time - t) < 0)
    dt = t - time;
  if (M.local && (lon <= A.lon[0] || lon >= A.lon[M.nx - 1] || lat <= M.latmin || lat >= M.latmax))
    dt = 0.0;
  return dt;
}

// module_position, mptrac.c:5445-5488
__device__ __forceinline__ void position(const DevMet &M, const Axes &A, Particle &P) {
  if (M.coord_type == 0) {
    double lon = fmod_trunc(P.lon, 360.);
    double lat = fmod_trunc(P.lat, 360.);
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
    P.lon = lon;
    P.lat = lat;
  } else {
    const double x0 = A.lon[0], x1 = A.lon[M.nx - 1], y0 = A.lat[0], y1 = A.lat[M.ny - 1];
    P.lon = (x0 < x1) ? dmin(dmax(P.lon, x0), x1) : dmin(dmax(P.lon, x1), x0);
    P.lat = (y0 < y1) ? dmin(dmax(P.lat, y0), y1) : dmin(dmax(P.lat, y1), y0);
  }
  const double ptop = A.p[M.np - 1];
  if (P.p < ptop) {
    P.p = ptop * ptop / P.p;
  } else if (P.p > 300.) {
    // INTPOL_2D(ps, 0) on the zeroed stencil of INTPOL_INIT (mptrac.c:5449,
    // 5484): indices 0, weights 0 -> the value at grid node [1][1].  Reference
    // behaviour, reproduced; every lane reads the same four records.
    const Stencil s = stencil_zero();
    SurfA c;
    load_sfa(M, s, c);
    const double ps = sfa_time_2d(c, s, time_weight(M, P.time), 0);
    if (P.p > ps)
      P.p = ps * ps / P.p;
  }
}

// module_advect, pressure-level branch, mptrac.c:3612-3677
// `hook(i)` runs right after the gathers of stage i were issued: work that does
// not depend on them (the random numbers of the later modules) fills the wait.
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

template <int ADVECT, class Hook>
__device__ __forceinline__ void advect_n(const DevMet &M, const Axes &A, Particle &P, Hook &hook, WindCache &wc) {
  const int ct = M.coord_type;
  const double dt = P.dt;
  double u = 0, v = 0, w = 0, um = 0, vm = 0, wm = 0;
  double x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
  for (int i = 0; i < ADVECT; i++) {
    double dts;
    if (i == 0) {
      dts = 0.0;
      x0 = P.lon;
      x1 = P.lat;
      x2 = P.p;
    } else {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2coord(ct, dts * u, P.lat);
      x1 = P.lat + dy2coord(ct, dts * v);
      x2 = P.p + dts * w;
    }
    const double tm = P.time + dts;
    Stencil s;
    stencil_3d(M, A, x2, x0, x1, s);
    if (wc.enabled) {
      load_wind_cached(M, s, wc);
      hook(i);
      wind_cache_wait(wc);
    } else {
      load_wind(M, s, wc.c);
      hook(i);
    }
    const WindCorners &c = wc.c;
    const double wt = time_weight(M, tm);
    u = wind_time_3d(c, s, wt, 0);
    v = wind_time_3d(c, s, wt, 1);
    w = wind_time_3d(c, s, wt, 2);
    double k = 1.0;
    if (ADVECT == 2)
      k = (i == 0 ? 0.0 : 1.0);
    else if (ADVECT == 4)
      k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
    um += k * u;
    vm += k * v;
    wm += k * w;
  }
  P.time += dt;
  P.lon += dx2coord(ct, dt * um, (ADVECT == 2 ? x1 : P.lat));
  P.lat += dy2coord(ct, dt * vm);
  P.p += dt * wm;
}

template <class Hook>
__device__ __forceinline__ void advect(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                       Hook &hook, WindCache &wc) {
  if (ctl.advect == 4)
    advect_n<4>(M, A, P, hook, wc);
  else if (ctl.advect == 2)
    advect_n<2>(M, A, P, hook, wc);
  else
    advect_n<1>(M, A, P, hook, wc);
}

__device__ __forceinline__ void advect(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                       WindCache &wc) {
  NoHook none;
  advect(ctl, M, A, P, none, wc);
}

// module_advect, zeta / eta branch (mptrac.c:3681-3757); zeta is the particle's
// vertical-coordinate quantity q[qnt_zeta | qnt_eta]
template <int ADVECT>
__device__ __forceinline__ void advect_ml_n(const DevMet &M, const Axes &A, Particle &P, double &zeta) {
  const int ct = M.coord_type;
  const double dt = P.dt;
  Stencil4 s;
  stencil_4d(M, A, M.pll[0], M.pll[1], P.time, P.p, P.lon, P.lat, s);
  zeta = ml_field(M, M.zl[0], M.zl[1], s);
  double u = 0, v = 0, wdot = 0, um = 0, vm = 0, wdotm = 0, x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
  for (int i = 0; i < ADVECT; i++) {
    double dts;
    if (i == 0) {
      dts = 0.0;
      x0 = P.lon;
      x1 = P.lat;
      x2 = zeta;
    } else {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2coord(ct, dts * u, P.lat);
      x1 = P.lat + dy2coord(ct, dts * v);
      x2 = zeta + dts * wdot;
    }
    stencil_4d(M, A, M.zl[0], M.zl[1], P.time + dts, x2, x0, x1, s);
    MlCorners c;
    load_ml(M, s, c);
    u = ml_packed(c, s, 0);
    v = ml_packed(c, s, 1);
    wdot = ml_packed(c, s, 2);
    double k = 1.0;
    if (ADVECT == 2)
      k = (i == 0 ? 0.0 : 1.0);
    else if (ADVECT == 4)
      k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
    um += k * u;
    vm += k * v;
    wdotm += k * wdot;
  }
  P.time += dt;
  P.lon += dx2coord(ct, dt * um, (ADVECT == 2 ? x1 : P.lat));
  P.lat += dy2coord(ct, dt * vm);
  zeta += dt * wdotm;
  stencil_4d(M, A, M.zl[0], M.zl[1], P.time, zeta, P.lon, P.lat, s);
  P.p = ml_field(M, M.pll[0], M.pll[1], s);
}

__device__ __forceinline__ void advect_ml(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                          double &zeta) {
  if (ctl.advect == 4)
    advect_ml_n<4>(M, A, P, zeta);
  else if (ctl.advect == 2)
    advect_ml_n<2>(M, A, P, zeta);
  else
    advect_ml_n<1>(M, A, P, zeta);
}

// module_advect with ADVECT_VERT_COORD 2 (mptrac.c:3609-3678, branch 3649-3659): the pressure-level
// integrator with u, v, omega interpolated from the model levels -- intpol_met_4d_zeta with the pressure
// field pl as the height variable, first call initialises the stencil, the two others re-use it
template <int ADVECT>
__device__ __forceinline__ void advect_mlp_n(const DevMet &M, const Axes &A, Particle &P) {
  const int ct = M.coord_type;
  const double dt = P.dt;
  Stencil4 s;
  double u = 0, v = 0, w = 0, um = 0, vm = 0, wm = 0, x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
  for (int i = 0; i < ADVECT; i++) {
    double dts;
    if (i == 0) {
      dts = 0.0;
      x0 = P.lon;
      x1 = P.lat;
      x2 = P.p;
    } else {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2coord(ct, dts * u, P.lat);
      x1 = P.lat + dy2coord(ct, dts * v);
      x2 = P.p + dts * w;
    }
    stencil_4d(M, A, M.pll[0], M.pll[1], P.time + dts, x2, x0, x1, s);
    MlCorners c;
    load_ml(M, s, c);
    u = ml_packed(c, s, 0);
    v = ml_packed(c, s, 1);
    w = ml_packed(c, s, 2);
    double k = 1.0;
    if (ADVECT == 2)
      k = (i == 0 ? 0.0 : 1.0);
    else if (ADVECT == 4)
      k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
    um += k * u;
    vm += k * v;
    wm += k * w;
  }
  P.time += dt;
  P.lon += dx2coord(ct, dt * um, (ADVECT == 2 ? x1 : P.lat));
  P.lat += dy2coord(ct, dt * vm);
  P.p += dt * wm;
}

__device__ __forceinline__ void advect_mlp(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P) {
  if (ctl.advect == 4)
    advect_mlp_n<4>(M, A, P);
  else if (ctl.advect == 2)
    advect_mlp_n<2>(M, A, P);
  else
    advect_mlp_n<1>(M, A, P);
}

// module_advect_init, mptrac.c:3777-3784: pressure consistent with zeta
__device__ __forceinline__ double pressure_from_zeta(const DevMet &M, const Axes &A, double time, double zeta,
                                                     double lon, double lat) {
  Stencil4 s;
  stencil_4d(M, A, M.zl[0], M.zl[1], time, zeta, lon, lat, s);
  return ml_field(M, M.pll[0], M.pll[1], s);
}

// ---- model-level interpolation, fast path -----------------------------------
// Same indices and the same arithmetic as stencil_4d / ml_field above, for height fields whose columns
// are strictly monotonic (checked on the device when the grids are packed; otherwise the kernels run the
// path above, which reproduces the reference's bisection read by read).  For a monotonic column
// locate_irr_float returns the unique bracketing index whatever its first guess, so the search may start
// from the previous result; the packed pairs {h0,h1} make one 16-byte gather return levels k and k + 1 of
// both snapshots, where the reference-shaped code issues four dword gathers.

// The column index type COL of the functions below is size_t (any grid) or uint32_t: the lean instantiations run
// only where launch_step has checked that 24 bytes x cells fit 32 bits, and a 32-bit index lets the loads take
// their base from SGPRs and one offset register per lane instead of a 64-bit address built per load.
template <class COL>
__device__ __forceinline__ COL col_ml_as(const DevMet &M, int ix, int iy) {
  return ((COL) ix * (COL) M.ny + (COL) iy) * (COL) M.npl;
}

// levels k, k + 1 of both snapshots of column `col`: {h0[k], h1[k], h0[k+1], h1[k+1]}
template <class COL>
__device__ __forceinline__ f32x4u load_h4(const float *__restrict__ h2, COL col, int k) {
  return *(const f32x4u *) ((const char *) h2 + (COL) 8 * (col + (COL) k));
}

// The searches compare the (double) height x with single-precision levels v.  With xd = the largest float <= x,
// x >= v is v <= xd and x < v is v > xd for every float v (v <= x < next float above xd), so one conversion per
// stencil replaces a conversion and a double-precision compare per level read.  (NaN: every compare false,
// as with the doubles.)
__device__ __forceinline__ float float_below(double x) {
  float f = (float) x;
  if ((double) f > x) {   // rounded up: one float down
    const uint32_t b = __float_as_uint(f);
    f = __uint_as_float(f > 0.f ? b - 1u : (f < 0.f ? b + 1u : 0x80000001u));
  }
  return f;
}

// -1: the index must decrease, +1: increase, 0: level pair (lo, hi) brackets x (bisection semantics of
// locate_irr_float: ascending lo <= x < hi, descending lo > x >= hi); xd = float_below(x)
__device__ __forceinline__ int bracket_dir(float lo, float hi, float xd) {
  if (lo < hi)
    return lo > xd ? -1 : (hi <= xd ? 1 : 0);
  return lo <= xd ? -1 : (hi > xd ? 1 : 0);
}

// index of snapshot t (0 / 1) in a monotonic column by bisection over the packed pairs
template <class COL>
__device__ __forceinline__ int bisect_pair(const float *__restrict__ h2, COL col, int n, float xd, int t) {
  int lo = 0, hi = n - 1;
  const int mid0 = (hi + lo) >> 1;
  const f32x4u m = load_h4(h2, col, mid0);
  const bool asc = m[t] < m[2 + t];
  while (hi > lo + 1) {
    const int mid = (hi + lo) >> 1;
    const float v = ((const float *) ((const char *) h2 + (COL) 8 * (col + (COL) mid)))[t];
    if (asc ? (v > xd) : (v <= xd))
      hi = mid;
    else
      lo = mid;
  }
  return lo;
}

// The level indices of the four columns of a stencil, searched side by side: every round probes one level pair of
// all four columns (four independent loads in flight; a column is done when both snapshots bracket, at most four
// rounds, then bisection).  The
// search is exact whatever its first guess (monotonic columns), so all four start from `guess`.  The state of a
// column is one word -- k0 | k1 << 10 | probe << 20 | found << 30 -- because the model-level kernels sit at their
// register limit (the records at the result are loaded again by the caller: one more round, but a parallel one).
constexpr int kLockstepMaxLevels = 1024;

__device__ __forceinline__ uint32_t pair_search_step(uint32_t S, const f32x4u v, int n, float x) {
  const int k = (int) ((S >> 20) & 1023u);
  const uint32_t found = S >> 30;
  const int d0 = bracket_dir(v[0], v[2], x), d1 = bracket_dir(v[1], v[3], x);
  const bool e0 = (d0 == 0) | ((d0 < 0) & (k == 0)) | ((d0 > 0) & (k == n - 2));
  const bool e1 = (d1 == 0) | ((d1 < 0) & (k == 0)) | ((d1 > 0) & (k == n - 2));
  const bool t0 = e0 & !(found & 1u), t1 = e1 & !(found & 2u);
  uint32_t k0 = S & 1023u, k1 = (S >> 10) & 1023u;
  k0 = t0 ? (uint32_t) k : k0;
  k1 = t1 ? (uint32_t) k : k1;
  const uint32_t f = found | (t0 ? 1u : 0u) | (t1 ? 2u : 0u);
  const int d = (f & 1u) ? d1 : d0;
  const int kn = k + (f == 3u ? 0 : d);
  return k0 | (k1 << 10) | ((uint32_t) kn << 20) | (f << 30);
}

template <class COL>
__device__ __forceinline__ void pair_search_finish(uint32_t S, const float *__restrict__ h2, COL col, int n, float x,
                                                   int &k0, int &k1) {
  k0 = (int) (S & 1023u);
  k1 = (int) ((S >> 10) & 1023u);
  if (!(S & (1u << 30)))
    k0 = bisect_pair(h2, col, n, x, 0);
  if (!(S & (2u << 30)))
    k1 = bisect_pair(h2, col, n, x, 1);
}

template <class COL>
__device__ __forceinline__ void locate_pairs4(const float *__restrict__ h2, COL ca, COL cb, COL cc, COL cd, int n,
                                              float x, int guess, int &kmin, int &kmax) {
  const uint32_t k = (uint32_t) (guess < 0 ? 0 : (guess > n - 2 ? n - 2 : guess));
  uint32_t A = k | (k << 10) | (k << 20), B = A, C = A, D = A;
#pragma unroll 1
  for (int it = 0; it < 4; it++) {
    // (the whole wave leaves together: a lane whose searches are done probes the same records again)
    const f32x4u va = load_h4(h2, ca, (int) ((A >> 20) & 1023u)), vb = load_h4(h2, cb, (int) ((B >> 20) & 1023u)),
                 vc = load_h4(h2, cc, (int) ((C >> 20) & 1023u)), vd = load_h4(h2, cd, (int) ((D >> 20) & 1023u));
    A = pair_search_step(A, va, n, x);
    B = pair_search_step(B, vb, n, x);
    C = pair_search_step(C, vc, n, x);
    D = pair_search_step(D, vd, n, x);
    if (__all((A & B & C & D) >> 30 == 3u))
      break;
  }
  int a0, a1, b0, b1, c0, c1, d0, d1;
  pair_search_finish(A, h2, ca, n, x, a0, a1);
  pair_search_finish(B, h2, cb, n, x, b0, b1);
  pair_search_finish(C, h2, cc, n, x, c0, c1);
  pair_search_finish(D, h2, cd, n, x, d0, d1);
  kmin = min(min(min(a0, a1), min(b0, b1)), min(min(c0, c1), min(d0, d1)));
  kmax = max(max(max(a0, a1), max(b0, b1)), max(max(c0, c1), max(d0, d1)));
}

__device__ __forceinline__ double level_pair_value(const Stencil4 &s, const f32x4u c00, const f32x4u c01,
                                                   const f32x4u c10, const f32x4u c11, int l) {
  const int e = 2 * l;   // l = 0: level k, l = 1: level k + 1
  const double v00 = s.wt * (double) (c00[e + 1] - c00[e]) + (double) c00[e];
  const double v01 = s.wt * (double) (c01[e + 1] - c01[e]) + (double) c01[e];
  const double v10 = s.wt * (double) (c10[e + 1] - c10[e]) + (double) c10[e];
  const double v11 = s.wt * (double) (c11[e + 1] - c11[e]) + (double) c11[e];
  const double a = s.wy * (v01 - v00) + v00;
  const double b = s.wy * (v11 - v10) + v10;
  return s.wx * (b - a) + a;
}

// ---- level window -----------------------------------------------------------
// A wave waits for its slowest lane: when each of the four columns is searched by probing one level pair per
// round, nearly every wave has a lane that needs a second probe, then the records at the lowest index, then
// another pair for the reference's while loop -- four or five dependent gather rounds per stencil, and the
// model-level kernels were idle 43 % of the time waiting for them.  So the first (and almost always only) round
// reads a WINDOW of four levels around the hint from every column, two 16-byte loads each: levels b .. b + 3,
// which hold the pairs b, b + 1, b + 2.  If all eight searches (four columns, two snapshots) end inside it, the
// indices, the records at the lowest index and the records the while loop steps through are all in registers.
// A lane with a search that leaves the window takes the lockstep search and loads what it needs.
#ifndef MPHIP_LEVEL_WINDOW
#define MPHIP_LEVEL_WINDOW 1
#endif

// index of snapshot t among the pairs b, b + 1, b + 2 (w0: levels b, b + 1; w1: b + 2, b + 3), or -1
__device__ __forceinline__ int window_index(const f32x4u w0, const f32x4u w1, int t, int b, int n, float xd) {
  const float l0 = w0[t], l1 = w0[2 + t], l2 = w1[t], l3 = w1[2 + t];
  const int d0 = bracket_dir(l0, l1, xd), d1 = bracket_dir(l1, l2, xd), d2 = bracket_dir(l2, l3, xd);
  int k = -1;
  k = ((d2 == 0) | ((d2 > 0) & (b + 2 == n - 2))) ? b + 2 : k;   // (ends of the column: as the bisection)
  k = (d1 == 0) ? b + 1 : k;
  k = ((d0 == 0) | ((d0 < 0) & (b == 0))) ? b : k;
  return k;
}

// the pair record {h0[k], h1[k], h0[k+1], h1[k+1]} of a column: from its window if it holds pair k
template <class COL>
__device__ __forceinline__ f32x4u window_record(const float *__restrict__ h2, COL col, const f32x4u w0, const f32x4u w1,
                                                int b, bool have, int k) {
  const int j = k - b;
  if (have && (unsigned) j <= 2u) {
    f32x4u r;
    r[0] = j == 0 ? w0[0] : (j == 1 ? w0[2] : w1[0]);
    r[1] = j == 0 ? w0[1] : (j == 1 ? w0[3] : w1[1]);
    r[2] = j == 0 ? w0[2] : (j == 1 ? w1[0] : w1[2]);
    r[3] = j == 0 ? w0[3] : (j == 1 ? w1[1] : w1[3]);
    return r;
  }
  return load_h4(h2, col, k);
}

// stencil_4d on a packed height field; `hint` is any earlier vertical index (e.g. of the previous stage)
#ifndef MPHIP_ML_HORIZ_FAST
#define MPHIP_ML_HORIZ_FAST 1
#endif
__device__ __forceinline__ bool lat_fast(const DevMet &M, const Axes &A, double lat2, int &iy, double &y1, double &yinv);
template <class COL = size_t>
__device__ __forceinline__ void stencil_4d_fast(const DevMet &M, const Axes &A, const float *__restrict__ h2, double ts,
                                                double height, double lon, double lat, int hint, Stencil4 &s) {
  double lon2, lat2;
  AxisHit hy;
  bool guessed = false;
#if MPHIP_ML_HORIZ_FAST
  if constexpr (std::is_same<COL, uint32_t>::value) {
    // the lean instantiations (32-bit columns: a lat/lon grid with the look-up tables, launch_step): the horizontal
    // indices from verified guesses, as the pressure-level kernels take them (lon_fast / lat_fast)
    lat2 = vmin_s(vmax_s(lat, M.latmin), M.latmax);
    lon2 = lon + (lon < M.lon_first ? 360.0 : (lon > M.lon_last ? -360.0 : 0.0));
#if MPHIP_EXACT_DIV
    s.ix = locate_reg(A.lon, M.nx, lon2);
#else
    s.ix = clamp0_s((int) ((lon2 - M.lon_first) * M.inv_dlon0), M.nx - 2);
#endif
    guessed = (fabs(lon) < 360.0) & lat_fast(M, A, lat2, hy.i, hy.x1, hy.inv);
    hy.x0 = A.lat[hy.i];
  }
#endif
  if (!guessed) {
    check_horizontal(M, A, lon, lat, lon2, lat2);
    hy = hit_lat(M, A, lat2);
    s.ix = locate_lon(M, A, lon2);
  }
  s.iy = hy.i;
  const int n = M.npl;
  const COL c00 = col_ml_as<COL>(M, s.ix, s.iy), c10 = col_ml_as<COL>(M, s.ix + 1, s.iy),
            c01 = col_ml_as<COL>(M, s.ix, s.iy + 1), c11 = col_ml_as<COL>(M, s.ix + 1, s.iy + 1);
  const float hd = float_below(height);
  int kmin, kmax;
#if MPHIP_LEVEL_WINDOW
  const int b = min(max(hint - 1, 0), max(n - 4, 0));
  f32x4u a0 = {}, a1 = {}, b0 = {}, b1 = {}, e0 = {}, e1 = {}, f0 = {}, f1 = {};   // windows of c00, c10, c01, c11
  bool have = false;
  if (n >= 4) {
    a0 = load_h4(h2, c00, b); a1 = load_h4(h2, c00, b + 2);
    b0 = load_h4(h2, c10, b); b1 = load_h4(h2, c10, b + 2);
    e0 = load_h4(h2, c01, b); e1 = load_h4(h2, c01, b + 2);
    f0 = load_h4(h2, c11, b); f1 = load_h4(h2, c11, b + 2);
    const int k0 = window_index(a0, a1, 0, b, n, hd), k1 = window_index(a0, a1, 1, b, n, hd);
    const int k2 = window_index(b0, b1, 0, b, n, hd), k3 = window_index(b0, b1, 1, b, n, hd);
    const int k4 = window_index(e0, e1, 0, b, n, hd), k5 = window_index(e0, e1, 1, b, n, hd);
    const int k6 = window_index(f0, f1, 0, b, n, hd), k7 = window_index(f0, f1, 1, b, n, hd);
    kmin = min(min(min(k0, k1), min(k2, k3)), min(min(k4, k5), min(k6, k7)));
    kmax = max(max(max(k0, k1), max(k2, k3)), max(max(k4, k5), max(k6, k7)));
    have = kmin >= 0;
  }
  if (!have)
    locate_pairs4(h2, c00, c10, c01, c11, n, hd, hint, kmin, kmax);
  s.iz = kmin;
  f32x4u q00 = window_record(h2, c00, a0, a1, b, have, s.iz), q10 = window_record(h2, c10, b0, b1, b, have, s.iz),
         q01 = window_record(h2, c01, e0, e1, b, have, s.iz), q11 = window_record(h2, c11, f0, f1, b, have, s.iz);
#else
  locate_pairs4(h2, c00, c10, c01, c11, n, hd, hint, kmin, kmax);
  s.iz = kmin;
  f32x4u q00 = load_h4(h2, c00, s.iz), q10 = load_h4(h2, c10, s.iz), q01 = load_h4(h2, c01, s.iz),
         q11 = load_h4(h2, c11, s.iz);
#endif
  s.wt = div_const(ts - M.time0, M.time1 - M.time0, M.inv_dtime);
  s.wx = div_const(lon2 - A.lon[s.ix], A.lon[s.ix + 1] - A.lon[s.ix], A.inv_lon[s.ix]);
  s.wy = div_const(lat2 - hy.x0, hy.x1 - hy.x0, hy.inv);
  double bot = level_pair_value(s, q00, q01, q10, q11, 0);
  double top = level_pair_value(s, q00, q01, q10, q11, 1);
  const float g0 = h2[0], g1 = h2[2];   // heights0[0][0][0], heights0[0][0][1]
  while (((g0 > g1) && ((bot <= height) || (top > height)) && (bot >= height) && (s.iz < kmax))
         || ((g0 < g1) && ((bot >= height) || (top < height)) && (bot <= height) && (s.iz < kmax))) {
    s.iz++;
    bot = top;
#if MPHIP_LEVEL_WINDOW
    q00 = window_record(h2, c00, a0, a1, b, have, s.iz);
    q01 = window_record(h2, c01, e0, e1, b, have, s.iz);
    q10 = window_record(h2, c10, b0, b1, b, have, s.iz);
    q11 = window_record(h2, c11, f0, f1, b, have, s.iz);
#else
    q00 = load_h4(h2, c00, s.iz);
    q01 = load_h4(h2, c01, s.iz);
    q10 = load_h4(h2, c10, s.iz);
    q11 = load_h4(h2, c11, s.iz);
#endif
    top = level_pair_value(s, q00, q01, q10, q11, 1);
  }
  s.wz = (height - bot) / (top - bot);
}

// ml_field on a packed pair array
template <class COL = size_t>
__device__ __forceinline__ double ml_field_fast(const DevMet &M, const float *__restrict__ a2, const Stencil4 &s) {
  double v[2][2][2];
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const f32x4u q = load_h4(a2, col_ml_as<COL>(M, s.ix + di, s.iy + dj), s.iz);
#pragma unroll
      for (int l = 0; l < 2; l++)
        v[di][dj][l] = s.wt * (double) (q[2 * l + 1] - q[2 * l]) + (double) q[2 * l];
    }
  return ml_combine(s, v[0][0][0], v[1][0][0], v[0][1][0], v[1][1][0], v[0][0][1], v[1][0][1], v[0][1][1],
                    v[1][1][1]);
}

// {ul,vl,zeta_dot} corners with the cell they belong to (as WindCache)
struct MlCache {
  MlCorners c;
  int ix, iy, iz;
};

__device__ __forceinline__ void ml_cache_reset(MlCache &w) {
  w.ix = w.iy = w.iz = -1;
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
#pragma unroll
      for (int k = 0; k < 3; k++)
        w.c.r[di][dj][k] = f32x4u{ 0.f, 0.f, 0.f, 0.f };
}

#ifndef MPHIP_ML_CACHE
#define MPHIP_ML_CACHE 0
#endif
template <class COL = size_t>
__device__ __forceinline__ void load_ml_cached(const DevMet &M, const Stencil4 &s, MlCache &w) {
#if !MPHIP_ML_CACHE
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      const COL off = (COL) 24 * (col_ml_as<COL>(M, s.ix + di, s.iy + dj) + (COL) s.iz);
#pragma unroll
      for (int k = 0; k < 3; k++)
        w.c.r[di][dj][k] = *(const f32x4u *) ((const char *) M.mlw + off + (COL) (16 * k));
    }
  return;
#endif
  if (s.ix != w.ix || s.iy != w.iy || s.iz != w.iz) {
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
      for (int dj = 0; dj < 2; dj++) {
        const float *q = M.mlw + 6 * (col_ml(M, s.ix + di, s.iy + dj) + (size_t) s.iz);
        asm volatile("global_load_dwordx4 %0, %3, off\n\t"
                     "global_load_dwordx4 %1, %3, off offset:16\n\t"
                     "global_load_dwordx4 %2, %3, off offset:32"
                     : "+v"(w.c.r[di][dj][0]), "+v"(w.c.r[di][dj][1]), "+v"(w.c.r[di][dj][2])
                     : "v"(q)
                     : "memory");
      }
    w.ix = s.ix;
    w.iy = s.iy;
    w.iz = s.iz;
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(w.c.r[0][0][0]), "+v"(w.c.r[0][0][1]), "+v"(w.c.r[0][0][2]), "+v"(w.c.r[0][1][0]),
                 "+v"(w.c.r[0][1][1]), "+v"(w.c.r[0][1][2]), "+v"(w.c.r[1][0][0]), "+v"(w.c.r[1][0][1]),
                 "+v"(w.c.r[1][0][2]), "+v"(w.c.r[1][1][0]), "+v"(w.c.r[1][1][1]), "+v"(w.c.r[1][1][2])
               :
               : "memory");
}

// module_advect, zeta / eta branch, on the packed height fields (same arithmetic as advect_ml_n)
template <int ADVECT, class COL>
__device__ __forceinline__ void advect_ml_fast_n(const DevMet &M, const Axes &A, Particle &P, double &zeta, int &kz) {
  const int ct = M.coord_type;
  const double dt = P.dt;
  Stencil4 s;
  stencil_4d_fast<COL>(M, A, M.pl2, P.time, P.p, P.lon, P.lat, kz, s);
  zeta = ml_field_fast<COL>(M, M.zl2, s);
  double u = 0, v = 0, wdot = 0, um = 0, vm = 0, wdotm = 0, x0 = 0, x1 = 0, x2 = 0;
  MlCache mc;
  ml_cache_reset(mc);
#pragma unroll
  for (int i = 0; i < ADVECT; i++) {
    double dts;
    if (i == 0) {
      dts = 0.0;
      x0 = P.lon;
      x1 = P.lat;
      x2 = zeta;
    } else {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2coord(ct, dts * u, P.lat);
      x1 = P.lat + dy2coord(ct, dts * v);
      x2 = zeta + dts * wdot;
    }
    stencil_4d_fast<COL>(M, A, M.zl2, P.time + dts, x2, x0, x1, s.iz, s);
    load_ml_cached<COL>(M, s, mc);
    u = ml_packed(mc.c, s, 0);
    v = ml_packed(mc.c, s, 1);
    wdot = ml_packed(mc.c, s, 2);
    double k = 1.0;
    if (ADVECT == 2)
      k = (i == 0 ? 0.0 : 1.0);
    else if (ADVECT == 4)
      k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
    um += k * u;
    vm += k * v;
    wdotm += k * wdot;
  }
  P.time += dt;
  P.lon += dx2coord(ct, dt * um, (ADVECT == 2 ? x1 : P.lat));
  P.lat += dy2coord(ct, dt * vm);
  zeta += dt * wdotm;
  stencil_4d_fast<COL>(M, A, M.zl2, P.time, zeta, P.lon, P.lat, s.iz, s);
  P.p = ml_field_fast<COL>(M, M.pl2, s);
  kz = s.iz;
}

template <class COL = size_t>
__device__ __forceinline__ void advect_ml_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                               double &zeta, int &kz) {
  if (ctl.advect == 4)
    advect_ml_fast_n<4, COL>(M, A, P, zeta, kz);
  else if (ctl.advect == 2)
    advect_ml_fast_n<2, COL>(M, A, P, zeta, kz);
  else
    advect_ml_fast_n<1, COL>(M, A, P, zeta, kz);
}

// ADVECT_VERT_COORD 2 on the packed pressure field (monotonic columns), as advect_mlp_n
template <int ADVECT, class COL>
__device__ __forceinline__ void advect_mlp_fast_n(const DevMet &M, const Axes &A, Particle &P, int &kz) {
  const int ct = M.coord_type;
  const double dt = P.dt;
  Stencil4 s;
  s.iz = kz;
  double u = 0, v = 0, w = 0, um = 0, vm = 0, wm = 0, x0 = 0, x1 = 0, x2 = 0;
  MlCache mc;
  ml_cache_reset(mc);
#pragma unroll
  for (int i = 0; i < ADVECT; i++) {
    double dts;
    if (i == 0) {
      dts = 0.0;
      x0 = P.lon;
      x1 = P.lat;
      x2 = P.p;
    } else {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2coord(ct, dts * u, P.lat);
      x1 = P.lat + dy2coord(ct, dts * v);
      x2 = P.p + dts * w;
    }
    stencil_4d_fast<COL>(M, A, M.pl2, P.time + dts, x2, x0, x1, s.iz, s);
    load_ml_cached<COL>(M, s, mc);
    u = ml_packed(mc.c, s, 0);
    v = ml_packed(mc.c, s, 1);
    w = ml_packed(mc.c, s, 2);
    double k = 1.0;
    if (ADVECT == 2)
      k = (i == 0 ? 0.0 : 1.0);
    else if (ADVECT == 4)
      k = (i == 0 || i == 3 ? 1.0 / 6.0 : 2.0 / 6.0);
    um += k * u;
    vm += k * v;
    wm += k * w;
  }
  P.time += dt;
  P.lon += dx2coord(ct, dt * um, (ADVECT == 2 ? x1 : P.lat));
  P.lat += dy2coord(ct, dt * vm);
  P.p += dt * wm;
  kz = s.iz;
}

template <class COL = size_t>
__device__ __forceinline__ void advect_mlp_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                                int &kz) {
  if (ctl.advect == 4)
    advect_mlp_fast_n<4, COL>(M, A, P, kz);
  else if (ctl.advect == 2)
    advect_mlp_fast_n<2, COL>(M, A, P, kz);
  else
    advect_mlp_fast_n<1, COL>(M, A, P, kz);
}

__device__ __forceinline__ double pressure_from_zeta_fast(const DevMet &M, const Axes &A, double time, double zeta,
                                                          double lon, double lat) {
  Stencil4 s;
  stencil_4d_fast(M, A, M.zl2, time, zeta, lon, lat, M.npl / 2, s);
  return ml_field_fast(M, M.pl2, s);
}

// the Kz blend evaluated at a displaced pressure, mptrac.c:4669-4688
__device__ __forceinline__ double kz_blend(const mphip_ctl_t &ctl, double pt, double p, double pbl, double ps) {
  const double wpbl = pbl_weight(ctl, p, pbl, ps);
  const double wtrop = tropo_weight_pt(pt, p) * (1.0 - wpbl);
  const double wstrat = 1.0 - wpbl - wtrop;
  return wpbl * ctl.turb_dz_pbl + wtrop * ctl.turb_dz_trop + wstrat * ctl.turb_dz_strat;
}

// module_diff_turb, mptrac.c:4603-4733
__device__ __forceinline__ void diff_turb(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevClim &C,
                                          Particle &P, uint64_t ctr, uint64_t g, const double *pre, const double *ltab) {
  const int ct = M.coord_type;
  Stencil s = stencil_zero();
  stencil_2d(M, A, P.lon, P.lat, s);
  SurfA c;
  load_sfa(M, s, c);
  const double wt = time_weight(M, P.time);
  const double pbl = sfa_time_2d(c, s, wt, 1);
  if (ctl.turb_pbl_scheme > 0 && P.p >= pbl)
    return;
  const double ps = sfa_time_2d(c, s, wt, 0);
  const double ptop = A.p[M.np - 1];

  const double wpbl = pbl_weight(ctl, P.p, pbl, ps);
  const double wtrop = tropo_weight(ctl, C, P.time, P.lat, P.p) * (1.0 - wpbl);
  const double wstrat = 1.0 - wpbl - wtrop;
  const double Kx = wpbl * ctl.turb_dx_pbl + wtrop * ctl.turb_dx_trop + wstrat * ctl.turb_dx_strat;
  const double Kz = wpbl * ctl.turb_dz_pbl + wtrop * ctl.turb_dz_trop + wstrat * ctl.turb_dz_strat;
  const double dt_abs = fabs(P.dt);

  double rs0, rs1, rs2;
  if (pre) {
    rs0 = pre[0];
    rs1 = pre[1];
    rs2 = pre[2];
  } else
    normal_triple(ltab, ctr, g, rs0, rs1, rs2);

  if (Kx > 0) {
    const double sigma_h = fsqrt(2.0 * Kx * dt_abs);
    P.lon += dx2coord(ct, rs0 * sigma_h, P.lat);
    P.lat += dy2coord(ct, rs1 * sigma_h);
  }
  if (Kz > 0) {
    const double sigma_z = fsqrt(2.0 * Kz * dt_abs) * 1e-3;
    const double p_save = P.p;
    const double eps_km = 0.01;
    const double p_up = p_save + dz2dp(eps_km, p_save);
    const double p_dn = p_save + dz2dp(-eps_km, p_save);
    const double pt = tropo_pressure(ctl, C, P.time, P.lat);   // latitude already displaced above
    const double Kz_up = kz_blend(ctl, pt, dmax(ptop, dmin(ps, p_up)), pbl, ps);
    const double Kz_dn = kz_blend(ctl, pt, dmax(ptop, dmin(ps, p_dn)), pbl, ps);
    const double dKz_dz = div_const(Kz_up - Kz_dn, 2.0 * eps_km * 1e3, 1.0 / (2.0 * eps_km * 1e3));
    const double dlnrho_dz = -1.0 / (1e3 * kH0);
    const double w_drift = dKz_dz + Kz * dlnrho_dz;
    const double dz_drift = w_drift * dt_abs * 1e-3;
    const double dz_tot = rs2 * sigma_z + dz_drift;
    double ptrial = p_save + dz2dp(dz_tot, p_save);
    for (int iter = 0; iter < 10; iter++) {
      if (ptrial > ps)
        ptrial = ps * ps / ptrial;
      else if (ptrial < ptop)
        ptrial = ptop * ptop / ptrial;
      else
        break;
    }
    P.p = dmax(ptop, dmin(ps, ptrial));
  }
}

// temporal correlation of module_diff_meso (mptrac.c:4310-4311): from the host for the regular time step
__device__ __forceinline__ void meso_coeffs(const mphip_ctl_t &ctl, const DevMet &M, double dt, double &r, double &r2) {
  if (fabs(dt) == M.meso_dt) {
    r = M.meso_r;
    r2 = M.meso_r2;
  } else {
    r = 1 - fdiv(2 * fabs(dt), ctl.dt_met);
    r2 = fsqrt(1 - r * r);
  }
}

// module_diff_meso, mptrac.c:4280-4338
__device__ __forceinline__ void diff_meso(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                          float &up, float &vp, float &wp, uint64_t ctr, uint64_t g,
                                          const double *pre, WindCache &wc, const double *ltab) {
  // HIP's __fadd_rn / __fmul_rn are plain operators, so contraction has to be
  // switched off here for the single-precision statistics to round like the
  // reference's separate multiply and add (the variance is a small difference
  // of large sums: an FMA changes sigma at the 1e-5 level).
#pragma clang fp contract(off)
  const int ct = M.coord_type;
  // raw (un-wrapped) coordinates, mptrac.c:4283-4285
  Stencil s;
  s.ix = locate_reg(A.lon, M.nx, P.lon);   // exact: sigma is not continuous across cells
  s.iy = locate_lat(M, A, P.lat);
  s.ip = locate_p(M, A, P.p);
  if (wc.enabled) {
    load_wind_cached(M, s, wc);
    wind_cache_wait(wc);
  } else
    load_wind(M, s, wc.c);
  const WindCorners &c = wc.c;

  // single-precision sums in the reference's order: i (lon), j (lat),
  // k (level), met0 before met1
  float mean[3] = { 0, 0, 0 }, sig[3] = { 0, 0, 0 };
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int q = 0; q < 3; q++) {
            const float a = wind_elem(c, i, j, k, t, q);
            mean[q] = mean[q] + a;
            sig[q] = sig[q] + a * a;
          }
  float sd[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float m16 = mean[q] / 16.f;
    const float var = sig[q] / 16.f - m16 * m16;
    sd[q] = (var > 0 ? sqrtf(var) : 0.f);
  }

  double r, r2;
  meso_coeffs(ctl, M, P.dt, r, r2);
  double rs0, rs1, rs2;
  if (pre) {
    rs0 = pre[0];
    rs1 = pre[1];
    rs2 = pre[2];
  } else
    normal_triple(ltab, ctr, g, rs0, rs1, rs2);

  if (ctl.turb_mesox > 0) {
    up = (float) (r * up + r2 * rs0 * ctl.turb_mesox * sd[0]);
    P.lon += dx2coord(ct, up * P.dt, P.lat);
    vp = (float) (r * vp + r2 * rs1 * ctl.turb_mesox * sd[1]);
    P.lat += dy2coord(ct, vp * P.dt);
  }
  if (ctl.turb_mesoz > 0) {
    wp = (float) (r * wp + r2 * rs2 * ctl.turb_mesoz * sd[2]);
    P.p += wp * P.dt;
  }
}

// temperature at (p, lon, lat): INTPOL_3D(t, 1)
__device__ __forceinline__ double temperature_at(const DevMet &M, const Axes &A, double time, double p, double lon,
                                                 double lat) {
  Stencil s;
  stencil_3d(M, A, p, lon, lat, s);
  return temp_time_3d(M, s, time_weight(M, time));
}

// module_convection, mptrac.c:4116-4170
__device__ __forceinline__ void convection(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                           uint64_t ctr, uint64_t g, const double *pre = nullptr) {
  Stencil s = stencil_zero();
  stencil_2d(M, A, P.lon, P.lat, s);
  SurfA c;
  load_sfa(M, s, c);
  const double wt = time_weight(M, P.time);
  const double ps = sfa_time_2d(c, s, wt, 0);
  double pbot = ps, ptop = ps;
  if (ctl.conv_mix_pbl) {
    const double pbl = sfa_time_2d(c, s, wt, 1);
    ptop = pbl - ctl.conv_pbl_trans * (ps - pbl);
  }
  if (ctl.conv_cape >= 0) {
    if (ctl.conv_cin <= 0) {   // CIN is not consulted: one 16-byte pair record per corner instead of two loads
      SurfA b;
      load_pair_2d(M.cp2, M, s, b);
      const double cape = sfa_time_2d(b, s, wt, 0);
      const double pel = sfa_time_2d(b, s, wt, 1);
      if (isfinite(cape) && cape >= ctl.conv_cape)
        ptop = dmin(ptop, pel);
    } else {
      SurfB b;
      load_sfb(M.sfb, M, s, b);
      const double cape = sfb_time_2d(b, s, wt, 0);
      const double cin = sfb_time_2d(b, s, wt, 1);
      const double pel = sfb_time_2d(b, s, wt, 2);
      if (isfinite(cape) && cape >= ctl.conv_cape && isfinite(cin) && cin >= ctl.conv_cin)
        ptop = dmin(ptop, pel);
    }
  }
  if (ptop != pbot && P.p >= ptop) {
    const double tbot = temperature_at(M, A, P.time, pbot, P.lon, P.lat);
    const double ttop = temperature_at(M, A, P.time, ptop, P.lon, P.lat);
    const double rhobot = fdiv(pbot, tbot);
    const double rhotop = fdiv(ptop, ttop);
    const double rs = pre ? *pre : uniform01(ctr + g);
    const double rho = rhobot + (rhotop - rhobot) * rs;
    P.p = lin(rhobot, pbot, rhotop, ptop, rho);
  }
}

// module_sedi, mptrac.c:5869-5882
__device__ __forceinline__ void sedimentation(const DevMet &M, const Axes &A, Particle &P, double rp, double rhop) {
  const double t = temperature_at(M, A, P.time, P.p, P.lon, P.lat);
  const double v_s = sedi(P.p, t, rp, rhop, libm_tables());
  P.p += dz2dp(div_const(v_s * P.dt, 1000., 1e-3), P.p);
}

// ---- module_meteo (mptrac.c:5062-5165) --------------------------------------
// Every field comes from a packed two-snapshot record (wind, temp, h2o, cloud, mx; sfa ... sfd, mx2).

constexpr double kLv = 2501000.;   // LV, mptrac.h:275

__device__ __forceinline__ double pw_of(double p, double h2o) {   // PW, mptrac.h:1859
  return p * dmax(h2o, 0.1e-6) / (1. + (1. - kEps) * dmax(h2o, 0.1e-6));
}

__device__ __forceinline__ double psat_of(double t) {   // PSAT, mptrac.h:1808
  return 6.112 * libm_exp(17.62 * (t - kT0) / (243.12 + t - kT0));
}

__device__ __forceinline__ double psice_of(double t) {   // PSICE, mptrac.h:1832
  return 6.112 * libm_exp(22.46 * (t - kT0) / (272.62 + t - kT0));
}

__device__ __forceinline__ double sh_of(double h2o) {   // SH, mptrac.h:2024
  return kEps * dmax(h2o, 0.1e-6);
}

__device__ __forceinline__ double tdew_of(double p, double h2o) {   // TDEW, mptrac.h:2075
  const double l = libm_log(pw_of(p, h2o) / 6.112);
  return kT0 + 243.12 * l / (17.62 - l);
}

__device__ __forceinline__ double tice_of(double p, double h2o) {   // TICE, mptrac.h:2100
  const double l = libm_log(pw_of(p, h2o) / 6.112);
  return kT0 + 272.62 * l / (22.46 - l);
}

__device__ __forceinline__ double theta_of(double p, double t) {   // THETA, mptrac.h:2124
  return t * libm_pow(1000. / p, kKappa);
}

__device__ __forceinline__ double zeta_of(double ps, double p, double t) {   // ZETA, mptrac.h:2293
  return (p / ps <= 0.3 ? 1. : libm_sin(kPi / 2. * (1. - p / ps) / (1. - 0.3))) * theta_of(p, t);
}

__device__ __forceinline__ double lapse_rate(double t, double h2o) {   // lapse_rate, mptrac.c:3324-3338
  const double a = kRA * (t * t), r = sh_of(h2o) / (1. - sh_of(h2o));   // (RA * SQR(t): the square first)
  return 1e3 * kG0 * (a + kLv * r * t) / (kCpd * a + kLv * kLv * r * kEps);
}

// ---- module_isosurf (mptrac.c:4886-5005) and module_bound_cond (mptrac.c:3789-3881) ----

// module_isosurf_init, modes 1-3: the conserved quantity of the particle
__device__ __forceinline__ double isosurf_value(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A,
                                                const Particle &P) {
  if (ctl.isosurf == 1)
    return P.p;
  const double t = temperature_at(M, A, P.time, P.p, P.lon, P.lat);
  return ctl.isosurf == 2 ? P.p / t : theta_of(P.p, t);
}

// module_isosurf: pressure that puts the particle back on its surface
__device__ __forceinline__ double isosurf_pressure(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A,
                                                   const DevAtm &a, const Particle &P, double iso_var) {
  if (ctl.isosurf == 1)
    return iso_var;
  if (ctl.isosurf == 2 || ctl.isosurf == 3) {
    const double t = temperature_at(M, A, P.time, P.p, P.lon, P.lat);
    return ctl.isosurf == 2 ? iso_var * t : 1000. * libm_pow(iso_var / t, -1. / kKappa);
  }
  if (ctl.isosurf == 4) {
    const int n = a.iso_n;
    if (P.time <= a.iso_ts[0])
      return a.iso_ps[0];
    if (P.time >= a.iso_ts[n - 1])
      return a.iso_ps[n - 1];
    const int idx = locate_irr(a.iso_ts, n, P.time, a.iso_ts[(n - 1) >> 1] < a.iso_ts[((n - 1) >> 1) + 1]);
    return lin(a.iso_ts[idx], a.iso_ps[idx], a.iso_ts[idx + 1], a.iso_ps[idx + 1], P.time);
  }
  return P.p;
}

// module_bound_cond: true if the particle lies in the boundary region (mptrac.c:3809-3846)
__device__ __forceinline__ bool in_boundary_region(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A,
                                                   const Particle &P) {
  if (P.lat < ctl.bound_lat0 || P.lat > ctl.bound_lat1 || P.p > ctl.bound_p0 || P.p < ctl.bound_p1)
    return false;
  if (ctl.bound_dps > 0 || ctl.bound_dzs > 0 || ctl.bound_zetas > 0 || ctl.bound_pbl) {
    Stencil s = stencil_zero();
    stencil_2d(M, A, P.lon, P.lat, s);
    SurfA c;
    load_sfa(M, s, c);
    const double wt = time_weight(M, P.time);
    const double ps = sfa_time_2d(c, s, wt, 0);
    if (ctl.bound_dps > 0 && P.p < ps - ctl.bound_dps)
      return false;
    if (ctl.bound_dzs > 0 && zfromp(P.p) > zfromp(ps) + ctl.bound_dzs)
      return false;
    if (ctl.bound_zetas > 0) {
      const double t = temperature_at(M, A, P.time, P.p, P.lon, P.lat);
      if (zeta_of(ps, P.p, t) > ctl.bound_zetas)
        return false;
    }
    if (ctl.bound_pbl && P.p < sfa_time_2d(c, s, wt, 1))
      return false;
  }
  return true;
}

// =============================================================================
// Lean versions for the specialised kernels (RK4 on pressure levels, lat/lon grid with a pressure look-up
// table -- launch_step checks that; everything else runs the general code above).  Same values as the
// general functions: every stencil is set up in straight-line code from a first guess and checked as a
// whole; a lane whose check fails (a coordinate exactly on a grid line, |lon| >= 360, two axis nodes in
// one table bin, NaN) recomputes it with the general function, so there is one rarely taken branch per
// stencil instead of one per search step and per special case.  On gfx950 every VALU instruction -- a
// move, a compare, an fp64 fma -- occupies the SIMD for the same four cycles, so what these versions save
// is instruction count: one-instruction min / max / med3 (the C ternaries compile to compare + two
// selects), axis end values from the kernel arguments instead of LDS, no per-axis direction branches.
// =============================================================================

// Metre -> degree factors of one latitude: DX2DEG(dx, lat) = dx * kx, DY2DEG(dy) = dy * ky with dx, dy in
// metres (mptrac.h:904-906, 922; the / 1000 of DX2COORD folded in).  One cosine and one reciprocal serve
// every conversion at that latitude (the four of a Runge-Kutta step use the same one, mptrac.c:3628, 3672).
// (MPHIP_EXACT_DIV: the record keeps the divisor pi RE cos(lat) and every conversion divides as the reference does.)
struct DegPerMetre {
  double kx;   // degrees per metre along x -- with MPHIP_EXACT_DIV: pi RE cos(lat); 0 next to the poles
};

__device__ __forceinline__ DegPerMetre deg_per_metre(double lat) {
#pragma clang fp contract(off)
  DegPerMetre d;
#if MPHIP_EXACT_DIV
  d.kx = (lat < -89.999 || lat > 89.999) ? 0.0 : kPi * kRE * libm_cos(deg2rad(lat));
#else
  const double c = kPi * kRE * cos_latitude_k(deg2rad(lat));
  d.kx = (lat < -89.999 || lat > 89.999) ? 0.0 : kMetresToDeg * frcp(c);
#endif
  return d;
}

// (contraction off: the product is rounded before the caller adds it, as in dx2coord / dy2coord of the general code)
__device__ __forceinline__ double dx2deg_k(const DegPerMetre &d, double dx_metres) {
#pragma clang fp contract(off)
#if MPHIP_EXACT_DIV
  return d.kx == 0.0 ? 0.0 : dx_metres / 1000.0 * 180. / d.kx;
#else
  return dx_metres * d.kx;
#endif
}

__device__ __forceinline__ double dy2deg_k(double dy_metres) {
#pragma clang fp contract(off)
#if MPHIP_EXACT_DIV
  return dy_metres / 1000.0 * 180. / (kPi * kRE);
#else
  return dy_metres * kDegPerMetreY;
#endif
}

// a stencil weight: numerator x the reciprocal interval width the host rounded once -- with MPHIP_EXACT_DIV the
// quotient by the width itself (the *_fast functions below hand back the one or the other as `inv`)
__device__ __forceinline__ double weight_of(double num, double inv) {
#if MPHIP_EXACT_DIV
  return num / inv;
#else
  return num * inv;
#endif
}

// ---- stencil set-up ---------------------------------------------------------

// longitude part of intpol_check_lon_lat + locate_reg + weight (mptrac.c:2762-2770, 3004-3018); false if the
// longitude needs FMOD (|lon| >= 360)
__device__ __forceinline__ bool lon_fast(const DevMet &M, const Axes &A, double lon, int &ix, double &wx) {
  double lon2 = lon + (lon < M.lon_first ? 360.0 : (lon > M.lon_last ? -360.0 : 0.0));
#if MPHIP_EXACT_DIV
  ix = locate_reg(A.lon, M.nx, lon2);
  const double lx1 = A.lon[ix + 1];
  wx = (lx1 - lon2) / (lx1 - A.lon[ix]);
#else
  ix = clamp0_s((int) ((lon2 - M.lon_first) * M.inv_dlon0), M.nx - 2);
  const double lx1 = A.lon[ix + 1], linv = A.inv_lon[ix];
  wx = (lx1 - lon2) * linv;
#endif
  return fabs(lon) < 360.0;
}

// index of the latitude interval (locate_irr semantics on either axis direction) for a latitude inside
// the axis range, with the interval's far node and reciprocal width; false if the guess was not it
__device__ __forceinline__ bool lat_fast(const DevMet &M, const Axes &A, double lat2, int &iy, double &y1,
                                         double &yinv) {
  const double lat_s = vmin_s(lat2, M.lat_search_max);   // a latitude on the last node belongs to the last interval
  iy = clamp0_s((int) ((lat2 - M.lat_x0) * M.lat_inv_dx), M.ny - 2);
  const double y0 = A.lat[iy];
  y1 = A.lat[iy + 1];
#if MPHIP_EXACT_DIV
  yinv = y1 - y0;
#else
  yinv = A.inv_lat[iy];
#endif
  return (vmin(y0, y1) <= lat_s) & (lat_s < vmax(y0, y1));
}

// the same for a pressure (any value: outside the axis the end intervals, as the bisection returns them)
__device__ __forceinline__ bool p_fast(const DevMet &M, const Axes &A, double p, int &ip, double &p1, double &pinv) {
  const double ps = vmin_s(vmax_s(p, M.p_min), M.p_search_max);
  const int g = (int) A.p_lut[(__double2hiint(ps) >> 13) - M.lut_base];   // index of the table bin's lower edge
  // the bin may hold an axis node between its edge and p: then the neighbouring interval is the one
  const double node = A.p[g + M.p_cmp_off];
  ip = g + (ps >= node ? M.p_step : 0);
  const double p0 = A.p[ip];
  p1 = A.p[ip + 1];
#if MPHIP_EXACT_DIV
  pinv = p1 - p0;
#else
  pinv = A.inv_p[ip];
#endif
  return (vmin(p0, p1) <= ps) & (ps < vmax(p0, p1)) & (p == p);   // (a NaN takes the general path: index n - 2)
}

// horizontal part of a stencil: indices and weights of intpol_met_space_2d / _3d (mptrac.c:2997-3021,
// 3059-3081); on return s.ix, s.iy, s.wx, s.wy are set (s.ip, s.wp untouched)
__device__ __forceinline__ void horiz_fast(const DevMet &M, const Axes &A, double lon, double lat, Stencil &s) {
  const double lat2 = vmin_s(vmax_s(lat, M.latmin), M.latmax);
  double y1, yinv;
  bool ok = lon_fast(M, A, lon, s.ix, s.wx);
  ok &= lat_fast(M, A, lat2, s.iy, y1, yinv);
  s.wy = weight_of(y1 - lat2, yinv);
  if (!ok) {
    Stencil g = stencil_zero();
    stencil_2d(M, A, lon, lat, g);
    s.ix = g.ix;
    s.iy = g.iy;
    s.wx = g.wx;
    s.wy = g.wy;
  }
}

// vertical part: s.ip, s.wp
__device__ __forceinline__ void vert_fast(const DevMet &M, const Axes &A, double p, Stencil &s) {
  double p1, pinv;
  const bool ok = p_fast(M, A, p, s.ip, p1, pinv);
  s.wp = weight_of(p1 - p, pinv);
  if (!ok) {
    const AxisHit hp = hit_p(M, A, p);
    s.ip = hp.i;
    s.wp = div_const(hp.x1 - p, hp.x1 - hp.x0, hp.inv);
  }
}

__device__ __forceinline__ void stencil_3d_fast(const DevMet &M, const Axes &A, double p, double lon, double lat,
                                                Stencil &s) {
  const double lat2 = vmin_s(vmax_s(lat, M.latmin), M.latmax);
  double y1, yinv, p1, pinv;
  bool ok = lon_fast(M, A, lon, s.ix, s.wx);
  ok &= lat_fast(M, A, lat2, s.iy, y1, yinv);
  ok &= p_fast(M, A, p, s.ip, p1, pinv);
  s.wy = weight_of(y1 - lat2, yinv);
  s.wp = weight_of(p1 - p, pinv);
  if (!ok)
    stencil_3d(M, A, p, lon, lat, s);
}

// raw indices of module_diff_meso / module_sort (mptrac.c:4283-4285, 5913-5917): locate_reg on the
// un-wrapped longitude with the reference's division -- the product with the reciprocal spacing decides
// unless it lands within 1e-9 of a whole number --, locate_irr on latitude and pressure
__device__ __forceinline__ void raw_cell_fast(const DevMet &M, const Axes &A, double lon, double lat, double p,
                                              Stencil &s) {
  const double q = (lon - M.lon_first) * M.inv_dlon0;
  const int i = (int) q;
  const double f = q - (double) i;
  bool ok = (q < -1e-9) | ((f > 1e-9) & (f < 1.0 - 1e-9));
  s.ix = clamp0_s(i, M.nx - 2);
  double y1, yinv, p1, pinv;
  ok &= lat_fast(M, A, vmin_s(vmax_s(lat, M.latmin), M.latmax), s.iy, y1, yinv);
  ok &= p_fast(M, A, p, s.ip, p1, pinv);
  if (!ok) {
    s.ix = locate_reg(A.lon, M.nx, lon);
    s.iy = locate_lat(M, A, lat);
    s.ip = locate_p(M, A, p);
  }
}

// ---- interpolation ----------------------------------------------------------

// Gathers of the lean kernels address their records as (wave-uniform base in an SGPR pair) + (32-bit byte
// offset per lane): one VGPR and one 32-bit multiply per address instead of a 64-bit multiply-add into a
// register pair (launch_step checks that every packed grid is smaller than 4 GB)
__device__ __forceinline__ unsigned cell32(const DevMet &M, const Stencil &s, int di, int dj) {
  const unsigned base = __umul24(__umul24((unsigned) s.ix, (unsigned) M.ny) + (unsigned) s.iy, (unsigned) M.np)
    + (unsigned) s.ip;
  return base + (unsigned) di * ((unsigned) M.ny * (unsigned) M.np) + (unsigned) dj * (unsigned) M.np;
}

template <class T>
__device__ __forceinline__ T load_at(const void *base, unsigned byte_offset) {
  return *(const T *) ((const char *) base + byte_offset);
}

// ... and for grids whose packed arrays are larger than 4 GB (the kBigGrid instantiations, BIG = true below): the cell
// index still fits 32 bits (launch_step checks), the byte offset does not
template <class T>
__device__ __forceinline__ T load_at64(const void *base, uint64_t byte_offset) {
  return *(const T *) ((const char *) base + byte_offset);
}

typedef float f32x2s __attribute__((ext_vector_type(2)));
// A box of wind records staged in LDS (north_star's "met grids staged through LDS tiles per thread-block", SURVEY x1;
// traj_tile_kernel): columns [x0, x0 + nx) x [y0, y0 + ny), levels [z0, z0 + nz), level index fastest as in the grid.
struct WindTile {
  const float *rec;   // 6 floats per cell {u0,v0,u1,v1,w0,w1}
  int x0, y0, z0, nx, ny, nz;
};

template <bool BIG = false>
__device__ __forceinline__ void load_wind_cached32(const DevMet &M, const Stencil &s, WindCache &w,
                                                   const WindTile *tile = nullptr) {
  if constexpr (BIG) {   // 64-bit addresses per lane: the gather of the general kernels
    load_wind_cached(M, s, w);
    return;
  }
  if (s.ix != w.ix || s.iy != w.iy || s.ip != w.ip) {
    if (tile) {   // (a compile-time null in every kernel but the one that stages a tile)
      const int tx = s.ix - tile->x0, ty = s.iy - tile->y0, tz = s.ip - tile->z0;
      if (tx >= 0 && ty >= 0 && tz >= 0 && tx + 1 < tile->nx && ty + 1 < tile->ny && tz + 1 < tile->nz) {
#pragma unroll
        for (int di = 0; di < 2; di++)
#pragma unroll
          for (int dj = 0; dj < 2; dj++) {
            // levels tz and tz + 1 of the column: twelve consecutive floats, 8-byte aligned
            const f32x2s *q = (const f32x2s *) (tile->rec + 6 * (((tx + di) * tile->ny + ty + dj) * tile->nz + tz));
#pragma unroll
            for (int k = 0; k < 3; k++) {
              const f32x2s lo = q[2 * k], hi = q[2 * k + 1];
              w.c.r[di][dj][k] = f32x4u{ lo[0], lo[1], hi[0], hi[1] };
            }
          }
      } else {
        // outside the tile: ordinary loads the compiler waits for by itself (the asynchronous gathers below rely on
        // registers nobody else touches until wind_cache_wait -- not the case where a second path writes them)
#pragma unroll
        for (int di = 0; di < 2; di++)
#pragma unroll
          for (int dj = 0; dj < 2; dj++) {
            const unsigned off = 24u * cell32(M, s, di, dj);
#pragma unroll
            for (int k = 0; k < 3; k++)
              w.c.r[di][dj][k] = load_at<f32x4u>(M.wind, off + 16u * (unsigned) k);
          }
      }
      w.ix = s.ix;
      w.iy = s.iy;
      w.ip = s.ip;
      return;
    }
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
      for (int dj = 0; dj < 2; dj++) {
        const unsigned off = 24u * cell32(M, s, di, dj);
        asm volatile("global_load_dwordx4 %0, %3, %4\n\t"
                     "global_load_dwordx4 %1, %3, %4 offset:16\n\t"
                     "global_load_dwordx4 %2, %3, %4 offset:32"
                     : "+v"(w.c.r[di][dj][0]), "+v"(w.c.r[di][dj][1]), "+v"(w.c.r[di][dj][2])
                     : "v"(off), "s"(M.wind)
                     : "memory");
      }
    w.ix = s.ix;
    w.iy = s.iy;
    w.ip = s.ip;
  }
}

__device__ __forceinline__ void load_pair_2d32(const f32x4 *__restrict__ g, const DevMet &M, const Stencil &s, SurfA &c) {
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++)
      c.v[di][dj] = load_at<f32x4>(g, 16u * col_of(M, s, di, dj));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef MPHIP_WIND_SERIAL
#define MPHIP_WIND_SERIAL 1
#endif

// Wind records of the lean kernels: {u0,v0,u1,v1,w0,w1} per level (snapshots 0 / 1), so that a level pair
// is twelve floats whose (u,v) pairs of one snapshot sit in an aligned register pair -- the single-
// precision corner differences of intpol_met_space_3d and the sums of module_diff_meso then run as packed
// two-float instructions.  Element e of a corner: 6 * level + {0,1,4: u,v,w of met0; 2,3,5: of met1}.
__device__ __forceinline__ int wind_e(int lvl, int t, int k) {
  return 6 * lvl + (k < 2 ? 2 * t + k : 4 + t);
}

__device__ __forceinline__ float wind_at(const WindCorners &c, int di, int dj, int e) {
  return c.r[di][dj][e >> 2][e & 3];
}

// u, v, w at the stencil: intpol_met_time_3d of the three components (mptrac.c:3023-3043, 3112-3137) with
// the corner differences (level ip minus level ip + 1, in single precision as the reference's float - float)
// formed two at a time
__device__ __forceinline__ void wind_uvw_fast(const WindCorners &c, const Stencil &s, double wt, double &u, double &v,
                                              double &w) {
  double val[6];
  // one register pair of elements at a time (u0 v0 | u1 v1 | w0 w1): with MPHIP_WIND_SERIAL the scheduler may
  // not start the next pair before this one is reduced to its two values -- 16 instead of 48 live registers
#pragma unroll
  for (int pr = 0; pr < 3; pr++) {
    double col[2][2][2];   // [di][dj][element of the pair]: wp * (lo - hi) + hi
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
      for (int dj = 0; dj < 2; dj++) {
        const f32x4u r0 = c.r[di][dj][0], r1 = c.r[di][dj][1], r2 = c.r[di][dj][2];
        // level ip: r0[0..3], r1[0..1]; level ip + 1: r1[2..3], r2[0..3]
        const f32x2 lo = pr == 0 ? __builtin_shufflevector(r0, r0, 0, 1)
          : (pr == 1 ? __builtin_shufflevector(r0, r0, 2, 3) : __builtin_shufflevector(r1, r1, 0, 1));
        const f32x2 hi = pr == 0 ? __builtin_shufflevector(r1, r1, 2, 3)
          : (pr == 1 ? __builtin_shufflevector(r2, r2, 0, 1) : __builtin_shufflevector(r2, r2, 2, 3));
        const f32x2 d = lo - hi;
        col[di][dj][0] = s.wp * (double) d[0] + (double) hi[0];
        col[di][dj][1] = s.wp * (double) d[1] + (double) hi[1];
      }
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const double r0 = s.wy * (col[0][0][e] - col[0][1][e]) + col[0][1][e];
      const double r1 = s.wy * (col[1][0][e] - col[1][1][e]) + col[1][1][e];
      val[2 * pr + e] = s.wx * (r0 - r1) + r1;
    }
#if MPHIP_WIND_SERIAL
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  u = wt * (val[0] - val[2]) + val[2];
  v = wt * (val[1] - val[3]) + val[3];
  w = wt * (val[4] - val[5]) + val[5];
}

// field f (0 / 1) of a {a0,b0,a1,b1} surface record at the stencil: intpol_met_space_2d at both snapshots
// and intpol_met_time_2d (mptrac.c:3083-3107, 3155-3169).  A corner that is not finite makes the plain
// result not finite (inf - inf, 0 * inf, inf + x), and only then do the nearest-neighbour rules apply, so
// one test of the result replaces the ten tests of the inputs.
__device__ __forceinline__ double pair_time_2d_fast(const SurfA &c, const Stencil &s, double wt, int f) {
  double v[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const double c00 = c.v[0][0][2 * t + f], c01 = c.v[0][1][2 * t + f], c10 = c.v[1][0][2 * t + f],
                 c11 = c.v[1][1][2 * t + f];
    const double r0 = s.wy * (c00 - c01) + c01;
    const double r1 = s.wy * (c10 - c11) + c11;
    v[t] = s.wx * (r0 - r1) + r1;
  }
  double r = wt * (v[0] - v[1]) + v[1];
  if (!isfinite(r))
    r = sfa_time_2d(c, s, wt, f);
  return r;
}

// temperature at a stencil: as pair_time_3d, corner differences two at a time
template <bool BIG = false>
__device__ __forceinline__ double temp_fast(const DevMet &M, const Stencil &s, double wt) {
  double col[2][2][2];
#pragma unroll
  for (int di = 0; di < 2; di++)
#pragma unroll
    for (int dj = 0; dj < 2; dj++) {
      f32x4u q;   // {t0,t1} at ip, {t0,t1} at ip + 1
      if constexpr (BIG)
        q = load_at64<f32x4u>(M.temp, (uint64_t) 8 * cell32(M, s, di, dj));
      else
        q = load_at<f32x4u>(M.temp, 8u * cell32(M, s, di, dj));
      const f32x2 d = __builtin_shufflevector(q, q, 0, 1) - __builtin_shufflevector(q, q, 2, 3);
      col[di][dj][0] = s.wp * (double) d[0] + (double) q[2];
      col[di][dj][1] = s.wp * (double) d[1] + (double) q[3];
    }
  double val[2];
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const double r0 = s.wy * (col[0][0][t] - col[0][1][t]) + col[0][1][t];
    const double r1 = s.wy * (col[1][0][t] - col[1][1][t]) + col[1][1][t];
    val[t] = s.wx * (r0 - r1) + r1;
  }
  return wt * (val[0] - val[1]) + val[1];
}

// ---- modules ----------------------------------------------------------------

// module_position (mptrac.c:5445-5488) on a lat/lon grid; the surface pressure it reflects at is the one of
// grid node [1][1] (quirk Q1): with zero weights intpol_met_space_2d returns that corner whatever the others are
__device__ __forceinline__ void position_fast(const DevMet &M, const Axes &A, Particle &P) {
  double lon = fmod_trunc(P.lon, 360.);
  double lat = fmod_trunc(P.lat, 360.);
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
  P.lon = lon;
  P.lat = lat;
  const double ptop = A.p[M.np - 1];
  if (P.p < ptop) {
    P.p = ptop * ptop / P.p;
  } else if (P.p > 300.) {
    const double ps = blend_time_2d((double) M.ps11[0], (double) M.ps11[1], time_weight(M, P.time));
    if (P.p > ps)
      P.p = ps * ps / P.p;
  }
}

// module_advect on pressure levels (mptrac.c:3612-3677), STAGES = 4: classical Runge-Kutta (ADVECT 4);
// STAGES = 2: the midpoint scheme (ADVECT 2, the reference's default) -- and, with `euler` set (wave-uniform,
// from the control parameters), its first stage alone (ADVECT 1).  `hook(i)` runs behind the gathers of stage i.
template <int STAGES, bool BIG = false, class Hook>
__device__ __forceinline__ void advect_fast(const DevMet &M, const Axes &A, Particle &P, Hook &hook, WindCache &wc,
                                            bool euler = false, const WindTile *tile = nullptr) {
  const double dt = P.dt;
  const DegPerMetre dm = deg_per_metre(P.lat);   // every stage converts at the latitude the step starts from
  double u = 0, v = 0, w = 0, um = 0, vm = 0, wm = 0;
  double x1 = P.lat;
#pragma unroll
  for (int i = 0; i < STAGES; i++) {
    if (STAGES == 2 && i == 1 && euler)
      break;
    double dts = 0.0, x0 = P.lon, x2 = P.p;
    x1 = P.lat;
    if (i > 0) {
      dts = (i == 3 ? 1.0 : 0.5) * dt;
      x0 = P.lon + dx2deg_k(dm, dts * u);
      x1 = P.lat + dy2deg_k(dts * v);
      x2 = P.p + dts * w;
    }
    Stencil s;
#if MPHIP_SETPRIO
    __builtin_amdgcn_s_setprio(MPHIP_SETPRIO);   // a wave on its way to a gather round goes first
#endif
    stencil_3d_fast(M, A, x2, x0, x1, s);
    load_wind_cached32<BIG>(M, s, wc, tile);
#if MPHIP_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    hook(i);
    wind_cache_wait(wc);
    wind_uvw_fast(wc.c, s, time_weight(M, P.time + dts), u, v, w);
    if (STAGES == 4) {
      const double k = (i == 0 || i == 3) ? 1.0 / 6.0 : 2.0 / 6.0;
      um += k * u;
      vm += k * v;
      wm += k * w;
    } else {   // the step is taken with the wind of the last stage that ran
      um = u;
      vm = v;
      wm = w;
    }
  }
  P.time += dt;
  if (STAGES == 2 && !euler) {
    // the midpoint scheme converts the final displacement at the latitude of the midpoint (mptrac.c:3672)
    const DegPerMetre dmid = deg_per_metre(x1);
    P.lon += dx2deg_k(dmid, dt * um);
  } else
    P.lon += dx2deg_k(dm, dt * um);
  P.lat += dy2deg_k(dt * vm);
  P.p += dt * wm;
}

template <class Hook>
__device__ __forceinline__ void advect_rk4_fast(const DevMet &M, const Axes &A, Particle &P, Hook &hook, WindCache &wc) {
  advect_fast<4>(M, A, P, hook, wc);
}

// module_diff_turb (mptrac.c:4603-4733).  The surface pressure and the boundary-layer top only matter near the
// ground: for a particle whose pressure (with a 1 % margin for the +-10 m probes of the Kz gradient) is below
// DevMet::turb_skip the boundary-layer weight is 0 wherever it is, and the surface can only come into play if the
// vertical displacement reaches it.  Such a particle runs with ps = pbl = the bound -- every expression below then
// evaluates exactly as with the interpolated values -- and gathers nothing; if its trial pressure does reach the
// bound, the reflection is redone with the interpolated surface pressure.
__device__ __forceinline__ void diff_turb_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevClim &C,
                                               Particle &P, uint64_t ctr, uint64_t g, const double *pre,
                                               const double *ltab) {
  const double wt = time_weight(M, P.time);
  const bool far = (P.p * 1.01 < M.turb_skip) & (wt >= 0.0) & (wt <= 1.0);
  double pbl = M.turb_skip, ps = M.turb_skip;
  const double lon0 = P.lon, lat0 = P.lat;
  if (!far) {
    Stencil s = stencil_zero();
    horiz_fast(M, A, P.lon, P.lat, s);
    SurfA c;
    load_pair_2d32(M.sfa, M, s, c);
    pbl = pair_time_2d_fast(c, s, wt, 1);
    if (ctl.turb_pbl_scheme > 0 && P.p >= pbl)
      return;
    ps = pair_time_2d_fast(c, s, wt, 0);
  }
  const double ptop = A.p[M.np - 1];

  const TropoTime tt = tropo_time(C, P.time);
  const double wpbl = pbl_weight(ctl, P.p, pbl, ps);
  double pt = tropo_pressure_at(ctl, C, tt, P.lat);
  const double wtrop = tropo_weight_pt(pt, P.p) * (1.0 - wpbl);
  const double wstrat = 1.0 - wpbl - wtrop;
  const double Kx = wpbl * ctl.turb_dx_pbl + wtrop * ctl.turb_dx_trop + wstrat * ctl.turb_dx_strat;
  const double Kz = wpbl * ctl.turb_dz_pbl + wtrop * ctl.turb_dz_trop + wstrat * ctl.turb_dz_strat;
  const double dt_abs = fabs(P.dt);

  // Without horizontal diffusion (Kx = 0: the stratosphere of the default parameters) only rs[3 g + 2] is used --
  // one Box-Muller pair instead of two -- and the latitude keeps the tropopause pressure found above
  double rs0 = 0, rs1 = 0, rs2;
  if (pre) {
    rs0 = pre[0];
    rs1 = pre[1];
    rs2 = pre[2];
  } else if (Kx > 0 || !MPHIP_TURB_SHORTCUT)
    normal_triple(ltab, ctr, g, rs0, rs1, rs2);
  else
    rs2 = normal_third(ltab, ctr, g);

  if (Kx > 0) {
    const double sigma_h = fsqrt(2.0 * Kx * dt_abs);
    const DegPerMetre dm = deg_per_metre(P.lat);
    P.lon += dx2deg_k(dm, rs0 * sigma_h);
    P.lat += dy2deg_k(rs1 * sigma_h);
    pt = tropo_pressure_at(ctl, C, tt, P.lat);   // (mptrac.c:4669: at the displaced latitude)
  }
  if (Kz > 0) {
    const double sigma_z = fsqrt(2.0 * Kz * dt_abs) * 1e-3;
    const double p_save = P.p;
    const double eps_km = 0.01;
    const double p_up = p_save + dz2dp(eps_km, p_save);
    const double p_dn = p_save + dz2dp(-eps_km, p_save);
    const double Kz_up = kz_blend(ctl, pt, vmax(ptop, vmin(ps, p_up)), pbl, ps);
    const double Kz_dn = kz_blend(ctl, pt, vmax(ptop, vmin(ps, p_dn)), pbl, ps);
    const double dKz_dz = div_const(Kz_up - Kz_dn, 2.0 * eps_km * 1e3, 1.0 / (2.0 * eps_km * 1e3));
    const double dlnrho_dz = -1.0 / (1e3 * kH0);
    const double w_drift = dKz_dz + Kz * dlnrho_dz;
    const double dz_drift = w_drift * dt_abs * 1e-3;
    const double dz_tot = rs2 * sigma_z + dz_drift;
    const double ptrial0 = p_save + dz2dp(dz_tot, p_save);
    double ptrial = ptrial0;
    bool at_surface = false;
    for (int iter = 0; iter < 10; iter++) {
      if (ptrial > ps) {
        at_surface = true;
        ptrial = ps * ps / ptrial;
      } else if (ptrial < ptop)
        ptrial = ptop * ptop / ptrial;
      else
        break;
    }
    if (far & (at_surface | !(ptrial <= ps))) {   // (rare: a displacement of kilometres) -- now the surface pressure is needed
      Stencil s = stencil_zero();
      horiz_fast(M, A, lon0, lat0, s);
      SurfA c;
      load_pair_2d32(M.sfa, M, s, c);
      ps = pair_time_2d_fast(c, s, wt, 0);
      ptrial = ptrial0;
      for (int iter = 0; iter < 10; iter++) {
        if (ptrial > ps)
          ptrial = ps * ps / ptrial;
        else if (ptrial < ptop)
          ptrial = ptop * ptop / ptrial;
        else
          break;
      }
    }
    P.p = dmax(ptop, dmin(ps, ptrial));
  }
}

// module_diff_meso (mptrac.c:4280-4338) on the {u0,v0,u1,v1,w0,w1} records
template <bool STREAM = false, bool BIG = false>
__device__ __forceinline__ void diff_meso_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                               float &up, float &vp, float &wp, uint64_t ctr, uint64_t g,
                                               const double *pre, WindCache &wc, const double *ltab) {
#pragma clang fp contract(off)
  Stencil s;
  raw_cell_fast(M, A, P.lon, P.lat, P.p, s);
  if (!STREAM) {
    load_wind_cached32<BIG>(M, s, wc);
    wind_cache_wait(wc);
  }
  const WindCorners &c = wc.c;

  // single-precision sums in the reference's order -- i (lon), j (lat), k (level), met0 before met1 --,
  // u and v side by side in one packed accumulator
  f32x2 mean_uv = { 0.f, 0.f }, sig_uv = { 0.f, 0.f };
  float mean_w = 0.f, sig_w = 0.f;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      f32x4u r0, r1, r2;
      if (STREAM) {   // (no corner cache in this instantiation: a corner is summed as it arrives)
        if constexpr (BIG) {
          const uint64_t off = (uint64_t) 24 * cell32(M, s, i, j);
          r0 = load_at64<f32x4u>(M.wind, off);
          r1 = load_at64<f32x4u>(M.wind, off + 16);
          r2 = load_at64<f32x4u>(M.wind, off + 32);
        } else {
          const unsigned off = 24u * cell32(M, s, i, j);
          r0 = load_at<f32x4u>(M.wind, off);
          r1 = load_at<f32x4u>(M.wind, off + 16u);
          r2 = load_at<f32x4u>(M.wind, off + 32u);
        }
      } else {
        r0 = c.r[i][j][0];
        r1 = c.r[i][j][1];
        r2 = c.r[i][j][2];
      }
      // level ip: uv0 = r0[0,1], uv1 = r0[2,3], w0 w1 = r1[0,1]; level ip + 1: uv0 = r1[2,3], uv1 = r2[0,1], w0 w1 = r2[2,3]
      const f32x2 uv[2][2] = { { __builtin_shufflevector(r0, r0, 0, 1), __builtin_shufflevector(r0, r0, 2, 3) },
                               { __builtin_shufflevector(r1, r1, 2, 3), __builtin_shufflevector(r2, r2, 0, 1) } };
      const float ww[2][2] = { { r1[0], r1[1] }, { r2[2], r2[3] } };
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
          mean_uv = mean_uv + uv[k][t];
          sig_uv = sig_uv + uv[k][t] * uv[k][t];
          mean_w = mean_w + ww[k][t];
          sig_w = sig_w + ww[k][t] * ww[k][t];
        }
    }
  const float mean[3] = { mean_uv[0], mean_uv[1], mean_w }, sig[3] = { sig_uv[0], sig_uv[1], sig_w };
  float sd[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const float m16 = mean[q] / 16.f;
    const float var = sig[q] / 16.f - m16 * m16;
    sd[q] = (var > 0 ? sqrtf(var) : 0.f);
  }

  double r, r2;
  meso_coeffs(ctl, M, P.dt, r, r2);
  double rs0, rs1, rs2;
  if (pre) {
    rs0 = pre[0];
    rs1 = pre[1];
    rs2 = pre[2];
  } else
    normal_triple(ltab, ctr, g, rs0, rs1, rs2);

  if (ctl.turb_mesox > 0) {
    up = (float) (r * up + r2 * rs0 * ctl.turb_mesox * sd[0]);
    const DegPerMetre dm = deg_per_metre(P.lat);
    P.lon += dx2deg_k(dm, up * P.dt);
    vp = (float) (r * vp + r2 * rs1 * ctl.turb_mesox * sd[1]);
    P.lat += dy2deg_k(vp * P.dt);
  }
  if (ctl.turb_mesoz > 0) {
    wp = (float) (r * wp + r2 * rs2 * ctl.turb_mesoz * sd[2]);
    P.p += wp * P.dt;
  }
}

// module_convection (mptrac.c:4116-4170) and module_sedi (mptrac.c:5869-5882): both work at the horizontal
// position module_diff_meso left, so one horizontal stencil serves the surface fields and the three
// temperature columns
template <bool BIG = false>
__device__ __forceinline__ void conv_sedi_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, Particle &P,
                                               unsigned mask, uint64_t ctr, uint64_t g, const double *pre, double rp,
                                               double rhop, const double *ltab) {
  Stencil s = stencil_zero();
  horiz_fast(M, A, P.lon, P.lat, s);
  const double wt = time_weight(M, P.time);
  // (a particle with p < DevMet::conv_skip lies above every top the convective column can have: nothing to do)
  if ((mask & MPHIP_MOD_CONVECTION) && !((P.p < M.conv_skip) & (wt >= 0.0) & (wt <= 1.0))) {
    SurfA c;
    load_pair_2d32(M.sfa, M, s, c);
    const double ps = pair_time_2d_fast(c, s, wt, 0);
    double pbot = ps, ptop = ps;
    if (ctl.conv_mix_pbl) {
      const double pbl = pair_time_2d_fast(c, s, wt, 1);
      ptop = pbl - ctl.conv_pbl_trans * (ps - pbl);
    }
    if (ctl.conv_cape >= 0) {
      if (ctl.conv_cin <= 0) {
        SurfA b;
        load_pair_2d32(M.cp2, M, s, b);
        const double cape = pair_time_2d_fast(b, s, wt, 0);
        const double pel = pair_time_2d_fast(b, s, wt, 1);
        if (isfinite(cape) && cape >= ctl.conv_cape)
          ptop = dmin(ptop, pel);
      } else {
        SurfB b;
        load_sfb(M.sfb, M, s, b);
        const double cape = sfb_time_2d(b, s, wt, 0);
        const double cin = sfb_time_2d(b, s, wt, 1);
        const double pel = sfb_time_2d(b, s, wt, 2);
        if (isfinite(cape) && cape >= ctl.conv_cape && isfinite(cin) && cin >= ctl.conv_cin)
          ptop = dmin(ptop, pel);
      }
    }
    if (ptop != pbot && P.p >= ptop) {
      vert_fast(M, A, pbot, s);
      const double tbot = temp_fast<BIG>(M, s, wt);
      vert_fast(M, A, ptop, s);
      const double ttop = temp_fast<BIG>(M, s, wt);
      const double rhobot = fdiv(pbot, tbot);
      const double rhotop = fdiv(ptop, ttop);
      const double rs = pre ? *pre : uniform01(ctr + g);
      const double rho = rhobot + (rhotop - rhobot) * rs;
      P.p = lin(rhobot, pbot, rhotop, ptop, rho);
    }
  }
  if (mask & MPHIP_MOD_SEDI) {
    vert_fast(M, A, P.p, s);
    const double t = temp_fast<BIG>(M, s, wt);
    const double v_s = sedi(P.p, t, rp, rhop, ltab);
    P.p += dz2dp(div_const(v_s * P.dt, 1000., 1e-3), P.p);
  }
}

// ---- module_diff_pbl (mptrac.c:4343-4584; TURB_PBL_SCHEME 1) ------------------------------------------------------
// Langevin turbulence inside the boundary layer with Hanna's (1982) profiles in the form FLEXPART uses them.  Per
// particle below the boundary-layer top the module needs
//   (1) the layer in metres above ground: its depth, the particle's height and relative height (PblLayer);
//   (2) the surface-layer scales from the stresses and the heat flux: friction velocity, Obukhov length and, for a
//       convective layer, the convective velocity (PblScales);
//   (3) standard deviations and Lagrangian time scales of the three velocity components for one of three stability
//       classes (PblTurbulence);
//   (4) one Ornstein-Uhlenbeck step of the three perturbations with the drift correction of the vertical one, the
//       displacement, and mirror reflections at the ground and at the layer top.
// Laid out for a wave whose lanes sit at different heights and over different surfaces: the classes are three
// functions that fill one record; the piecewise vertical profile of the convective class is a choice of (base,
// exponents, factors) followed by ONE evaluation of the pair of powers instead of a ladder of branches with a pow
// pair in each.  Every exp, log and pow is the C library's (libm_*, mphip_libm.h) on the reference's operands, and
// products keep the reference's association: the values that are rounded to the single-precision perturbations are
// then the reference's doubles whenever the particle's position and the interpolated fields are.
// A particle above every boundary-layer top of the two snapshots (DevMet::turb_skip, exact: a blend with
// weights in [0, 1] is not below its smallest corner) leaves without a gather.
__device__ __forceinline__ double clampd(double v, double lo, double hi) {   // CLAMP, mptrac.h:756
  return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ double tvirt(double t, double h2o) {   // TVIRT, mptrac.h:2199
#pragma clang fp contract(off)
  return t * (1. + (1. - kEps) * dmax(h2o, 0.1e-6));
}

struct PblLayer {
  double depth;   // boundary-layer depth [m]
  double h;       // height above ground inside [0, depth] [m]
  double h1;      // ... but at least one metre
  double eta;     // h / depth inside [1e-6, 1 - 1e-6]
};

struct PblScales {
  double ustar;     // friction velocity, at least 1e-4 m/s
  double obukhov;   // Obukhov length [m]; 1e12 without a heat flux
  double wstar;     // convective velocity scale [m/s] (convective class only)
};

struct PblTurbulence {   // component 0 / 1 / 2: along x, along y, vertical
  double sigma[3];       // standard deviations [m/s]
  double tl[3];          // Lagrangian time scales [s]
  double dsigma_w;       // d sigma_w / dz [1/s]
};

// (The perturbations are stored in single precision: a difference in the last bits of the double that is rounded to
// them can flip a float, and the 1e-7 of that flip is what the positions then inherit.  The expressions below therefore
// keep the reference's operands and IEEE divisions / square roots wherever a value feeds the perturbations; the
// restructuring is in what is evaluated, not in how a value is rounded.)
// mptrac.c:4439-4449
__device__ __forceinline__ void hanna_neutral(const PblLayer &L, const PblScales &K, PblTurbulence &T, const double *lt) {
#pragma clang fp contract(off)
  const double x = L.h1 / K.ustar;
  const double sw = 1.3 * K.ustar * libm_exp(lt, -2e-4 * x);
  T.sigma[0] = dmax(2.0 * K.ustar * libm_exp(lt, -3e-4 * x), 1e-5);
  T.sigma[1] = T.sigma[2] = dmax(sw, 1e-5);
  T.dsigma_w = -2e-4 * sw / K.ustar;
  T.tl[0] = T.tl[1] = T.tl[2] = 0.5 * L.h1 / T.sigma[2] / (1.0 + 1.5e-3 * x);
}

// mptrac.c:4451-4510.  sigma_w / w* over the relative height eta is piecewise c B^e:
//   eta < 0.03, and up to 0.4 while it is the smaller one:  0.96  (3 eta - L / D)^(1/3)    "free convection"
//   otherwise up to 0.4:                                    0.763 eta^0.175
//   up to 0.96:                                             0.722 (1 - eta)^0.207
//   above:                                                  0.37
// and D / w*^2 d(sigma_w^2)/dz has the same shape with (1.8432, -1/3), (0.203759, -0.65), (-0.215812, -0.586), 0.
// Which factor meets w* first differs between the pieces in the reference (c w* B^e below 0.03 and above 0.4,
// w* (c B^e) between, where the two candidates are compared as c B^e): kept, a product is rounded once per step.
__device__ __forceinline__ void hanna_convective(const PblLayer &L, const PblScales &K, PblTurbulence &T, const double *lt) {
#pragma clang fp contract(off)
  const double third = 1.0 / 3.0;
  T.sigma[0] = T.sigma[1] = dmax(K.ustar * libm_pow(lt, dmax(12.0 - 0.5 * L.depth / K.obukhov, 0.0), third), 1e-6);
  const bool low = L.eta < 0.4, mid = L.eta < 0.96, candidates = low & !(L.eta < 0.03);
  const double base_free = dmax(3.0 * L.eta - K.obukhov / L.depth, 1e-12);
  // the power of the piece's own base ... (above 0.96 none is needed: base 1)
  const double pw_piece = libm_pow(lt, low ? base_free : (mid ? 1.0 - L.eta : 1.0), low ? third : 0.207);
  // ... and, where two candidates compete, the other one's
  const double s1 = 0.96 * pw_piece;
  const double s2 = candidates ? 0.763 * libm_pow(lt, L.eta, 0.175) : 0.0;
  const bool free = low & (!candidates | (s1 < s2));
  const double sig_w = low ? (candidates ? K.wstar * (free ? s1 : s2) : 0.96 * K.wstar * pw_piece)
                           : (mid ? 0.722 * K.wstar * pw_piece : 0.37 * K.wstar);
  const double base_g = low ? (free ? base_free : L.eta) : 1.0 - L.eta;
  const double e_g = low ? (free ? -third : -0.65) : -0.586;
  const double c_g = low ? (free ? 1.8432 : 0.203759) : -0.215812;
  const double grad = mid ? c_g * (K.wstar * K.wstar) / L.depth * libm_pow(lt, base_g, e_g) : 0.0;   // d(sigma_w^2)/dz
  T.sigma[2] = dmax(sig_w, 1e-6);
  T.dsigma_w = 0.5 * grad / T.sigma[2];
  T.tl[0] = T.tl[1] = 0.15 * L.depth / dmax(T.sigma[0], 1e-12);
  const double near_ground = 0.1 * L.h1 / (T.sigma[2] * dmax(0.55 - 0.38 * fabs(L.h1 / K.obukhov), 0.05));
  const double aloft = L.eta < 0.1 ? 0.59 * L.h1 / T.sigma[2] : 0.15 * L.depth / T.sigma[2] * (1.0 - libm_exp(lt, -5.0 * L.eta));
  T.tl[2] = L.h1 < fabs(K.obukhov) ? near_ground : aloft;
}

// mptrac.c:4512-4522
__device__ __forceinline__ void hanna_stable(const PblLayer &L, const PblScales &K, PblTurbulence &T, const double *lt) {
#pragma clang fp contract(off)
  const double fade = 1.0 - L.eta;
  T.sigma[0] = dmax(2.0 * K.ustar * fade, 1e-6);
  T.sigma[1] = T.sigma[2] = dmax(1.3 * K.ustar * fade, 1e-6);
  T.dsigma_w = -1.3 * K.ustar / L.depth;
  T.tl[0] = 0.15 * L.depth / T.sigma[0] * sqrt(L.eta);
  T.tl[1] = 0.467 * T.tl[0];
  T.tl[2] = 0.1 * L.depth / T.sigma[2] * libm_pow(lt, L.eta, 0.8);
}

// LEAN: the stencils and gathers of the lean kernels (lat/lon grid with the pressure table, 32-bit offsets)
template <bool LEAN = false>
__device__ __forceinline__ void diff_pbl(const DevMet &M, const Axes &A, Particle &P, float &up, float &vp,
                                         float &wp, uint64_t ctr, uint64_t g, const double *ltab) {
#pragma clang fp contract(off)
  const double wt = time_weight(M, P.time);
  if ((P.p < M.turb_skip) & (wt >= 0.0) & (wt <= 1.0))
    return;
  // (1) the layer (mptrac.c:4372-4398)
  Stencil col = stencil_zero();
  SurfA sa;
  double ps, pbl;
  if constexpr (LEAN) {
    horiz_fast(M, A, P.lon, P.lat, col);
    load_pair_2d32(M.sfa, M, col, sa);
    pbl = pair_time_2d_fast(sa, col, wt, 1);
  } else {
    stencil_2d(M, A, P.lon, P.lat, col);
    load_sfa(M, col, sa);
    pbl = sfa_time_2d(sa, col, wt, 1);
  }
  if (P.p < pbl)
    return;
  ps = LEAN ? pair_time_2d_fast(sa, col, wt, 0) : sfa_time_2d(sa, col, wt, 0);
  if (!(ps > 0.0 && pbl > 0.0 && ps > pbl))
    return;
  const double p_in = dmin(P.p, ps);
  const double ground_km = kH0 * libm_log(ltab, kP0 / ps);                 // Z(), mptrac.h:2243
  PblLayer L;
  L.depth = 1e3 * (kH0 * libm_log(ltab, kP0 / pbl) - ground_km);
  if (!(L.depth > 1.0))
    return;
  L.h = clampd(1e3 * (kH0 * libm_log(ltab, kP0 / p_in) - ground_km), 0.0, L.depth);
  L.eta = clampd(L.h / L.depth, 1e-6, 1.0 - 1e-6);
  L.h1 = dmax(L.h, 1.0);

  // (2) the scales (mptrac.c:4400-4436): air density and virtual potential temperature at the particle, stresses and
  // heat flux of the surface
  SurfB sd;
  load_sfb(M.sfd, M, col, sd);
  const double stress_x = sfb_time_2d(sd, col, wt, 0), stress_y = sfb_time_2d(sd, col, wt, 1);
  const double heat_flux = sfb_time_2d(sd, col, wt, 2);
  Stencil cell = col;
  double t, h2o;
  if constexpr (LEAN) {
    vert_fast(M, A, p_in, cell);
    t = temp_fast(M, cell, wt);
  } else {
    stencil_3d(M, A, p_in, P.lon, P.lat, cell);
    t = temp_time_3d(M, cell, wt);
  }
  h2o = pair_time_3d(M.h2o, M, cell, wt);
  const double rho = 100. * p_in / (kRA * tvirt(t, h2o));   // RHO, mptrac.h:1961, as the IEEE quotient (rho_air's is not)
  if (!(rho > 0.0))
    return;
  const double theta_v = tvirt(t * libm_pow(ltab, 1000. / p_in, kKappa), dmax(h2o, 0.1e-6));   // THETAVIRT, mptrac.h:2153
  PblScales K;
  K.ustar = dmax(1e-4, sqrt(dmax(sqrt(stress_x * stress_x + stress_y * stress_y) / rho, 0.0)));
  K.obukhov = fabs(heat_flux) > 1e-6 ? theta_v * rho * kCpd * (K.ustar * K.ustar) * K.ustar / (kKarman * kG0 * heat_flux) : 1e12;

  // (3) the class (mptrac.c:4438-4523): neutral while the layer is shallower than |L|, else by the sign of L
  PblTurbulence T;
  if (L.depth / fabs(K.obukhov) < 1.0)
    hanna_neutral(L, K, T, ltab);
  else if (K.obukhov < 0.0) {
    K.wstar = libm_pow(ltab, dmax(-kG0 / theta_v * heat_flux / (rho * kCpd) * L.depth, 0.0), 1.0 / 3.0);
    hanna_convective(L, K, T, ltab);
  } else
    hanna_stable(L, K, T, ltab);
  T.tl[0] = dmax(T.tl[0], 10.0);
  T.tl[1] = dmax(T.tl[1], 10.0);
  T.tl[2] = dmax(T.tl[2], 30.0);
  bool usable = true;
#pragma unroll
  for (int k = 0; k < 3; k++)
    usable &= (T.sigma[k] > 0.0) & (T.tl[k] > 0.0);
  if (!usable)
    return;

  // (4) the step (mptrac.c:4531-4580)
  double xi[3];
  normal_triple(ltab, ctr, g, xi[0], xi[1], xi[2]);
  double vel[3] = { (double) up, (double) vp, (double) wp };
  const double dt = P.dt, span = fabs(P.dt);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double keep = libm_exp(ltab, -span / T.tl[k]);
    double v = vel[k] * keep + T.sigma[k] * sqrt(dmax(0.0, 1.0 - keep * keep)) * xi[k];
    if (k == 2)   // drift of the vertical component: well-mixed condition + density gradient -1 / H
      v += T.tl[2] * (1.0 - keep) * (2.0 * T.sigma[2] * T.dsigma_w + (-1.0 / (1e3 * kH0)) * (T.sigma[2] * T.sigma[2]));
    vel[k] = (double) (float) v;      // (the perturbations are stored in single precision, mptrac.h:3633)
  }
  P.lon += dx2coord(M.coord_type, vel[0] * dt, P.lat);
  P.lat += dy2coord(M.coord_type, vel[1] * dt);
  double h = L.h + vel[2] * dt;
  bool flipped = false;
  // mirrors at the ground and at the layer top, one after the other as the reference takes them (mptrac.c:4562-4573);
  // a height the reference would mirror for ever (infinite: its loop does not end) or for very long is left to a bound
  for (int mirrors = 0; mirrors < 4096 && (h < 0.0 || h > L.depth); mirrors++) {
    h = h < 0.0 ? -h : 2.0 * L.depth - h;
    flipped = !flipped;
  }
  up = (float) vel[0];
  vp = (float) vel[1];
  wp = flipped ? -(float) vel[2] : (float) vel[2];
  P.p = clampd(kP0 * libm_exp(ltab, -(ground_km + h / 1000.0) / kH0), pbl, ps);   // P(z), mptrac.h:1784
}

// The closure as a function call.  Inlined into the fused step kernel its registers push the whole kernel into scratch
// (the gated instantiation with the closure inlined: +0.35 ms per step on workload C3p although 98 % of the particles
// never enter it, tools/gpu_pbl_cost.py); called, the kernel around it keeps the registers and the schedule of the
// instantiation without the closure, and only the few waves with a particle near the ground pay for the call.
// Arguments and results travel by value (a particle passed by reference would have to live in scratch memory); the
// meteo descriptor is read through a pointer into the kernel's argument segment.
struct PblState {
  double time, lon, lat, p, dt;
  float up, vp, wp;
};

__device__ __attribute__((noinline)) PblState diff_pbl_call(const DevMet *M, Axes A, PblState in, uint64_t ctr, uint64_t g,
                                                            const double *ltab) {
  Particle P;
  P.time = in.time;
  P.lon = in.lon;
  P.lat = in.lat;
  P.p = in.p;
  P.dt = in.dt;
  diff_pbl<true>(*M, A, P, in.up, in.vp, in.wp, ctr, g, ltab);
  in.lon = P.lon;
  in.lat = P.lat;
  in.p = P.p;
  return in;
}

// a particle above every boundary-layer top of the two snapshots (exact, see DevMet::turb_skip) has nothing to do there
__device__ __forceinline__ bool above_every_boundary_layer(const DevMet &M, const Particle &P) {
  const double wt = time_weight(M, P.time);
  return (P.p < M.turb_skip) & (wt >= 0.0) & (wt <= 1.0);
}

// module_isosurf for the same instantiations, as a call as well
__device__ __attribute__((noinline)) double isosurf_call(const mphip_ctl_t *ctl, const DevMet *M, Axes A, const DevAtm *a,
                                                         double time, double p, double lon, double lat, double iso_var) {
  Particle P;
  P.time = time;
  P.p = p;
  P.lon = lon;
  P.lat = lat;
  P.dt = 0;
  return isosurf_pressure(*ctl, *M, A, *a, P, iso_var);
}

}   // namespace mphip
