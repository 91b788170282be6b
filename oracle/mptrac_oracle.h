/*
 * mptrac_oracle.h -- CPU oracle for the MPTRAC per-particle time-step loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C (C99 + OpenMP) restatement of
 * the reference algorithm (slcs-jsc/mptrac, src/mptrac.c / src/mptrac.h); it is
 * the checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg compare the HIP path against.  Nothing in the product path
 * (mptrac_amd/, include/) may include, link or call it.
 *
 * Parity status: the reference itself cannot be compiled in this image
 * (mptrac.h:174-182 includes GSL and netCDF headers unconditionally and the
 * image has neither), so the oracle is pinned against the reference's own
 * golden files instead -- see oracle/README.md for the list of pins.
 *
 * Every function cites the reference file:line it follows.  Arithmetic is kept
 * in the reference's operation order so that results are bit-identical to a
 * gcc -O3 (no FMA) build of the reference on x86-64.
 */
#ifndef MPTRAC_ORACLE_H
#define MPTRAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NQ_MAX 16

/* 3-D met fields, each float [nx][ny][np] (level index fastest, mptrac.h:3964) */
enum { ORC_U = 0, ORC_V, ORC_W, ORC_T, ORC_LWC, ORC_RWC, ORC_IWC, ORC_SWC,
       /* model-level fields, float [nx][ny][npl] (mptrac.h:3997-4012) */
       ORC_PL, ORC_UL, ORC_VL, ORC_ZETAL, ORC_ZETA_DOTL,
       ORC_H2O,
       /* module_meteo only (INTPOL_TIME_ALL, mptrac.h:1278-1318) */
       ORC_Z, ORC_PV, ORC_O3, ORC_CC,
       ORC_WL,      /* vertical velocity on model levels, float [nx][ny][npl] (ADVECT_VERT_COORD 2) */
       ORC_N3D };
/* 2-D met fields, each float [nx][ny] */
enum { ORC_PS = 0, ORC_PBL, ORC_CAPE, ORC_CIN, ORC_PEL, ORC_PCT, ORC_PCB, ORC_CL,
       ORC_ESS, ORC_NSS, ORC_SHF,
       ORC_TS, ORC_ZS, ORC_US, ORC_VS, ORC_LSM, ORC_SST, ORC_PT, ORC_TT, ORC_ZT, ORC_H2OT,
       ORC_PLCL, ORC_PLFC, ORC_O3C, ORC_N2D };
/* quantities module_meteo sets, in the order of its SET_ATM list (mptrac.c:5091-5157) */
enum { ORC_MQ_PS = 0, ORC_MQ_TS, ORC_MQ_ZS, ORC_MQ_US, ORC_MQ_VS, ORC_MQ_ESS, ORC_MQ_NSS, ORC_MQ_SHF,
       ORC_MQ_LSM, ORC_MQ_SST, ORC_MQ_PBL, ORC_MQ_PT, ORC_MQ_TT, ORC_MQ_ZT, ORC_MQ_H2OT, ORC_MQ_ZG, ORC_MQ_P,
       ORC_MQ_T, ORC_MQ_RHO, ORC_MQ_U, ORC_MQ_V, ORC_MQ_W, ORC_MQ_H2O, ORC_MQ_O3, ORC_MQ_LWC, ORC_MQ_RWC,
       ORC_MQ_IWC, ORC_MQ_SWC, ORC_MQ_CC, ORC_MQ_PCT, ORC_MQ_PCB, ORC_MQ_CL, ORC_MQ_PLCL, ORC_MQ_PLFC,
       ORC_MQ_PEL, ORC_MQ_CAPE, ORC_MQ_CIN, ORC_MQ_O3C, ORC_MQ_VH, ORC_MQ_VZ, ORC_MQ_PSAT, ORC_MQ_PSICE,
       ORC_MQ_PW, ORC_MQ_SH, ORC_MQ_RH, ORC_MQ_RHICE, ORC_MQ_THETA, ORC_MQ_ZETA_D, ORC_MQ_TVIRT, ORC_MQ_LAPSE,
       ORC_MQ_PV, ORC_MQ_TDEW, ORC_MQ_TICE,
       ORC_MQ_HNO3, ORC_MQ_OH, ORC_MQ_H2O2, ORC_MQ_HO2, ORC_MQ_O1D, ORC_MQ_TNAT, ORC_MQ_TSTS, ORC_NMQ };
/* zonal-mean climatologies of clim_t that module_meteo samples (mptrac.h:3805-3817) */
enum { ORC_ZM_HNO3 = 0, ORC_ZM_OH, ORC_ZM_H2O2, ORC_ZM_HO2, ORC_ZM_O1D, ORC_NZM };
/* trace gases with a surface time series (clim_ts_t members ccl4, ccl3f, ccl2f2, n2o, sf6, mptrac.h:3820-3832) */
enum { ORC_TR_CCL4 = 0, ORC_TR_CCL3F, ORC_TR_CCL2F2, ORC_TR_N2O, ORC_TR_SF6, ORC_NTR };

/* Hot-path subset of ctl_t (mptrac.h:2494-3553).  Field names follow the
 * reference.  Layout is mirrored 1:1 by the Python ctypes class. */
typedef struct {
  /* time control (mptrac.c:7015-7024) */
  int direction;
  int met_coord_type;
  double t_start, t_stop, dt_mod, dt_met;
  double met_utm_ref_lat;
  /* quantities (mptrac.c:6737-6971); -1 = not present */
  int nq;
  int qnt_m, qnt_vmr, qnt_rp, qnt_rhop, qnt_ens;
  int qnt_loss_rate, qnt_mloss_decay, qnt_mloss_wet, qnt_mloss_dry;
  int qnt_zeta, qnt_eta;
  int nens;
  /* modules */
  int advect;             /* 1, 2 or 4 (mptrac.c:7218) */
  int advect_vert_coord;  /* 0/2 pressure levels, 1 zeta, 3 eta (mptrac.c:3609, 3681) */
  int rng_type;           /* only 1 (Squares) restated */
  int diffusion;
  int turb_pbl_scheme;
  int conv_mix_pbl;
  double turb_dx_pbl, turb_dx_trop, turb_dx_strat;
  double turb_dz_pbl, turb_dz_trop, turb_dz_strat;
  double turb_mesox, turb_mesoz, turb_pbl_trans;
  double conv_pbl_trans, conv_cape, conv_cin, conv_dt;
  double sort_dt;
  double tdec_trop, tdec_strat;
  double mixing_dt, mixing_trop, mixing_strat;
  double mixing_z0, mixing_z1, mixing_lon0, mixing_lon1, mixing_lat0, mixing_lat1;
  int mixing_nx, mixing_ny, mixing_nz;
  int pad0;
  double wet_depo_pre[2];
  double wet_depo_ic_a, wet_depo_ic_b, wet_depo_bc_a, wet_depo_bc_b;
  double wet_depo_ic_h[2], wet_depo_bc_h[2];
  double wet_depo_so2_ph, wet_depo_ic_ret_ratio, wet_depo_bc_ret_ratio;
  double dry_depo_vdep, dry_depo_dp;
  /* gridded output (mptrac.c:7631-7648) */
  double grid_z0, grid_z1, grid_lon0, grid_lon1, grid_lat0, grid_lat1;
  int grid_nx, grid_ny, grid_nz;
  int pad1;
  /* module_meteo (mptrac.c:7197, 7921-7924); qnt_met[k] = ctl->qnt_<name>, -1 = not present */
  double met_dt_out;
  int qnt_met[ORC_NMQ];
  /* module_isosurf (mptrac.c:7208) and module_bound_cond (mptrac.c:7266-7289) */
  int isosurf;            /* 0 none, 1 pressure, 2 density, 3 potential temperature, 4 balloon time series */
  int bound_pbl;
  int qnt_aoa;            /* age of air: set by module_bound_cond, mixed by module_mixing */
  int pad3;
  double bound_mass, bound_mass_trend, bound_vmr, bound_vmr_trend;
  double bound_lat0, bound_lat1, bound_p0, bound_p1, bound_dps, bound_dzs, bound_zetas;
  /* clim_oh (mptrac.c:89-120) */
  double oh_chem_beta;
  double met_utm_ref_lon;
  /* qnt_Cccl4, qnt_Cccl3f, qnt_Cccl2f2, qnt_Cn2o, qnt_Csf6 */
  int qnt_tracer[ORC_NTR];
  int pad4;
} orc_ctl_t;

/* One meteo snapshot: compact view of met_t (mptrac.h:3844-4014). */
typedef struct {
  double time;
  int coord_type;
  int nx, ny, np;
  int npl;                 /* number of model levels (mptrac.h:3862) */
  const double *lon, *lat, *p;
  const float *f3[ORC_N3D];
  const float *f2[ORC_N2D];
} orc_met_t;

/* Particle state, SoA as atm_t (mptrac.h:3563-3583). */
typedef struct {
  int np;
  int nq;
  double *time, *p, *lon, *lat;
  double *q[ORC_NQ_MAX];
} orc_atm_t;

/* Per-particle scratch as cache_t (mptrac.h:3618-3641) + the file-static
 * RNG counter (mptrac.c:35). */
typedef struct {
  double *dt;     /* [np]       */
  double *rs;     /* [3*np + 1] */
  float *uvwp;    /* [np][3]    */
  uint64_t rng_ctr;
  /* module_isosurf state (mptrac.h:3620-3632) */
  double *iso_var;  /* [np] */
  double *iso_ts;   /* [iso_n] balloon time series (ISOSURF 4) */
  double *iso_ps;   /* [iso_n] */
  int iso_n;
  /* Subsample mode (NULL = off): `atm` holds np particles picked from a run with np_global particles, particle
   * ip being that run's particle ip_global[ip].  The stochastic modules then draw, for every particle, the random
   * numbers the full run binds to its slot (rs[3 * ip_global + k], mptrac.c:4645-4647) and advance the counter as
   * the full run does, so a few thousand particles of a 10^7 or 10^8 run can be checked.  Only meaningful for
   * per-particle modules (no module_sort, which rebinds slots, and no module_mixing). */
  const int64_t *ip_global;
  int64_t np_global;
} orc_cache_t;

/* A zonal-mean climatology, view of clim_zm_t (mptrac.h:3745-3776): vmr[ntime][np][nlat], compact. */
typedef struct {
  int ntime, np, nlat;
  int pad;
  const double *time, *p, *lat, *vmr;
} orc_zm_t;

/* A trace-gas time series, view of clim_ts_t (mptrac.h:3729-3743). */
typedef struct {
  int ntime;
  int pad;
  const double *time, *vmr;
} orc_ts_t;

/* Climatological tropopause part of clim_t (mptrac.h:3785-3800), the zonal means module_meteo samples and the
 * time series module_bound_cond applies. */
typedef struct {
  int tropo_ntime, tropo_nlat;
  double tropo_time[12];
  double tropo_lat[73];
  double tropo[12][73];
  orc_zm_t zm[ORC_NZM];
  orc_ts_t ts[ORC_NTR];   /* ntime = 0: the reference's CLIM_*_TIMESERIES = "-" */
} orc_clim_t;

size_t orc_sizeof_ctl(void);
/* OpenMP team size of the module loops (OMP_NUM_THREADS of the reference run);
 * returns the value in effect. */
int orc_set_num_threads(int n);

/* --- helpers ------------------------------------------------------------- */
int orc_locate_irr(const double *xx, int n, double x);           /* mptrac.c:3495 */
int orc_locate_reg(const double *xx, int n, double x);           /* mptrac.c:3559 */
double orc_clim_tropo(const orc_clim_t *clim, double t, double lat);  /* mptrac.c:213 */
double orc_sedi(double p, double T, double rp, double rhop);     /* mptrac.c:12506 */
double orc_tropo_weight(const orc_ctl_t *ctl, const orc_clim_t *clim,
                        double time, double lat, double p);      /* mptrac.c:12748 */
double orc_pbl_weight(const orc_ctl_t *ctl, double p, double pbl, double ps); /* mptrac.c:8358 */
uint64_t orc_squares(uint64_t ctr);                              /* mptrac.c:5797-5809 */
/* the C library's cosf / sinf, as module_rng calls them (mptrac.c:5824-5825) */
void orc_libm_sincosf(const float *x, size_t n, float *cos_out, float *sin_out);
void orc_libm_f64(int op, const double *x, const double *y, size_t n, double *out);
void orc_intpol_met_time_3d(const orc_met_t *met0, const orc_met_t *met1, int field,
                            double ts, double p, double lon, double lat, double *var);
void orc_intpol_met_time_2d(const orc_met_t *met0, const orc_met_t *met1, int field,
                            double ts, double lon, double lat, double *var);

/* thermodynamic macros used by module_meteo: RH, RHICE, TDEW, TICE (mptrac.h:1906, 1936, 2075, 2100),
 * THETA, ZETA, TVIRT (mptrac.h:2124, 2293, 2199) and lapse_rate (mptrac.c:3324) */
double orc_rh(double p, double t, double h2o);
double orc_rhice(double p, double t, double h2o);
double orc_tdew(double p, double h2o);
double orc_tice(double p, double h2o);
double orc_theta(double p, double t);
double orc_zeta(double ps, double p, double t);
double orc_lapse_rate(double t, double h2o);
/* the climatology part of module_meteo: clim_zm (mptrac.c:414), cos_sza (mptrac.c:1857), clim_oh (mptrac.c:89),
 * nat_temperature (mptrac.c:8334) */
double orc_clim_zm(const orc_zm_t *zm, double t, double lat, double p);
double orc_cos_sza(double sec, double lon, double lat);
double orc_clim_oh(const orc_ctl_t *ctl, const orc_clim_t *clim, double t, double lon, double lat, double p);
double orc_nat_temperature(double p, double h2o, double hno3);
double orc_clim_ts(const orc_ts_t *ts, double t);   /* mptrac.c:394-410 */

/* --- modules (mptrac.c:3598-6293) ---------------------------------------- */
void orc_module_rng(const orc_ctl_t *ctl, orc_cache_t *cache, size_t n, int method);
void orc_module_timesteps(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                          const orc_atm_t *atm, double t);
void orc_module_timesteps_init(orc_ctl_t *ctl, const orc_atm_t *atm);
void orc_module_position(const orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm);
void orc_module_advect(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                       const orc_met_t *met1, orc_atm_t *atm);
void orc_module_advect_init(const orc_ctl_t *ctl, const orc_met_t *met0, const orc_met_t *met1,
                            orc_atm_t *atm);   /* mptrac.c:3762 */
void orc_module_diff_turb(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_clim_t *clim,
                          const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm);
void orc_module_diff_pbl(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm);   /* mptrac.c:4343 */
void orc_module_diff_meso(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                          const orc_met_t *met1, orc_atm_t *atm);
void orc_module_convection(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                           const orc_met_t *met1, orc_atm_t *atm);
void orc_module_sedi(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                     const orc_met_t *met1, orc_atm_t *atm);
void orc_module_decay(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_clim_t *clim,
                      orc_atm_t *atm);
void orc_module_mixing(const orc_ctl_t *ctl, const orc_clim_t *clim, orc_atm_t *atm, double t);
void orc_module_wet_depo(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm);
void orc_module_dry_depo(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                         const orc_met_t *met1, orc_atm_t *atm);
void orc_module_isosurf_init(const orc_ctl_t *ctl, orc_cache_t *cache, const orc_met_t *met0,
                             const orc_met_t *met1, const orc_atm_t *atm);   /* mptrac.c:4886 (modes 1-3) */
void orc_module_isosurf(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_met_t *met0,
                        const orc_met_t *met1, orc_atm_t *atm);              /* mptrac.c:4956 */
void orc_module_bound_cond(const orc_ctl_t *ctl, const orc_cache_t *cache, const orc_clim_t *clim,
                           const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm);   /* mptrac.c:3789 */
void orc_module_meteo(const orc_ctl_t *ctl, const orc_clim_t *clim, const orc_met_t *met0, const orc_met_t *met1,
                      orc_atm_t *atm);   /* mptrac.c:5062 */
/* keys[np] (as the reference's double keys, exact integers) and perm[np] are
 * optional outputs (may be NULL).  Ties are ordered by original index. */
void orc_module_sort(const orc_ctl_t *ctl, const orc_met_t *met0, orc_atm_t *atm,
                     double *keys, int *perm);

/* --- scheduler (mptrac.c:7851-8001) -------------------------------------- */
void orc_run_timestep(orc_ctl_t *ctl, orc_cache_t *cache, const orc_clim_t *clim,
                      const orc_met_t *met0, const orc_met_t *met1, orc_atm_t *atm, double t);

/* --- write_grid binning (mptrac.c:13815-13872), kernel weight = 1 --------- */
/* cnt[ncell] (int), mean[nq][ncell], sigma[nq][ncell] hold the raw sums
 * (before the divide at mptrac.c:13906-13913). */
void orc_grid_sums(const orc_ctl_t *ctl, const orc_atm_t *atm, double t,
                   int *cnt, double *mean, double *sigma);
/* ... with the vertical weighting function of GRID_KERNEL (nk nodes kz [km], kw; nk < 2: weight one) */
void orc_grid_sums_kernel(const orc_ctl_t *ctl, const orc_atm_t *atm, double t, int nk, const double *kz,
                          const double *kw, int *cnt, double *mean, double *sigma);

#ifdef __cplusplus
}
#endif
#endif
