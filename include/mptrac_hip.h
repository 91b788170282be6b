/*
 * mptrac_hip.h -- C ABI of the MI355X (gfx950) back end for MPTRAC's
 * per-particle time-step loop.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or framework
 * types.  Each entry point names the reference interface it stands in for
 * (paths relative to the reference repository; mptrac.c = src/mptrac.c,
 * mptrac.h = src/mptrac.h).  INTEGRATION.md shows the few lines a maintainer
 * adds to mptrac.c to route the OpenACC data/compute regions through it.
 *
 * Residency contract (same as the reference's OpenACC build, mptrac.c:8005-8113,
 * 8248-8255): after mphip_update_atm() the device copy of the particles is
 * authoritative; the host sees it again only through mphip_get_atm().
 *
 * Error convention: every int-returning call returns 0 on success and a
 * non-zero code otherwise; mphip_last_error() gives the text.  The reference
 * has no error returns on this path (ERRMSG prints and exits,
 * mptrac.h:2406-2410); a host mirroring that behaviour prints the text and
 * calls exit(EXIT_FAILURE).  There is no CPU fallback: without a usable HIP
 * device mphip_create() fails.
 */
#ifndef MPTRAC_HIP_H
#define MPTRAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPHIP_NQ_MAX 16

/* 3-D meteo fields (met_t, mptrac.h:3962-4012), float [ix][iy][ip] */
enum { MPHIP_U = 0, MPHIP_V, MPHIP_W, MPHIP_T, MPHIP_LWC, MPHIP_RWC, MPHIP_IWC, MPHIP_SWC,
       /* model-level fields (met_t pl, ul, vl, zetal, zeta_dotl; mptrac.h:3997-4012), float [ix][iy][npl] */
       MPHIP_PL, MPHIP_UL, MPHIP_VL, MPHIP_ZETAL, MPHIP_ZETA_DOTL,
       MPHIP_H2O,   /* water vapour on pressure levels (module_diff_pbl, module_meteo) */
       /* read by module_meteo only (INTPOL_TIME_ALL, mptrac.h:1278-1318) */
       MPHIP_Z, MPHIP_PV, MPHIP_O3, MPHIP_CC,
       MPHIP_WL,    /* vertical velocity on model levels (met_t wl), float [ix][iy][npl]: ADVECT_VERT_COORD 2 */
       MPHIP_N3D };
/* 2-D meteo fields (met_t, mptrac.h:3886-3958), float [ix][iy] */
enum { MPHIP_PS = 0, MPHIP_PBL, MPHIP_CAPE, MPHIP_CIN, MPHIP_PEL, MPHIP_PCT, MPHIP_PCB, MPHIP_CL,
       MPHIP_ESS, MPHIP_NSS, MPHIP_SHF,   /* surface stresses and sensible heat flux (module_diff_pbl) */
       /* read by module_meteo only */
       MPHIP_TS, MPHIP_ZS, MPHIP_US, MPHIP_VS, MPHIP_LSM, MPHIP_SST, MPHIP_PT, MPHIP_TT, MPHIP_ZT, MPHIP_H2OT,
       MPHIP_PLCL, MPHIP_PLFC, MPHIP_O3C,
       MPHIP_N2D };

/* Quantities module_meteo fills (SET_ATM list, mptrac.c:5091-5157, in that order); mphip_ctl_t::qnt_met[k]
 * is the reference's ctl->qnt_<name> (-1 = not requested).  The last seven come from the zonal-mean
 * climatologies of clim_t (mphip_update_clim_zm): hno3, oh, h2o2, ho2, o1d = clim_zm / clim_oh at the particle,
 * tnat = nat_temperature(p, h2o, hno3), tsts = (tice + tnat) / 2 (needs both, mptrac.c:5074-5076).  The
 * self-assignments of the list (zeta, zeta_dot, eta, eta_dot) have no entry. */
enum { MPHIP_MQ_PS = 0, MPHIP_MQ_TS, MPHIP_MQ_ZS, MPHIP_MQ_US, MPHIP_MQ_VS, MPHIP_MQ_ESS, MPHIP_MQ_NSS,
       MPHIP_MQ_SHF, MPHIP_MQ_LSM, MPHIP_MQ_SST, MPHIP_MQ_PBL, MPHIP_MQ_PT, MPHIP_MQ_TT, MPHIP_MQ_ZT,
       MPHIP_MQ_H2OT, MPHIP_MQ_ZG, MPHIP_MQ_P, MPHIP_MQ_T, MPHIP_MQ_RHO, MPHIP_MQ_U, MPHIP_MQ_V, MPHIP_MQ_W,
       MPHIP_MQ_H2O, MPHIP_MQ_O3, MPHIP_MQ_LWC, MPHIP_MQ_RWC, MPHIP_MQ_IWC, MPHIP_MQ_SWC, MPHIP_MQ_CC,
       MPHIP_MQ_PCT, MPHIP_MQ_PCB, MPHIP_MQ_CL, MPHIP_MQ_PLCL, MPHIP_MQ_PLFC, MPHIP_MQ_PEL, MPHIP_MQ_CAPE,
       MPHIP_MQ_CIN, MPHIP_MQ_O3C, MPHIP_MQ_VH, MPHIP_MQ_VZ, MPHIP_MQ_PSAT, MPHIP_MQ_PSICE, MPHIP_MQ_PW,
       MPHIP_MQ_SH, MPHIP_MQ_RH, MPHIP_MQ_RHICE, MPHIP_MQ_THETA, MPHIP_MQ_ZETA_D, MPHIP_MQ_TVIRT,
       MPHIP_MQ_LAPSE, MPHIP_MQ_PV, MPHIP_MQ_TDEW, MPHIP_MQ_TICE,
       MPHIP_MQ_HNO3, MPHIP_MQ_OH, MPHIP_MQ_H2O2, MPHIP_MQ_HO2, MPHIP_MQ_O1D, MPHIP_MQ_TNAT, MPHIP_MQ_TSTS,
       MPHIP_NMQ };

/* Zonal-mean climatologies of clim_t that module_meteo samples (clim_zm_t members hno3, oh, h2o2, ho2, o1d,
 * mptrac.h:3745-3776, 3805-3817) */
enum { MPHIP_ZM_HNO3 = 0, MPHIP_ZM_OH, MPHIP_ZM_H2O2, MPHIP_ZM_HO2, MPHIP_ZM_O1D, MPHIP_NZM };

/* Trace gases module_bound_cond sets from a surface time series and module_mixing mixes (clim_ts_t members ccl4,
 * ccl3f, ccl2f2, n2o, sf6 of clim_t, mptrac.h:3820-3832; quantities Cccl4, Cccl3f, Cccl2f2, Cn2o, Csf6) */
enum { MPHIP_TR_CCL4 = 0, MPHIP_TR_CCL3F, MPHIP_TR_CCL2F2, MPHIP_TR_N2O, MPHIP_TR_SF6, MPHIP_NTR };

/* Module bits for mphip_module(); one bit per reference module_* function
 * (declarations mptrac.h:6140-7205). */
enum {
  MPHIP_MOD_TIMESTEPS  = 1 << 0,   /* module_timesteps   mptrac.c:5999 */
  MPHIP_MOD_POSITION   = 1 << 1,   /* module_position    mptrac.c:5435 (first call, mptrac.c:7884) */
  MPHIP_MOD_ADVECT     = 1 << 2,   /* module_advect      mptrac.c:3598 */
  MPHIP_MOD_DIFF_TURB  = 1 << 3,   /* module_diff_turb   mptrac.c:4588 */
  MPHIP_MOD_DIFF_MESO  = 1 << 4,   /* module_diff_meso   mptrac.c:4266 */
  MPHIP_MOD_CONVECTION = 1 << 5,   /* module_convection  mptrac.c:4102 */
  MPHIP_MOD_SEDI       = 1 << 6,   /* module_sedi        mptrac.c:5859 */
  MPHIP_MOD_POSITION2  = 1 << 7,   /* module_position, second call (mptrac.c:7919) */
  MPHIP_MOD_LOSS_ZERO  = 1 << 8,   /* q[loss_rate] = 0   mptrac.c:7932-7936 */
  MPHIP_MOD_DECAY      = 1 << 9,   /* module_decay       mptrac.c:4227 */
  MPHIP_MOD_WET_DEPO   = 1 << 10,  /* module_wet_depo    mptrac.c:6155 */
  MPHIP_MOD_DRY_DEPO   = 1 << 11,  /* module_dry_depo    mptrac.c:4738 */
  MPHIP_MOD_ADVECT_INIT = 1 << 12, /* module_advect_init mptrac.c:3762 (not guarded by dt) */
  MPHIP_MOD_DIFF_PBL   = 1 << 13,  /* module_diff_pbl    mptrac.c:4343 (runs between diff_turb and diff_meso) */
  MPHIP_MOD_METEO      = 1 << 14,  /* module_meteo       mptrac.c:5062 (own kernel; not guarded by dt) */
  MPHIP_MOD_ISOSURF    = 1 << 15,  /* module_isosurf     mptrac.c:4956 (not guarded by dt; between sedi and position2) */
  MPHIP_MOD_SORT       = 1 << 16,  /* module_sort        mptrac.c:5887 (own kernels) */
  MPHIP_MOD_MIXING     = 1 << 17,  /* module_mixing      mptrac.c:5169 (own kernels) */
  MPHIP_MOD_BOUND_COND = 1 << 18,  /* module_bound_cond  mptrac.c:3789, first call (mptrac.c:7929) */
  MPHIP_MOD_BOUND_COND2 = 1 << 19, /* module_bound_cond, second call (mptrac.c:8000) */
  MPHIP_MOD_ISOSURF_INIT = 1 << 20 /* module_isosurf_init mptrac.c:4886, modes 1-3 (not guarded by dt) */
};

/* Hot-path subset of ctl_t (mptrac.h:2494-3553); same field names, meaning
 * and defaults as mptrac_read_ctl (mptrac.c:6723-7741).  A compact POD is
 * passed instead of the 469 kB reference struct. */
typedef struct {
  int direction;
  int met_coord_type;
  double t_start, t_stop, dt_mod, dt_met;
  double met_utm_ref_lat;
  int nq;
  int qnt_m, qnt_vmr, qnt_rp, qnt_rhop, qnt_ens;
  int qnt_loss_rate, qnt_mloss_decay, qnt_mloss_wet, qnt_mloss_dry;
  int qnt_zeta, qnt_eta;
  int nens;
  int advect;
  int advect_vert_coord;   /* 0 pressure levels, 1 zeta, 2 pressure with model-level winds (pl, ul, vl, wl), 3 eta (mptrac.c:3609-3757) */
  int rng_type;
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
  double grid_z0, grid_z1, grid_lon0, grid_lon1, grid_lat0, grid_lat1;
  int grid_nx, grid_ny, grid_nz;
  int pad1;
  /* module_meteo (mptrac.c:7921-7924): runs when met_dt_out > 0 and (met_dt_out < dt_mod or
   * fmod(t, met_dt_out) == 0); reference default 0.1 (mptrac.c:7197) */
  double met_dt_out;
  int qnt_met[MPHIP_NMQ];
  /* module_isosurf (ISOSURF, mptrac.c:7208) and module_bound_cond (BOUND_*, mptrac.c:7266-7289) */
  int isosurf;            /* 0 none, 1 pressure, 2 density, 3 potential temperature, 4 balloon time series */
  int bound_pbl;
  int qnt_aoa;            /* age of air: set by module_bound_cond, mixed by module_mixing */
  int pad3;
  double bound_mass, bound_mass_trend, bound_vmr, bound_vmr_trend;
  double bound_lat0, bound_lat1, bound_p0, bound_p1, bound_dps, bound_dzs, bound_zetas;
  /* module_meteo's OH quantity (clim_oh, mptrac.c:89-120): exponent of the diurnal scaling (OH_CHEM_BETA, 0 = none)
   * and the reference longitude a Cartesian grid is centred on (MET_UTM_REF_LON) */
  double oh_chem_beta;
  double met_utm_ref_lon;
  /* ctl->qnt_Cccl4, qnt_Cccl3f, qnt_Cccl2f2, qnt_Cn2o, qnt_Csf6 (MPHIP_TR_* order; -1 = not present): mixed by
   * module_mixing (mptrac.c:5223-5230), set by module_bound_cond where a time series was uploaded
   * (mphip_update_clim_ts; mptrac.c:3857-3875) */
  int qnt_tracer[MPHIP_NTR];
  int pad4;
} mphip_ctl_t;

/* View of one met_t snapshot (mptrac.h:3844-4014).  The arrays stay where the
 * host has them; strides describe the reference's fixed-extent layout
 * (float u[EX][EY][EP]: sx = EY*EP, sy = EP; float ps[EX][EY]: sx2 = EY) or a
 * compact one (sx = ny*np, sy = np, sx2 = ny).  A NULL field is "not
 * provided"; modules that need it then fail with an error. */
typedef struct {
  double time;
  int coord_type;
  int nx, ny, np;
  int npl;                 /* number of model levels of the MPHIP_PL ... fields (0 = none) */
  const double *lon, *lat, *p;
  long long sx, sy, sx2;
  long long sx_ml, sy_ml;  /* strides of the model-level arrays (same as sx, sy for met_t's [EX][EY][EP]) */
  const float *f3[MPHIP_N3D];
  const float *f2[MPHIP_N2D];
} mphip_met_t;

typedef struct mphip_ctx mphip_ctx;

/* Collective hook: called on the host, with the context's stream idle, when a
 * device buffer of `count` doubles must be summed over all ranks (write_grid
 * sums, mptrac.c:13862-13872; module_mixing cell sums, mptrac.c:5289-5303).
 * A single-process run leaves it unset. */
typedef int (*mphip_allreduce_fn)(void *device_buffer, size_t count, void *user);

size_t mphip_sizeof_ctl(void);
size_t mphip_sizeof_met(void);
/* "mptrac_amd <version> (gfx950)" -- or "(gfx950, reference rounding)" from libmptrac_hip_exact.so, the build of the
 * same sources and the same ABI whose results are the CPU reference's bits ("Two libraries") */
const char *mphip_version(void);

/* mptrac_alloc / mptrac_free: device side (acc enter/exit data,
 * mptrac.c:6336-6372, 6398-6430).  `device` is the HIP device ordinal. */
int mphip_create(mphip_ctx **ctx, int device);
void mphip_destroy(mphip_ctx *ctx);
const char *mphip_last_error(const mphip_ctx *ctx);

/* mptrac_update_device(ctl, ...), mptrac.c:8013-8018 */
int mphip_update_ctl(mphip_ctx *ctx, const mphip_ctl_t *ctl);
/* mptrac_update_device(..., clim, ...), mptrac.c:8027-8032: tropopause part of
 * clim_t (mptrac.h:3785-3800); tropo is [ntime][ld] with ld >= nlat. */
int mphip_update_clim(mphip_ctx *ctx, int ntime, int nlat, const double *tropo_time,
                      const double *tropo_lat, const double *tropo, int ld);
/* ... and one of its zonal-mean climatologies (clim_zm_t, mptrac.h:3745-3776; `which` = MPHIP_ZM_*): monthly
 * times [s since the start of the year], descending pressures [hPa], ascending latitudes [deg] and the volume
 * mixing ratios vmr[ntime][np][nlat] (the reference's index order, compact).  module_meteo needs the table of
 * every climatology quantity that is requested (hno3: also for tnat); ntime = 0 removes a table. */
int mphip_update_clim_zm(mphip_ctx *ctx, int which, int ntime, int np, int nlat, const double *time,
                         const double *p, const double *lat, const double *vmr);
/* ... and one of its trace-gas time series (clim_ts_t, mptrac.h:3729-3743; `which` = MPHIP_TR_*): ascending times
 * [s] and volume mixing ratios; module_bound_cond sets the quantity of a series that is present to clim_ts at the
 * particle's time (constant beyond the ends of the series), ntime = 0 removes it -- the reference's
 * CLIM_*_TIMESERIES = "-". */
int mphip_update_clim_ts(mphip_ctx *ctx, int which, int ntime, const double *time, const double *vmr);
/* mptrac_update_device(..., met0, met1, ...), mptrac.c:8034-8048; slot 0 = met0,
 * slot 1 = met1. */
int mphip_update_met(mphip_ctx *ctx, int slot, const mphip_met_t *met);
/* the met0/met1 pointer swap in mptrac_get_met, mptrac.c:6488-6491 */
int mphip_swap_met(mphip_ctx *ctx);
/* The same hand-over of mptrac_get_met (read the next file into the old met0
 * buffer, swap, mptrac.c:6479-6503) with the upload taken off the stepping
 * path: mphip_prefetch_met() starts the host-to-device copies of the NEXT
 * snapshot into a third staging slot on a copy stream and returns at once
 * (an uploader thread of the library issues the copies, which keep their
 * calling thread busy for ordinary host memory; the caller's arrays must
 * stay untouched until the commit); time steps keep running on
 * met0 / met1 meanwhile.  mphip_commit_met() makes old met1 the new met0 and
 * the prefetched snapshot the new met1: the next kernel waits for the copy on
 * the device, the host does not block.  mphip_prefetch_done() = 1 once the
 * copies have finished.  mphip_prefetch_met() is the one entry point that may
 * be called from a second thread (a file reader) while another thread steps;
 * everything else, the commit included, belongs to the stepping thread (the
 * reference's interface is not re-entrant either).  Same grid dimensions as the resident snapshots
 * ("Meteo grid dimensions do not match!" otherwise, mptrac.c:6543-6546). */
int mphip_prefetch_met(mphip_ctx *ctx, const mphip_met_t *met);
int mphip_commit_met(mphip_ctx *ctx);
int mphip_prefetch_done(mphip_ctx *ctx);
/* drop a prefetched snapshot that will not be used (waits for its copies) */
int mphip_discard_prefetch(mphip_ctx *ctx);

/* mptrac_update_device(..., atm), mptrac.c:8050-8055.  This process owns the
 * particles [ip0, ip0 + np) of a simulation with np_total particles; random
 * numbers are drawn for the global index so results do not depend on the
 * sharding.  q[iq] may be NULL for iq >= nq. */
int mphip_update_atm(mphip_ctx *ctx, long long np, long long ip0, long long np_total, int nq,
                     const double *time, const double *p, const double *lon, const double *lat,
                     const double *const *q);
/* One quantity array of the particles, in the caller's order: for a host-side writer that changes a single
 * quantity of the model state (write_station sets the station flag, mptrac.c:15143-15145 -- on the reference's
 * CPU path that is the state the next time step sees). */
int mphip_update_quantity(mphip_ctx *ctx, int iq, const double *q);
/* mptrac_update_host(..., atm), mptrac.c:8105-8110 */
int mphip_get_atm(mphip_ctx *ctx, double *time, double *p, double *lon, double *lat,
                  double *const *q);
/* mptrac_update_device / _host (cache), mptrac.c:8020-8025, 8076-8081, plus the
 * file-static rng_ctr (mptrac.c:35).  uvwp is [np][3] as cache_t
 * (mptrac.h:3633); any pointer may be NULL. */
int mphip_update_cache(mphip_ctx *ctx, const float *uvwp, const uint64_t *rng_ctr);
int mphip_get_cache(mphip_ctx *ctx, float *uvwp, double *dt, uint64_t *rng_ctr);
/* The isosurface part of cache_t (iso_var[np], iso_ts / iso_ps [iso_n]; mptrac.h:3620-3632), as
 * mptrac_update_device / _host move it.  iso_var (per particle slot; filled on the device by
 * module_isosurf_init for ISOSURF 1-3) and the balloon time series read by module_isosurf_init for
 * ISOSURF 4 (mptrac.c:4925-4951); any pointer may be NULL. */
int mphip_update_iso(mphip_ctx *ctx, const double *iso_var, const double *iso_ts, const double *iso_ps, int iso_n);
int mphip_get_iso(mphip_ctx *ctx, double *iso_var);

/* mptrac_run_timestep, mptrac.c:7851-8001: the reference's module order and
 * gating, fused into as few launches as the order allows. */
/* module_mixing (mptrac.c:5169-5347) mixes the quantities of the hot path -- mass, volume mixing ratio, age
 * of air (qnt_m, qnt_vmr, qnt_aoa) -- in one pass; the chemistry and radionuclide quantities of the
 * reference's list (mptrac.c:5223-5230) are not part of this back end and are left untouched. */
int mphip_run_timestep(mphip_ctx *ctx, double t);
/* The time loop of the reference's driver (trac.c:204-226: `for (t = t_start; ...; t += direction * dt_mod)
 * mptrac_run_timestep(...)`) for `nsteps` consecutive steps starting at t_first: same results as nsteps calls of
 * mphip_run_timestep.  Steps with nothing scheduled between them may share one kernel launch (option
 * "multi_step" = most steps per launch, default 64; 0 = never) -- what small particle counts need, where a time
 * step is shorter than a launch.  A driver calls it for the steps up to its next output.
 * What shares launches: every integrator (ADVECT 1 / 2 / 4) and every subset of turbulent / mesoscale diffusion,
 * convection, sedimentation, with the loss / decay / deposition modules and boundary conditions, on pressure and on
 * model levels.  A step at which module_sort, module_mixing or (CONV_DT > 0) module_convection is due runs on its own
 * and a batch ends before it; module_meteo is deferred as in mphip_run_timestep, so a batch ends only behind a step that
 * schedules it when the next one does not.  module_isosurf and the boundary-layer closure (TURB_PBL_SCHEME 1) share
 * launches on pressure-level winds.  Single steps throughout: the first step (t == T_START), ADVECT 0, ISOSURF or
 * TURB_PBL_SCHEME 1 with winds from the model levels, the option "generic_kernel". */
int mphip_run_timesteps(mphip_ctx *ctx, double t_first, int nsteps);
/* One reference module_* on its own (same state hand-over through the device
 * copy of cache->dt); `modules` is one MPHIP_MOD_* bit or an OR of the
 * per-particle bits in reference order. */
int mphip_module(mphip_ctx *ctx, unsigned modules, double t);
/* Keys (as the reference's double keys) and permutation of the last
 * module_sort call, for order checks. */
int mphip_get_sort(mphip_ctx *ctx, double *keys, int *perm);

/* write_grid's binning loop (mptrac.c:13815-13872) on the device: cnt[ncell],
 * mean[nq][ncell], sigma[nq][ncell] raw sums (added in ascending particle index like the reference's loop,
 * option "deterministic_sums"), summed over the ranks through the communicator or the all-reduce hook if
 * one is set. */
int mphip_grid_sums(mphip_ctx *ctx, double t, int *cnt, double *mean, double *sigma);
/* The vertical weighting function of write_grid (GRID_KERNEL; read_kernel + kernel_weight, mptrac.c:8846-8883,
 * 3298-3320, used at mptrac.c:13866): nk nodes (height [km] ascending, weight -- already scaled to a largest
 * weight of one as read_kernel does); every summand of mphip_grid_sums is then kernel * q (and its square).
 * nk < 2 switches it off (weight one, the default). */
int mphip_set_grid_kernel(mphip_ctx *ctx, int nk, const double *kz, const double *kw);

int mphip_set_allreduce(mphip_ctx *ctx, mphip_allreduce_fn fn, void *user);

/* Multi-GPU without a host language in the data path: one process per GPU, every process owns one context
 * with an index range of the particles (mphip_update_atm: ip0, np_total) and the same meteo data.  The only
 * exchanges are the cell sums of module_mixing (mptrac.c:5289-5316: one grouped all-reduce per mixing step --
 * the sums of all mixed quantities as doubles, the cell counts once as 32-bit integers) and write_grid's sums
 * (mptrac.c:13862-13872).  With a communicator they are RCCL all-reduces issued on the context's own stream;
 * the host thread never waits for them.  librccl is loaded with dlopen at the first call, single-GPU callers
 * do not need it.  Replaces the rank -> device binding of the reference's driver (src/trac.c:70-81).
 *   mphip_comm_unique_id: rank 0 creates the 128-byte identifier (ncclGetUniqueId) and hands it to the other
 *     ranks by whatever the host has (MPI_Bcast, a TCP socket as host/trac.c does, torch.distributed);
 *   mphip_comm_init: collective over all ranks (ncclCommInitRank), after hipSetDevice of mphip_create;
 *   mphip_comm_destroy: back to a single rank (mphip_destroy does it too).
 * A communicator takes precedence over an all-reduce hook. */
int mphip_comm_unique_id(void *id128);
int mphip_comm_init(mphip_ctx *ctx, int nranks, int rank, const void *id128);
int mphip_comm_destroy(mphip_ctx *ctx);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank): *nranks = 0 without a communicator.
 * A launcher prints it next to its results so that a run that was meant to span N GPUs can be told from N
 * independent ones (the reference prints its MPI rank / size the same way, trac.c:70-81). */
int mphip_comm_query(mphip_ctx *ctx, int *nranks, int *rank);

/* Tuning knobs without a reference counterpart.
 *   "locality_sort_interval" (default 60): the device keeps the particles stored
 *   in meteo-grid-cell order and re-sorts every this many time steps so that
 *   the gathers of neighbouring particles share cache lines.  The order is
 *   internal: random numbers follow the external slot index and every download
 *   returns the caller's order, so results do not depend on the value.
 *   0 switches it off.
 *   "locality_tile" (default 0 = 4, or 8 with model-level winds): edge, in grid columns, of the horizontal tiles of
 *   that order (tile, then level, then column within the tile).
 *   "step_blocks" (default 8192): upper bound of the step kernel's grid;
 *   "xcd_map" (default 1): give each XCD one contiguous eighth of the particles;
 *   "lazy_meteo" (default 1): module_meteo (mptrac.c:7921-7924) writes quantities
 *   no module reads, so the launch a time step schedules is held back until
 *   its result can be seen -- mphip_get_atm, mphip_grid_sums, mphip_module --
 *   or its inputs change (meteo / control uploads), and is dropped when the
 *   next time step would overwrite it unseen; a time step that does not run
 *   module_meteo evaluates a pending one first, before the particles move.
 *   Downloads are bit-identical either way.  0 = launch it inside every step;
 *   "fuse_sort" (default 1): inside mphip_run_timestep the gather of time, p,
 *   lon, lat that module_sort ends with (mptrac.c:5944-5949) happens in the step
 *   launch that follows; 0 = every array is re-ordered in module_sort's own pass;
 *   "pin_host_atm" (default 0): page-lock the caller's particle arrays handed to
 *   mphip_update_atm / mphip_get_atm with one registration spanning them (for
 *   a persistent atm_t whose arrays lie in one allocation);
 *   "deterministic_sums" (default 1): the cell sums of module_mixing and of mphip_grid_sums add every cell's
 *   summands in ascending particle index, as the reference's serial loops do (mptrac.c:5289-5303,
 *   13862-13872): same bits as the serial code, from run to run and for any storage order.  0 = floating-point
 *   atomics (order of arrival; faster when single cells hold very many particles);
 *   "sort_bits" (default 0 = the width with the fewest passes; 8, 9, 10): digit width of the radix sort;
 *   "compact_depo" (default 1): a launch of module_wet_depo / module_dry_depo alone (what follows module_mixing in
 *   a time step) first packs the particles with anything to do into full waves; 0 = tail of the fused kernel.
 *   Not observable;
 *   "sort_ahead" (default 1): keys, module_timesteps and radix sort of the next time step's module_sort start on a
 *   second stream as soon as this step's particles have moved, beside module_mixing and the deposition modules;
 *   taken over by the next mphip_run_timestep if it comes with the expected time and nothing they depend on was
 *   touched in between, dropped otherwise.  Not observable;
 *   "grid_records" (default 1): mphip_grid_sums first interleaves the quantities of every particle into one record
 *   (one pass), so that the ordered sums pull one or two cache lines per particle instead of one per quantity;
 *   0 = gather from the quantity arrays.  Not observable;
 *   "locality_zorder" (default 0): number the tiles of the internal order along a Z-order curve instead of row by
 *   row (measured: no effect on the step kernel);
 *   "sum_path" (default 0 = by crowding; 1, 2): tests -- force the group / the chain algorithm of the ordered sums;
 *   "chain_blocks": tuning -- workgroups of the chain walk of the ordered sums;
 *   "generic_kernel" (default 0): tuning aid, never pick a specialised kernel;
 *   "mix_exchange_levels" (default 0): with several ranks, exchange the cell sums of module_mixing only for the band of
 *     grid levels that holds particles on any rank (a small all-reduce of the per-level occupancy, read by the host, then
 *     the band); same results, a third of the bytes on the default grid, one host synchronisation per mixing step;
 *   "lds_tile" (default 0 = off; 64 ... 2400): cells of an LDS tile of wind records that runs of pure trajectory steps
 *     (mphip_run_timesteps with module_timesteps, module_position, module_advect only) stage per workgroup; same results,
 *     measured slower than the default gathers;
 *   "emit_keys" (default 1): in a time step with module_mixing, the launch that moves the particles also writes the keys
 *     of the next step's module_sort, its module_timesteps and module_mixing's box index (otherwise a kernel of their own
 *     behind it); same values.  Taken for the headline module set only -- ADVECT 4 with turbulent + mesoscale diffusion,
 *     convection and sedimentation, no module_bound_cond; every other set (ADVECT 2 / 1, subsets of the movers, the
 *     boundary condition) keeps the separate key kernel;
 *   "sort_repair" (default 1): the module_sort that runs ahead repairs the order of the previous module_sort (only the
 *     particles that changed their cell are sorted, then merged with the others) instead of sorting from scratch; same
 *     permutation
 *   "big_grid" (default 0): tests -- take the instantiations with 64-bit byte offsets into the packed meteo records (what a
 *     grid with more than 4 GB of wind records -- 178e6 cells -- takes by itself) on a grid that fits 32 bits too: same bits. */
int mphip_set_option(mphip_ctx *ctx, const char *name, double value);
int mphip_synchronize(mphip_ctx *ctx);

/* Timing of the fused step kernel with HIP events on the context's stream:
 * between begin and end every launch is bracketed; end returns the launch
 * count and the summed device time. */
int mphip_profile_begin(mphip_ctx *ctx);
int mphip_profile_end(mphip_ctx *ctx, long long *launches, double *kernel_ms);

/* Device self-tests used by tests/ (single-precision sine/cosine of the
 * Box-Muller step over a range of float bit patterns; uniform and normal
 * random numbers for given counters). */
int mphip_test_sincosf(mphip_ctx *ctx, uint32_t bits_first, uint32_t count, float *cos_out,
                       float *sin_out);
int mphip_test_rng(mphip_ctx *ctx, uint64_t ctr, long long n, int method, double *out);
/* out[i] = exp(x[i]) (op 0), log(x[i]) (1), pow(x[i], y[i]) (2), sqrt(x[i]) (3), cos(x[i]) (4), sin(x[i]) (5) as the
 * kernels evaluate them: the restatement of the C library's exp / log / pow the reference's CPU build links
 * (src/mptrac.c:4531-4546, 5822; csrc/mphip_libm.h), the square root of the Box-Muller radius, and the library's
 * cos / sin of DX2DEG / ZETA (mptrac.h:904, 2293; the reference-rounding build and module_meteo); op + 16 reads the
 * exp / log / pow tables from an LDS copy.
 * x, y, out are host arrays of n doubles. */
int mphip_test_libm(mphip_ctx *ctx, int op, const double *x, const double *y, long long n, double *out);
/* Profiling aid: run building block `piece` of the step kernel (stencil set-up, one Runge-Kutta stage's
 * interpolation, the random-number triple, ...; list in mphip_kernels.hpp:piece_kernel) `reps` times per
 * resident particle; tools/piece_cost.py reads the instruction counters of these launches. */
int mphip_test_piece(mphip_ctx *ctx, int piece, int reps, double *checksum);
#ifdef __cplusplus
}
#endif
#endif
