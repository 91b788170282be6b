/*
 * mptrac.h -- host-side mirror of MPTRAC's high-level interface for the
 * per-particle time-step loop, on top of the MI355X back end
 * (include/mptrac_hip.h).
 *
 * Same struct names, field names and function signatures as the reference
 * (src/mptrac.h) for the hot-path subset, so that a driver written against
 * the reference's high-level interface (docs/manual/high-level-interface.md,
 * src/trac.c) compiles against this header unchanged.  The structs are NOT
 * layout-compatible with the reference's (fields outside the hot path are
 * absent); host code must be compiled against this header.
 *
 * All physics runs on the device; there is no CPU implementation of any
 * module in this library.
 */
#ifndef MPTRAC_AMD_HOST_H
#define MPTRAC_AMD_HOST_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- compile-time dimensions (reference: mptrac.h:543-595) -------------- */
#ifndef EP
#define EP 140
#endif
#ifndef EX
#define EX 1444
#endif
#ifndef EY
#define EY 724
#endif
#ifndef NP
#define NP 10000000
#endif
#ifndef NQ
#define NQ 15
#endif
#ifndef LEN
#define LEN 5000
#endif
#ifndef LOGLEV
#define LOGLEV 2
#endif

/* ---- constants and macros (reference: mptrac.h:255-345, 819-989, ...) ---- */
#ifndef H0
#define H0 7.0
#endif
#ifndef P0
#define P0 1013.25
#endif
#ifndef RE
#define RE 6367.421
#endif
#ifndef MA
#define MA 28.9644            /* molar mass of dry air [g/mol], mptrac.h:297 */
#endif
#ifndef RI
#define RI 8.3144598          /* ideal gas constant [J/(mol K)], mptrac.h:322 */
#endif
#ifndef MH2O
#define MH2O 18.01528         /* molar mass of water vapour [g/mol], mptrac.h:295 */
#endif
#ifndef MO3
#define MO3 48.00             /* molar mass of ozone [g/mol], mptrac.h:300 */
#endif
#define RA (1e3 * RI / MA)    /* specific gas constant of dry air, mptrac.h:317 */
#define FMOD(x, y) ((x) - (int) ((x) / (y)) * (y))   /* mptrac.h:1121 */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define P(z) (P0 * exp(-(z) / H0))
#define Z(p) (H0 * log(P0 / (p)))
#define SQR(x) ((x) * (x))
#define DEG2RAD(deg) ((deg) * (M_PI / 180.0))
#define ARRAY_3D(ix, iy, ny, iz, nz) (((ix) * (ny) + (iy)) * (nz) + (iz))
#define ARRAY_2D(ix, iy, ny) ((ix) * (ny) + (iy))
#define RAD2DEG(rad) ((rad) * (180.0 / M_PI))
#ifndef NENS
#define NENS 2000             /* ensemble members of the analysis outputs, mptrac.h:584 */
#endif
#ifndef NCSI
#define NCSI 1000000          /* grid boxes in the verification statistics, mptrac.h:579 */
#endif
#ifndef NOBS
#define NOBS 10000000         /* observations, mptrac.h:589 */
#endif

/* logging as the reference (mptrac.h:2303-2410): ERRMSG prints and exits */
#define LOG(level, ...) {                                               \
    if (level >= 2) printf("  ");                                       \
    if (level <= LOGLEV) { printf(__VA_ARGS__); printf("\n"); }         \
  }
#define WARN(...) {                                                     \
    printf("\nWarning (%s, %s, l%d): ", __FILE__, __func__, __LINE__);  \
    LOG(0, __VA_ARGS__);                                                \
  }
#define ERRMSG(...) {                                                   \
    printf("\nError (%s, %s, l%d): ", __FILE__, __func__, __LINE__);    \
    LOG(0, __VA_ARGS__);                                                \
    exit(EXIT_FAILURE);                                                 \
  }
#define ALLOC(ptr, type, n)                                             \
  if ((ptr = calloc((size_t) (n), sizeof(type))) == NULL)               \
    ERRMSG("Out of memory!");

/* quantities module_meteo fills: X(name, unit), in the order of its SET_ATM list (mptrac.c:5091-5157) =
 * the MPHIP_MQ_* order of include/mptrac_hip.h; units as in the SET_QNT table (mptrac.c:6853-6969) */
#define MPTRAC_METEO_QNT(X)                                                                            \
  X(ps, "hPa") X(ts, "K") X(zs, "km") X(us, "m/s") X(vs, "m/s") X(ess, "N/m^2") X(nss, "N/m^2")         \
  X(shf, "W/m^2") X(lsm, "1") X(sst, "K") X(pbl, "hPa") X(pt, "hPa") X(tt, "K") X(zt, "km")             \
  X(h2ot, "ppv") X(zg, "km") X(p, "hPa") X(t, "K") X(rho, "kg/m^3") X(u, "m/s") X(v, "m/s")             \
  X(w, "hPa/s") X(h2o, "ppv") X(o3, "ppv") X(lwc, "kg/kg") X(rwc, "kg/kg") X(iwc, "kg/kg")              \
  X(swc, "kg/kg") X(cc, "1") X(pct, "hPa") X(pcb, "hPa") X(cl, "kg/m^2") X(plcl, "hPa") X(plfc, "hPa")  \
  X(pel, "hPa") X(cape, "J/kg") X(cin, "J/kg") X(o3c, "DU") X(vh, "m/s") X(vz, "m/s") X(psat, "hPa")    \
  X(psice, "hPa") X(pw, "hPa") X(sh, "kg/kg") X(rh, "%") X(rhice, "%") X(theta, "K") X(zeta_d, "K")     \
  X(tvirt, "K") X(lapse, "K/km") X(pv, "PVU") X(tdew, "K") X(tice, "K")                                 \
  X(hno3, "ppv") X(oh, "ppv") X(h2o2, "ppv") X(ho2, "ppv") X(o1d, "ppv") X(tnat, "K") X(tsts, "K")

/* ---- structs -------------------------------------------------------------- */

/* control parameters, hot-path subset of the reference's ctl_t
 * (mptrac.h:2494-3553) */
typedef struct {
  /* quantities */
  int nq;
  char qnt_name[NQ][LEN], qnt_longname[NQ][LEN], qnt_unit[NQ][LEN], qnt_format[NQ][LEN];
  int qnt_m, qnt_vmr, qnt_rp, qnt_rhop, qnt_ens, qnt_stat, qnt_loss_rate;
  int qnt_mloss_decay, qnt_mloss_wet, qnt_mloss_dry, qnt_zeta, qnt_eta, qnt_aoa;
  int qnt_Cccl4, qnt_Cccl3f, qnt_Cccl2f2, qnt_Cn2o, qnt_Csf6;   /* trace gases: boundary condition from a time series, mixing */
  /* module_meteo outputs: qnt_ps, qnt_ts, ..., qnt_tice (mptrac.h:2518-2740) */
#define X(n, u) int qnt_##n;
  MPTRAC_METEO_QNT(X)
#undef X
  /* time and meteo input */
  int direction, met_coord_type, met_type;
  int met_nc_scale, met_pbl, met_cape;   /* netCDF input (MET_TYPE 0): packed data, source of pbl / cape */
  double t_start, t_stop, dt_mod, dt_met, met_utm_ref_lat, met_utm_ref_lon, met_dt_out;
  /* zonal-mean climatologies module_meteo samples (mptrac.c:7461-7470) and the diurnal scaling of OH (mptrac.c:7394) */
  char clim_hno3_filename[LEN], clim_oh_filename[LEN], clim_h2o2_filename[LEN], clim_ho2_filename[LEN],
    clim_o1d_filename[LEN];
  double oh_chem_beta;
  /* surface time series of the trace gases (mptrac.c:7471-7480) */
  char clim_ccl4_timeseries[LEN], clim_ccl3f_timeseries[LEN], clim_ccl2f2_timeseries[LEN], clim_n2o_timeseries[LEN],
    clim_sf6_timeseries[LEN];
  char metbase[LEN];
  /* modules */
  double sort_dt;
  int rng_type, advect, advect_vert_coord, diffusion, turb_pbl_scheme;
  double turb_dx_pbl, turb_dx_trop, turb_dx_strat, turb_dz_pbl, turb_dz_trop, turb_dz_strat;
  double turb_mesox, turb_mesoz, turb_pbl_trans;
  int conv_mix_pbl;
  double conv_pbl_trans, conv_cape, conv_cin, conv_dt;
  int isosurf;
  char balloon[LEN];
  double bound_mass, bound_mass_trend, bound_vmr, bound_vmr_trend, bound_lat0, bound_lat1, bound_p0, bound_p1;
  double bound_dps, bound_dzs, bound_zetas;
  int bound_pbl;
  double tdec_trop, tdec_strat;
  int nens;
  double mixing_dt, mixing_trop, mixing_strat, mixing_z0, mixing_z1;
  double mixing_lon0, mixing_lon1, mixing_lat0, mixing_lat1;
  int mixing_nx, mixing_ny, mixing_nz;
  double wet_depo_pre[2], wet_depo_ic_a, wet_depo_ic_b, wet_depo_bc_a, wet_depo_bc_b;
  double wet_depo_ic_h[2], wet_depo_bc_h[2], wet_depo_so2_ph;
  double wet_depo_ic_ret_ratio, wet_depo_bc_ret_ratio;
  double dry_depo_vdep, dry_depo_dp;
  /* output */
  char atm_basename[LEN];
  double atm_dt_out;
  int atm_filter, atm_stride, atm_type, atm_type_out;
  char grid_basename[LEN], grid_kernel[LEN];
  double grid_dt_out;
  int grid_sparse, grid_stddev;
  double grid_z0, grid_z1, grid_lon0, grid_lon1, grid_lat0, grid_lat1;
  int grid_nx, grid_ny, grid_nz, grid_type;
  /* the other writers of mptrac_write_output (mptrac.h:3227-3423) */
  int obs_type;
  char csi_basename[LEN], csi_kernel[LEN], csi_obsfile[LEN];
  double csi_dt_out, csi_obsmin, csi_modmin, csi_z0, csi_z1, csi_lon0, csi_lon1, csi_lat0, csi_lat1;
  int csi_nx, csi_ny, csi_nz;
  char ens_basename[LEN];
  double ens_dt_out;
  char prof_basename[LEN], prof_obsfile[LEN];
  double prof_z0, prof_z1, prof_lon0, prof_lon1, prof_lat0, prof_lat1;
  int prof_nx, prof_ny, prof_nz;
  char sample_basename[LEN], sample_kernel[LEN], sample_obsfile[LEN];
  double sample_dx, sample_dz;
  char stat_basename[LEN];
  double stat_lon, stat_lat, stat_r, stat_t0, stat_t1;
  char vtk_basename[LEN];
  double vtk_dt_out, vtk_scale, vtk_offset;
  int vtk_stride, vtk_sphere;
  double molmass;
  char species[LEN];
  /* back-end options (no reference counterpart) */
  int hip_device;
  int hip_locality_interval;
  int hip_met_prefetch;   /* HIP_MET_PREFETCH: read and upload the next meteo file beside the time steps */
} ctl_t;

/* air parcels, as the reference (mptrac.h:3563-3583) */
typedef struct {
  int np;
  double time[NP];
  double p[NP];
  double lon[NP];
  double lat[NP];
  double q[NQ][NP];
} atm_t;

/* per-parcel cache, as the reference (mptrac.h:3618-3641) */
typedef struct {
  double iso_var[NP];
  double iso_ps[NP];
  double iso_ts[NP];
  int iso_n;
  float uvwp[NP][3];
  double rs[3 * NP + 1];
  double dt[NP];
} cache_t;

/* extents of the zonal-mean climatologies (reference: mptrac.h:599-620) */
#ifndef CP
#define CP 70
#endif
#ifndef CY
#define CY 250
#endif
#ifndef CT
#define CT 12
#endif

#ifndef CTS
#define CTS 1000
#endif

/* a climatological time series, as the reference (clim_ts_t, mptrac.h:3733-3744) */
typedef struct {
  int ntime;
  double time[CTS];
  double vmr[CTS];
} clim_ts_t;

/* a zonal-mean climatology, as the reference (clim_zm_t, mptrac.h:3745-3776) */
typedef struct {
  int ntime;
  int nlat;
  int np;
  double time[CT];
  double lat[CY];
  double p[CP];
  double vmr[CT][CP][CY];
} clim_zm_t;

/* climatological data: the tropopause part of the reference's clim_t (mptrac.h:3785-3800) and the zonal means
 * module_meteo samples (mptrac.h:3805-3817); photolysis rates and the tracer time series belong to the
 * chemistry modules and are absent */
typedef struct {
  int tropo_ntime;
  int tropo_nlat;
  double tropo_time[12];
  double tropo_lat[73];
  double tropo[12][73];
  clim_zm_t hno3, oh, h2o2, ho2, o1d;
  clim_ts_t ccl4, ccl3f, ccl2f2, n2o, sf6;   /* surface time series of module_bound_cond's trace gases */
} clim_t;

/* meteo snapshot: the fields of the reference's met_t (mptrac.h:3844-4014)
 * that the hot path reads, with the reference's fixed extents */
typedef struct {
  double time;
  int coord_type;
  int nx, ny, np, npl;
  double lon[EX], lat[EY], p[EP];
  float ps[EX][EY], pbl[EX][EY], cape[EX][EY], cin[EX][EY], pel[EX][EY];
  float pct[EX][EY], pcb[EX][EY], cl[EX][EY], ess[EX][EY], nss[EX][EY], shf[EX][EY];
  /* sampled by module_meteo only */
  float ts[EX][EY], zs[EX][EY], us[EX][EY], vs[EX][EY], lsm[EX][EY], sst[EX][EY], pt[EX][EY], tt[EX][EY];
  float zt[EX][EY], h2ot[EX][EY], plcl[EX][EY], plfc[EX][EY], o3c[EX][EY];
  float z[EX][EY][EP], pv[EX][EY][EP], o3[EX][EY][EP], cc[EX][EY][EP];
  float h2o[EX][EY][EP];
  float t[EX][EY][EP], u[EX][EY][EP], v[EX][EY][EP], w[EX][EY][EP];
  float lwc[EX][EY][EP], rwc[EX][EY][EP], iwc[EX][EY][EP], swc[EX][EY][EP];
  /* model levels (mptrac.h:3997-4012); not part of the MET_TYPE 1 file format, filled by the caller */
  float pl[EX][EY][EP], ul[EX][EY][EP], vl[EX][EY][EP], zetal[EX][EY][EP], zeta_dotl[EX][EY][EP];
} met_t;

/* not used on the hot path; kept so that the reference's signatures hold */
typedef struct {
  int unused;
} depo_t;
typedef struct {
  int unused;
} dd_t;

/* ---- high-level interface (reference: mptrac.h:7246-7736) ----------------- */

void mptrac_alloc(ctl_t **ctl, cache_t **cache, clim_t **clim, met_t **met0, met_t **met1, atm_t **atm,
                  depo_t **depo, dd_t **dd);                                   /* mptrac.c:6294 */
void mptrac_free(ctl_t *ctl, cache_t *cache, clim_t *clim, met_t *met0, met_t *met1, atm_t *atm,
                 depo_t *depo, dd_t *dd);                                      /* mptrac.c:6377 */
void mptrac_read_ctl(const char *filename, int argc, char *argv[], ctl_t *ctl);  /* mptrac.c:6723 */
void mptrac_read_clim(const ctl_t *ctl, clim_t *clim);                           /* mptrac.c:6663 */
int mptrac_read_atm(const char *filename, const ctl_t *ctl, atm_t *atm);         /* mptrac.c:6588 */
int mptrac_read_met(const char *filename, const ctl_t *ctl, const clim_t *clim, met_t *met,
                    dd_t *dd);                                                  /* mptrac.c:7742 */
void mptrac_init(ctl_t *ctl, cache_t *cache, clim_t *clim, atm_t *atm, depo_t *depo,
                 const int ntask);                                              /* mptrac.c:6563 */
void mptrac_get_met(ctl_t *ctl, clim_t *clim, const double t, met_t **met0, met_t **met1,
                    dd_t *dd);                                                  /* mptrac.c:6438 */
void mptrac_run_timestep(ctl_t *ctl, cache_t *cache, clim_t *clim, met_t **met0, met_t **met1,
                         atm_t *atm, depo_t *depo, double t, dd_t *dd);         /* mptrac.c:7851 */
void mptrac_update_device(const ctl_t *ctl, const cache_t *cache, const clim_t *clim, met_t **met0,
                          met_t **met1, const atm_t *atm);                      /* mptrac.c:8005 */
void mptrac_update_host(const ctl_t *ctl, const cache_t *cache, const clim_t *clim, met_t **met0,
                        met_t **met1, const atm_t *atm);                        /* mptrac.c:8061 */
void mptrac_write_atm(const char *filename, const ctl_t *ctl, const atm_t *atm,
                      const double t);                                          /* mptrac.c:8117 */
void mptrac_write_met(const char *filename, const ctl_t *ctl, met_t *met);      /* mptrac.c:8159 */
void mptrac_write_output(const char *dirname, const ctl_t *ctl, met_t *met0, met_t *met1, atm_t *atm,
                         depo_t *depo, const double t);                         /* mptrac.c:8230 */

/* utilities of the reference that the driver and tools use */
double scan_ctl(const char *filename, int argc, char *argv[], const char *varname, const int arridx,
                const char *defvalue, char *value);                             /* mptrac.c:12434 */
void ctlfile_invalidate(void);   /* forget the parsed control file (scan_ctl reads it again) */
void jsec2time(const double jsec, int *year, int *mon, int *day, int *hour, int *min, int *sec,
               double *remain);                                                 /* mptrac.c:3265 */
void time2jsec(const int year, const int mon, const int day, const int hour, const int min,
               const int sec, const double remain, double *jsec);               /* mptrac.c:12607 */
void clim_tropo_init(clim_t *clim);                                             /* mptrac.c:241 */
void module_timesteps_init(ctl_t *ctl, const atm_t *atm);                       /* mptrac.c:6046 */
void write_grid(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm,
                const double t);                                                /* mptrac.c:13751 */
/* analysis outputs (host/output.c); they work on the particles the caller downloaded */
void write_csi(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t);       /* mptrac.c:13188 */
void write_ens(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t);       /* mptrac.c:13475 */
void write_prof(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm,
                const double t);                                                /* mptrac.c:14683 */
void write_sample(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm,
                  const double t);                                              /* mptrac.c:14918 */
void write_station(const char *filename, const ctl_t *ctl, atm_t *atm, const double t);         /* mptrac.c:15083 */
void write_vtk(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t);       /* mptrac.c:15172 */
void read_obs(const char *filename, const ctl_t *ctl, double *rt, double *rz, double *rlon, double *rlat,
              double *robs, int *nobs);                                         /* mptrac.c:12333 */
void read_kernel(const char *filename, double kz[EP], double kw[EP], int *nk);  /* mptrac.c:8846 */
double kernel_weight(const double kz[EP], const double kw[EP], const int nk, const double p);   /* mptrac.c:3298 */
void geo2cart(const double z, const double lon, const double lat, double *x);   /* mptrac.c:2601 */
void cart2geo(const double *x, double *z, double *lon, double *lat);            /* mptrac.c:74 */
int mptrac_amd_read_obs_nc(const char *filename, double *rt, double *rz, double *rlon, double *rlat, double *robs);
int mptrac_amd_world(void);   /* processes that share this run (1 without a launcher) */
/* a level field (selected by its offset in met_t) at a point, both snapshots blended in time: what
 * intpol_met_time_3d returns (mptrac.c:3112-3137) -- for the few profile / sample points of the writers */
double mptrac_amd_intpol_3d(const met_t *met0, const met_t *met1, size_t field_offset, double ts, double p,
                            double lon, double lat);

/* ---- one process per GPU (no reference counterpart) -------------------------- */
/* The reference's driver binds MPI ranks to devices and gives every rank its own work directories
 * (src/trac.c:70-98).  Here the ranks of a job share ONE simulation: every rank keeps an index range of the
 * particles, the meteo data are replicated, and the gridded sums (module_mixing, write_grid) are all-reduced
 * by RCCL inside the back end.  Rank and world size come from the launcher's environment (RANK, WORLD_SIZE,
 * LOCAL_RANK, MASTER_ADDR, MASTER_PORT). */
typedef struct {
  int rank, world, local_rank, port;
  char addr[64];
} mptrac_amd_job_t;
void mptrac_amd_job_from_env(mptrac_amd_job_t *job);
/* keep particles [np * rank / world, np * (rank + 1) / world) of a freshly read atm_t; uploads announce the range */
void mptrac_amd_shard(atm_t *atm, const mptrac_amd_job_t *job);
/* RCCL communicator of the job (collective; after mptrac_init) */
void mptrac_amd_comm_init(const ctl_t *ctl, const mptrac_amd_job_t *job);
int mptrac_amd_bcast(void *buf, size_t n, int rank, int world, const char *addr, int port);

#ifdef __cplusplus
}
#endif
#endif
