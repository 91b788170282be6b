/*
 * mptrac.c -- host side of the MI355X build: the reference's high-level
 * interface (src/mptrac.h:7246-7736) implemented on the C ABI of the HIP back
 * end (include/mptrac_hip.h).  Control-file parsing, particle and meteo file
 * I/O and the output writers run on the host; every module of the time-step
 * loop runs on the device -- this file contains no physics.
 *
 * Reference citations: mptrac.c = src/mptrac.c of the reference repository.
 */
#define _GNU_SOURCE
#include "mptrac.h"
#include <stddef.h>

#include <pthread.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "../../include/mptrac_hip.h"
#include "nc_classic.h"

/* process-global device state; the reference's interface is not re-entrant
 * either (file-static RNG state, mptrac.c:32-40) */
static mphip_ctx *g_ctx;
static const met_t *g_met_host[2];     /* host snapshots mirrored in device slots met0 / met1 */
/* meteo read-ahead (HIP_MET_PREFETCH), see start_read_ahead() */
static struct {
  met_t *met;              /* the spare buffer */
  pthread_t thread;
  int active;              /* a reader thread was started for `filename` */
  int joined;              /* ... and has been joined */
  int done;                /* set by the reader when it is through (atomic) */
  int file_ok;             /* the file was read into `met` */
  int ok;                  /* its upload was started (mphip_prefetch_met) */
  ctl_t *ctl;
  clim_t *clim;
  char filename[LEN];
} g_ahead;
static long long g_ip0, g_np_total = -1; /* index range of this rank's particles (mptrac_amd_shard); -1: all of them */
static double g_release_first, g_release_last; /* release times of the whole run, taken before the shard was cut */
static int g_rank, g_world = 1;
static int g_nq;                        /* ctl->nq of the last control upload */
static int g_isosurf;                   /* ctl->isosurf of the last control upload */
static int g_meteo_fields;              /* a module_meteo quantity is requested: upload the fields only it reads */

/* Time steps handed over by mptrac_run_timestep wait here while they follow each other at the model's stride;
 * they go to the device as one mphip_run_timesteps call -- steps with nothing scheduled between them share a kernel
 * launch there -- as soon as anything else needs the device (every other call of this layer goes through HIP()),
 * the run of steps breaks, or the list is full.  The caller's loop stays the reference's: one call per step.
 * HIP_STEP_BATCH in the environment sets the most steps held back (default 64; 0 or 1: every step at once). */
static struct {
  int n, cap;
  double t_first, t_next;
} g_steps = { 0, -1, 0, 0 };

static void flush_steps(void) {
  if (!g_steps.n)
    return;
  const int n = g_steps.n;
  g_steps.n = 0;
  if (mphip_run_timesteps(g_ctx, g_steps.t_first, n) != 0)
    ERRMSG("HIP back end: %s", mphip_last_error(g_ctx));
}

#define HIP(call) {                                                     \
    flush_steps();                                                      \
    if ((call) != 0)                                                    \
      ERRMSG("HIP back end: %s", mphip_last_error(g_ctx));              \
  }

/* -------------------------------------------------------------------------- */
/* small utilities                                                            */
/* -------------------------------------------------------------------------- */

/* scan_ctl, jsec2time, time2jsec: ctlfile.c */

/* a condition the input has to meet, else the run stops with the message */
#define REQUIRE(ok, ...) { if (!(ok)) ERRMSG(__VA_ARGS__); }

/* -------------------------------------------------------------------------- */
/* alloc / free                                                               */
/* -------------------------------------------------------------------------- */

void mptrac_alloc(ctl_t **ctl, cache_t **cache, clim_t **clim, met_t **met0, met_t **met1, atm_t **atm,
                  depo_t **depo, dd_t **dd) {
  /* one calloc per struct as the reference (mptrac.c:6294-6372); the device
   * context is created in mptrac_init, once the device ordinal is known */
#define ZEROED(pp) ALLOC(*(pp), __typeof__(**(pp)), 1)
  ZEROED(ctl); ZEROED(cache); ZEROED(clim); ZEROED(met0); ZEROED(met1); ZEROED(atm);
  if (depo)
    ZEROED(depo);
  if (dd)
    ZEROED(dd);
#undef ZEROED
}

void mptrac_free(ctl_t *ctl, cache_t *cache, clim_t *clim, met_t *met0, met_t *met1, atm_t *atm,
                 depo_t *depo, dd_t *dd) {
  if (g_ahead.active && !g_ahead.joined)
    pthread_join(g_ahead.thread, NULL);
  g_ahead.active = g_ahead.joined = 0;
  if (g_ctx) {
    flush_steps();
    mphip_destroy(g_ctx);
    g_ctx = NULL;
  }
  free(g_ahead.met);
  g_ahead.met = NULL;
  g_met_host[0] = g_met_host[1] = NULL;
  g_ip0 = 0;   /* the next run of this process is cut (or not) by its own mptrac_amd_shard */
  g_np_total = -1;
  free(ctl);
  free(cache);
  free(clim);
  free(met0);
  free(met1);
  free(atm);
  free(depo);
  free(dd);
}

/* -------------------------------------------------------------------------- */
/* control parameters                                                         */
/* -------------------------------------------------------------------------- */

static const struct {
  const char *name, *longname, *unit;
} qnt_table[] = {
  /* names, descriptions and units of the reference's SET_QNT table (mptrac.c:6857-6969): the quantities of
   * the hot path and of module_meteo */
  { "idx", "particle index", "-" }, { "ens", "ensemble index", "-" }, { "stat", "station flag", "-" },
  { "m", "mass", "kg" }, { "vmr", "volume mixing ratio", "ppv" }, { "rp", "particle radius", "microns" },
  { "rhop", "particle density", "kg/m^3" }, { "loss_rate", "total loss rate", "s^-1" },
  { "mloss_decay", "mass loss due to exponential decay", "kg" }, { "mloss_wet", "mass loss due to wet deposition", "kg" },
  { "mloss_dry", "mass loss due to dry deposition", "kg" }, { "zeta", "zeta coordinate", "K" },
  { "eta", "eta coordinate", "1" }, { "aoa", "age of air", "s" },
  { "ps", "surface pressure", "hPa" }, { "ts", "surface temperature", "K" }, { "zs", "surface height", "km" },
  { "us", "surface zonal wind", "m/s" }, { "vs", "surface meridional wind", "m/s" },
  { "ess", "eastward turbulent surface stress", "N/m^2" }, { "nss", "northward turbulent surface stress", "N/m^2" },
  { "shf", "surface sensible heat flux", "W/m^2" }, { "lsm", "land-sea mask", "1" },
  { "sst", "sea surface temperature", "K" }, { "pbl", "planetary boundary layer", "hPa" },
  { "pt", "tropopause pressure", "hPa" }, { "tt", "tropopause temperature", "K" },
  { "zt", "tropopause geopotential height", "km" }, { "h2ot", "tropopause water vapor", "ppv" },
  { "zg", "geopotential height", "km" }, { "p", "pressure", "hPa" }, { "t", "temperature", "K" },
  { "rho", "air density", "kg/m^3" }, { "u", "zonal wind", "m/s" }, { "v", "meridional wind", "m/s" },
  { "w", "vertical velocity", "hPa/s" }, { "h2o", "water vapor", "ppv" }, { "o3", "ozone", "ppv" },
  { "lwc", "cloud liquid water content", "kg/kg" }, { "rwc", "cloud rain water content", "kg/kg" },
  { "iwc", "cloud ice water content", "kg/kg" }, { "swc", "cloud snow water content", "kg/kg" },
  { "cc", "cloud cover", "1" }, { "pct", "cloud top pressure", "hPa" }, { "pcb", "cloud bottom pressure", "hPa" },
  { "cl", "total column cloud water", "kg/m^2" }, { "plcl", "lifted condensation level", "hPa" },
  { "plfc", "level of free convection", "hPa" }, { "pel", "equilibrium level", "hPa" },
  { "cape", "convective available potential energy", "J/kg" }, { "cin", "convective inhibition", "J/kg" },
  { "o3c", "total column ozone", "DU" }, { "psat", "saturation pressure over water", "hPa" },
  { "psice", "saturation pressure over ice", "hPa" }, { "pw", "partial water vapor pressure", "hPa" },
  { "sh", "specific humidity", "kg/kg" }, { "rh", "relative humidity", "%" },
  { "rhice", "relative humidity over ice", "%" }, { "theta", "potential temperature", "K" },
  { "zeta_d", "diagnosed zeta coordinate", "K" }, { "tvirt", "virtual temperature", "K" },
  { "lapse", "temperature lapse rate", "K/km" }, { "vh", "horizontal velocity", "m/s" },
  { "vz", "vertical velocity", "m/s" }, { "pv", "potential vorticity", "PVU" },
  { "tdew", "dew point temperature", "K" }, { "tice", "frost point temperature", "K" },
  { "hno3", "nitric acid", "ppv" }, { "oh", "hydroxyl radical", "ppv" }, { "h2o2", "hydrogen peroxide", "ppv" },
  { "ho2", "hydroperoxyl radical", "ppv" }, { "o1d", "atomic oxygen", "ppv" },
  { "tsts", "STS existence temperature", "K" }, { "tnat", "NAT existence temperature", "K" },
  { "Cccl4", "CCl4 (CFC-10) volume mixing ratio", "ppv" }, { "Cccl3f", "CCl3F (CFC-11) volume mixing ratio", "ppv" },
  { "Cccl2f2", "CCl2F2 (CFC-12) volume mixing ratio", "ppv" }, { "Cn2o", "N2O volume mixing ratio", "ppv" },
  { "Csf6", "SF6 volume mixing ratio", "ppv" },
};

static const char *unsupported_qnt[] = {
  /* quantities only the chemistry, radioactive-decay and domain-decomposition code of the reference fills or
   * mixes (SET_QNT table, mptrac.c:6905-6969): they would be carried along unchanged here */
  "mloss_oh", "mloss_h2o2", "mloss_kpp", "Cx", "Ch2o", "Co3", "Cco", "Coh", "Ch", "Cho2", "Ch2o2", "Co1d", "Co3p",
  "Arn222", "Apb210", "Abe7", "Acs137", "Ai131", "Axe133",
  "current_subdomain", "target_subdomain", NULL
};

/* Scalar control parameters of the hot path: (kind, ctl_t member, key, default) -- defaults as the reference's
 * control-parameter documentation (docs/manual/control-parameters.md; src/mptrac.c:6974-7648).  D: double,
 * I: integer, S: text.  Keys whose default depends on other settings are read explicitly in mptrac_read_ctl. */
#define CTL_SCALARS(D, I, S) \
  I(met_coord_type, "MET_COORD_TYPE", "0") \
  I(direction, "DIRECTION", "1") \
  D(t_stop, "T_STOP", "1e100") \
  D(dt_mod, "DT_MOD", "180") \
  S(metbase, "METBASE", "-") \
  S(clim_hno3_filename, "CLIM_HNO3_FILENAME", "../../data/gozcards_HNO3.nc") \
  S(clim_oh_filename, "CLIM_OH_FILENAME", "../../data/clams_radical_species_vmr.nc") \
  S(clim_h2o2_filename, "CLIM_H2O2_FILENAME", "../../data/cams_H2O2.nc") \
  S(clim_ho2_filename, "CLIM_HO2_FILENAME", "../../data/clams_radical_species_vmr.nc") \
  S(clim_o1d_filename, "CLIM_O1D_FILENAME", "../../data/clams_radical_species_vmr.nc") \
  D(oh_chem_beta, "OH_CHEM_BETA", "0") \
  S(clim_ccl4_timeseries, "CLIM_CCL4_TIMESERIES", "../../data/noaa_gml_ccl4.tab") \
  S(clim_ccl3f_timeseries, "CLIM_CCL3F_TIMESERIES", "../../data/noaa_gml_cfc11.tab") \
  S(clim_ccl2f2_timeseries, "CLIM_CCL2F2_TIMESERIES", "../../data/noaa_gml_cfc12.tab") \
  S(clim_n2o_timeseries, "CLIM_N2O_TIMESERIES", "../../data/noaa_gml_n2o.tab") \
  S(clim_sf6_timeseries, "CLIM_SF6_TIMESERIES", "../../data/noaa_gml_sf6.tab") \
  D(dt_met, "DT_MET", "3600") \
  I(met_type, "MET_TYPE", "0") \
  D(met_dt_out, "MET_DT_OUT", "0.1") \
  I(met_nc_scale, "MET_NC_SCALE", "1") \
  I(met_pbl, "MET_PBL", "3") \
  I(met_cape, "MET_CAPE", "1") \
  D(sort_dt, "SORT_DT", "-999") \
  I(rng_type, "RNG_TYPE", "1") \
  I(advect, "ADVECT", "2") \
  I(advect_vert_coord, "ADVECT_VERT_COORD", "0") \
  I(diffusion, "DIFFUSION", "0") \
  I(turb_pbl_scheme, "TURB_PBL_SCHEME", "0") \
  D(turb_dx_pbl, "TURB_DX_PBL", "50") \
  D(turb_dx_trop, "TURB_DX_TROP", "50") \
  D(turb_dx_strat, "TURB_DX_STRAT", "0") \
  D(turb_dz_pbl, "TURB_DZ_PBL", "0") \
  D(turb_dz_trop, "TURB_DZ_TROP", "0") \
  D(turb_dz_strat, "TURB_DZ_STRAT", "0.1") \
  D(turb_mesox, "TURB_MESOX", "0.16") \
  D(turb_mesoz, "TURB_MESOZ", "0.16") \
  D(turb_pbl_trans, "TURB_PBL_TRANS", "0") \
  I(conv_mix_pbl, "CONV_MIX_PBL", "0") \
  D(conv_pbl_trans, "CONV_PBL_TRANS", "0") \
  D(conv_cape, "CONV_CAPE", "-999") \
  D(conv_cin, "CONV_CIN", "-999") \
  D(conv_dt, "CONV_DT", "-999") \
  I(isosurf, "ISOSURF", "0") \
  S(balloon, "BALLOON", "-") \
  D(bound_mass, "BOUND_MASS", "-999") \
  D(bound_mass_trend, "BOUND_MASS_TREND", "0") \
  D(bound_vmr, "BOUND_VMR", "-999") \
  D(bound_vmr_trend, "BOUND_VMR_TREND", "0") \
  D(bound_lat0, "BOUND_LAT0", "-999") \
  D(bound_lat1, "BOUND_LAT1", "-999") \
  D(bound_p0, "BOUND_P0", "-999") \
  D(bound_p1, "BOUND_P1", "-999") \
  D(bound_dps, "BOUND_DPS", "-999") \
  D(bound_dzs, "BOUND_DZS", "-999") \
  D(bound_zetas, "BOUND_ZETAS", "-999") \
  I(bound_pbl, "BOUND_PBL", "0") \
  D(wet_depo_so2_ph, "WET_DEPO_SO2_PH", "0") \
  D(wet_depo_ic_a, "WET_DEPO_IC_A", "0") \
  D(wet_depo_ic_b, "WET_DEPO_IC_B", "0") \
  D(wet_depo_bc_a, "WET_DEPO_BC_A", "0") \
  D(wet_depo_bc_b, "WET_DEPO_BC_B", "0") \
  D(wet_depo_ic_ret_ratio, "WET_DEPO_IC_RET_RATIO", "1") \
  D(wet_depo_bc_ret_ratio, "WET_DEPO_BC_RET_RATIO", "1") \
  D(dry_depo_vdep, "DRY_DEPO_VDEP", "0") \
  D(dry_depo_dp, "DRY_DEPO_DP", "30") \
  D(mixing_dt, "MIXING_DT", "3600.") \
  D(mixing_trop, "MIXING_TROP", "-999") \
  D(mixing_strat, "MIXING_STRAT", "-999") \
  D(mixing_z0, "MIXING_Z0", "-5") \
  D(mixing_z1, "MIXING_Z1", "85") \
  I(mixing_nz, "MIXING_NZ", "90") \
  D(mixing_lon0, "MIXING_LON0", "-180") \
  D(mixing_lon1, "MIXING_LON1", "180") \
  I(mixing_nx, "MIXING_NX", "360") \
  D(mixing_lat0, "MIXING_LAT0", "-90") \
  D(mixing_lat1, "MIXING_LAT1", "90") \
  I(mixing_ny, "MIXING_NY", "180") \
  D(tdec_trop, "TDEC_TROP", "0") \
  D(tdec_strat, "TDEC_STRAT", "0") \
  I(nens, "NENS", "0") \
  S(atm_basename, "ATM_BASENAME", "-") \
  D(atm_dt_out, "ATM_DT_OUT", "86400") \
  I(atm_filter, "ATM_FILTER", "0") \
  I(atm_stride, "ATM_STRIDE", "1") \
  I(atm_type, "ATM_TYPE", "0") \
  I(atm_type_out, "ATM_TYPE_OUT", "-1") \
  S(grid_basename, "GRID_BASENAME", "-") \
  S(grid_kernel, "GRID_KERNEL", "-") \
  D(grid_dt_out, "GRID_DT_OUT", "86400") \
  I(grid_sparse, "GRID_SPARSE", "0") \
  I(grid_stddev, "GRID_STDDEV", "0") \
  D(grid_z0, "GRID_Z0", "-5") \
  D(grid_z1, "GRID_Z1", "85") \
  I(grid_nz, "GRID_NZ", "1") \
  D(grid_lon0, "GRID_LON0", "-180") \
  D(grid_lon1, "GRID_LON1", "180") \
  I(grid_nx, "GRID_NX", "360") \
  D(grid_lat0, "GRID_LAT0", "-90") \
  D(grid_lat1, "GRID_LAT1", "90") \
  I(grid_ny, "GRID_NY", "180") \
  I(grid_type, "GRID_TYPE", "0") \
  I(obs_type, "OBS_TYPE", "0") \
  S(csi_basename, "CSI_BASENAME", "-") \
  S(csi_kernel, "CSI_KERNEL", "-") \
  D(csi_dt_out, "CSI_DT_OUT", "86400") \
  S(csi_obsfile, "CSI_OBSFILE", "-") \
  D(csi_obsmin, "CSI_OBSMIN", "0") \
  D(csi_modmin, "CSI_MODMIN", "0") \
  D(csi_z0, "CSI_Z0", "-5") \
  D(csi_z1, "CSI_Z1", "85") \
  I(csi_nz, "CSI_NZ", "1") \
  D(csi_lon0, "CSI_LON0", "-180") \
  D(csi_lon1, "CSI_LON1", "180") \
  I(csi_nx, "CSI_NX", "360") \
  D(csi_lat0, "CSI_LAT0", "-90") \
  D(csi_lat1, "CSI_LAT1", "90") \
  I(csi_ny, "CSI_NY", "180") \
  S(ens_basename, "ENS_BASENAME", "-") \
  D(ens_dt_out, "ENS_DT_OUT", "86400") \
  S(prof_basename, "PROF_BASENAME", "-") \
  S(prof_obsfile, "PROF_OBSFILE", "-") \
  D(prof_z0, "PROF_Z0", "0") \
  D(prof_z1, "PROF_Z1", "60") \
  I(prof_nz, "PROF_NZ", "60") \
  D(prof_lon0, "PROF_LON0", "-180") \
  D(prof_lon1, "PROF_LON1", "180") \
  I(prof_nx, "PROF_NX", "360") \
  D(prof_lat0, "PROF_LAT0", "-90") \
  D(prof_lat1, "PROF_LAT1", "90") \
  I(prof_ny, "PROF_NY", "180") \
  S(sample_basename, "SAMPLE_BASENAME", "-") \
  S(sample_kernel, "SAMPLE_KERNEL", "-") \
  S(sample_obsfile, "SAMPLE_OBSFILE", "-") \
  D(sample_dx, "SAMPLE_DX", "50") \
  D(sample_dz, "SAMPLE_DZ", "-999") \
  S(stat_basename, "STAT_BASENAME", "-") \
  D(stat_lon, "STAT_LON", "0") \
  D(stat_lat, "STAT_LAT", "0") \
  D(stat_r, "STAT_R", "50") \
  D(stat_t0, "STAT_T0", "-1e100") \
  D(stat_t1, "STAT_T1", "1e100") \
  S(vtk_basename, "VTK_BASENAME", "-") \
  D(vtk_dt_out, "VTK_DT_OUT", "86400") \
  I(vtk_stride, "VTK_STRIDE", "1") \
  D(vtk_scale, "VTK_SCALE", "1.0") \
  D(vtk_offset, "VTK_OFFSET", "0.0") \
  I(vtk_sphere, "VTK_SPHERE", "0") \
  I(hip_device, "HIP_DEVICE", "0") \
  I(hip_locality_interval, "HIP_LOCALITY_SORT_INTERVAL", "60") \
  I(hip_met_prefetch, "HIP_MET_PREFETCH", "0")

void mptrac_read_ctl(const char *filename, int argc, char *argv[], ctl_t *ctl) {
  LOG(1, "\nMassive-Parallel Trajectory Calculations (MPTRAC), MI355X build (%s)\n", mphip_version());
  ctlfile_invalidate();   /* read the control file as it is now, not as an earlier call found it */

  /* quantities, mptrac.c:6737-6971 */
  ctl->qnt_m = ctl->qnt_vmr = ctl->qnt_rp = ctl->qnt_rhop = ctl->qnt_ens = ctl->qnt_loss_rate = -1;
  ctl->qnt_mloss_decay = ctl->qnt_mloss_wet = ctl->qnt_mloss_dry = ctl->qnt_zeta = ctl->qnt_eta = -1;
  ctl->qnt_aoa = ctl->qnt_stat = -1;
  ctl->qnt_Cccl4 = ctl->qnt_Cccl3f = ctl->qnt_Cccl2f2 = ctl->qnt_Cn2o = ctl->qnt_Csf6 = -1;
#define X(n, u) ctl->qnt_##n = -1;
  MPTRAC_METEO_QNT(X)
#undef X
  ctl->nq = (int) scan_ctl(filename, argc, argv, "NQ", -1, "0", NULL);
  if (ctl->nq > NQ || ctl->nq > MPHIP_NQ_MAX)
    ERRMSG("Too many quantities!");
  for (int iq = 0; iq < ctl->nq; iq++) {
    scan_ctl(filename, argc, argv, "QNT_NAME", iq, "", ctl->qnt_name[iq]);
    scan_ctl(filename, argc, argv, "QNT_FORMAT", iq, "%g", ctl->qnt_format[iq]);
    if (strcasecmp(ctl->qnt_name[iq], "aoa") == 0)   /* mptrac.c:6852-6853 */
      sprintf(ctl->qnt_format[iq], "%%.2f");
    int known = 0;
    ctl->qnt_longname[iq][0] = '\0';
    for (size_t k = 0; k < sizeof(qnt_table) / sizeof(qnt_table[0]); k++)
      if (strcasecmp(ctl->qnt_name[iq], qnt_table[k].name) == 0) {
        sprintf(ctl->qnt_longname[iq], "%s", qnt_table[k].longname);
        sprintf(ctl->qnt_unit[iq], "%s", qnt_table[k].unit);
        known = 1;
      }
    for (int k = 0; unsupported_qnt[k]; k++)
      if (strcasecmp(ctl->qnt_name[iq], unsupported_qnt[k]) == 0)
        ERRMSG("Quantity %s is filled by the chemistry / climatology / decomposition code of the reference, "
               "which this build does not provide!", ctl->qnt_name[iq]);
    if (!known)   /* a quantity the model does not know is carried along; its unit has to be given (mptrac.c:6970) */
      scan_ctl(filename, argc, argv, "QNT_UNIT", iq, "", ctl->qnt_unit[iq]);
    const char *n = ctl->qnt_name[iq];
    if (!strcasecmp(n, "m")) ctl->qnt_m = iq;
    else if (!strcasecmp(n, "stat")) ctl->qnt_stat = iq;
    else if (!strcasecmp(n, "vmr")) ctl->qnt_vmr = iq;
    else if (!strcasecmp(n, "rp")) ctl->qnt_rp = iq;
    else if (!strcasecmp(n, "rhop")) ctl->qnt_rhop = iq;
    else if (!strcasecmp(n, "ens")) ctl->qnt_ens = iq;
    else if (!strcasecmp(n, "loss_rate")) ctl->qnt_loss_rate = iq;
    else if (!strcasecmp(n, "mloss_decay")) ctl->qnt_mloss_decay = iq;
    else if (!strcasecmp(n, "mloss_wet")) ctl->qnt_mloss_wet = iq;
    else if (!strcasecmp(n, "mloss_dry")) ctl->qnt_mloss_dry = iq;
    else if (!strcasecmp(n, "zeta")) ctl->qnt_zeta = iq;
    else if (!strcasecmp(n, "eta")) ctl->qnt_eta = iq;
    else if (!strcasecmp(n, "aoa")) ctl->qnt_aoa = iq;
    else if (!strcasecmp(n, "Cccl4")) ctl->qnt_Cccl4 = iq;
    else if (!strcasecmp(n, "Cccl3f")) ctl->qnt_Cccl3f = iq;
    else if (!strcasecmp(n, "Cccl2f2")) ctl->qnt_Cccl2f2 = iq;
    else if (!strcasecmp(n, "Cn2o")) ctl->qnt_Cn2o = iq;
    else if (!strcasecmp(n, "Csf6")) ctl->qnt_Csf6 = iq;
#define X(nm, u) else if (!strcasecmp(n, #nm)) ctl->qnt_##nm = iq;
    MPTRAC_METEO_QNT(X)
#undef X
  }

  /* every scalar key of the table above */
#define SCAN_D(field, key, def) ctl->field = scan_ctl(filename, argc, argv, key, -1, def, NULL);
#define SCAN_I(field, key, def) ctl->field = (int) scan_ctl(filename, argc, argv, key, -1, def, NULL);
#define SCAN_S(field, key, def) scan_ctl(filename, argc, argv, key, -1, def, ctl->field);
  CTL_SCALARS(SCAN_D, SCAN_I, SCAN_S)
#undef SCAN_D
#undef SCAN_I
#undef SCAN_S

  /* checks and the keys that depend on others */
  ctl->met_utm_ref_lat = (ctl->met_coord_type != 0)
    ? scan_ctl(filename, argc, argv, "MET_UTM_REF_LAT", -1, "", NULL) : 0;
  ctl->met_utm_ref_lon = (ctl->met_coord_type != 0)
    ? scan_ctl(filename, argc, argv, "MET_UTM_REF_LON", -1, "", NULL) : 0;
  REQUIRE(ctl->direction == -1 || ctl->direction == 1, "Set DIRECTION to -1 or 1!");
  if (ctl->met_type == 0) {
    /* netCDF input is taken as stored: no down-sampling, smoothing, detrending, re-gridding (mptrac.c:7770-7830) */
    static const struct {
      const char *key, *def;
    } untouched[] = { { "MET_DX", "1" }, { "MET_DY", "1" }, { "MET_DP", "1" }, { "MET_SX", "1" }, { "MET_SY", "1" },
      { "MET_SP", "1" }, { "MET_DETREND", "-999" }, { "MET_NP", "0" }, { "MET_VERT_COORD", "0" }, { "MET_CLAMS", "0" },
      { "MET_CONVENTION", "0" }, { "MET_RELHUM", "0" }, { NULL, NULL } };
    for (int k = 0; untouched[k].key; k++)
      if (scan_ctl(filename, argc, argv, untouched[k].key, -1, untouched[k].def, NULL) != atof(untouched[k].def))
        ERRMSG("%s: the meteo preprocessing of the reference is not part of this build (netCDF input is used as stored)!",
               untouched[k].key);
  }

  /* modules (mptrac.c:7196-7263) */
  REQUIRE(ctl->advect == 1 || ctl->advect == 2 || ctl->advect == 4, "Set ADVECT to 1, 2, or 4!");
  REQUIRE(ctl->turb_pbl_trans >= 0 && ctl->turb_pbl_trans <= 1, "TURB_PBL_TRANS must be in the range [0, 1]!");
  /* isosurface and boundary conditions (mptrac.c:7207-7209, 7266-7289) */
  /* SPECIES presets (values: mptrac.c:7291-7383): molar mass and Henry's-law constants become the defaults of
   * MOLMASS / WET_DEPO_*_H below; a species that also switches the OH chemistry on is only accepted with that
   * chemistry explicitly off, because this build does not provide it */
  static const struct {
    const char *name;
    double molmass, henry_ref, henry_temp;
    int oh_reaction;
  } species[] = {
    { "CF2Cl2", 120.907, 3e-5, 3500.0, 0 }, { "CFCl3", 137.359, 1.1e-4, 3300.0, 0 }, { "CH4", 16.043, 1.4e-5, 1600.0, 2 },
    { "CO", 28.01, 9.7e-6, 1300.0, 3 }, { "CO2", 44.009, 3.3e-4, 2400.0, 0 }, { "H2O", 18.01528, 0, 0, 0 },
    { "N2O", 44.013, 2.4e-4, 2600.0, 0 }, { "NH3", 17.031, 5.9e-1, 4200.0, 2 }, { "HNO3", 63.012, 2.1e3, 8700.0, 0 },
    { "NO", 30.006, 1.9e-5, 1600.0, 3 }, { "NO2", 46.005, 1.2e-4, 2400.0, 3 }, { "O3", 47.997, 1e-4, 2800.0, 2 },
    { "SF6", 146.048, 2.4e-6, 3100.0, 0 }, { "SO2", 64.066, 1.3e-2, 2900.0, 3 }, { NULL, 0, 0, 0, 0 }
  };
  char defstr[LEN];
  double molmass_default = 0, henry_default[2] = { 0, 0 };   /* calloc'ed ctl_t of the reference */
  int oh_default = 0;
  scan_ctl(filename, argc, argv, "SPECIES", -1, "-", ctl->species);
  for (int k = 0; species[k].name; k++)
    if (strcasecmp(ctl->species, species[k].name) == 0) {
      molmass_default = species[k].molmass;
      henry_default[0] = species[k].henry_ref;
      henry_default[1] = species[k].henry_temp;
      oh_default = species[k].oh_reaction;
    }
  sprintf(defstr, "%g", molmass_default);
  ctl->molmass = scan_ctl(filename, argc, argv, "MOLMASS", -1, defstr, NULL);
  sprintf(defstr, "%d", oh_default);
  if ((int) scan_ctl(filename, argc, argv, "OH_CHEM_REACTION", -1, defstr, NULL) != 0)
    ERRMSG("OH_CHEM_REACTION%s: the OH chemistry is not implemented in this build (set OH_CHEM_REACTION 0 to run "
           "the species as a passive tracer with its deposition parameters)!", oh_default ? " (switched on by SPECIES)" : "");

  /* wet / dry deposition, decay, mixing (mptrac.c:7425-7543): the species sets the defaults of the in-cloud
   * constants and of the below-cloud WET_DEPO_BC_H[0]; WET_DEPO_BC_H[1] is not a control parameter */
  for (int k = 0; k < 2; k++) {
    sprintf(defstr, "%g", henry_default[k]);
    ctl->wet_depo_ic_h[k] = scan_ctl(filename, argc, argv, "WET_DEPO_IC_H", k, defstr, NULL);
  }
  sprintf(defstr, "%g", henry_default[0]);
  ctl->wet_depo_bc_h[0] = scan_ctl(filename, argc, argv, "WET_DEPO_BC_H", 0, defstr, NULL);
  ctl->wet_depo_bc_h[1] = henry_default[1];
  ctl->wet_depo_pre[0] = scan_ctl(filename, argc, argv, "WET_DEPO_PRE", 0, "0.5", NULL);
  ctl->wet_depo_pre[1] = scan_ctl(filename, argc, argv, "WET_DEPO_PRE", 1, "0.36", NULL);
  REQUIRE(ctl->mixing_nx >= 1 && ctl->mixing_ny >= 1 && ctl->mixing_nz >= 1 && ctl->mixing_lon0 < ctl->mixing_lon1
          && ctl->mixing_lat0 < ctl->mixing_lat1 && ctl->mixing_z0 < ctl->mixing_z1 && ctl->mixing_lat0 >= -90
          && ctl->mixing_lat1 <= 90, "Invalid mixing grid!");

  /* output (mptrac.c:7551-7648) */
  if (ctl->atm_type_out < 0)   /* -1: as the input */
    ctl->atm_type_out = ctl->atm_type;
  REQUIRE(ctl->grid_nx >= 1 && ctl->grid_ny >= 1 && ctl->grid_nz >= 1, "Invalid output grid dimensions!");
  REQUIRE(ctl->grid_lon0 < ctl->grid_lon1 && ctl->grid_lat0 < ctl->grid_lat1 && ctl->grid_z0 < ctl->grid_z1
          && ctl->grid_lat0 >= -90 && ctl->grid_lat1 <= 90, "Invalid output grid boundaries!");

  /* back-end options */

  /* what this host layer / the device do not implement must not be requested silently: the reference's
   * other output writers (mptrac.c:7574-7713), chemistry and radioactive decay switches (7386-7411),
   * kernel-weighted and netCDF gridded output, domain decomposition */
  {
    static const char *const names[] = { "DEPO_BASENAME", "ATM_GPFILE", "GRID_GPFILE", NULL };
    char val[LEN];
    for (int k = 0; names[k]; k++) {
      scan_ctl(filename, argc, argv, names[k], -1, "-", val);
      if (val[0] != '-')
        ERRMSG("%s is not implemented in this host layer!", names[k]);
    }
    static const char *const switches[] = { "H2O2_CHEM_REACTION", "KPP_CHEM", "TRACER_CHEM", "RADIO_DECAY", "RADIO_DEPO",
      "DD", "MET_MPI_SHARE", NULL };
    for (int k = 0; switches[k]; k++)
      if ((int) scan_ctl(filename, argc, argv, switches[k], -1, "0", NULL) != 0)
        ERRMSG("%s is not implemented in this host layer!", switches[k]);
  }
  /* netCDF outputs are written in the classic format (host/nc_classic.c): no compression (accepted, without
   * effect), and no lossy quantisation -- that would change the stored values, so it is refused */
  {
    char key[LEN];
    for (int iq = 0; iq < ctl->nq; iq++)
      for (int k = 0; k < 2; k++) {
        sprintf(key, "%s", k ? "GRID_NC_QUANT" : "ATM_NC_QUANT");
        if ((int) scan_ctl(filename, argc, argv, key, iq, "0", NULL) != 0)
          ERRMSG("%s: quantisation of netCDF output needs the netCDF-4 library, which this build does not use!", key);
      }
    if ((int) scan_ctl(filename, argc, argv, "ATM_NC_LEVEL", -1, "0", NULL) != 0
        || (int) scan_ctl(filename, argc, argv, "GRID_NC_LEVEL", -1, "0", NULL) != 0)
      WARN("netCDF output is written uncompressed (classic format): ATM_NC_LEVEL / GRID_NC_LEVEL have no effect");
  }
  REQUIRE(ctl->grid_type == 0 || ctl->grid_type == 1, "Set GRID_TYPE to 0 or 1!");
  REQUIRE(ctl->csi_nx >= 1 && ctl->csi_ny >= 1 && ctl->csi_nz >= 1 && ctl->csi_lon0 < ctl->csi_lon1
          && ctl->csi_lat0 < ctl->csi_lat1 && ctl->csi_z0 < ctl->csi_z1 && ctl->csi_lat0 >= -90 && ctl->csi_lat1 <= 90,
          "Invalid CSI grid!");
  REQUIRE(ctl->prof_nx >= 1 && ctl->prof_ny >= 1 && ctl->prof_nz >= 1 && ctl->prof_lon0 < ctl->prof_lon1
          && ctl->prof_lat0 < ctl->prof_lat1 && ctl->prof_z0 < ctl->prof_z1 && ctl->prof_lat0 >= -90
          && ctl->prof_lat1 <= 90, "Invalid profile grid!");
  if (ctl->rng_type != 1)
    ERRMSG("This build implements RNG_TYPE 1 (Squares) only!");
  REQUIRE(ctl->advect_vert_coord >= 0 && ctl->advect_vert_coord <= 3, "Set ADVECT_VERT_COORD to 0, 1, 2, or 3!");
  REQUIRE(ctl->advect_vert_coord != 1 || ctl->qnt_zeta >= 0,                   /* messages: mptrac.c:6992-6994 */
          "Please add zeta to your quantities for diabatic calculations!");
  REQUIRE(ctl->advect_vert_coord != 3 || ctl->qnt_eta >= 0, "Please add eta to your quantities for etadot calculations!");
  /* MET_TYPE 1 files carry pressure-level fields only (mptrac.c:8887-9043); the model-level options of
   * the back end are reached through the C ABI (mphip_update_met with pl, ul, vl, wl, zetal, zeta_dotl) */
  if (ctl->advect_vert_coord == 2)
    ERRMSG("Using ADVECT_VERT_COORD = 2 requires meteo data on model levels!");            /* mptrac.c:7002 */
  if (ctl->advect_vert_coord == 1)
    ERRMSG("Please use meteo files in netcdf format for diabatic calculations.");           /* mptrac.c:7031 */
  if (ctl->advect_vert_coord == 3)
    ERRMSG("Please use meteo files in netcdf format for etadot calculations.");             /* mptrac.c:7034 */
}

/* -------------------------------------------------------------------------- */
/* climatology                                                                */
/* -------------------------------------------------------------------------- */

void clim_tropo_init(clim_t *clim) {
  /* The reference carries the NCEP/NCAR Reanalysis-1 table as a literal
   * (mptrac.c:241-371); here it is read from the data file that
   * tools/extract_clim_tropo.py wrote from that literal. */
  const char *path = getenv("MPTRAC_AMD_CLIM_TROPO");
  char buf[LEN];
  if (!path) {
    const char *dir = getenv("MPTRAC_AMD_DATA");
    snprintf(buf, LEN, "%s/clim_tropo_ncep.tab", dir ? dir : MPTRAC_AMD_DATA_DIR);
    path = buf;
  }
  FILE *in = fopen(path, "r");
  if (!in)
    ERRMSG("Cannot open tropopause climatology %s (set MPTRAC_AMD_DATA)!", path);
  char line[65536];
  int row = -3;
  while (fgets(line, sizeof(line), in)) {
    if (line[0] == '#' || line[0] == '\n')
      continue;
    char *tok = strtok(line, " \t\n");
    if (row == -3) {
      clim->tropo_ntime = atoi(tok);
      clim->tropo_nlat = atoi(strtok(NULL, " \t\n"));
      if (clim->tropo_ntime != 12 || clim->tropo_nlat != 73)
        ERRMSG("Unexpected tropopause climatology dimensions!");
    } else {
      for (int k = 0; tok; k++, tok = strtok(NULL, " \t\n")) {
        if (row == -2)
          clim->tropo_time[k] = atof(tok);
        else if (row == -1)
          clim->tropo_lat[k] = atof(tok);
        else if (row < 12 && k < 73)
          clim->tropo[row][k] = atof(tok);
      }
    }
    row++;
  }
  fclose(in);
  if (row != 12)
    ERRMSG("Error while reading tropopause climatology!");
}

/* Position of the sun at `sec` seconds since 2000-01-01 00:00 UTC -- the low-precision almanac formulae the
 * reference's cos_sza evaluates (mptrac.c:1857-1897) --, split into what depends on the time only and what depends
 * on the observer: the day / night correction below asks for 360 longitudes per (month, latitude). */
typedef struct {
  double sin_dec, cos_dec;   /* declination of the sun */
  double ra;                 /* right ascension [rad] */
  double gmst;               /* Greenwich mean sidereal time [h] */
} sun_t;

static sun_t sun_at(const double sec) {
  sun_t sun;
  const double days = sec / 86400 - 0.5;   /* since 2000-01-01 12:00 UTC */
  const double anomaly = DEG2RAD(357.529 + 0.98560028 * days);
  const double ecl_lon = DEG2RAD(280.459 + 0.98564736 * days + 1.915 * sin(anomaly) + 0.020 * sin(2 * anomaly));
  const double obliquity = DEG2RAD(23.439 - 0.00000036 * days);
  sun.sin_dec = sin(obliquity) * sin(ecl_lon);
  sun.cos_dec = sqrt(1 - SQR(sun.sin_dec));
  sun.ra = atan2(cos(obliquity) * sin(ecl_lon), cos(ecl_lon));
  sun.gmst = 18.697374558 + 24.06570982441908 * days;
  return sun;
}

/* cosine of the solar zenith angle for an observer at (lon, lat) */
static double cos_zenith(const sun_t *sun, const double lon, const double sin_lat, const double cos_lat) {
  const double hour_angle = (sun->gmst + lon / 15) / 12 * M_PI - sun->ra;
  return sin_lat * sun->sin_dec + cos_lat * sun->cos_dec * cos(hour_angle);
}

/* One zonal-mean climatology (monthly means) from a netCDF file with the dimensions time (12), press, lat and
 * the variable `varname`[time][press][lat] -- read_clim_zm, mptrac.c:8747-8843.  A file that cannot be opened is
 * a warning (the table stays empty and module_meteo refuses the quantities that need it); negative entries are
 * gaps, filled from the same column (lowest valid level, then overwritten by the highest one -- the reference's
 * two loops, in their order). */
static void read_clim_zm(const char *filename, const char *varname, clim_zm_t *zm) {
  static const double month_mid[12] = { 1209600.00, 3888000.00, 6393600.00, 9072000.00, 11664000.00, 14342400.00,
    16934400.00, 19612800.00, 22291200.00, 24883200.00, 27561600.00, 30153600.00 };
  LOG(1, "Read %s data: %s", varname, filename);
  char why[256];
  ncc_file *nc = ncc_open(filename, why, sizeof(why));
  if (!nc) {
    WARN("%s climatology data are missing! (%s)", varname, why);
    return;
  }
  long long np, nlat, nt;
  if (ncc_find_dim(nc, "press", &np) < 0 || np < 2 || np > CP)
    ERRMSG("Dimension press is missing or out of range!");
  if (ncc_find_dim(nc, "lat", &nlat) < 0 || nlat < 2 || nlat > CY)
    ERRMSG("Dimension lat is missing or out of range!");
  if (ncc_find_dim(nc, "time", &nt) < 0 || nt != 12)
    ERRMSG("Dimension time is missing or out of range!");
  const int vp = ncc_find_var(nc, "press"), vlat = ncc_find_var(nc, "lat"), var = ncc_find_var(nc, varname);
  if (vp < 0 || vlat < 0 || var < 0)
    ERRMSG("Cannot find the variables press, lat and %s!", varname);
  zm->np = (int) np;
  zm->nlat = (int) nlat;
  if (!ncc_read_double(nc, vp, 0, 0, np, zm->p) || !ncc_read_double(nc, vlat, 0, 0, nlat, zm->lat))
    ERRMSG("netCDF: %s", ncc_error(nc));
  if (zm->p[0] < zm->p[1])
    ERRMSG("Pressure data are not descending!");
  if (zm->lat[0] > zm->lat[1])
    ERRMSG("Latitude data are not ascending!");
  zm->ntime = 12;
  for (int it = 0; it < 12; it++)
    zm->time[it] = month_mid[it];
  double *help;
  ALLOC(help, double, (size_t) (np * nlat));
  const int by_record = ncc_var_is_record(nc, var);
  for (int it = 0; it < zm->ntime; it++) {
    if (!ncc_read_double(nc, var, by_record ? it : 0, by_record ? 0 : (long long) it * np * nlat, np * nlat, help))
      ERRMSG("netCDF: %s", ncc_error(nc));
    for (int iz = 0; iz < zm->np; iz++)
      for (int iy = 0; iy < zm->nlat; iy++)
        zm->vmr[it][iz][iy] = help[(size_t) iz * (size_t) zm->nlat + (size_t) iy];
  }
  free(help);
  ncc_close(nc);
  double lo = 1e99, hi = -1e99;
  for (int it = 0; it < zm->ntime; it++)
    for (int iy = 0; iy < zm->nlat; iy++)
      for (int iz = 0; iz < zm->np; iz++) {
        if (zm->vmr[it][iz][iy] < 0) {
          for (int k = 0; k < zm->np; k++)
            if (zm->vmr[it][k][iy] >= 0) {
              zm->vmr[it][iz][iy] = zm->vmr[it][k][iy];
              break;
            }
          for (int k = zm->np - 1; k >= 0; k--)
            if (zm->vmr[it][k][iy] >= 0) {
              zm->vmr[it][iz][iy] = zm->vmr[it][k][iy];
              break;
            }
        }
        lo = fmin(lo, zm->vmr[it][iz][iy]);
        hi = fmax(hi, zm->vmr[it][iz][iy]);
      }
  LOG(2, "Number of time steps: %d", zm->ntime);
  LOG(2, "Number of pressure levels: %d", zm->np);
  LOG(2, "Pressure levels: %g, %g ... %g hPa", zm->p[0], zm->p[1], zm->p[zm->np - 1]);
  LOG(2, "Number of latitudes: %d", zm->nlat);
  LOG(2, "Latitudes: %g, %g ... %g deg", zm->lat[0], zm->lat[1], zm->lat[zm->nlat - 1]);
  LOG(2, "%s volume mixing ratio range: %g ... %g ppv", varname, lo, hi);
}

/* clim_oh_diurnal_correction (mptrac.c:122-152): the OH table divided by the zonal mean of the day / night
 * factor exp(-beta / cos(sza)) that clim_oh applies again per particle.  The mean depends on month and latitude
 * only: one row of factors per month, applied to every level. */
static void clim_oh_diurnal_correction(const ctl_t *ctl, clim_t *clim) {
  const double floor_csza = cos(DEG2RAD(85.));
  for (int it = 0; it < clim->oh.ntime; it++) {
    const sun_t sun = sun_at(clim->oh.time[it]);
    for (int iy = 0; iy < clim->oh.nlat; iy++) {
      const double sin_lat = sin(DEG2RAD(clim->oh.lat[iy])), cos_lat = cos(DEG2RAD(clim->oh.lat[iy]));
      double sum = 0;
      for (int k = 0; k < 360; k++) {
        const double csza = cos_zenith(&sun, -180.0 + k, sin_lat, cos_lat);
        sum += exp(-ctl->oh_chem_beta / fmax(csza, floor_csza));
      }
      const double zonal_mean = sum / 360.0;
      for (int iz = 0; iz < clim->oh.np; iz++)
        clim->oh.vmr[it][iz][iy] /= zonal_mean;
    }
  }
}

/* Text table of two numeric columns; lines that do not start with two numbers (headers, comments) are skipped.
 * Returns the number of rows stored (at most `max`; one more row is an error), or -1 if the file cannot be opened. */
static int read_two_columns(const char *filename, double *x, double *y, const int max) {
  FILE *in = fopen(filename, "r");
  if (!in)
    return -1;
  int n = 0;
  char line[LEN];
  while (fgets(line, LEN, in)) {
    char *end;
    const double a = strtod(line, &end);
    if (end == line)
      continue;
    char *end2;
    const double b = strtod(end, &end2);
    if (end2 == end)
      continue;
    if (n >= max) {
      fclose(in);
      ERRMSG("Too many data points!");
    }
    x[n] = a;
    y[n] = b;
    n++;
  }
  fclose(in);
  return n;
}

/* A trace-gas time series "year vmr" (read_clim_ts, mptrac.c:8693-8743); years become seconds since 2000-01-01.
 * A file that cannot be opened is a warning: the gas then has no boundary condition. */
static int read_clim_ts(const char *filename, clim_ts_t *ts) {
  LOG(1, "Read climatological time series: %s", filename);
  const int n = read_two_columns(filename, ts->time, ts->vmr, CTS - 1);
  if (n < 0) {
    WARN("Cannot open file!");
    return 0;
  }
  if (n < 2)
    ERRMSG("Not enough data points!");
  double lo = ts->vmr[0], hi = ts->vmr[0];
  for (int i = 0; i < n; i++) {
    ts->time[i] = (ts->time[i] - 2000.0) * 365.25 * 86400.;
    if (i > 0 && ts->time[i] <= ts->time[i - 1])
      ERRMSG("Time series must be ascending!");
    lo = fmin(lo, ts->vmr[i]);
    hi = fmax(hi, ts->vmr[i]);
  }
  ts->ntime = n;
  LOG(2, "Number of time steps: %d", ts->ntime);
  LOG(2, "Time steps: %.2f, %.2f ... %.2f s", ts->time[0], ts->time[1], ts->time[n - 1]);
  LOG(2, "Volume mixing ratio range: %g ... %g ppv", lo, hi);
  return 1;
}

/* mptrac.c:6663-6719: the tropopause climatology and the zonal means a requested module_meteo quantity needs
 * (the reference reads all its climatologies whatever the quantities; its photolysis rates and tracer time
 * series feed the chemistry modules, which are not part of this build) */
void mptrac_read_clim(const ctl_t *ctl, clim_t *clim) {
  clim_tropo_init(clim);
  /* zonal means: read when a module_meteo quantity needs the table and a file is named */
  const struct {
    int wanted;
    const char *file, *var;
    clim_zm_t *zm;
  } zonal[] = {
    { ctl->qnt_hno3 >= 0 || ctl->qnt_tnat >= 0, ctl->clim_hno3_filename, "HNO3", &clim->hno3 },
    { ctl->qnt_oh >= 0, ctl->clim_oh_filename, "OH", &clim->oh },
    { ctl->qnt_h2o2 >= 0, ctl->clim_h2o2_filename, "H2O2", &clim->h2o2 },
    { ctl->qnt_ho2 >= 0, ctl->clim_ho2_filename, "HO2", &clim->ho2 },
    { ctl->qnt_o1d >= 0, ctl->clim_o1d_filename, "O1D", &clim->o1d },
  };
  for (size_t k = 0; k < sizeof(zonal) / sizeof(zonal[0]); k++)
    if (zonal[k].wanted && zonal[k].file[0] != '-') {
      read_clim_zm(zonal[k].file, zonal[k].var, zonal[k].zm);
      if (zonal[k].zm == &clim->oh && ctl->oh_chem_beta > 0)
        clim_oh_diurnal_correction(ctl, clim);
    }
  /* surface time series of the trace gases that are carried (module_bound_cond, mptrac.c:3857-3875) */
  const struct {
    int qnt;
    const char *file;
    clim_ts_t *ts;
  } series[] = {
    { ctl->qnt_Cccl4, ctl->clim_ccl4_timeseries, &clim->ccl4 },     { ctl->qnt_Cccl3f, ctl->clim_ccl3f_timeseries, &clim->ccl3f },
    { ctl->qnt_Cccl2f2, ctl->clim_ccl2f2_timeseries, &clim->ccl2f2 }, { ctl->qnt_Cn2o, ctl->clim_n2o_timeseries, &clim->n2o },
    { ctl->qnt_Csf6, ctl->clim_sf6_timeseries, &clim->sf6 },
  };
  for (size_t k = 0; k < sizeof(series) / sizeof(series[0]); k++)
    if (series[k].qnt >= 0 && series[k].file[0] != '-')
      read_clim_ts(series[k].file, series[k].ts);
}

/* -------------------------------------------------------------------------- */
/* particle I/O                                                               */
/* -------------------------------------------------------------------------- */

/* binary records: `count` items of `size` bytes, all or nothing */
static void get_items(FILE *f, void *dst, size_t size, size_t count) {
  if (fread(dst, size, count, f) != count)
    ERRMSG("Error while reading!");
}

static void put_items(FILE *f, const void *src, size_t size, size_t count) {
  if (fwrite(src, size, count, f) != count)
    ERRMSG("Error while writing!");
}

static int get_int(FILE *f) {
  int v;
  get_items(f, &v, sizeof(int), 1);
  return v;
}

/* the per-particle columns of an atm_t in file order: time, pressure, longitude, latitude, quantities */
static int atm_columns(const ctl_t *ctl, atm_t *atm, double *col[4 + NQ]) {
  col[0] = atm->time;
  col[1] = atm->p;
  col[2] = atm->lon;
  col[3] = atm->lat;
  for (int iq = 0; iq < ctl->nq; iq++)
    col[4 + iq] = atm->q[iq];
  return 4 + ctl->nq;
}

/* ASCII table (reference format: docs/manual/input-data.md; src/mptrac.c read_atm_asc): one particle per line --
 * time, altitude [km], longitude, latitude, then the quantities; lines that do not start with a number are
 * comments, a line that starts with numbers but ends early is an error */
static int read_atm_asc(const char *filename, const ctl_t *ctl, atm_t *atm) {
  FILE *in = fopen(filename, "r");
  if (!in) {
    WARN("Cannot open file!");
    return 0;
  }
  double *col[4 + NQ];
  const int ncol = atm_columns(ctl, atm, col);
  char line[LEN];
  while (fgets(line, LEN, in)) {
    double v[4 + NQ];
    int got = 0;
    for (char *tok = strtok(line, " \t"); tok && got < ncol && sscanf(tok, "%lg", &v[got]) == 1; tok = strtok(NULL, " \t"))
      got++;
    if (got == 0 || (got < ncol && line[0] == '#'))
      continue;
    if (got < ncol)
      ERRMSG("Error while reading!");
    if (atm->np >= NP)
      ERRMSG("Too many data points!");
    v[1] = P(v[1]);   /* altitude -> pressure */
    for (int k = 0; k < ncol; k++)
      col[k][atm->np] = v[k];
    atm->np++;
  }
  fclose(in);
  return 1;
}

/* binary particle file, version 100: [100] [np] then the columns as double[np] each, [999] */
static int read_atm_bin(const char *filename, const ctl_t *ctl, atm_t *atm) {
  FILE *in = fopen(filename, "r");
  if (!in)
    return 0;
  if (get_int(in) != 100)
    ERRMSG("Wrong version of binary data!");
  atm->np = get_int(in);
  if (atm->np < 0 || atm->np > NP)
    ERRMSG("Too many data points!");
  double *col[4 + NQ];
  const int ncol = atm_columns(ctl, atm, col);
  for (int k = 0; k < ncol; k++)
    get_items(in, col[k], sizeof(double), (size_t) atm->np);
  if (get_int(in) != 999)
    ERRMSG("Error while reading binary data!");
  fclose(in);
  return 1;
}

/* ---- netCDF particle files (classic format; host/nc_classic.c) ---- */

/* a whole one-dimensional (or [1][n]) variable as doubles; `need`: stop if it is missing, else warn and leave
 * `dst` alone.  1 = read */
static int nc_column(ncc_file *nc, const char *name, long long n, double *dst, int need) {
  const int var = ncc_find_var(nc, name);
  if (var < 0) {
    if (need)
      ERRMSG("Cannot find variable %s in the netCDF file!", name);
    WARN("netCDF variable %s is missing!", name);
    return 0;
  }
  if (!ncc_read_double(nc, var, 0, 0, n, dst))
    ERRMSG("Cannot read variable %s: %s", name, ncc_error(nc));
  return 1;
}

static ncc_file *nc_open_or_null(const char *filename) {
  char why[256];
  ncc_file *nc = ncc_open(filename, why, sizeof(why));
  if (!nc) {
    FILE *probe = fopen(filename, "r");
    if (probe) {   /* the file exists but is not a classic netCDF file: say why instead of "not found" */
      fclose(probe);
      ERRMSG("%s: %s", filename, why);
    }
  }
  return nc;
}

static long long nc_particles(ncc_file *nc, const char *dim) {
  long long n = 0;
  if (ncc_find_dim(nc, dim, &n) < 0)
    ERRMSG("Cannot find dimension %s in the netCDF file!", dim);
  if (n < 1 || n > NP)
    ERRMSG("Dimension %s is out of range!", dim);
  return n;
}

/* ATM_TYPE 2 (mptrac.c:8541-8573): dimension obs; time, press, lon, lat and one variable per quantity */
static int read_atm_nc(const char *filename, const ctl_t *ctl, atm_t *atm) {
  ncc_file *nc = nc_open_or_null(filename);
  if (!nc)
    return 0;
  const long long n = nc_particles(nc, "obs");
  atm->np = (int) n;
  nc_column(nc, "time", n, atm->time, 1);
  nc_column(nc, "press", n, atm->p, 1);
  nc_column(nc, "lon", n, atm->lon, 1);
  nc_column(nc, "lat", n, atm->lat, 1);
  for (int iq = 0; iq < ctl->nq; iq++)
    nc_column(nc, ctl->qnt_name[iq], n, atm->q[iq], 0);
  ncc_close(nc);
  return 1;
}

/* ATM_TYPE 3 / 4, CLaMS position files (mptrac.c:8478-8537): dimension NPARTS; LON, LAT, TIME_INIT (or one
 * scalar time), PRESS_INIT or PRESS, ZETA with diabatic advection, quantities by name */
static int read_atm_clams(const char *filename, const ctl_t *ctl, atm_t *atm) {
  if (ctl->met_coord_type != 0)
    ERRMSG("CLaMS atmospheric files support only lat/lon grids");
  ncc_file *nc = nc_open_or_null(filename);
  if (!nc)
    return 0;
  const long long n = nc_particles(nc, "NPARTS");
  atm->np = (int) n;
  if (ncc_find_var(nc, "TIME_INIT") >= 0)
    nc_column(nc, "TIME_INIT", n, atm->time, 1);
  else {
    WARN("TIME_INIT not found use time instead!");
    double t_file;
    nc_column(nc, "time", 1, &t_file, 1);
    for (long long ip = 0; ip < n; ip++)
      atm->time[ip] = t_file;
  }
  if (ctl->advect_vert_coord == 1) {
    nc_column(nc, "ZETA", n, atm->q[ctl->qnt_zeta], 1);
    nc_column(nc, "PRESS", n, atm->p, 0);
  } else if (ncc_find_var(nc, "PRESS_INIT") >= 0)
    nc_column(nc, "PRESS_INIT", n, atm->p, 1);
  else {
    WARN("PRESS_INIT not found use PRESS instead!");
    nc_column(nc, "PRESS", n, atm->p, 1);
  }
  for (int iq = 0; iq < ctl->nq; iq++)
    nc_column(nc, ctl->qnt_name[iq], n, atm->q[iq], 0);
  nc_column(nc, "LON", n, atm->lon, 1);
  nc_column(nc, "LAT", n, atm->lat, 1);
  ncc_close(nc);
  return 1;
}

int mptrac_amd_read_obs_nc(const char *filename, double *rt, double *rz, double *rlon, double *rlat, double *robs) {
  ncc_file *nc = nc_open_or_null(filename);
  if (!nc)
    ERRMSG("Cannot open file!");
  long long n = 0;
  if (ncc_find_dim(nc, "nobs", &n) < 0 || n < 1 || n > NOBS)
    ERRMSG("Dimension nobs is missing or out of range!");
  nc_column(nc, "time", n, rt, 1);
  nc_column(nc, "alt", n, rz, 1);
  nc_column(nc, "lon", n, rlon, 1);
  nc_column(nc, "lat", n, rlat, 1);
  nc_column(nc, "obs", n, robs, 1);
  ncc_close(nc);
  return (int) n;
}

#define NCW(call) { if ((call) < 0) ERRMSG("netCDF output failed: %s", nccw_error(w)); }

static int ncw_var(nccw_file *w, const char *name, int type, int ndims, const int *dims, const char *longname,
                   const char *units) {
  const int v = nccw_def_var(w, name, type, ndims, dims);
  NCW(v);
  NCW(nccw_put_att_text(w, v, "long_name", longname));
  NCW(nccw_put_att_text(w, v, "units", units));
  return v;
}

static void ncw_put(nccw_file *w, const char *name, long long rec, const double *data) {
  NCW(nccw_put_double(w, nccw_find_var(w, name), rec, data));
}

static nccw_file *ncw_create(const char *filename) {
  nccw_file *w = nccw_create(filename);
  if (!w)
    ERRMSG("Cannot create file!");
  return w;
}

/* ATM_TYPE_OUT 2 (mptrac.c:13139-13185) */
static void write_atm_nc(const char *filename, const ctl_t *ctl, const atm_t *atm) {
  nccw_file *w = ncw_create(filename);
  const int obs = nccw_def_dim(w, "obs", atm->np);
  NCW(obs);
  ncw_var(w, "time", NCC_DOUBLE, 1, &obs, "time", "seconds since 2000-01-01 00:00:00 UTC");
  ncw_var(w, "press", NCC_DOUBLE, 1, &obs, "pressure", "hPa");
  ncw_var(w, "lon", NCC_DOUBLE, 1, &obs, "longitude", "degrees_east");
  ncw_var(w, "lat", NCC_DOUBLE, 1, &obs, "latitude", "degrees_north");
  for (int iq = 0; iq < ctl->nq; iq++)
    ncw_var(w, ctl->qnt_name[iq], NCC_DOUBLE, 1, &obs, ctl->qnt_longname[iq], ctl->qnt_unit[iq]);
  NCW(nccw_put_att_text(w, -1, "featureType", "point"));
  NCW(nccw_enddef(w));
  ncw_put(w, "time", 0, atm->time);
  ncw_put(w, "press", 0, atm->p);
  ncw_put(w, "lon", 0, atm->lon);
  ncw_put(w, "lat", 0, atm->lat);
  for (int iq = 0; iq < ctl->nq; iq++)
    ncw_put(w, ctl->qnt_name[iq], 0, atm->q[iq]);
  NCW(nccw_close(w));
}

/* the variables of a CLaMS file: positions per particle (`per_time`: with a leading time dimension), quantities
 * always [time][NPARTS] */
static void clams_define(nccw_file *w, const ctl_t *ctl, int tid, int pid, int per_time) {
  const int both[2] = { tid, pid };
  ncw_var(w, "time", NCC_DOUBLE, 1, &tid, "Time", "seconds since 2000-01-01 00:00:00 UTC");
  static const char *const pos[4][3] = { { "LAT", "Latitude", "deg" }, { "LON", "Longitude", "deg" },
    { "PRESS", "Pressure", "hPa" }, { "ZETA", "Zeta", "K" } };
  for (int k = 0; k < 4; k++)
    ncw_var(w, pos[k][0], NCC_DOUBLE, per_time ? 2 : 1, per_time ? both : &pid, pos[k][1], pos[k][2]);
  for (int iq = 0; iq < ctl->nq; iq++)
    ncw_var(w, ctl->qnt_name[iq], NCC_DOUBLE, 2, both, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
  NCW(nccw_put_att_text(w, -1, "exp_VERTCOOR_name", "zeta"));
  NCW(nccw_put_att_text(w, -1, "model", "MPTRAC"));
  NCW(nccw_enddef(w));
}

/* the vertical coordinate a CLaMS file carries as ZETA: the advected zeta, else the diagnosed one */
static const double *clams_zeta(const ctl_t *ctl, const atm_t *atm, int diagnosed_only) {
  if (!diagnosed_only && ctl->advect_vert_coord == 1)
    return atm->q[ctl->qnt_zeta];
  if (ctl->qnt_zeta_d >= 0 && (diagnosed_only || ctl->qnt_zeta >= 0))
    return atm->q[ctl->qnt_zeta_d];
  return NULL;
}

static void clams_put(nccw_file *w, const ctl_t *ctl, const atm_t *atm, long long rec, const double *zeta) {
  ncw_put(w, "time", rec, atm->time);   /* (one value: the time of the first particle, as the reference writes it) */
  ncw_put(w, "LAT", rec, atm->lat);
  ncw_put(w, "LON", rec, atm->lon);
  ncw_put(w, "PRESS", rec, atm->p);
  if (zeta)
    ncw_put(w, "ZETA", rec, zeta);
  for (int iq = 0; iq < ctl->nq; iq++)
    ncw_put(w, ctl->qnt_name[iq], rec, atm->q[iq]);
}

/* ATM_TYPE_OUT 4: CLaMS position file (mptrac.c:12922-12974) */
static void write_atm_clams(const char *filename, const ctl_t *ctl, const atm_t *atm) {
  if (ctl->met_coord_type != 0)
    ERRMSG("CLaMS atmospheric files support only lat/lon grids");
  /* Without the quantity zeta_d the reference writes atm->q[-1] as ZETA -- the array in front of q in its atm_t,
   * the latitudes (mptrac.c:12968; its tests/interoper_test converts a file without quantities this way).  The
   * variable is kept, with the same values, so that such files stay readable where ZETA is mandatory. */
  const double *zeta = clams_zeta(ctl, atm, 1);
  if (!zeta) {
    WARN("Quantity zeta_d is missing: ZETA of the position file is not a vertical coordinate!");
    zeta = atm->lat;
  }
  nccw_file *w = ncw_create(filename);
  const int tid = nccw_def_dim(w, "time", 1), pid = nccw_def_dim(w, "NPARTS", atm->np);
  NCW(tid);
  NCW(pid);
  clams_define(w, ctl, tid, pid, 0);
  clams_put(w, ctl, atm, 0, zeta);
  NCW(nccw_close(w));
}

/* ATM_TYPE_OUT 3: CLaMS trajectory file traj_fix_3d_<start>_<stop>.nc in the directory of `filename`, one record
 * per output time, and at the stop time the position file init_fix_<stop>.nc (mptrac.c:12978-13135) */
static void write_atm_clams_traj(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  static nccw_file *traj;
  if (ctl->met_coord_type != 0)
    ERRMSG("CLaMS atmospheric files support only lat/lon grids");
  char dir[2 * LEN], path[3 * LEN];
  snprintf(dir, sizeof(dir), "%s", filename);
  char *slash = strrchr(dir, '/');
  if (slash)
    *slash = '\0';
  else
    sprintf(dir, ".");
  int y0, m0, d0, h0, y1, m1, d1, h1, y, m, d, h, mi, se;
  double r;
  jsec2time(ctl->t_start, &y0, &m0, &d0, &h0, &mi, &se, &r);
  jsec2time(ctl->t_stop, &y1, &m1, &d1, &h1, &mi, &se, &r);
  jsec2time(t, &y, &m, &d, &h, &mi, &se, &r);
  if (!traj) {
    snprintf(path, sizeof(path), "%s/traj_fix_3d_%02d%02d%02d%02d_%02d%02d%02d%02d.nc", dir, y0 % 100, m0, d0, h0,
             y1 % 100, m1, d1, h1);
    LOG(1, "Write traj file: %s", path);
    traj = ncw_create(path);
    nccw_file *w = traj;
    const int tid = nccw_def_dim(w, "time", 0), pid = nccw_def_dim(w, "NPARTS", atm->np);
    NCW(tid);
    NCW(pid);
    NCW(nccw_def_dim(w, "TMDT", 7));
    clams_define(w, ctl, tid, pid, 1);
  }
  clams_put(traj, ctl, atm, nccw_numrecs(traj), clams_zeta(ctl, atm, 0));
  if (y == y1 && m == m1 && d == d1 && h == h1) {
    nccw_file *w = traj;
    NCW(nccw_close(w));
    traj = NULL;
    snprintf(path, sizeof(path), "%s/init_fix_%02d%02d%02d%02d.nc", dir, y1 % 100, m1, d1, h1);
    LOG(1, "Write init file: %s", path);
    write_atm_clams(path, ctl, atm);
  }
}

int mptrac_read_atm(const char *filename, const ctl_t *ctl, atm_t *atm) {
  LOG(1, "Read atmospheric data: %s", filename);
  atm->np = 0;
  int ok;
  switch (ctl->atm_type) {
  case 0: ok = read_atm_asc(filename, ctl, atm); break;
  case 1: ok = read_atm_bin(filename, ctl, atm); break;
  case 2: ok = read_atm_nc(filename, ctl, atm); break;
  case 3: case 4: ok = read_atm_clams(filename, ctl, atm); break;
  default: ERRMSG("Atmospheric data type not supported!");
  }
  if (!ok)
    return 0;
  if (atm->np < 1)
    ERRMSG("Can not read any data!");
  LOG(2, "Number of particles: %d", atm->np);
  return 1;
}

/* ASCII table as the reference writes it: numbered column legend, blank line, one particle per line;
 * ATM_FILTER 1 blanks (NaN), 2 drops the particles whose time is not within half a step of t */
static void write_atm_asc(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  FILE *out = fopen(filename, "w");
  REQUIRE(out, "Cannot create file!");
  const int cartesian = ctl->met_coord_type != 0;
  const char *legend[4] = { "time [s]", "altitude [km]", cartesian ? "x [m]" : "longitude [deg]",
    cartesian ? "y [m]" : "latitude [deg]" };
  for (int k = 0; k < 4; k++)
    fprintf(out, "# $%d = %s\n", k + 1, legend[k]);
  for (int iq = 0; iq < ctl->nq; iq++)
    fprintf(out, "# $%i = %s [%s]\n", iq + 5, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
  fputc('\n', out);
  const char *position = cartesian ? "%.2f %g %.2f %.2f" : "%.2f %g %g %g";
  for (int ip = 0; ip < atm->np; ip += ctl->atm_stride) {
    const int outside = atm->time[ip] < t - 0.5 * ctl->dt_mod || atm->time[ip] > t + 0.5 * ctl->dt_mod;
    if (outside && ctl->atm_filter == 2)
      continue;
    fprintf(out, position, atm->time[ip], Z(atm->p[ip]), atm->lon[ip], atm->lat[ip]);
    for (int iq = 0; iq < ctl->nq; iq++) {
      fputc(' ', out);
      fprintf(out, ctl->qnt_format[iq], (outside && ctl->atm_filter == 1) ? NAN : atm->q[iq][ip]);
    }
    fputc('\n', out);
  }
  fclose(out);
}

static void write_atm_bin(const char *filename, const ctl_t *ctl, const atm_t *atm) {
  FILE *out = fopen(filename, "w");
  REQUIRE(out, "Cannot create file!");
  const int head[2] = { 100, atm->np }, tail = 999;
  put_items(out, head, sizeof(int), 2);
  double *col[4 + NQ];
  const int ncol = atm_columns(ctl, (atm_t *) atm, col);
  for (int k = 0; k < ncol; k++)
    put_items(out, col[k], sizeof(double), (size_t) atm->np);
  put_items(out, &tail, sizeof(int), 1);
  fclose(out);
}

void mptrac_write_atm(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  LOG(1, "Write atmospheric data: %s", filename);
  switch (ctl->atm_type_out) {   /* ATM_TYPE_OUT: 0 ASCII, 1 binary, 2 netCDF, 3 CLaMS trajectories, 4 CLaMS positions */
  case 0: write_atm_asc(filename, ctl, atm, t); break;
  case 1: write_atm_bin(filename, ctl, atm); break;
  case 2: write_atm_nc(filename, ctl, atm); break;
  case 3: write_atm_clams_traj(filename, ctl, atm, t); break;
  case 4: write_atm_clams(filename, ctl, atm); break;
  default: ERRMSG("Atmospheric data type not supported!");
  }
}

/* every column (i, j) of the grid */
#define EACH_COLUMN(met, i, j) for (int i = 0; i < (met)->nx; i++) for (int j = 0; j < (met)->ny; j++)

/* -------------------------------------------------------------------------- */
/* meteo I/O: the reference's raw binary format (MET_TYPE 1, version 104)     */
/* -------------------------------------------------------------------------- */

/* Level fields are read by a few threads side by side: each takes a slab of longitudes, reads its part of the block
 * with pread() and clamps / scatters it into the member of met_t -- a 0.5 degree x 137 level snapshot is 1.9 GB, and one
 * thread moves that at 1.4 GB/s (profiles/r06_trac_dropin.txt: the drop-in driver at 10^7 particles is bound by this
 * reader, not by the time steps).  MPTRAC_AMD_IO_THREADS sets the number (default: 8 or the cores there are; 1 = serial). */
typedef struct {
  int fd;
  off_t off;             /* file offset of the block */
  const met_t *met;
  float *field, *buf;
  size_t lev, stride;
  float lo, hi;
  int ix0, ix1, ok;
} met_slab_t;

static void *met_slab_main(void *arg) {
  met_slab_t *s = (met_slab_t *) arg;
  const size_t per_ix = (size_t) s->met->ny * s->lev;
  char *dst = (char *) (s->buf + (size_t) s->ix0 * per_ix);
  size_t left = (size_t) (s->ix1 - s->ix0) * per_ix * sizeof(float);
  off_t at = s->off + (off_t) ((size_t) s->ix0 * per_ix * sizeof(float));
  s->ok = 1;
  while (left > 0) {
    const ssize_t got = pread(s->fd, dst, left, at);
    if (got <= 0) {
      s->ok = 0;
      return NULL;
    }
    dst += got;
    at += got;
    left -= (size_t) got;
  }
  if (s->field) {
    size_t k = (size_t) s->ix0 * per_ix;
    for (int ix = s->ix0; ix < s->ix1; ix++)
      for (int iy = 0; iy < s->met->ny; iy++) {
        float *cell = s->field + ((size_t) ix * EY + (size_t) iy) * s->stride;
        for (size_t ip = 0; ip < s->lev; ip++, k++)
          cell[ip] = fminf(fmaxf(s->buf[k], s->lo), s->hi);
      }
  }
  return NULL;
}

static int met_io_threads(void) {
  static int n;
  if (!n) {
    const char *e = getenv("MPTRAC_AMD_IO_THREADS");
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    n = e ? atoi(e) : (int) (cores < 8 ? cores : 8);
    if (n < 1)
      n = 1;
    if (n > 64)
      n = 64;
  }
  return n;
}

/* One field of a MET_TYPE 1 file: compact [nx][ny] or [nx][ny][np] floats <-> the fixed-extent member of met_t
 * (nlev = 0: surface field).  On reading, level fields are limited to [lo, hi] as the reference does
 * (read_met_bin_3d); dst == NULL skips the block, src == NULL writes zeros. */
static void met_block(FILE *f, const int write, const met_t *met, float *field, const int nlev, const float lo,
                      const float hi, float *buf) {
  const size_t lev = nlev > 0 ? (size_t) nlev : 1, stride = nlev > 0 ? EP : 1;
  const size_t n = (size_t) met->nx * (size_t) met->ny * lev;
  const int nthreads = met_io_threads();
  if (!write && nlev > 0 && nthreads > 1 && met->nx >= 2 * nthreads) {
    met_slab_t slab[64];
    pthread_t th[64];
    const off_t off = ftello(f);
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
      slab[t] = (met_slab_t) { fileno(f), off, met, field, buf, lev, stride, lo, hi,
                               (int) ((long) met->nx * t / nthreads), (int) ((long) met->nx * (t + 1) / nthreads), 0 };
      if (pthread_create(&th[t], NULL, met_slab_main, &slab[t]) != 0)
        break;
      started++;
    }
    for (int t = started; t < nthreads; t++)     /* (threads that could not be started: their slabs here) */
      met_slab_main(&slab[t]);
    int ok = 1;
    for (int t = 0; t < nthreads; t++) {
      if (t < started)
        pthread_join(th[t], NULL);
      ok &= slab[t].ok;
    }
    if (!ok || fseeko(f, off + (off_t) (n * sizeof(float)), SEEK_SET) != 0)
      ERRMSG("Error while reading!");
    return;
  }
  if (!write)
    get_items(f, buf, sizeof(float), n);
  size_t k = 0;
  for (int ix = 0; ix < met->nx; ix++)
    for (int iy = 0; iy < met->ny; iy++) {
      float *cell = field ? field + ((size_t) ix * EY + (size_t) iy) * stride : NULL;
      for (size_t ip = 0; ip < lev; ip++, k++) {
        if (write)
          buf[k] = cell ? cell[ip] : 0.f;
        else if (cell)
          cell[ip] = nlev > 0 ? fminf(fmaxf(buf[k], lo), hi) : buf[k];
      }
    }
  if (write)
    put_items(f, buf, sizeof(float), n);
}

/* the 24 surface and 13 level fields in file order (version 104 of the format) */
static void met_bin_body(FILE *f, int write, met_t *met) {
  float *buf;
  if ((buf = malloc((size_t) met->nx * (size_t) met->ny * (size_t) met->np * sizeof(float))) == NULL)     /* (not zeroed: every block is read or written in full) */
    ERRMSG("Out of memory!");
  float *surface[24] = { &met->ps[0][0], &met->ts[0][0], &met->zs[0][0], &met->us[0][0], &met->vs[0][0], &met->ess[0][0],
    &met->nss[0][0], &met->shf[0][0], &met->lsm[0][0], &met->sst[0][0], &met->pbl[0][0], &met->pt[0][0], &met->tt[0][0],
    &met->zt[0][0], &met->h2ot[0][0], &met->pct[0][0], &met->pcb[0][0], &met->cl[0][0], &met->plcl[0][0],
    &met->plfc[0][0], &met->pel[0][0], &met->cape[0][0], &met->cin[0][0], &met->o3c[0][0] };
  for (int k = 0; k < 24; k++)
    met_block(f, write, met, surface[k], 0, 0.f, 0.f, buf);
  const struct {
    float *field, lo, hi;
  } level[13] = { { &met->z[0][0][0], -1e34f, 1e34f }, { &met->t[0][0][0], 0, 1e34f }, { &met->u[0][0][0], -1e34f, 1e34f },
    { &met->v[0][0][0], -1e34f, 1e34f }, { &met->w[0][0][0], -1e34f, 1e34f }, { &met->pv[0][0][0], -1e34f, 1e34f },
    { &met->h2o[0][0][0], 0, 1e34f }, { &met->o3[0][0][0], 0, 1e34f }, { &met->lwc[0][0][0], 0, 1e34f },
    { &met->rwc[0][0][0], 0, 1e34f }, { &met->iwc[0][0][0], 0, 1e34f }, { &met->swc[0][0][0], 0, 1e34f },
    { &met->cc[0][0][0], 0, 1 } };
  for (int k = 0; k < 13; k++)
    met_block(f, write, met, level[k].field, met->np, level[k].lo, level[k].hi, buf);
  free(buf);
}

static int read_met_nc(const char *filename, const ctl_t *ctl, met_t *met);

static int extent_from_file(FILE *in, const int max, const char *what) {
  const int n = get_int(in);
  if (n < 2 || n > max)
    ERRMSG("Number of %s out of range!", what);
  return n;
}

/* MET_TYPE 1: [1] [104] time nx ny np lon[] lat[] p[] fields [999] (reference: read_met_bin, write_met_bin) */
int mptrac_read_met(const char *filename, const ctl_t *ctl, const clim_t *clim, met_t *met, dd_t *dd) {
  (void) clim;
  (void) dd;
  LOG(1, "Read meteo data: %s", filename);
  if (ctl->met_type == 0)
    return read_met_nc(filename, ctl, met);
  if (ctl->met_type != 1)
    ERRMSG("This build reads MET_TYPE 0 (classic netCDF, grids without preprocessing) and 1 (raw binary) meteo files!");
  FILE *in = fopen(filename, "r");
  if (!in) {
    WARN("Cannot open file!");
    return 0;
  }
  if (get_int(in) != ctl->met_type)
    ERRMSG("Wrong MET_TYPE of binary data!");
  if (get_int(in) != 104)
    ERRMSG("Wrong version of binary data!");
  get_items(in, &met->time, sizeof(double), 1);
  met->coord_type = ctl->met_coord_type;
  met->nx = extent_from_file(in, EX, "longitudes");
  met->ny = extent_from_file(in, EY, "latitudes");
  met->np = extent_from_file(in, EP, "levels");
  get_items(in, met->lon, sizeof(double), (size_t) met->nx);
  get_items(in, met->lat, sizeof(double), (size_t) met->ny);
  get_items(in, met->p, sizeof(double), (size_t) met->np);
  met_bin_body(in, 0, met);
  if (get_int(in) != 999)
    ERRMSG("Error while reading binary data!");
  fclose(in);
  return 1;
}

static void write_met_nc(const char *filename, const ctl_t *ctl, met_t *met);

void mptrac_write_met(const char *filename, const ctl_t *ctl, met_t *met) {
  LOG(1, "Write meteo data: %s", filename);
  if (ctl->met_type == 0) {
    write_met_nc(filename, ctl, met);
    return;
  }
  if (ctl->met_type != 1)
    ERRMSG("This build writes MET_TYPE 0 (netCDF, classic format) and 1 (raw binary) meteo files!");
  FILE *out = fopen(filename, "w");
  REQUIRE(out, "Cannot create file!");
  const int head[2] = { ctl->met_type, 104 }, dims[3] = { met->nx, met->ny, met->np }, tail = 999;
  put_items(out, head, sizeof(int), 2);
  put_items(out, &met->time, sizeof(double), 1);
  put_items(out, dims, sizeof(int), 3);
  put_items(out, met->lon, sizeof(double), (size_t) met->nx);
  put_items(out, met->lat, sizeof(double), (size_t) met->ny);
  put_items(out, met->p, sizeof(double), (size_t) met->np);
  met_bin_body(out, 1, met);
  put_items(out, &tail, sizeof(int), 1);
  fclose(out);
}

/* MET_TYPE 0 output (mptrac.c:14440-14618): dimensions time, lon / lat (x / y on Cartesian grids), lev; the
 * surface fields as [time][lat][lon], the level fields as [time][lev][lat][lon], single precision, in the units
 * of the ECMWF archives the reference reads (pressures in Pa, heights in m or as geopotential, humidity and ozone
 * as mass mixing ratios) */
static void write_met_nc(const char *filename, const ctl_t *ctl, met_t *met) {
  (void) ctl;
  static const struct {
    const char *name, *longname, *units;
    size_t offset;
    double scale;
  } surface[] = {
    { "sp", "Surface pressure", "Pa", offsetof(met_t, ps), 100. },
    { "z", "Geopotential", "m**2 s**-2", offsetof(met_t, zs), 1000. * 9.80665 },
    { "t2m", "2 metre temperature", "K", offsetof(met_t, ts), 1. },
    { "u10m", "10 metre U wind component", "m s**-1", offsetof(met_t, us), 1. },
    { "v10m", "10 metre V wind component", "m s**-1", offsetof(met_t, vs), 1. },
    { "iews", "Instantaneous eastward turbulent surface stress", "N m**-2", offsetof(met_t, ess), 1. },
    { "inss", "Instantaneous northward turbulent surface stress", "N m**-2", offsetof(met_t, nss), 1. },
    { "ishf", "Instantaneous surface sensible heat flux", "W m**-2", offsetof(met_t, shf), 1. },
    { "lsm", "Land/sea mask", "-", offsetof(met_t, lsm), 1. },
    { "sstk", "Sea surface temperature", "K", offsetof(met_t, sst), 1. },
    { "blp", "Boundary layer pressure", "Pa", offsetof(met_t, pbl), 100. },
    { "pt", "Tropopause pressure", "Pa", offsetof(met_t, pt), 100. },
    { "tt", "Tropopause temperature", "K", offsetof(met_t, tt), 1. },
    { "zt", "Tropopause height", "m", offsetof(met_t, zt), 1000. },
    { "h2ot", "Tropopause water vapor", "ppv", offsetof(met_t, h2ot), 1. },
    { "pct", "Cloud top pressure", "Pa", offsetof(met_t, pct), 100. },
    { "pcb", "Cloud bottom pressure", "Pa", offsetof(met_t, pcb), 100. },
    { "cl", "Total column cloud water", "kg m**2", offsetof(met_t, cl), 1. },
    { "plcl", "Pressure at lifted condensation level (LCL)", "Pa", offsetof(met_t, plcl), 100. },
    { "plfc", "Pressure at level of free convection (LFC)", "Pa", offsetof(met_t, plfc), 100. },
    { "pel", "Pressure at equilibrium level (EL)", "Pa", offsetof(met_t, pel), 100. },
    { "cape", "Convective available potential energy", "J kg**-1", offsetof(met_t, cape), 1. },
    { "cin", "Convective inhibition", "J kg**-1", offsetof(met_t, cin), 1. },
    { "o3c", "Total column ozone", "DU", offsetof(met_t, o3c), 1. },
  }, level[] = {
    { "t", "Temperature", "K", offsetof(met_t, t), 1. },
    { "u", "U velocity", "m s**-1", offsetof(met_t, u), 1. },
    { "v", "V velocity", "m s**-1", offsetof(met_t, v), 1. },
    { "w", "Vertical velocity", "Pa s**-1", offsetof(met_t, w), 100. },
    { "q", "Specific humidity", "kg kg**-1", offsetof(met_t, h2o), MH2O / MA },
    { "o3", "Ozone mass mixing ratio", "kg kg**-1", offsetof(met_t, o3), MO3 / MA },
    { "clwc", "Cloud liquid water content", "kg kg**-1", offsetof(met_t, lwc), 1. },
    { "crwc", "Cloud rain water content", "kg kg**-1", offsetof(met_t, rwc), 1. },
    { "ciwc", "Cloud ice water content", "kg kg**-1", offsetof(met_t, iwc), 1. },
    { "cswc", "Cloud snow water content", "kg kg**-1", offsetof(met_t, swc), 1. },
    { "cc", "Cloud cover", "-", offsetof(met_t, cc), 1. },
  };
  const int nsurface = (int) (sizeof(surface) / sizeof(surface[0])), nlevel = (int) (sizeof(level) / sizeof(level[0]));
  const int lonlat = met->coord_type == 0;
  nccw_file *w = ncw_create(filename);
  int tid, xid, yid, lid;
  NCW(tid = nccw_def_dim(w, "time", 1));
  NCW(xid = nccw_def_dim(w, lonlat ? "lon" : "x", met->nx));
  NCW(yid = nccw_def_dim(w, lonlat ? "lat" : "y", met->ny));
  ncw_var(w, lonlat ? "lon" : "x", NCC_DOUBLE, 1, &xid, lonlat ? "longitude" : "x", lonlat ? "degrees_east" : "easting");
  ncw_var(w, lonlat ? "lat" : "y", NCC_DOUBLE, 1, &yid, lonlat ? "latitude" : "y", lonlat ? "degrees_north" : "northing");
  NCW(lid = nccw_def_dim(w, "lev", met->np));
  ncw_var(w, "time", NCC_DOUBLE, 1, &tid, "time", "seconds since 2000-01-01 00:00:00 UTC");
  ncw_var(w, "lev", NCC_DOUBLE, 1, &lid, "pressure", "Pa");
  const int dims2[3] = { tid, yid, xid }, dims3[4] = { tid, lid, yid, xid };
  for (int k = 0; k < nsurface; k++)
    ncw_var(w, surface[k].name, NCC_FLOAT, 3, dims2, surface[k].longname, surface[k].units);
  for (int k = 0; k < nlevel; k++)
    ncw_var(w, level[k].name, NCC_FLOAT, 4, dims3, level[k].longname, level[k].units);
  NCW(nccw_enddef(w));
  ncw_put(w, "time", 0, &met->time);
  ncw_put(w, lonlat ? "lon" : "x", 0, met->lon);
  ncw_put(w, lonlat ? "lat" : "y", 0, met->lat);
  double pa[EP];
  for (int k = 0; k < met->np; k++)
    pa[k] = 100. * met->p[k];
  ncw_put(w, "lev", 0, pa);
  double *turned;
  ALLOC(turned, double, (size_t) met->nx * (size_t) met->ny * (size_t) met->np);
  for (int k = 0; k < nsurface; k++) {   /* [x][y] of met_t -> [y][x]; scaled in single precision like the stored value */
    const float (*f)[EY] = (const float (*)[EY]) ((const char *) met + surface[k].offset);
    EACH_COLUMN(met, i, j)
      turned[(size_t) j * (size_t) met->nx + (size_t) i] = (double) ((float) surface[k].scale * f[i][j]);
    LOG(2, "Write 2-D variable: %s (netCDF)", surface[k].name);
    ncw_put(w, surface[k].name, 0, turned);
  }
  for (int k = 0; k < nlevel; k++) {
    const float (*f)[EY][EP] = (const float (*)[EY][EP]) ((const char *) met + level[k].offset);
    EACH_COLUMN(met, i, j)
      for (int l = 0; l < met->np; l++)
        turned[((size_t) l * (size_t) met->ny + (size_t) j) * (size_t) met->nx + (size_t) i] =
          (double) ((float) level[k].scale * f[i][j][l]);
    LOG(2, "Write 3-D variable: %s (netCDF)", level[k].name);
    ncw_put(w, level[k].name, 0, turned);
  }
  free(turned);
  NCW(nccw_close(w));
}

/* -------------------------------------------------------------------------- */
/* meteo I/O: netCDF (MET_TYPE 0), classic format, for grids that need no       */
/* preprocessing                                                               */
/* -------------------------------------------------------------------------- */

/* The reference's netCDF reader (read_met_nc_grid / _surface / _levels, mptrac.c:9638-10250) is followed by its
 * meteo preprocessing (geopotential heights, boundary layer, tropopause, cloud and CAPE diagnostics, polar
 * winds, periodic column, ...; mptrac.c:7770-7830), which is outside this repository's scope.  What is
 * read here is what needs none of that: Cartesian / UTM grids (MET_COORD_TYPE 1, the reference's
 * tests/coord_test) with the fields stored in the file.  Rules restated from the reference's reader:
 *   - time from the file name (<metbase>_YYYY_MM_DD_HH.nc), axes x / y in metres, levels in Pa -> hPa, descending;
 *   - a field is looked up under alternative names, layout [time][level][y][x], first time;
 *   - value = scl * stored (single precision); packed shorts: scl * (stored * scale_factor + add_offset);
 *     _FillValue, missing_value (when non-zero) and |value| >= 1e14 give NaN;
 *   - read_met_extrapolate (mptrac.c:9470-9506): levels at and below the lowest one where t, u, v or w is
 *     not finite are filled from the level above. */

typedef struct {
  ncc_file *nc;
  const ctl_t *ctl;
  const met_t *met;
  int nc_scale;
} nc_reader;

static int nc_lookup(const nc_reader *r, const char *const *names) {
  for (int k = 0; names[k]; k++) {
    const int v = ncc_find_var(r->nc, names[k]);
    if (v >= 0)
      return v;
  }
  return -1;
}

/* one level field ([np][ny][nx] in the file) or surface field (nlev = 0: [ny][nx]) into dest[(ix * EY + iy) * stride + ip] */
static int nc_field(const nc_reader *r, const char *const *names, const int nlev, const float scl, float *dest,
                    const size_t stride) {
  const int var = nc_lookup(r, names);
  if (var < 0)
    return 0;
  const met_t *met = r->met;
  const int nd = ncc_var_ndims(r->nc, var), lev = nlev > 0 ? nlev : 1;
  const long long n = (long long) lev * met->ny * met->nx;
  /* [time] [level] y x: check the trailing extents */
  if (nd < (nlev > 0 ? 3 : 2) || ncc_var_dim(r->nc, var, nd - 1, NULL) != met->nx
      || ncc_var_dim(r->nc, var, nd - 2, NULL) != met->ny || (nlev > 0 && ncc_var_dim(r->nc, var, nd - 3, NULL) != nlev))
    ERRMSG("Meteo field has unexpected dimensions!");
  double fill = 0, miss = 0, scale = 1, offset = 0;
  const int has_fill = ncc_get_att(r->nc, var, "_FillValue", &fill);
  const int has_miss = ncc_get_att(r->nc, var, "missing_value", &miss);
  const int packed = r->nc_scale && ncc_get_att(r->nc, var, "add_offset", &offset)
    && ncc_get_att(r->nc, var, "scale_factor", &scale);
  float *help;
  ALLOC(help, float, (size_t) n);
  if (packed) {
    short *raw;
    ALLOC(raw, short, (size_t) n);
    if (!ncc_read_short(r->nc, var, 0, 0, n, raw))
      ERRMSG("netCDF: %s", ncc_error(r->nc));
    const short fillval = has_fill ? (short) fill : 0, missval = has_miss ? (short) miss : 0;
    const float scalfac = (float) scale, off = (float) offset;
    for (long long i = 0; i < n; i++) {
      const float v = raw[i] * scalfac + off;
      help[i] = ((fillval == 0 || raw[i] != fillval) && (missval == 0 || raw[i] != missval) && fabsf(v) < 1e14f)
        ? scl * v : NAN;
    }
    free(raw);
  } else {
    if (!ncc_read_float(r->nc, var, 0, 0, n, help))
      ERRMSG("netCDF: %s", ncc_error(r->nc));
    const float fillval = has_fill ? (float) fill : 0.f, missval = has_miss ? (float) miss : 0.f;
    for (long long i = 0; i < n; i++) {
      const float v = help[i];
      help[i] = ((fillval == 0 || v != fillval) && (missval == 0 || v != missval) && fabsf(v) < 1e14f) ? scl * v : NAN;
    }
  }
  /* file order [level][y][x] -> met_t order [x][y][level] */
  EACH_COLUMN(met, i, j)
    for (int k = 0; k < lev; k++)
      dest[((size_t) i * EY + (size_t) j) * stride + (size_t) k] = help[((size_t) k * met->ny + j) * met->nx + i];
  free(help);
  return 1;
}

#define NAMES(...) ((const char *const[]) { __VA_ARGS__, NULL })
#define NC_3D(field, scl, ...) nc_field(&r, NAMES(__VA_ARGS__), met->np, scl, &met->field[0][0][0], EP)
#define NC_2D(field, scl, ...) nc_field(&r, NAMES(__VA_ARGS__), 0, scl, &met->field[0][0], 1)

/* The wind at a pole has no direction of its own.  On a grid that reaches both poles (read_met_polar_winds,
 * mptrac.c:11775-11833) each pole row gets the mean wind vector of the row next to it: averaged in a frame fixed
 * at the pole (every longitude's (u, v) turned by its longitude), then turned back into the local directions. */
static void pole_row_from_neighbour(met_t *met, const int pole, const int next, const double *c, const double *s) {
  const int nx = met->nx;
  for (int ip = 0; ip < met->np; ip++) {
    double mx = 0, my = 0;   /* mean vector in the polar frame */
    for (int ix = 0; ix < nx; ix++) {
      const double u = met->u[ix][next][ip], v = met->v[ix][next][ip];
      mx += (u * c[ix] - v * s[ix]) / nx;
      my += (u * s[ix] + v * c[ix]) / nx;
    }
    for (int ix = 0; ix < nx; ix++) {
      met->u[ix][pole][ip] = (float) (mx * c[ix] + my * s[ix]);
      met->v[ix][pole][ip] = (float) (my * c[ix] - mx * s[ix]);
    }
  }
}

static void met_polar_winds(met_t *met) {
  const int last = met->ny - 1;
  if (fabs(met->lat[0]) < 89.999 || fabs(met->lat[last]) < 89.999)
    return;
  double *c, *s;
  ALLOC(c, double, met->nx);
  ALLOC(s, double, met->nx);
  const int rows[2][2] = { { 0, 1 }, { last, last - 1 } };   /* { pole row, its neighbour } */
  for (int k = 0; k < 2; k++) {
    /* the frame turns with the longitude at the north pole, against it at the south pole */
    const double turn = met->lat[rows[k][0]] < 0 ? -1 : 1;
    for (int ix = 0; ix < met->nx; ix++) {
      c[ix] = cos(turn * DEG2RAD(met->lon[ix]));
      s[ix] = sin(turn * DEG2RAD(met->lon[ix]));
    }
    pole_row_from_neighbour(met, rows[k][0], rows[k][1], c, s);
  }
  free(c);
  free(s);
}

/* read_met_periodic (mptrac.c:11714-11771): a global grid gets one more longitude, a copy of the first column */
static void met_periodic(met_t *met) {
  if (!(fabs(met->lon[met->nx - 1] - met->lon[0] + met->lon[1] - met->lon[0] - 360) < 0.01))
    return;
  if ((++met->nx) >= EX)
    ERRMSG("Cannot create periodic boundary conditions!");
  const int last = met->nx - 1;
  met->lon[last] = met->lon[last - 1] + met->lon[1] - met->lon[0];
  float (*f2[])[EY] = { met->ps, met->zs, met->ts, met->us, met->vs, met->ess, met->nss, met->shf, met->lsm, met->sst,
    met->pbl, met->cape, met->cin };
  float (*f3[])[EY][EP] = { met->t, met->u, met->v, met->w, met->h2o, met->o3, met->lwc, met->rwc, met->iwc, met->swc,
    met->cc };
  for (size_t f = 0; f < sizeof(f2) / sizeof(f2[0]); f++)
    memcpy(f2[f][last], f2[f][0], sizeof(f2[f][0]));
  for (size_t f = 0; f < sizeof(f3) / sizeof(f3[0]); f++)
    memcpy(f3[f][last], f3[f][0], sizeof(f3[f][0]));
}

static int read_met_nc(const char *filename, const ctl_t *ctl, met_t *met) {
  char err[256];
  ncc_file *nc = ncc_open(filename, err, sizeof(err));
  if (!nc) {
    if (strcmp(err, "cannot open file") == 0) {
      WARN("Cannot open file!");
      return 0;
    }
    ERRMSG("%s: %s", filename, err);
  }
  met->coord_type = ctl->met_coord_type;
  nc_reader r = { nc, ctl, met, ctl->met_nc_scale };

  /* time from the file name: ..._YYYY_MM_DD_HH.nc */
  const size_t len = strlen(filename);
  int year, mon, day, hour;
  if (len < 16 || sscanf(filename + len - 16, "%4d_%2d_%2d_%2d", &year, &mon, &day, &hour) != 4 || year < 1900
      || year > 2100 || mon < 1 || mon > 12 || day < 1 || day > 31 || hour < 0 || hour > 23)
    ERRMSG("Cannot read time from filename!");
  time2jsec(year, mon, day, hour, 0, 0, 0, &met->time);

  /* axes */
  long long nx, ny;
  const char *xname = ctl->met_coord_type == 0 ? "lon" : "x", *yname = ctl->met_coord_type == 0 ? "lat" : "y";
  if (ncc_find_dim(nc, xname, &nx) < 0 || ncc_find_dim(nc, yname, &ny) < 0)
    ERRMSG("Cannot read netCDF dimension %s / %s!", xname, yname);
  if (nx < 2 || nx >= EX || ny < 2 || ny > EY)      /* (one column is kept for the periodic boundary) */
    ERRMSG("Dimension %s / %s is out of range!", xname, yname);
  met->nx = (int) nx;
  met->ny = (int) ny;
  const int vx = ncc_find_var(nc, xname), vy = ncc_find_var(nc, yname);
  if (vx < 0 || vy < 0 || !ncc_read_double(nc, vx, 0, 0, nx, met->lon) || !ncc_read_double(nc, vy, 0, 0, ny, met->lat))
    ERRMSG("Cannot read the %s / %s coordinates!", xname, yname);
  const int vu = nc_lookup(&r, NAMES("u", "U"));
  if (vu < 0)
    ERRMSG("Variable 'u' or 'U' not found, cannot determine vertical dimension!");
  const int nd = ncc_var_ndims(nc, vu);
  if (nd != 3 && nd != 4)
    ERRMSG("Cannot determine vertical dimension!");
  const char *levname;
  const long long np = ncc_var_dim(nc, vu, nd == 4 ? 1 : 0, &levname);
  if (np < 2 || np > EP)
    ERRMSG("Number of levels out of range!");
  met->np = (int) np;
  const int vl = ncc_find_var(nc, levname);
  if (vl < 0 || !ncc_read_double(nc, vl, 0, 0, np, met->p))
    ERRMSG("Cannot read the pressure levels!");
  /* (the level axis is the one `u` names: a reader that had to guess axis names by their length must not have
   * guessed a horizontal one, and what was read must look like pressures) */
  if (strcmp(levname, xname) == 0 || strcmp(levname, yname) == 0)
    ERRMSG("Cannot determine vertical dimension!");
  for (int k = 0; k < met->np; k++) {
    met->p[k] /= 100.0;   /* Pa -> hPa */
    REQUIRE(isfinite(met->p[k]) && met->p[k] > 0 && met->p[k] < 2000, "The levels of '%s' are not pressures!", levname);
  }
  const double dx = fabs(met->lon[1] - met->lon[0]);
  for (int i = 2; i < met->nx; i++)
    REQUIRE(fabs(fabs(met->lon[i] - met->lon[i - 1]) - dx) <= 0.001, "No regular grid spacing in longitudes!");
  LOG(2, "Grid: %d x %d x %d, %g ... %g hPa", met->nx, met->ny, met->np, met->p[0], met->p[met->np - 1]);

  /* surface fields */
  if (NC_2D(ps, 1.0f, "lnsp", "LNSP")) {   /* logarithm of the surface pressure in Pa */
    EACH_COLUMN(met, i, j)
      met->ps[i][j] = (float) (exp(met->ps[i][j]) / 100.);
  } else if (!NC_2D(ps, 0.01f, "ps", "PS", "sp", "SP")) {
    WARN("Cannot not read surface pressure data (use lowest level)!");
    EACH_COLUMN(met, i, j)
      met->ps[i][j] = (float) met->p[0];
  }
  (void) NC_2D(zs, (float) (1. / (1000. * 9.80665)), "z", "Z");
  (void) NC_2D(ts, 1.0f, "t2m", "T2M", "2t", "2T", "t2", "T2");
  (void) NC_2D(us, 1.0f, "u10m", "U10M", "10u", "10U", "u10", "U10");
  (void) NC_2D(vs, 1.0f, "v10m", "V10M", "10v", "10V", "v10", "V10");
  (void) NC_2D(ess, 1.0f, "iews", "IEWS");
  (void) NC_2D(nss, 1.0f, "inss", "INSS");
  (void) NC_2D(shf, 1.0f, "ishf", "ISHF");
  (void) NC_2D(lsm, 1.0f, "lsm", "LSM");
  (void) NC_2D(sst, 1.0f, "sstk", "SSTK", "sst", "SST");
  const int have_pbl = ctl->met_pbl == 0 && NC_2D(pbl, 0.01f, "blp", "BLP");
  const int have_cape = ctl->met_cape == 0 && NC_2D(cape, 1.0f, "cape", "CAPE") && NC_2D(cin, 1.0f, "cin", "CIN");

  /* level fields */
  REQUIRE(NC_3D(t, 1.0f, "t", "T", "temp", "TEMP"), "Cannot read temperature!");
  REQUIRE(NC_3D(u, 1.0f, "u", "U"), "Cannot read zonal wind!");
  REQUIRE(NC_3D(v, 1.0f, "v", "V"), "Cannot read meridional wind!");
  if (!NC_3D(w, 0.01f, "w", "W", "omega", "OMEGA"))   /* Pa/s -> hPa/s */
    WARN("Cannot read vertical velocity!");
  if (!NC_3D(h2o, (float) (MA / MH2O), "q", "Q", "sh", "SH"))   /* mass -> volume mixing ratio */
    WARN("Cannot read specific humidity!");
  if (!NC_3D(o3, (float) (MA / MO3), "o3", "O3"))
    WARN("Cannot read ozone data!");
  const int have_cloud = NC_3D(lwc, 1.0f, "clwc", "CLWC") & NC_3D(rwc, 1.0f, "crwc", "CRWC")
    & NC_3D(iwc, 1.0f, "ciwc", "CIWC") & NC_3D(swc, 1.0f, "cswc", "CSWC");
  (void) NC_3D(cc, 1.0f, "cc", "CC");
  ncc_close(nc);
  for (int k = 1; k < met->np; k++)
    REQUIRE(met->p[k - 1] >= met->p[k], "Pressure levels must be descending!");

  /* below the lowest level with valid t, u, v, w every level field continues with that level's value
   * (reference: read_met_extrapolate) */
  EACH_COLUMN(met, i, j) {
    int low = met->np - 1;
    while (low >= 0 && isfinite(met->t[i][j][low]) && isfinite(met->u[i][j][low]) && isfinite(met->v[i][j][low])
           && isfinite(met->w[i][j][low]))
      low--;
    float (*f3[11])[EY][EP] = { met->t, met->u, met->v, met->w, met->h2o, met->o3, met->lwc, met->rwc, met->iwc,
      met->swc, met->cc };
    for (int k = low; k >= 0; k--)
      for (int f = 0; f < 11; f++)
        f3[f][i][j][k] = k + 1 < EP ? f3[f][i][j][k + 1] : 0.f;
  }

  if (met->coord_type == 0) {
    met_polar_winds(met);
    met_periodic(met);
  }

  /* Fields the reference derives in its preprocessing.  The boundary-layer pressure enters module_diff_turb
   * only through weights that multiply TURB_DX_PBL / TURB_DZ_PBL against TURB_DX_TROP / TURB_DZ_TROP: with
   * equal values (the defaults) and none of the other consumers active any finite value below the
   * tropopause gives the reference's result, and surface pressure - 100 hPa is used.  Everything else that
   * would need a derived field is refused. */
  if (!have_pbl) {
    if ((ctl->diffusion && (ctl->turb_dx_pbl != ctl->turb_dx_trop || ctl->turb_dz_pbl != ctl->turb_dz_trop
                            || ctl->turb_pbl_scheme != 0))
        || ctl->conv_mix_pbl || ctl->bound_pbl || ctl->qnt_pbl >= 0)
      ERRMSG("This configuration uses the boundary-layer pressure, which the reference derives in its meteo "
             "preprocessing (not provided): supply it in the file (MET_PBL 0, variable blp) or use MET_TYPE 1 files!");
    EACH_COLUMN(met, i, j)
      met->pbl[i][j] = met->ps[i][j] - 100.f;
  }
  (void) have_cloud;
  if (ctl->conv_cape >= 0)
    ERRMSG("CONV_CAPE needs the equilibrium level from the reference's meteo preprocessing (not provided)!");
  if (ctl->wet_depo_ic_a > 0 || ctl->wet_depo_ic_h[0] > 0)
    ERRMSG("Wet deposition needs the cloud diagnostics (pct, pcb, cl) of the reference's meteo preprocessing (not provided)!");
  static const char *const derived[] = { "pt", "tt", "zt", "h2ot", "zg", "pv", "pct", "pcb", "cl", "plcl", "plfc", "pel",
    "cape", "cin", "o3c", NULL };
  for (int iq = 0; iq < ctl->nq; iq++)
    for (int k = 0; derived[k]; k++)
      if (strcasecmp(ctl->qnt_name[iq], derived[k]) == 0 && !(have_cape && (k == 12 || k == 13)))
        ERRMSG("Quantity %s comes from the reference's meteo preprocessing, which netCDF input does not get here!",
               ctl->qnt_name[iq]);
  return 1;
}

static void get_met_filename(const ctl_t *ctl, const double t, const int direct, char *filename) {
  /* <metbase>_YYYY_MM_DD_HH.bin on the DT_MET raster (mptrac.c:2620-2700) */
  double t6, r;
  int year, mon, day, hour, min, sec;
  if (direct == -1)
    t6 = floor(t / ctl->dt_met) * ctl->dt_met;
  else
    t6 = ceil(t / ctl->dt_met) * ctl->dt_met;
  jsec2time(t6, &year, &mon, &day, &hour, &min, &sec, &r);
  sprintf(filename, "%s_%d_%02d_%02d_%02d.%s", ctl->metbase, year, mon, day, hour, ctl->met_type == 0 ? "nc" : "bin");
}

/* -------------------------------------------------------------------------- */
/* device mirrors                                                             */
/* -------------------------------------------------------------------------- */

static void to_device_ctl(const ctl_t *c, mphip_ctl_t *d) {
  memset(d, 0, sizeof(*d));
#define TAKE(f) d->f = c->f
  TAKE(direction); TAKE(met_coord_type); TAKE(t_start); TAKE(t_stop); TAKE(dt_mod); TAKE(dt_met); TAKE(met_utm_ref_lat);
  TAKE(met_utm_ref_lon); TAKE(oh_chem_beta);
  TAKE(nq); TAKE(qnt_m); TAKE(qnt_vmr); TAKE(qnt_rp); TAKE(qnt_rhop); TAKE(qnt_ens); TAKE(qnt_loss_rate);
  TAKE(qnt_mloss_decay); TAKE(qnt_mloss_wet); TAKE(qnt_mloss_dry); TAKE(nens); TAKE(advect); TAKE(advect_vert_coord);
  TAKE(rng_type); TAKE(diffusion); TAKE(turb_pbl_scheme); TAKE(conv_mix_pbl);
  TAKE(turb_dx_pbl); TAKE(turb_dx_trop); TAKE(turb_dx_strat); TAKE(turb_dz_pbl); TAKE(turb_dz_trop); TAKE(turb_dz_strat);
  TAKE(turb_mesox); TAKE(turb_mesoz); TAKE(turb_pbl_trans); TAKE(conv_pbl_trans); TAKE(conv_cape); TAKE(conv_cin);
  TAKE(conv_dt); TAKE(sort_dt); TAKE(tdec_trop); TAKE(tdec_strat); TAKE(mixing_dt); TAKE(mixing_trop); TAKE(mixing_strat);
  TAKE(mixing_z0); TAKE(mixing_z1); TAKE(mixing_lon0); TAKE(mixing_lon1); TAKE(mixing_lat0); TAKE(mixing_lat1);
  TAKE(mixing_nx); TAKE(mixing_ny); TAKE(mixing_nz);
  TAKE(wet_depo_ic_a); TAKE(wet_depo_ic_b); TAKE(wet_depo_bc_a); TAKE(wet_depo_bc_b); TAKE(wet_depo_so2_ph);
  TAKE(wet_depo_ic_ret_ratio); TAKE(wet_depo_bc_ret_ratio); TAKE(dry_depo_vdep); TAKE(dry_depo_dp);
  TAKE(grid_z0); TAKE(grid_z1); TAKE(grid_lon0); TAKE(grid_lon1); TAKE(grid_lat0); TAKE(grid_lat1);
  TAKE(grid_nx); TAKE(grid_ny); TAKE(grid_nz);
#undef TAKE
  d->qnt_zeta = c->qnt_zeta;
  d->qnt_eta = c->qnt_eta;
  d->met_dt_out = c->met_dt_out;
  d->qnt_aoa = c->qnt_aoa;
  d->qnt_tracer[MPHIP_TR_CCL4] = c->qnt_Cccl4;
  d->qnt_tracer[MPHIP_TR_CCL3F] = c->qnt_Cccl3f;
  d->qnt_tracer[MPHIP_TR_CCL2F2] = c->qnt_Cccl2f2;
  d->qnt_tracer[MPHIP_TR_N2O] = c->qnt_Cn2o;
  d->qnt_tracer[MPHIP_TR_SF6] = c->qnt_Csf6;
  d->isosurf = c->isosurf;
  d->bound_pbl = c->bound_pbl;
  d->bound_mass = c->bound_mass;
  d->bound_mass_trend = c->bound_mass_trend;
  d->bound_vmr = c->bound_vmr;
  d->bound_vmr_trend = c->bound_vmr_trend;
  d->bound_lat0 = c->bound_lat0;
  d->bound_lat1 = c->bound_lat1;
  d->bound_p0 = c->bound_p0;
  d->bound_p1 = c->bound_p1;
  d->bound_dps = c->bound_dps;
  d->bound_dzs = c->bound_dzs;
  d->bound_zetas = c->bound_zetas;
  g_isosurf = c->isosurf;
  int k_mq = 0;
#define X(n, u) d->qnt_met[k_mq++] = c->qnt_##n;
  MPTRAC_METEO_QNT(X)
#undef X
  g_meteo_fields = 0;
  for (int k = 0; k < MPHIP_NMQ; k++)
    if (d->qnt_met[k] >= 0)
      g_meteo_fields = 1;
  for (int k = 0; k < 2; k++) {
    d->wet_depo_pre[k] = c->wet_depo_pre[k];
    d->wet_depo_ic_h[k] = c->wet_depo_ic_h[k];
    d->wet_depo_bc_h[k] = c->wet_depo_bc_h[k];
  }
}

static void need_ctx(const ctl_t *ctl) {
  if (g_ctx)
    return;
  if (mphip_create(&g_ctx, ctl ? ctl->hip_device : 0) != 0)
    ERRMSG("Cannot initialise the HIP device (this build has no CPU path)!");
  if (ctl)
    HIP(mphip_set_option(g_ctx, "locality_sort_interval", ctl->hip_locality_interval));
  /* atm_t lives from mptrac_alloc to mptrac_free: page-lock its arrays for the particle transfers */
  HIP(mphip_set_option(g_ctx, "pin_host_atm", 1));
}

/* Device copies of the meteo snapshots are keyed on the host addresses, as
 * OpenACC's present table does for the reference: when the caller has swapped
 * its met0/met1 pointers (mptrac_get_met, mptrac.c:6488-6491) the device slots
 * are swapped, not re-uploaded. */
static void map_met_slot(const met_t *host, int slot) {
  if (g_met_host[slot] == host)
    return;
  if (g_met_host[1 - slot] == host) {
    HIP(mphip_swap_met(g_ctx));
    const met_t *tmp = g_met_host[0];
    g_met_host[0] = g_met_host[1];
    g_met_host[1] = tmp;
    return;
  }
  g_met_host[slot] = host;
}

static void describe_met(met_t *met, mphip_met_t *out) {
  mphip_met_t m;
  memset(&m, 0, sizeof(m));
  m.time = met->time;
  m.coord_type = met->coord_type;
  m.nx = met->nx;
  m.ny = met->ny;
  m.np = met->np;
  m.lon = met->lon;
  m.lat = met->lat;
  m.p = met->p;
  m.sx = (long long) EY * EP;
  m.sy = EP;
  m.sx2 = EY;
  m.npl = met->npl;
  m.sx_ml = m.sx;
  m.sy_ml = m.sy;
  if (met->npl > 0) {
    m.f3[MPHIP_PL] = &met->pl[0][0][0];
    m.f3[MPHIP_UL] = &met->ul[0][0][0];
    m.f3[MPHIP_VL] = &met->vl[0][0][0];
    m.f3[MPHIP_ZETAL] = &met->zetal[0][0][0];
    m.f3[MPHIP_ZETA_DOTL] = &met->zeta_dotl[0][0][0];
  }
  m.f3[MPHIP_U] = &met->u[0][0][0];
  m.f3[MPHIP_V] = &met->v[0][0][0];
  m.f3[MPHIP_W] = &met->w[0][0][0];
  m.f3[MPHIP_T] = &met->t[0][0][0];
  m.f3[MPHIP_LWC] = &met->lwc[0][0][0];
  m.f3[MPHIP_RWC] = &met->rwc[0][0][0];
  m.f3[MPHIP_IWC] = &met->iwc[0][0][0];
  m.f3[MPHIP_SWC] = &met->swc[0][0][0];
  m.f2[MPHIP_PS] = &met->ps[0][0];
  m.f2[MPHIP_PBL] = &met->pbl[0][0];
  m.f2[MPHIP_CAPE] = &met->cape[0][0];
  m.f2[MPHIP_CIN] = &met->cin[0][0];
  m.f2[MPHIP_PEL] = &met->pel[0][0];
  m.f2[MPHIP_PCT] = &met->pct[0][0];
  m.f2[MPHIP_PCB] = &met->pcb[0][0];
  m.f2[MPHIP_CL] = &met->cl[0][0];
  m.f2[MPHIP_ESS] = &met->ess[0][0];
  m.f2[MPHIP_NSS] = &met->nss[0][0];
  m.f2[MPHIP_SHF] = &met->shf[0][0];
  m.f3[MPHIP_H2O] = &met->h2o[0][0][0];
  /* module_meteo-only fields: uploaded when a quantity it fills was requested */
  if (g_meteo_fields) {
    m.f3[MPHIP_Z] = &met->z[0][0][0];
    m.f3[MPHIP_PV] = &met->pv[0][0][0];
    m.f3[MPHIP_O3] = &met->o3[0][0][0];
    m.f3[MPHIP_CC] = &met->cc[0][0][0];
    m.f2[MPHIP_TS] = &met->ts[0][0];
    m.f2[MPHIP_ZS] = &met->zs[0][0];
    m.f2[MPHIP_US] = &met->us[0][0];
    m.f2[MPHIP_VS] = &met->vs[0][0];
    m.f2[MPHIP_LSM] = &met->lsm[0][0];
    m.f2[MPHIP_SST] = &met->sst[0][0];
    m.f2[MPHIP_PT] = &met->pt[0][0];
    m.f2[MPHIP_TT] = &met->tt[0][0];
    m.f2[MPHIP_ZT] = &met->zt[0][0];
    m.f2[MPHIP_H2OT] = &met->h2ot[0][0];
    m.f2[MPHIP_PLCL] = &met->plcl[0][0];
    m.f2[MPHIP_PLFC] = &met->plfc[0][0];
    m.f2[MPHIP_O3C] = &met->o3c[0][0];
  }
  *out = m;
}

static void upload_met(met_t *met, int slot) {
  map_met_slot(met, slot);
  mphip_met_t m;
  describe_met(met, &m);
  HIP(mphip_update_met(g_ctx, slot, &m));
}

/* Read-ahead of the next meteo file (HIP_MET_PREFETCH 1, forward runs): while the time steps of the current
 * interval run, a reader thread loads the file after met1 into a third met_t.  The thread touches nothing but
 * that buffer; the stepping thread notices that the file has arrived (poll_read_ahead, called once per time
 * step), starts its upload on the back end's copy stream (mphip_prefetch_met) and, at the hand-over,
 * mptrac_get_met rotates the three host pointers and commits instead of reading and uploading on the stepping
 * path -- so every call into the back end comes from one thread.  Costs one more met_t of host memory, hence
 * off by default. */

static void *read_ahead_main(void *arg) {
  (void) arg;
  FILE *probe = fopen(g_ahead.filename, "r");   /* past the last file of the run: nothing to do */
  if (probe) {
    fclose(probe);
    if (mptrac_read_met(g_ahead.filename, g_ahead.ctl, g_ahead.clim, g_ahead.met, NULL))
      g_ahead.file_ok = 1;
  }
  __atomic_store_n(&g_ahead.done, 1, __ATOMIC_RELEASE);
  return NULL;
}

static void start_read_ahead(ctl_t *ctl, clim_t *clim, const met_t *met1) {
  if (!ctl->hip_met_prefetch || ctl->direction != 1 || g_ahead.active)
    return;
  if (!g_ahead.met)
    ALLOC(g_ahead.met, met_t, 1);
  g_ahead.ctl = ctl;
  g_ahead.clim = clim;
  g_ahead.file_ok = g_ahead.ok = 0;
  g_ahead.done = 0;
  get_met_filename(ctl, met1->time + 1, 1, g_ahead.filename);
  if (pthread_create(&g_ahead.thread, NULL, read_ahead_main, NULL) == 0)
    g_ahead.active = 1;
}

/* the reader has finished: join it and hand the snapshot to the back end's copy stream (stepping thread) */
static void poll_read_ahead(int wait) {
  if (!g_ahead.active || g_ahead.joined)
    return;
  if (!wait && !__atomic_load_n(&g_ahead.done, __ATOMIC_ACQUIRE))
    return;
  pthread_join(g_ahead.thread, NULL);
  g_ahead.joined = 1;
  if (g_ahead.file_ok) {
    mphip_met_t m;
    describe_met(g_ahead.met, &m);
    flush_steps();
    if (mphip_prefetch_met(g_ctx, &m) != 0) {
      WARN("Meteo read-ahead: %s", mphip_last_error(g_ctx));
    } else
      g_ahead.ok = 1;
  }
}

static void cancel_read_ahead(void) {
  if (!g_ahead.active)
    return;
  poll_read_ahead(1);
  g_ahead.active = g_ahead.joined = 0;
  if (g_ahead.ok)
    HIP(mphip_discard_prefetch(g_ctx));
}

/* 1: the prefetched snapshot is the one `filename` names and now is *met1 */
static int take_read_ahead(const char *filename, met_t **met0, met_t **met1) {
  if (!g_ahead.active)
    return 0;
  poll_read_ahead(1);
  g_ahead.active = g_ahead.joined = 0;
  if (!g_ahead.ok)
    return 0;
  if (strcmp(filename, g_ahead.filename) != 0) {   /* not the file this hand-over needs */
    HIP(mphip_discard_prefetch(g_ctx));
    return 0;
  }
  met_t *old0 = *met0;
  *met0 = *met1;
  *met1 = g_ahead.met;
  g_ahead.met = old0;
  HIP(mphip_commit_met(g_ctx));
  LOG(1, "Meteo data from the read-ahead: %s", filename);
  g_met_host[0] = *met0;
  g_met_host[1] = *met1;
  return 1;
}

void mptrac_update_device(const ctl_t *ctl, const cache_t *cache, const clim_t *clim, met_t **met0,
                          met_t **met1, const atm_t *atm) {
  /* NULL = skip, as the reference (mptrac.c:8005-8057) */
  need_ctx(ctl);
  if (ctl) {
    mphip_ctl_t d;
    to_device_ctl(ctl, &d);
    HIP(mphip_update_ctl(g_ctx, &d));
    g_nq = ctl->nq;
  }
  if (clim) {
    HIP(mphip_update_clim(g_ctx, clim->tropo_ntime, clim->tropo_nlat, clim->tropo_time, clim->tropo_lat,
                          &clim->tropo[0][0], 73));
    const clim_zm_t *zm[MPHIP_NZM] = { &clim->hno3, &clim->oh, &clim->h2o2, &clim->ho2, &clim->o1d };
    for (int k = 0; k < MPHIP_NZM; k++) {   /* compact copies of the tables that were read */
      double *v = NULL;
      if (zm[k]->ntime > 0) {
        ALLOC(v, double, (size_t) zm[k]->ntime * (size_t) zm[k]->np * (size_t) zm[k]->nlat);
        size_t n = 0;
        for (int it = 0; it < zm[k]->ntime; it++)
          for (int iz = 0; iz < zm[k]->np; iz++)
            for (int iy = 0; iy < zm[k]->nlat; iy++)
              v[n++] = zm[k]->vmr[it][iz][iy];
      }
      HIP(mphip_update_clim_zm(g_ctx, k, zm[k]->ntime, zm[k]->np, zm[k]->nlat, zm[k]->time, zm[k]->p, zm[k]->lat, v));
      free(v);
    }
    const clim_ts_t *ts[MPHIP_NTR] = { &clim->ccl4, &clim->ccl3f, &clim->ccl2f2, &clim->n2o, &clim->sf6 };
    for (int k = 0; k < MPHIP_NTR; k++)
      HIP(mphip_update_clim_ts(g_ctx, k, ts[k]->ntime, ts[k]->time, ts[k]->vmr));
  }
  if (met0)
    upload_met(*met0, 0);
  if (met1)
    upload_met(*met1, 1);
  if (atm) {
    const double *q[MPHIP_NQ_MAX] = { 0 };
    for (int iq = 0; iq < g_nq; iq++)
      q[iq] = atm->q[iq];
    HIP(mphip_update_atm(g_ctx, atm->np, g_ip0, g_np_total >= 0 ? g_np_total : atm->np, g_nq, atm->time, atm->p,
                         atm->lon, atm->lat, q));
  }
  if (cache) {
    HIP(mphip_update_cache(g_ctx, &cache->uvwp[0][0], NULL));
    if (g_isosurf >= 1 && g_isosurf <= 3)
      HIP(mphip_update_iso(g_ctx, cache->iso_var, NULL, NULL, 0));
    if (cache->iso_n > 0)
      HIP(mphip_update_iso(g_ctx, NULL, cache->iso_ts, cache->iso_ps, cache->iso_n));
  }
}

void mptrac_update_host(const ctl_t *ctl, const cache_t *cache, const clim_t *clim, met_t **met0,
                        met_t **met1, const atm_t *atm) {
  /* only atm and cache change on the device (mptrac.c:8061-8113) */
  (void) ctl;
  (void) clim;
  (void) met0;
  (void) met1;
  if (!g_ctx)
    return;
  if (atm) {
    atm_t *a = (atm_t *) atm;   /* the reference's signature is const; the data are refreshed */
    double *q[MPHIP_NQ_MAX] = { 0 };
    for (int iq = 0; iq < g_nq; iq++)
      q[iq] = a->q[iq];
    HIP(mphip_get_atm(g_ctx, a->time, a->p, a->lon, a->lat, q));
  }
  if (cache) {
    cache_t *c = (cache_t *) cache;
    HIP(mphip_get_cache(g_ctx, &c->uvwp[0][0], c->dt, NULL));
    if (g_isosurf >= 1 && g_isosurf <= 3)
      (void) mphip_get_iso(g_ctx, c->iso_var);   /* not on the device before the first time step */
  }
}

/* -------------------------------------------------------------------------- */
/* init / meteo handling / time step                                          */
/* -------------------------------------------------------------------------- */

/* Start and stop time of a run (reference interface: src/mptrac.h module_timesteps_init): the run starts at
 * the first release time in the direction of travel, rounded outwards to the DT_MOD raster, and -- unless
 * T_STOP was given -- ends at the last one. */
static void release_time_range(const atm_t *atm, double *first, double *last) {
  *first = *last = atm->time[0];
  for (int ip = 1; ip < atm->np; ip++) {
    *first = fmin(*first, atm->time[ip]);
    *last = fmax(*last, atm->time[ip]);
  }
}

void module_timesteps_init(ctl_t *ctl, const atm_t *atm) {
  double first, last;
  if (g_np_total >= 0) {
    /* one of N processes: every rank must step through the same model times (the all-reduces of
     * module_mixing and of the gridded output pair up step by step), so the range is the one of the whole
     * particle file, not of this rank's index range -- which may even be empty */
    first = g_release_first;
    last = g_release_last;
  } else
    release_time_range(atm, &first, &last);
  const int forward = ctl->direction == 1;
  const double begin = forward ? first : last, end = forward ? last : first;
  if (ctl->t_stop > 1e99)   /* not set in the control file */
    ctl->t_stop = end;
  if (ctl->direction * (ctl->t_stop - begin) <= 0)
    ERRMSG("Nothing to do! Check T_STOP and DIRECTION!");
  ctl->t_start = (forward ? floor(begin / ctl->dt_mod) : ceil(begin / ctl->dt_mod)) * ctl->dt_mod;
}

void mptrac_init(ctl_t *ctl, cache_t *cache, clim_t *clim, atm_t *atm, depo_t *depo, const int ntask) {
  /* mptrac.c:6563-6584; ntask seeds the GSL generators of RNG_TYPE 0 only */
  (void) depo;
  (void) ntask;
  module_timesteps_init(ctl, atm);
  mptrac_update_device(ctl, cache, clim, NULL, NULL, atm);
}

/* one snapshot of the DT_MET raster: the file at or before (side = -1) / at or after (side = +1) time t */
static void load_snapshot(ctl_t *ctl, clim_t *clim, const double t, const int side, met_t *met, dd_t *dd) {
  char filename[LEN];
  get_met_filename(ctl, t, side, filename);
  if (!mptrac_read_met(filename, ctl, clim, met, dd))
    ERRMSG("Cannot open file!");
}

static void swap_snapshots(met_t **a, met_t **b) {
  met_t *tmp = *a;
  *a = *b;
  *b = tmp;
}

/* both snapshots must describe the same grid (axes to 1e-3) */
static void check_same_grid(const met_t *a, const met_t *b) {
  if (a->coord_type != b->coord_type)
    ERRMSG("Coordinate types do not match!");
  if (a->nx == 0 || b->nx == 0)
    return;
  if (a->nx != b->nx || a->ny != b->ny || a->np != b->np)
    ERRMSG("Meteo grid dimensions do not match!");
  const struct {
    const double *x, *y;
    int n;
    const char *what;
  } axes[3] = { { a->lon, b->lon, a->nx, "longitudes" }, { a->lat, b->lat, a->ny, "latitudes" },
    { a->p, b->p, a->np, "pressure levels" } };
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < axes[k].n; i++)
      if (fabs(axes[k].x[i] - axes[k].y[i]) > 0.001)
        ERRMSG("Meteo grid %s do not match!", axes[k].what);
}

/* Keeps met0 / met1 bracketing the model time (reference interface: src/mptrac.h mptrac_get_met): at the start
 * of a run both snapshots are read -- met0 at or before, met1 at or after t on the DT_MET raster, with the
 * start time itself belonging to the interval the run moves into --; when t leaves the interval the snapshot
 * left behind is replaced by the next file in that direction and the two pointers trade places, so that the
 * caller (and the device, which follows the pointers) never re-reads the snapshot that stays. */
void mptrac_get_met(ctl_t *ctl, clim_t *clim, const double t, met_t **met0, met_t **met1, dd_t *dd) {
  static int primed;
  const int forward = ctl->direction == 1;
  if (!primed || t == ctl->t_start) {
    primed = 1;
    cancel_read_ahead();
    load_snapshot(ctl, clim, forward ? t : t - 1, -1, *met0, dd);
    load_snapshot(ctl, clim, forward ? t + 1 : t, +1, *met1, dd);
    mptrac_update_device(NULL, NULL, NULL, met0, met1, NULL);
    start_read_ahead(ctl, clim, *met1);
  }
  if (t > (*met1)->time) {          /* moved past the later snapshot */
    char filename[LEN];
    get_met_filename(ctl, t, +1, filename);
    if (!take_read_ahead(filename, met0, met1)) {
      swap_snapshots(met0, met1);
      load_snapshot(ctl, clim, t, +1, *met1, dd);
      map_met_slot(*met0, 0);       /* the device slots follow the pointers */
      mptrac_update_device(NULL, NULL, NULL, NULL, met1, NULL);
    }
    start_read_ahead(ctl, clim, *met1);
  }
  if (t < (*met0)->time) {          /* moved before the earlier one (backward runs) */
    swap_snapshots(met0, met1);
    load_snapshot(ctl, clim, t, -1, *met0, dd);
    map_met_slot(*met1, 1);
    mptrac_update_device(NULL, NULL, NULL, met0, NULL, NULL);
  }
  check_same_grid(*met0, *met1);
}

void mptrac_run_timestep(ctl_t *ctl, cache_t *cache, clim_t *clim, met_t **met0, met_t **met1, atm_t *atm,
                         depo_t *depo, double t, dd_t *dd) {
  /* The module order and gating of mptrac.c:7851-8001 live in the back end
   * (mphip_run_timestep); the structs stay resident on the device. */
  (void) clim;
  (void) atm;
  (void) depo;
  (void) dd;
  need_ctx(ctl);
  map_met_slot(*met0, 0);
  map_met_slot(*met1, 1);
  poll_read_ahead(0);   /* a meteo file that has arrived starts its upload beside this and the following steps */
  /* module_isosurf_init, ISOSURF 4: read the balloon pressure time series (mptrac.c:4925-4951);
   * modes 1-3 are evaluated on the device */
  if (t == ctl->t_start && ctl->isosurf == 4) {
    LOG(1, "Read balloon pressure data: %s", ctl->balloon);
    cache->iso_n = read_two_columns(ctl->balloon, cache->iso_ts, cache->iso_ps, NP);
    if (cache->iso_n < 0)
      ERRMSG("Cannot open file!");
    if (cache->iso_n < 1)
      ERRMSG("Could not read any data!");
    HIP(mphip_update_iso(g_ctx, NULL, cache->iso_ts, cache->iso_ps, cache->iso_n));
  }
  /* queue the step (see g_steps) */
  if (g_steps.cap < 0) {
    const char *e = getenv("HIP_STEP_BATCH");
    g_steps.cap = e ? atoi(e) : 64;
  }
  if (g_steps.n && (t != g_steps.t_next || g_steps.n >= g_steps.cap))
    flush_steps();
  if (!g_steps.n)
    g_steps.t_first = t;
  g_steps.n++;
  g_steps.t_next = t + ctl->direction * ctl->dt_mod;
  if (g_steps.cap <= 1)
    flush_steps();
}

/* -------------------------------------------------------------------------- */
/* output                                                                     */
/* -------------------------------------------------------------------------- */

/* Temperature at an output cell centre for the implicit volume mixing ratio of
 * write_grid: intpol_met_time_3d(..., init = 1) of the reference
 * (mptrac.c:3112-3137) = intpol_met_space_3d on both snapshots with met0's
 * indices and weights (mptrac.c:2985-3044; intpol_check_lon_lat 2755-2778,
 * locate_irr 3495-3521, locate_reg 3559-3574).  One call per output cell and
 * output time, on the host like the reference; particles are never
 * interpolated here. */
/* interval of a monotonic axis (either direction) that the reference's bisection (locate_irr) selects for x:
 * halve [lo, hi] until the two nodes are neighbours; a node equal to x belongs to the upper interval on an
 * ascending axis and to the lower one on a descending axis */
static int grid_locate_irr(const double *xx, const int n, const double x) {
  int lo = 0, hi = n - 1;
  const int mid0 = (lo + hi) / 2;
  const int ascending = xx[mid0] < xx[mid0 + 1];
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    const int beyond = ascending ? xx[mid] > x : xx[mid] <= x;
    if (beyond)
      hi = mid;
    else
      lo = mid;
  }
  return lo;
}

double mptrac_amd_intpol_3d(const met_t *met0, const met_t *met1, size_t field_offset, const double ts,
                            const double p, const double lon, const double lat) {
  double lon2 = FMOD(lon, 360.);
  if (lon2 < met0->lon[0])
    lon2 += 360;
  else if (lon2 > met0->lon[met0->nx - 1])
    lon2 -= 360;
  double lat2 = lat;
  if (met0->lat[0] < met0->lat[met0->ny - 1])
    lat2 = fmin(fmax(lat, met0->lat[0]), met0->lat[met0->ny - 1]);
  else
    lat2 = fmin(fmax(lat, met0->lat[met0->ny - 1]), met0->lat[0]);
  const int ip = grid_locate_irr(met0->p, met0->np, p);
  int ix = (int) ((lon2 - met0->lon[0]) / (met0->lon[1] - met0->lon[0]));
  ix = ix < 0 ? 0 : (ix > met0->nx - 2 ? met0->nx - 2 : ix);
  const int iy = grid_locate_irr(met0->lat, met0->ny, lat2);
  const double wp = (met0->p[ip + 1] - p) / (met0->p[ip + 1] - met0->p[ip]);
  const double wx = (met0->lon[ix + 1] - lon2) / (met0->lon[ix + 1] - met0->lon[ix]);
  const double wy = (met0->lat[iy + 1] - lat2) / (met0->lat[iy + 1] - met0->lat[iy]);
  double v[2];
  const met_t *mm[2] = { met0, met1 };
  for (int k = 0; k < 2; k++) {
    const float (*f)[EY][EP] = (const float (*)[EY][EP]) ((const char *) mm[k] + field_offset);
    double a00 = wp * (f[ix][iy][ip] - f[ix][iy][ip + 1]) + f[ix][iy][ip + 1];
    const double a01 = wp * (f[ix][iy + 1][ip] - f[ix][iy + 1][ip + 1]) + f[ix][iy + 1][ip + 1];
    double a10 = wp * (f[ix + 1][iy][ip] - f[ix + 1][iy][ip + 1]) + f[ix + 1][iy][ip + 1];
    const double a11 = wp * (f[ix + 1][iy + 1][ip] - f[ix + 1][iy + 1][ip + 1]) + f[ix + 1][iy + 1][ip + 1];
    a00 = wy * (a00 - a01) + a01;
    a10 = wy * (a10 - a11) + a11;
    v[k] = wx * (a00 - a10) + a10;
  }
  const double wt = (met1->time - ts) / (met1->time - met0->time);
  return wt * (v[0] - v[1]) + v[1];
}

/* what write_grid derives per box from the device's sums: column density, implicit volume mixing ratio, number of
 * particles, mean (and standard deviation) of every quantity -- then laid out as text or as netCDF */
typedef struct {
  size_t ncell;
  double step_z, step_lon, step_lat;
  double *z, *lon, *lat, *area;   /* box centres and column areas */
  double *cd, *vmr_impl;
  int *np;
  double *mean, *sigma;            /* [nq][ncell] */
} grid_result;

static void grid_result_free(grid_result *g) {
  free(g->z);
  free(g->lon);
  free(g->lat);
  free(g->area);
  free(g->cd);
  free(g->vmr_impl);
  free(g->np);
  free(g->mean);
  free(g->sigma);
}

static void write_grid_asc(const char *filename, const ctl_t *ctl, const grid_result *g, const double t) {
  FILE *out = fopen(filename, "w");
  REQUIRE(out, "Cannot create file!");
  static const char *const legend[9] = { "time [s]", "altitude [km]", "longitude [deg]", "latitude [deg]",
    "surface area [km^2]", "layer depth [km]", "column density (implicit) [kg/m^2]", "volume mixing ratio (implicit) [ppv]",
    "number of particles [1]" };
  int column = 0;
  for (int k = 0; k < 9; k++)
    fprintf(out, "# $%d = %s\n", ++column, legend[k]);
  for (int pass = 0; pass < (ctl->grid_stddev ? 2 : 1); pass++)
    for (int q = 0; q < ctl->nq; q++)
      fprintf(out, "# $%i = %s (%s) [%s]\n", ++column, ctl->qnt_name[q], pass ? "stddev" : "mean", ctl->qnt_unit[q]);
  fputc('\n', out);
  /* blocks of the table are separated by blank lines (one per new longitude if there are several latitudes, one
   * per new latitude if there are several levels); sparse tables have none */
  const int blocks = !ctl->grid_sparse;
  for (size_t col = 0; col < (size_t) ctl->grid_nx * (size_t) ctl->grid_ny; col++) {
    const int i = (int) (col / (size_t) ctl->grid_ny), j = (int) (col % (size_t) ctl->grid_ny);
    if (blocks && j == 0 && i > 0 && ctl->grid_ny > 1)
      fputc('\n', out);
    if (blocks && j > 0 && ctl->grid_nz > 1)
      fputc('\n', out);
    for (int k = 0; k < ctl->grid_nz; k++) {
      const size_t cell = (size_t) ARRAY_3D(i, j, ctl->grid_ny, k, ctl->grid_nz);
      if (ctl->grid_sparse && !(g->vmr_impl[cell] > 0))   /* sparse output keeps cells with vmr_impl > 0 only */
        continue;
      fprintf(out, "%.2f %g %g %g %g %g %g %g %d", t, g->z[k], g->lon[i], g->lat[j], g->area[j], g->step_z, g->cd[cell],
              g->vmr_impl[cell], g->np[cell]);
      for (int pass = 0; pass < (ctl->grid_stddev ? 2 : 1); pass++)
        for (int q = 0; q < ctl->nq; q++) {
          fputc(' ', out);
          fprintf(out, ctl->qnt_format[q], (pass ? g->sigma : g->mean)[(size_t) q * g->ncell + cell]);
        }
      fputc('\n', out);
    }
  }
  fclose(out);
}

/* GRID_TYPE 1 (mptrac.c:14058-14183): dimensions time, z, lat, lon (and dz); the box fields as [time][z][lat][lon] */
static void write_grid_nc(const char *filename, const ctl_t *ctl, const grid_result *g, const double t) {
  nccw_file *w = ncw_create(filename);
  int dim[5];
  NCW(dim[0] = nccw_def_dim(w, "time", 1));
  NCW(dim[1] = nccw_def_dim(w, "z", ctl->grid_nz));
  NCW(dim[2] = nccw_def_dim(w, "lat", ctl->grid_ny));
  NCW(dim[3] = nccw_def_dim(w, "lon", ctl->grid_nx));
  NCW(dim[4] = nccw_def_dim(w, "dz", 1));
  ncw_var(w, "time", NCC_DOUBLE, 1, &dim[0], "time", "seconds since 2000-01-01 00:00:00 UTC");
  ncw_var(w, "z", NCC_DOUBLE, 1, &dim[1], "altitude", "km");
  ncw_var(w, "lat", NCC_DOUBLE, 1, &dim[2], "latitude", "degrees_north");
  ncw_var(w, "lon", NCC_DOUBLE, 1, &dim[3], "longitude", "degrees_east");
  ncw_var(w, "dz", NCC_DOUBLE, 1, &dim[1], "layer depth", "km");
  ncw_var(w, "area", NCC_DOUBLE, 1, &dim[2], "surface area", "km**2");
  ncw_var(w, "cd", NCC_FLOAT, 4, dim, "column density", "kg m**-2");
  ncw_var(w, "vmr_impl", NCC_FLOAT, 4, dim, "volume mixing ratio (implicit)", "ppv");
  ncw_var(w, "np", NCC_INT, 4, dim, "number of particles", "1");
  char name[2 * LEN], longname[2 * LEN];
  for (int pass = 0; pass < (ctl->grid_stddev ? 2 : 1); pass++)
    for (int q = 0; q < ctl->nq; q++) {
      sprintf(name, "%s_%s", ctl->qnt_name[q], pass ? "stddev" : "mean");
      sprintf(longname, "%s (%s)", ctl->qnt_longname[q], pass ? "stddev" : "mean");
      ncw_var(w, name, NCC_DOUBLE, 4, dim, longname, ctl->qnt_unit[q]);
    }
  NCW(nccw_enddef(w));
  ncw_put(w, "time", 0, &t);
  ncw_put(w, "lon", 0, g->lon);
  ncw_put(w, "lat", 0, g->lat);
  ncw_put(w, "z", 0, g->z);
  ncw_put(w, "area", 0, g->area);
  {
    /* (the variable dz has nz elements; the reference writes the one layer depth, the rest stays unset) */
    double *depth;
    ALLOC(depth, double, ctl->grid_nz);
    depth[0] = g->step_z;
    ncw_put(w, "dz", 0, depth);
    free(depth);
  }
  /* [lon][lat][z] of the sums -> [z][lat][lon] of the file */
  double *turned;
  ALLOC(turned, double, g->ncell);
#define TURN(expr)                                                                                  \
  for (int i = 0; i < ctl->grid_nx; i++)                                                            \
    for (int j = 0; j < ctl->grid_ny; j++)                                                          \
      for (int k = 0; k < ctl->grid_nz; k++) {                                                      \
        const size_t cell = (size_t) ARRAY_3D(i, j, ctl->grid_ny, k, ctl->grid_nz);                 \
        turned[ARRAY_3D(k, j, ctl->grid_ny, i, ctl->grid_nx)] = (expr);                             \
      }
  TURN(g->cd[cell]);
  ncw_put(w, "cd", 0, turned);
  TURN(g->vmr_impl[cell]);
  ncw_put(w, "vmr_impl", 0, turned);
  TURN((double) g->np[cell]);
  ncw_put(w, "np", 0, turned);
  for (int pass = 0; pass < (ctl->grid_stddev ? 2 : 1); pass++)
    for (int q = 0; q < ctl->nq; q++) {
      sprintf(name, "%s_%s", ctl->qnt_name[q], pass ? "stddev" : "mean");
      TURN((pass ? g->sigma : g->mean)[(size_t) q * g->ncell + cell]);
      ncw_put(w, name, 0, turned);
    }
#undef TURN
  free(turned);
  NCW(nccw_close(w));
}

void write_grid(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm,
                const double t) {
  /* Binning and sums on the device (mphip_grid_sums; mptrac.c:13815-13872),
   * post-processing as mptrac.c:13875-13918, layout as write_grid_asc / write_grid_nc;
   * the implicit volume mixing ratio (MOLMASS set, mass quantity present) uses
   * the temperature at the cell centre, interpolated on the host as in the
   * reference. */
  (void) atm;
  if (ctl->met_coord_type != 0)
    ERRMSG("Only lat/lon grid supported");
  LOG(1, "Write grid data: %s", filename);
  if (t == ctl->t_start) {   /* the vertical weighting function, once per run (mptrac.c:13776-13781) */
    static double kz[EP], kw[EP];
    int nk = 0;
    if (ctl->grid_kernel[0] != '-')
      read_kernel(ctl->grid_kernel, kz, kw, &nk);
    HIP(mphip_set_grid_kernel(g_ctx, nk, kz, kw));
  }
  grid_result g;
  memset(&g, 0, sizeof(g));
  g.ncell = (size_t) ctl->grid_nx * (size_t) ctl->grid_ny * (size_t) ctl->grid_nz;
  const size_t nqa = (size_t) (ctl->nq > 0 ? ctl->nq : 1);
  ALLOC(g.np, int, g.ncell);
  ALLOC(g.mean, double, g.ncell * nqa);
  ALLOC(g.sigma, double, g.ncell * nqa);
  HIP(mphip_grid_sums(g_ctx, t, g.np, g.mean, g.sigma));   /* summed over the ranks of the job */
  if (g_rank != 0) {                                        /* rank 0 writes the file */
    grid_result_free(&g);
    return;
  }
  g.step_z = (ctl->grid_z1 - ctl->grid_z0) / ctl->grid_nz;
  g.step_lon = (ctl->grid_lon1 - ctl->grid_lon0) / ctl->grid_nx;
  g.step_lat = (ctl->grid_lat1 - ctl->grid_lat0) / ctl->grid_ny;
  ALLOC(g.z, double, ctl->grid_nz);
  ALLOC(g.lon, double, ctl->grid_nx);
  ALLOC(g.lat, double, ctl->grid_ny);
  ALLOC(g.area, double, ctl->grid_ny);
  ALLOC(g.cd, double, g.ncell);
  ALLOC(g.vmr_impl, double, g.ncell);
  for (int k = 0; k < ctl->grid_nz; k++)
    g.z[k] = ctl->grid_z0 + g.step_z * (k + 0.5);
  for (int i = 0; i < ctl->grid_nx; i++)
    g.lon[i] = ctl->grid_lon0 + g.step_lon * (i + 0.5);
  for (int j = 0; j < ctl->grid_ny; j++) {
    g.lat[j] = ctl->grid_lat0 + g.step_lat * (j + 0.5);
    g.area[j] = g.step_lat * g.step_lon * SQR(RE * M_PI / 180.) * cos(DEG2RAD(g.lat[j]));
  }
  for (int i = 0; i < ctl->grid_nx; i++)
    for (int j = 0; j < ctl->grid_ny; j++)
      for (int k = 0; k < ctl->grid_nz; k++) {
        const size_t cell = (size_t) ARRAY_3D(i, j, ctl->grid_ny, k, ctl->grid_nz);
        const double mass = ctl->qnt_m >= 0 ? g.mean[(size_t) ctl->qnt_m * g.ncell + cell] : NAN;
        g.cd[cell] = mass / (1e6 * g.area[j]);
        /* implicit volume mixing ratio from the mass in the cell and the air density at its centre (mptrac.c:13885-13900) */
        g.vmr_impl[cell] = NAN;
        if (ctl->qnt_m >= 0 && ctl->molmass > 0 && met0 && met1) {
          g.vmr_impl[cell] = 0;
          if (mass > 0) {
            const double press = P(g.z[k]);
            const double temp = mptrac_amd_intpol_3d(met0, met1, offsetof(met_t, t), t, press, g.lon[i], g.lat[j]);
            g.vmr_impl[cell] = MA / ctl->molmass * g.cd[cell] / (100. * press / (RA * temp) * g.step_z * 1e3);
          }
        }
        /* sums -> mean and standard deviation (after the column density: it uses the summed mass) */
        for (int q = 0; q < ctl->nq; q++) {
          double *mean = &g.mean[(size_t) q * g.ncell + cell], *sigma = &g.sigma[(size_t) q * g.ncell + cell];
          if (g.np[cell] > 0) {
            *mean /= g.np[cell];
            const double var = *sigma / g.np[cell] - SQR(*mean);
            *sigma = var > 0 ? sqrt(var) : 0;
          } else
            *mean = *sigma = NAN;
        }
      }
  if (ctl->grid_type == 0)
    write_grid_asc(filename, ctl, &g, t);
  else
    write_grid_nc(filename, ctl, &g, t);
  grid_result_free(&g);
}

int mptrac_amd_world(void) {
  return g_world;
}

void mptrac_write_output(const char *dirname, const ctl_t *ctl, met_t *met0, met_t *met1, atm_t *atm,
                         depo_t *depo, const double t) {
  /* mptrac.c:8230-8330 without the radioactive deposition output */
  (void) depo;
  char filename[2 * LEN], stamp[64];
  double r;
  int year, mon, day, hour, min, sec;
  jsec2time(t, &year, &mon, &day, &hour, &min, &sec, &r);
  sprintf(stamp, "%04d_%02d_%02d_%02d_%02d_%02d", year, mon, day, hour, min, sec);
  const int atm_due = ctl->atm_basename[0] != '-' && (fmod(t, ctl->atm_dt_out) == 0 || t == ctl->t_stop);
  const int grid_due = ctl->grid_basename[0] != '-' && fmod(t, ctl->grid_dt_out) == 0;
  const int ens_due = ctl->ens_basename[0] != '-' && fmod(t, ctl->ens_dt_out) == 0;
  const int vtk_due = ctl->vtk_basename[0] != '-' && fmod(t, ctl->vtk_dt_out) == 0;
  /* the analysis outputs that look at the particles in every time step */
  const int every_step = ctl->csi_basename[0] != '-' || ctl->prof_basename[0] != '-' || ctl->sample_basename[0] != '-'
    || ctl->stat_basename[0] != '-';
  if ((ens_due || vtk_due || every_step) && g_world > 1)
    ERRMSG("CSI, ensemble, profile, sample, station and VTK output need all particles in one process!");
  if (atm_due || ens_due || vtk_due || every_step)
    mptrac_update_host(NULL, NULL, NULL, NULL, NULL, atm);
  if (atm_due) {
    sprintf(filename, "%s/%s_%s.%s", dirname, ctl->atm_basename, stamp,
            ctl->atm_type_out == 0 ? "tab" : ctl->atm_type_out == 1 ? "bin" : "nc");
    if (g_world > 1)   /* every rank writes its index range: <name>.rank<k> */
      sprintf(filename + strlen(filename), ".rank%d", g_rank);
    mptrac_write_atm(filename, ctl, atm, t);
  }
  if (grid_due) {
    sprintf(filename, "%s/%s_%s.%s", dirname, ctl->grid_basename, stamp, ctl->grid_type == 0 ? "tab" : "nc");
    write_grid(filename, ctl, met0, met1, atm, t);
  }
  if (ctl->csi_basename[0] != '-') {
    sprintf(filename, "%s/%s.tab", dirname, ctl->csi_basename);
    write_csi(filename, ctl, atm, t);
  }
  if (ens_due) {
    sprintf(filename, "%s/%s_%s.tab", dirname, ctl->ens_basename, stamp);
    write_ens(filename, ctl, atm, t);
  }
  if (ctl->prof_basename[0] != '-') {
    sprintf(filename, "%s/%s.tab", dirname, ctl->prof_basename);
    write_prof(filename, ctl, met0, met1, atm, t);
  }
  if (ctl->sample_basename[0] != '-') {
    sprintf(filename, "%s/%s.tab", dirname, ctl->sample_basename);
    write_sample(filename, ctl, met0, met1, atm, t);
  }
  if (ctl->stat_basename[0] != '-') {
    sprintf(filename, "%s/%s.tab", dirname, ctl->stat_basename);
    write_station(filename, ctl, atm, t);
    /* write_station marks the particles it has listed in the quantity "stat" of the host copy; on the
     * reference's CPU path that is the model state, so the marks go back to the device */
    if (ctl->qnt_stat >= 0)
      HIP(mphip_update_quantity(g_ctx, ctl->qnt_stat, atm->q[ctl->qnt_stat]));
  }
  if (vtk_due) {
    static int nvtk;
    if (t == ctl->t_start)
      nvtk = 0;
    sprintf(filename, "%s/%s_%05d.vtk", dirname, ctl->vtk_basename, ++nvtk);
    write_vtk(filename, ctl, atm, t);
  }
}

/* -------------------------------------------------------------------------- */
/* one process per GPU                                                        */
/* -------------------------------------------------------------------------- */

void mptrac_amd_job_from_env(mptrac_amd_job_t *job) {
  const char *e;
  memset(job, 0, sizeof(*job));
  job->world = 1;
  if ((e = getenv("WORLD_SIZE")))
    job->world = atoi(e) > 0 ? atoi(e) : 1;
  if ((e = getenv("RANK")))
    job->rank = atoi(e);
  job->local_rank = job->rank;
  if ((e = getenv("LOCAL_RANK")))
    job->local_rank = atoi(e);
  snprintf(job->addr, sizeof(job->addr), "%s", (e = getenv("MASTER_ADDR")) ? e : "127.0.0.1");
  job->port = ((e = getenv("MASTER_PORT")) ? atoi(e) : 29511) + 1;
  if (job->rank < 0 || job->rank >= job->world)
    ERRMSG("RANK / WORLD_SIZE of the environment do not fit!");
}

void mptrac_amd_shard(atm_t *atm, const mptrac_amd_job_t *job) {
  g_rank = job->rank;
  g_world = job->world;
  if (job->world <= 1)
    return;
  if (atm->np <= 0)
    ERRMSG("Need at least one particle!");
  release_time_range(atm, &g_release_first, &g_release_last);
  const long long n = atm->np, lo = n * job->rank / job->world, hi = n * (job->rank + 1) / job->world;
  const size_t bytes = (size_t) (hi - lo) * sizeof(double);
  memmove(atm->time, atm->time + lo, bytes);
  memmove(atm->p, atm->p + lo, bytes);
  memmove(atm->lon, atm->lon + lo, bytes);
  memmove(atm->lat, atm->lat + lo, bytes);
  for (int iq = 0; iq < NQ; iq++)
    memmove(atm->q[iq], atm->q[iq] + lo, bytes);
  atm->np = (int) (hi - lo);
  g_ip0 = lo;
  g_np_total = n;
  LOG(1, "Rank %d of %d: particles %lld ... %lld of %lld", job->rank, job->world, lo, hi - 1, n);
}

void mptrac_amd_comm_init(const ctl_t *ctl, const mptrac_amd_job_t *job) {
  if (job->world <= 1)
    return;
  need_ctx(ctl);
  char id[128];
  memset(id, 0, sizeof(id));
  if (job->rank == 0 && mphip_comm_unique_id(id) != 0)
    ERRMSG("Cannot create the RCCL identifier (is librccl.so available?)");
  if (!mptrac_amd_bcast(id, sizeof(id), job->rank, job->world, job->addr, job->port))
    ERRMSG("Rendezvous of the ranks at %s:%d failed!", job->addr, job->port);
  HIP(mphip_comm_init(g_ctx, job->world, job->rank, id));
}
