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

#include <pthread.h>
#include <strings.h>
#include <time.h>

#include "../../include/mptrac_hip.h"

/* process-global device state; the reference's interface is not re-entrant
 * either (file-static RNG state, mptrac.c:32-40) */
static mphip_ctx *g_ctx;
static const met_t *g_met_host[2];     /* host snapshots mirrored in device slots met0 / met1 */
/* meteo read-ahead (HIP_MET_PREFETCH), see start_read_ahead() */
static struct {
  met_t *met;              /* the spare buffer */
  pthread_t thread;
  int active;              /* a reader thread is running or waits to be joined */
  int ok;                  /* file read and upload started */
  ctl_t *ctl;
  clim_t *clim;
  char filename[LEN];
} g_ahead;
static int g_nq;                        /* ctl->nq of the last control upload */
static int g_isosurf;                   /* ctl->isosurf of the last control upload */
static int g_meteo_fields;              /* a module_meteo quantity is requested: upload the fields only it reads */

#define HIP(call) {                                                     \
    if ((call) != 0)                                                    \
      ERRMSG("HIP back end: %s", mphip_last_error(g_ctx));              \
  }

/* -------------------------------------------------------------------------- */
/* small utilities                                                            */
/* -------------------------------------------------------------------------- */

void jsec2time(const double jsec, int *year, int *mon, int *day, int *hour, int *min, int *sec,
               double *remain) {
  /* seconds since 2000-01-01T00:00Z, mptrac.c:3265-3294 */
  struct tm t0 = { 0 }, *t1;
  t0.tm_year = 100;
  t0.tm_mday = 1;
  const time_t jsec0 = (time_t) jsec + timegm(&t0);
  t1 = gmtime(&jsec0);
  *year = t1->tm_year + 1900;
  *mon = t1->tm_mon + 1;
  *day = t1->tm_mday;
  *hour = t1->tm_hour;
  *min = t1->tm_min;
  *sec = t1->tm_sec;
  *remain = jsec - floor(jsec);
}

void time2jsec(const int year, const int mon, const int day, const int hour, const int min,
               const int sec, const double remain, double *jsec) {
  struct tm t0 = { 0 }, t1 = { 0 };
  t0.tm_year = 100;
  t0.tm_mday = 1;
  t1.tm_year = year - 1900;
  t1.tm_mon = mon - 1;
  t1.tm_mday = day;
  t1.tm_hour = hour;
  t1.tm_min = min;
  t1.tm_sec = sec;
  *jsec = (double) timegm(&t1) - (double) timegm(&t0) + remain;
}

double scan_ctl(const char *filename, int argc, char *argv[], const char *varname, const int arridx,
                const char *defvalue, char *value) {
  /* Same look-up rules as mptrac.c:12434-12502: "KEY = VALUE" lines of the
   * control file, overridden by trailing "KEY VALUE" arguments; KEY[i] and
   * KEY[*] for arrays; a file name ending in '-' means arguments only. */
  FILE *in = NULL;
  char fullname1[LEN], fullname2[LEN], rval[LEN];
  int contain = 0;

  if (filename[strlen(filename) - 1] != '-')
    if (!(in = fopen(filename, "r")))
      ERRMSG("Cannot open file!");
  if (arridx >= 0) {
    sprintf(fullname1, "%s[%d]", varname, arridx);
    sprintf(fullname2, "%s[*]", varname);
  } else {
    sprintf(fullname1, "%s", varname);
    sprintf(fullname2, "%s", varname);
  }
  if (in != NULL) {
    char dummy[LEN], line[LEN], rvarname[LEN];
    while (fgets(line, LEN, in))
      if (sscanf(line, "%4999s %4999s %4999s", rvarname, dummy, rval) == 3)
        if (strcasecmp(rvarname, fullname1) == 0 || strcasecmp(rvarname, fullname2) == 0) {
          contain = 1;
          break;
        }
    fclose(in);
  }
  for (int i = 1; i < argc - 1; i++)
    if (strcasecmp(argv[i], fullname1) == 0 || strcasecmp(argv[i], fullname2) == 0) {
      sprintf(rval, "%s", argv[i + 1]);
      contain = 1;
      break;
    }
  if (!contain) {
    if (strlen(defvalue) > 0)
      sprintf(rval, "%s", defvalue);
    else
      ERRMSG("Missing variable %s!\n", fullname1);
  }
  LOG(1, "%s = %s", fullname1, rval);
  if (value != NULL)
    sprintf(value, "%s", rval);
  return atof(rval);
}

/* -------------------------------------------------------------------------- */
/* alloc / free                                                               */
/* -------------------------------------------------------------------------- */

void mptrac_alloc(ctl_t **ctl, cache_t **cache, clim_t **clim, met_t **met0, met_t **met1, atm_t **atm,
                  depo_t **depo, dd_t **dd) {
  /* one calloc per struct as the reference (mptrac.c:6294-6372); the device
   * context is created in mptrac_init, once the device ordinal is known */
  ALLOC(*ctl, ctl_t, 1);
  ALLOC(*cache, cache_t, 1);
  ALLOC(*clim, clim_t, 1);
  ALLOC(*met0, met_t, 1);
  ALLOC(*met1, met_t, 1);
  ALLOC(*atm, atm_t, 1);
  if (depo)
    ALLOC(*depo, depo_t, 1);
  if (dd)
    ALLOC(*dd, dd_t, 1);
}

void mptrac_free(ctl_t *ctl, cache_t *cache, clim_t *clim, met_t *met0, met_t *met1, atm_t *atm,
                 depo_t *depo, dd_t *dd) {
  if (g_ahead.active) {
    pthread_join(g_ahead.thread, NULL);
    g_ahead.active = 0;
  }
  if (g_ctx) {
    mphip_destroy(g_ctx);
    g_ctx = NULL;
  }
  free(g_ahead.met);
  g_ahead.met = NULL;
  g_met_host[0] = g_met_host[1] = NULL;
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
  const char *name, *unit;
} qnt_units[] = {
  /* names and units of the reference's SET_QNT table (mptrac.c:6853-6969),
   * hot-path quantities only */
  { "ens", "-" }, { "m", "kg" }, { "vmr", "ppv" }, { "rp", "microns" }, { "rhop", "kg/m^3" },
  { "loss_rate", "s^-1" }, { "mloss_decay", "kg" }, { "mloss_wet", "kg" }, { "mloss_dry", "kg" },
  { "idx", "-" }, { "stat", "-" }, { "zeta", "K" }, { "eta", "1" }, { "aoa", "s" },
#define X(n, u) { #n, u },
  MPTRAC_METEO_QNT(X)
#undef X
};

static const char *unsupported_qnt[] = {
  /* quantities module_meteo takes from the chemistry climatologies (mptrac.c:5129-5140, 5158-5163);
   * not provided */
  "hno3", "oh", "h2o2", "ho2", "o1d", "tsts", "tnat",
  /* quantities only the chemistry, radioactive-decay and domain-decomposition code of the reference fills or
   * mixes (SET_QNT table, mptrac.c:6905-6969): they would be carried along unchanged here */
  "mloss_oh", "mloss_h2o2", "mloss_kpp", "Cx", "Ch2o", "Co3", "Cco", "Coh", "Ch", "Cho2", "Ch2o2", "Co1d", "Co3p",
  "Cccl4", "Cccl3f", "Cccl2f2", "Cn2o", "Csf6", "Arn222", "Apb210", "Abe7", "Acs137", "Ai131", "Axe133",
  "current_subdomain", "target_subdomain", NULL
};

void mptrac_read_ctl(const char *filename, int argc, char *argv[], ctl_t *ctl) {
  LOG(1, "\nMassive-Parallel Trajectory Calculations (MPTRAC), MI355X build (%s)\n", mphip_version());

  /* quantities, mptrac.c:6737-6971 */
  ctl->qnt_m = ctl->qnt_vmr = ctl->qnt_rp = ctl->qnt_rhop = ctl->qnt_ens = ctl->qnt_loss_rate = -1;
  ctl->qnt_mloss_decay = ctl->qnt_mloss_wet = ctl->qnt_mloss_dry = ctl->qnt_zeta = ctl->qnt_eta = -1;
  ctl->qnt_aoa = -1;
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
    sprintf(ctl->qnt_unit[iq], "-");
    for (size_t k = 0; k < sizeof(qnt_units) / sizeof(qnt_units[0]); k++)
      if (strcasecmp(ctl->qnt_name[iq], qnt_units[k].name) == 0)
        sprintf(ctl->qnt_unit[iq], "%s", qnt_units[k].unit);
    for (int k = 0; unsupported_qnt[k]; k++)
      if (strcasecmp(ctl->qnt_name[iq], unsupported_qnt[k]) == 0)
        ERRMSG("Quantity %s is filled by the chemistry / climatology / decomposition code of the reference, "
               "which this build does not provide!", ctl->qnt_name[iq]);
    const char *n = ctl->qnt_name[iq];
    if (!strcasecmp(n, "m")) ctl->qnt_m = iq;
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
#define X(nm, u) else if (!strcasecmp(n, #nm)) ctl->qnt_##nm = iq;
    MPTRAC_METEO_QNT(X)
#undef X
  }

  /* coordinates, time steps, meteo input (mptrac.c:6974-7030) */
  ctl->met_coord_type = (int) scan_ctl(filename, argc, argv, "MET_COORD_TYPE", -1, "0", NULL);
  ctl->met_utm_ref_lat = (ctl->met_coord_type != 0)
    ? scan_ctl(filename, argc, argv, "MET_UTM_REF_LAT", -1, "", NULL) : 0;
  ctl->direction = (int) scan_ctl(filename, argc, argv, "DIRECTION", -1, "1", NULL);
  if (ctl->direction != -1 && ctl->direction != 1)
    ERRMSG("Set DIRECTION to -1 or 1!");
  ctl->t_stop = scan_ctl(filename, argc, argv, "T_STOP", -1, "1e100", NULL);
  ctl->dt_mod = scan_ctl(filename, argc, argv, "DT_MOD", -1, "180", NULL);
  scan_ctl(filename, argc, argv, "METBASE", -1, "-", ctl->metbase);
  ctl->dt_met = scan_ctl(filename, argc, argv, "DT_MET", -1, "3600", NULL);
  ctl->met_type = (int) scan_ctl(filename, argc, argv, "MET_TYPE", -1, "0", NULL);
  ctl->met_dt_out = scan_ctl(filename, argc, argv, "MET_DT_OUT", -1, "0.1", NULL);

  /* modules (mptrac.c:7196-7263) */
  ctl->sort_dt = scan_ctl(filename, argc, argv, "SORT_DT", -1, "-999", NULL);
  ctl->rng_type = (int) scan_ctl(filename, argc, argv, "RNG_TYPE", -1, "1", NULL);
  ctl->advect = (int) scan_ctl(filename, argc, argv, "ADVECT", -1, "2", NULL);
  if (!(ctl->advect == 1 || ctl->advect == 2 || ctl->advect == 4))
    ERRMSG("Set ADVECT to 1, 2, or 4!");
  ctl->advect_vert_coord = (int) scan_ctl(filename, argc, argv, "ADVECT_VERT_COORD", -1, "0", NULL);
  ctl->diffusion = (int) scan_ctl(filename, argc, argv, "DIFFUSION", -1, "0", NULL);
  ctl->turb_pbl_scheme = (int) scan_ctl(filename, argc, argv, "TURB_PBL_SCHEME", -1, "0", NULL);
  ctl->turb_dx_pbl = scan_ctl(filename, argc, argv, "TURB_DX_PBL", -1, "50", NULL);
  ctl->turb_dx_trop = scan_ctl(filename, argc, argv, "TURB_DX_TROP", -1, "50", NULL);
  ctl->turb_dx_strat = scan_ctl(filename, argc, argv, "TURB_DX_STRAT", -1, "0", NULL);
  ctl->turb_dz_pbl = scan_ctl(filename, argc, argv, "TURB_DZ_PBL", -1, "0", NULL);
  ctl->turb_dz_trop = scan_ctl(filename, argc, argv, "TURB_DZ_TROP", -1, "0", NULL);
  ctl->turb_dz_strat = scan_ctl(filename, argc, argv, "TURB_DZ_STRAT", -1, "0.1", NULL);
  ctl->turb_mesox = scan_ctl(filename, argc, argv, "TURB_MESOX", -1, "0.16", NULL);
  ctl->turb_mesoz = scan_ctl(filename, argc, argv, "TURB_MESOZ", -1, "0.16", NULL);
  ctl->turb_pbl_trans = scan_ctl(filename, argc, argv, "TURB_PBL_TRANS", -1, "0", NULL);
  if (ctl->turb_pbl_trans < 0 || ctl->turb_pbl_trans > 1)
    ERRMSG("TURB_PBL_TRANS must be in the range [0, 1]!");
  ctl->conv_mix_pbl = (int) scan_ctl(filename, argc, argv, "CONV_MIX_PBL", -1, "0", NULL);
  ctl->conv_pbl_trans = scan_ctl(filename, argc, argv, "CONV_PBL_TRANS", -1, "0", NULL);
  ctl->conv_cape = scan_ctl(filename, argc, argv, "CONV_CAPE", -1, "-999", NULL);
  ctl->conv_cin = scan_ctl(filename, argc, argv, "CONV_CIN", -1, "-999", NULL);
  ctl->conv_dt = scan_ctl(filename, argc, argv, "CONV_DT", -1, "-999", NULL);
  /* isosurface and boundary conditions (mptrac.c:7207-7209, 7266-7289) */
  ctl->isosurf = (int) scan_ctl(filename, argc, argv, "ISOSURF", -1, "0", NULL);
  scan_ctl(filename, argc, argv, "BALLOON", -1, "-", ctl->balloon);
  ctl->bound_mass = scan_ctl(filename, argc, argv, "BOUND_MASS", -1, "-999", NULL);
  ctl->bound_mass_trend = scan_ctl(filename, argc, argv, "BOUND_MASS_TREND", -1, "0", NULL);
  ctl->bound_vmr = scan_ctl(filename, argc, argv, "BOUND_VMR", -1, "-999", NULL);
  ctl->bound_vmr_trend = scan_ctl(filename, argc, argv, "BOUND_VMR_TREND", -1, "0", NULL);
  ctl->bound_lat0 = scan_ctl(filename, argc, argv, "BOUND_LAT0", -1, "-999", NULL);
  ctl->bound_lat1 = scan_ctl(filename, argc, argv, "BOUND_LAT1", -1, "-999", NULL);
  ctl->bound_p0 = scan_ctl(filename, argc, argv, "BOUND_P0", -1, "-999", NULL);
  ctl->bound_p1 = scan_ctl(filename, argc, argv, "BOUND_P1", -1, "-999", NULL);
  ctl->bound_dps = scan_ctl(filename, argc, argv, "BOUND_DPS", -1, "-999", NULL);
  ctl->bound_dzs = scan_ctl(filename, argc, argv, "BOUND_DZS", -1, "-999", NULL);
  ctl->bound_zetas = scan_ctl(filename, argc, argv, "BOUND_ZETAS", -1, "-999", NULL);
  ctl->bound_pbl = (int) scan_ctl(filename, argc, argv, "BOUND_PBL", -1, "0", NULL);
  ctl->molmass = scan_ctl(filename, argc, argv, "MOLMASS", -1, "-999", NULL);

  /* wet / dry deposition, decay, mixing (mptrac.c:7425-7543) */
  char defstr[LEN];
  for (int k = 0; k < 2; k++) {
    ctl->wet_depo_ic_h[k] = scan_ctl(filename, argc, argv, "WET_DEPO_IC_H", k, "0", NULL);
    ctl->wet_depo_bc_h[k] = k == 0 ? scan_ctl(filename, argc, argv, "WET_DEPO_BC_H", k, "0", NULL)
      : ctl->wet_depo_ic_h[1];
  }
  /* the reference scans WET_DEPO_BC_H[0] only (mptrac.c:7431-7435); [1] is set per species */
  sprintf(defstr, "%g", ctl->wet_depo_ic_h[1]);
  ctl->wet_depo_bc_h[1] = scan_ctl(filename, argc, argv, "WET_DEPO_BC_H", 1, defstr, NULL);
  ctl->wet_depo_so2_ph = scan_ctl(filename, argc, argv, "WET_DEPO_SO2_PH", -1, "0", NULL);
  ctl->wet_depo_ic_a = scan_ctl(filename, argc, argv, "WET_DEPO_IC_A", -1, "0", NULL);
  ctl->wet_depo_ic_b = scan_ctl(filename, argc, argv, "WET_DEPO_IC_B", -1, "0", NULL);
  ctl->wet_depo_bc_a = scan_ctl(filename, argc, argv, "WET_DEPO_BC_A", -1, "0", NULL);
  ctl->wet_depo_bc_b = scan_ctl(filename, argc, argv, "WET_DEPO_BC_B", -1, "0", NULL);
  ctl->wet_depo_pre[0] = scan_ctl(filename, argc, argv, "WET_DEPO_PRE", 0, "0.5", NULL);
  ctl->wet_depo_pre[1] = scan_ctl(filename, argc, argv, "WET_DEPO_PRE", 1, "0.36", NULL);
  ctl->wet_depo_ic_ret_ratio = scan_ctl(filename, argc, argv, "WET_DEPO_IC_RET_RATIO", -1, "1", NULL);
  ctl->wet_depo_bc_ret_ratio = scan_ctl(filename, argc, argv, "WET_DEPO_BC_RET_RATIO", -1, "1", NULL);
  ctl->dry_depo_vdep = scan_ctl(filename, argc, argv, "DRY_DEPO_VDEP", -1, "0", NULL);
  ctl->dry_depo_dp = scan_ctl(filename, argc, argv, "DRY_DEPO_DP", -1, "30", NULL);
  ctl->mixing_dt = scan_ctl(filename, argc, argv, "MIXING_DT", -1, "3600.", NULL);
  ctl->mixing_trop = scan_ctl(filename, argc, argv, "MIXING_TROP", -1, "-999", NULL);
  ctl->mixing_strat = scan_ctl(filename, argc, argv, "MIXING_STRAT", -1, "-999", NULL);
  ctl->mixing_z0 = scan_ctl(filename, argc, argv, "MIXING_Z0", -1, "-5", NULL);
  ctl->mixing_z1 = scan_ctl(filename, argc, argv, "MIXING_Z1", -1, "85", NULL);
  ctl->mixing_nz = (int) scan_ctl(filename, argc, argv, "MIXING_NZ", -1, "90", NULL);
  ctl->mixing_lon0 = scan_ctl(filename, argc, argv, "MIXING_LON0", -1, "-180", NULL);
  ctl->mixing_lon1 = scan_ctl(filename, argc, argv, "MIXING_LON1", -1, "180", NULL);
  ctl->mixing_nx = (int) scan_ctl(filename, argc, argv, "MIXING_NX", -1, "360", NULL);
  ctl->mixing_lat0 = scan_ctl(filename, argc, argv, "MIXING_LAT0", -1, "-90", NULL);
  ctl->mixing_lat1 = scan_ctl(filename, argc, argv, "MIXING_LAT1", -1, "90", NULL);
  ctl->mixing_ny = (int) scan_ctl(filename, argc, argv, "MIXING_NY", -1, "180", NULL);
  if (ctl->mixing_nx < 1 || ctl->mixing_ny < 1 || ctl->mixing_nz < 1 || ctl->mixing_lon0 >= ctl->mixing_lon1
      || ctl->mixing_lat0 >= ctl->mixing_lat1 || ctl->mixing_z0 >= ctl->mixing_z1
      || ctl->mixing_lat0 < -90 || ctl->mixing_lat1 > 90)
    ERRMSG("Invalid mixing grid!");
  ctl->tdec_trop = scan_ctl(filename, argc, argv, "TDEC_TROP", -1, "0", NULL);
  ctl->tdec_strat = scan_ctl(filename, argc, argv, "TDEC_STRAT", -1, "0", NULL);
  ctl->nens = (int) scan_ctl(filename, argc, argv, "NENS", -1, "0", NULL);

  /* output (mptrac.c:7551-7648) */
  scan_ctl(filename, argc, argv, "ATM_BASENAME", -1, "-", ctl->atm_basename);
  ctl->atm_dt_out = scan_ctl(filename, argc, argv, "ATM_DT_OUT", -1, "86400", NULL);
  ctl->atm_filter = (int) scan_ctl(filename, argc, argv, "ATM_FILTER", -1, "0", NULL);
  ctl->atm_stride = (int) scan_ctl(filename, argc, argv, "ATM_STRIDE", -1, "1", NULL);
  ctl->atm_type = (int) scan_ctl(filename, argc, argv, "ATM_TYPE", -1, "0", NULL);
  ctl->atm_type_out = (int) scan_ctl(filename, argc, argv, "ATM_TYPE_OUT", -1, "-1", NULL);
  if (ctl->atm_type_out == -1)
    ctl->atm_type_out = ctl->atm_type;
  scan_ctl(filename, argc, argv, "GRID_BASENAME", -1, "-", ctl->grid_basename);
  ctl->grid_dt_out = scan_ctl(filename, argc, argv, "GRID_DT_OUT", -1, "86400", NULL);
  ctl->grid_sparse = (int) scan_ctl(filename, argc, argv, "GRID_SPARSE", -1, "0", NULL);
  ctl->grid_stddev = (int) scan_ctl(filename, argc, argv, "GRID_STDDEV", -1, "0", NULL);
  ctl->grid_z0 = scan_ctl(filename, argc, argv, "GRID_Z0", -1, "-5", NULL);
  ctl->grid_z1 = scan_ctl(filename, argc, argv, "GRID_Z1", -1, "85", NULL);
  ctl->grid_nz = (int) scan_ctl(filename, argc, argv, "GRID_NZ", -1, "1", NULL);
  ctl->grid_lon0 = scan_ctl(filename, argc, argv, "GRID_LON0", -1, "-180", NULL);
  ctl->grid_lon1 = scan_ctl(filename, argc, argv, "GRID_LON1", -1, "180", NULL);
  ctl->grid_nx = (int) scan_ctl(filename, argc, argv, "GRID_NX", -1, "360", NULL);
  ctl->grid_lat0 = scan_ctl(filename, argc, argv, "GRID_LAT0", -1, "-90", NULL);
  ctl->grid_lat1 = scan_ctl(filename, argc, argv, "GRID_LAT1", -1, "90", NULL);
  ctl->grid_ny = (int) scan_ctl(filename, argc, argv, "GRID_NY", -1, "180", NULL);
  if (ctl->grid_nx < 1 || ctl->grid_ny < 1 || ctl->grid_nz < 1)
    ERRMSG("Invalid output grid dimensions!");
  if (ctl->grid_lon0 >= ctl->grid_lon1 || ctl->grid_lat0 >= ctl->grid_lat1 || ctl->grid_z0 >= ctl->grid_z1
      || ctl->grid_lat0 < -90 || ctl->grid_lat1 > 90)
    ERRMSG("Invalid output grid boundaries!");

  /* back-end options */
  ctl->hip_device = (int) scan_ctl(filename, argc, argv, "HIP_DEVICE", -1, "0", NULL);
  ctl->hip_locality_interval = (int) scan_ctl(filename, argc, argv, "HIP_LOCALITY_SORT_INTERVAL", -1, "60", NULL);
  ctl->hip_met_prefetch = (int) scan_ctl(filename, argc, argv, "HIP_MET_PREFETCH", -1, "0", NULL);

  /* what this host layer / the device do not implement must not be requested silently: the reference's
   * other output writers (mptrac.c:7574-7713), chemistry and radioactive decay switches (7386-7411),
   * kernel-weighted and netCDF gridded output, domain decomposition */
  {
    static const char *const names[] = { "DEPO_BASENAME", "CSI_BASENAME", "ENS_BASENAME", "PROF_BASENAME",
      "SAMPLE_BASENAME", "STAT_BASENAME", "VTK_BASENAME", "GRID_KERNEL", "ATM_GPFILE", "GRID_GPFILE", NULL };
    char val[LEN];
    for (int k = 0; names[k]; k++) {
      scan_ctl(filename, argc, argv, names[k], -1, "-", val);
      if (val[0] != '-')
        ERRMSG("%s is not implemented in this host layer!", names[k]);
    }
    static const char *const switches[] = { "OH_CHEM_REACTION", "H2O2_CHEM_REACTION", "KPP_CHEM", "TRACER_CHEM",
      "RADIO_DECAY", "GRID_TYPE", "DD", NULL };
    for (int k = 0; switches[k]; k++)
      if ((int) scan_ctl(filename, argc, argv, switches[k], -1, "0", NULL) != 0)
        ERRMSG("%s is not implemented in this host layer!", switches[k]);
  }
  if (ctl->rng_type != 1)
    ERRMSG("This build implements RNG_TYPE 1 (Squares) only!");
  if (ctl->advect_vert_coord < 0 || ctl->advect_vert_coord > 3)
    ERRMSG("Set ADVECT_VERT_COORD to 0, 1, 2, or 3!");
  if (ctl->advect_vert_coord == 1 && ctl->qnt_zeta < 0)
    ERRMSG("Please add zeta to your quantities for diabatic calculations!");   /* mptrac.c:6992 */
  if (ctl->advect_vert_coord == 3 && ctl->qnt_eta < 0)
    ERRMSG("Please add eta to your quantities for etadot calculations!");      /* mptrac.c:6994 */
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

void mptrac_read_clim(const ctl_t *ctl, clim_t *clim) {
  (void) ctl;   /* the chemistry climatologies of mptrac.c:6663-6719 are not on the hot path */
  clim_tropo_init(clim);
}

/* -------------------------------------------------------------------------- */
/* particle I/O                                                               */
/* -------------------------------------------------------------------------- */

#define FREAD(ptr, type, size, in) {                                    \
    if (fread(ptr, sizeof(type), size, in) != size)                     \
      ERRMSG("Error while reading!");                                   \
  }
#define FWRITE(ptr, type, size, out) {                                  \
    if (fwrite(ptr, sizeof(type), size, out) != size)                   \
      ERRMSG("Error while writing!");                                   \
  }

static int read_atm_asc(const char *filename, const ctl_t *ctl, atm_t *atm) {
  /* columns: time, altitude [km], lon, lat, q[0..nq) (mptrac.c:8380-8418) */
  FILE *in;
  if (!(in = fopen(filename, "r"))) {
    WARN("Cannot open file!");
    return 0;
  }
  char line[LEN];
  while (fgets(line, LEN, in)) {
    char *tok = strtok(line, " \t");
    double v[4 + NQ];
    int k = 0;
    for (; tok && k < 4 + ctl->nq; k++, tok = strtok(NULL, " \t"))
      if (sscanf(tok, "%lg", &v[k]) != 1)
        break;
    if (k == 0 && (line[0] == '#' || line[0] == '\n' || line[0] == '\0'))
      continue;
    if (k < 4 + ctl->nq) {
      if (line[0] == '#' || k == 0)
        continue;
      ERRMSG("Error while reading!");
    }
    atm->time[atm->np] = v[0];
    atm->p[atm->np] = P(v[1]);
    atm->lon[atm->np] = v[2];
    atm->lat[atm->np] = v[3];
    for (int iq = 0; iq < ctl->nq; iq++)
      atm->q[iq][atm->np] = v[4 + iq];
    if ((++atm->np) > NP)
      ERRMSG("Too many data points!");
  }
  fclose(in);
  return 1;
}

static int read_atm_bin(const char *filename, const ctl_t *ctl, atm_t *atm) {
  /* version 100 layout, mptrac.c:8422-8475 */
  FILE *in;
  if (!(in = fopen(filename, "r")))
    return 0;
  int version;
  FREAD(&version, int, 1, in);
  if (version != 100)
    ERRMSG("Wrong version of binary data!");
  FREAD(&atm->np, int, 1, in);
  if (atm->np < 0 || atm->np > NP)
    ERRMSG("Too many data points!");
  FREAD(atm->time, double, (size_t) atm->np, in);
  FREAD(atm->p, double, (size_t) atm->np, in);
  FREAD(atm->lon, double, (size_t) atm->np, in);
  FREAD(atm->lat, double, (size_t) atm->np, in);
  for (int iq = 0; iq < ctl->nq; iq++)
    FREAD(atm->q[iq], double, (size_t) atm->np, in);
  int final;
  FREAD(&final, int, 1, in);
  if (final != 999)
    ERRMSG("Error while reading binary data!");
  fclose(in);
  return 1;
}

int mptrac_read_atm(const char *filename, const ctl_t *ctl, atm_t *atm) {
  /* mptrac.c:6588-6659 */
  int result;
  atm->np = 0;
  LOG(1, "Read atmospheric data: %s", filename);
  if (ctl->atm_type == 0)
    result = read_atm_asc(filename, ctl, atm);
  else if (ctl->atm_type == 1)
    result = read_atm_bin(filename, ctl, atm);
  else
    ERRMSG("Atmospheric data type not supported (this build reads ATM_TYPE 0 and 1)!");
  if (result != 1)
    return 0;
  if (atm->np < 1)
    ERRMSG("Can not read any data!");
  LOG(2, "Number of particles: %d", atm->np);
  return result;
}

static void write_atm_asc(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  /* mptrac.c:12774-12868 */
  FILE *out;
  const double t0 = t - 0.5 * ctl->dt_mod, t1 = t + 0.5 * ctl->dt_mod;
  if (!(out = fopen(filename, "w")))
    ERRMSG("Cannot create file!");
  if (ctl->met_coord_type == 0)
    fprintf(out, "# $1 = time [s]\n# $2 = altitude [km]\n# $3 = longitude [deg]\n# $4 = latitude [deg]\n");
  else
    fprintf(out, "# $1 = time [s]\n# $2 = altitude [km]\n# $3 = x [m]\n# $4 = y [m]\n");
  for (int iq = 0; iq < ctl->nq; iq++)
    fprintf(out, "# $%i = %s [%s]\n", iq + 5, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
  fprintf(out, "\n");
  for (int ip = 0; ip < atm->np; ip += ctl->atm_stride) {
    if (ctl->atm_filter == 2 && (atm->time[ip] < t0 || atm->time[ip] > t1))
      continue;
    if (ctl->met_coord_type == 0)
      fprintf(out, "%.2f %g %g %g", atm->time[ip], Z(atm->p[ip]), atm->lon[ip], atm->lat[ip]);
    else
      fprintf(out, "%.2f %g %.2f %.2f", atm->time[ip], Z(atm->p[ip]), atm->lon[ip], atm->lat[ip]);
    for (int iq = 0; iq < ctl->nq; iq++) {
      fprintf(out, " ");
      if (ctl->atm_filter == 1 && (atm->time[ip] < t0 || atm->time[ip] > t1))
        fprintf(out, ctl->qnt_format[iq], NAN);
      else
        fprintf(out, ctl->qnt_format[iq], atm->q[iq][ip]);
    }
    fprintf(out, "\n");
  }
  fclose(out);
}

static void write_atm_bin(const char *filename, const ctl_t *ctl, const atm_t *atm) {
  /* mptrac.c:12872-12918 */
  FILE *out;
  if (!(out = fopen(filename, "w")))
    ERRMSG("Cannot create file!");
  int version = 100, final = 999;
  FWRITE(&version, int, 1, out);
  FWRITE(&atm->np, int, 1, out);
  FWRITE(atm->time, double, (size_t) atm->np, out);
  FWRITE(atm->p, double, (size_t) atm->np, out);
  FWRITE(atm->lon, double, (size_t) atm->np, out);
  FWRITE(atm->lat, double, (size_t) atm->np, out);
  for (int iq = 0; iq < ctl->nq; iq++)
    FWRITE(atm->q[iq], double, (size_t) atm->np, out);
  FWRITE(&final, int, 1, out);
  fclose(out);
}

void mptrac_write_atm(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  /* mptrac.c:8117-8155 */
  LOG(1, "Write atmospheric data: %s", filename);
  if (ctl->atm_type_out == 0)
    write_atm_asc(filename, ctl, atm, t);
  else if (ctl->atm_type_out == 1)
    write_atm_bin(filename, ctl, atm);
  else
    ERRMSG("Atmospheric data type not supported (this build writes ATM_TYPE_OUT 0 and 1)!");
}

/* -------------------------------------------------------------------------- */
/* meteo I/O: the reference's raw binary format (MET_TYPE 1, version 104)     */
/* -------------------------------------------------------------------------- */

static void bin_2d(FILE *f, int write, const met_t *met, float var[EX][EY], float *help) {
  /* one [nx][ny] float block, read or skipped (var == NULL) or written */
  const size_t n = (size_t) met->nx * (size_t) met->ny;
  if (write) {
    for (int ix = 0; ix < met->nx; ix++)
      for (int iy = 0; iy < met->ny; iy++)
        help[(size_t) ix * met->ny + iy] = var ? var[ix][iy] : 0.f;
    FWRITE(help, float, n, f);
  } else {
    FREAD(help, float, n, f);
    if (var)
      for (int ix = 0; ix < met->nx; ix++)
        for (int iy = 0; iy < met->ny; iy++)
          var[ix][iy] = help[(size_t) ix * met->ny + iy];
  }
}

static void bin_3d(FILE *f, int write, const met_t *met, float var[EX][EY][EP], float *help, float lo, float hi) {
  const size_t n = (size_t) met->nx * (size_t) met->ny * (size_t) met->np;
  if (write) {
    for (int ix = 0; ix < met->nx; ix++)
      for (int iy = 0; iy < met->ny; iy++)
        for (int ip = 0; ip < met->np; ip++)
          help[((size_t) ix * met->ny + iy) * met->np + ip] = var ? var[ix][iy][ip] : 0.f;
    FWRITE(help, float, n, f);
  } else {
    FREAD(help, float, n, f);
    if (var)
      for (int ix = 0; ix < met->nx; ix++)
        for (int iy = 0; iy < met->ny; iy++)
          for (int ip = 0; ip < met->np; ip++) {
            /* bounds check of read_met_bin_3d, mptrac.c:9172-9177 */
            float v = help[((size_t) ix * met->ny + iy) * met->np + ip];
            var[ix][iy][ip] = v < lo ? lo : (v > hi ? hi : v);
          }
  }
}

static void met_bin_body(FILE *f, int write, met_t *met) {
  /* field order of read_met_bin / write_met_bin (mptrac.c:8990-9028, 14245-14300):
   * 24 surface fields, 13 level fields */
  float *help;
  ALLOC(help, float, (size_t) met->nx * (size_t) met->ny * (size_t) met->np);
  float (*s2[24])[EY] = { met->ps, met->ts, met->zs, met->us, met->vs, met->ess, met->nss, met->shf, met->lsm,
    met->sst, met->pbl, met->pt, met->tt, met->zt, met->h2ot, met->pct, met->pcb, met->cl, met->plcl, met->plfc,
    met->pel, met->cape, met->cin, met->o3c };
  for (int k = 0; k < 24; k++)
    bin_2d(f, write, met, s2[k], help);
  float (*s3[13])[EY][EP] = { met->z, met->t, met->u, met->v, met->w, met->pv, met->h2o, met->o3, met->lwc, met->rwc,
    met->iwc, met->swc, met->cc };
  const float lo[13] = { -1e34f, 0, -1e34f, -1e34f, -1e34f, -1e34f, 0, 0, 0, 0, 0, 0, 0 };
  const float hi[13] = { 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1e34f, 1 };
  for (int k = 0; k < 13; k++)
    bin_3d(f, write, met, s3[k], help, lo[k], hi[k]);
  free(help);
}

int mptrac_read_met(const char *filename, const ctl_t *ctl, const clim_t *clim, met_t *met, dd_t *dd) {
  (void) clim;
  (void) dd;
  LOG(1, "Read meteo data: %s", filename);
  if (ctl->met_type != 1)
    ERRMSG("This build reads MET_TYPE 1 (raw binary) meteo files only!");
  FILE *in;
  if (!(in = fopen(filename, "r"))) {
    WARN("Cannot open file!");
    return 0;
  }
  int met_type, version;
  FREAD(&met_type, int, 1, in);
  if (met_type != ctl->met_type)
    ERRMSG("Wrong MET_TYPE of binary data!");
  FREAD(&version, int, 1, in);
  if (version != 104)
    ERRMSG("Wrong version of binary data!");
  FREAD(&met->time, double, 1, in);
  met->coord_type = ctl->met_coord_type;
  FREAD(&met->nx, int, 1, in);
  if (met->nx < 2 || met->nx > EX)
    ERRMSG("Number of longitudes out of range!");
  FREAD(&met->ny, int, 1, in);
  if (met->ny < 2 || met->ny > EY)
    ERRMSG("Number of latitudes out of range!");
  FREAD(&met->np, int, 1, in);
  if (met->np < 2 || met->np > EP)
    ERRMSG("Number of levels out of range!");
  FREAD(met->lon, double, (size_t) met->nx, in);
  FREAD(met->lat, double, (size_t) met->ny, in);
  FREAD(met->p, double, (size_t) met->np, in);
  met_bin_body(in, 0, met);
  int final;
  FREAD(&final, int, 1, in);
  if (final != 999)
    ERRMSG("Error while reading binary data!");
  fclose(in);
  return 1;
}

void mptrac_write_met(const char *filename, const ctl_t *ctl, met_t *met) {
  /* write_met_bin, mptrac.c:14188-14314 */
  LOG(1, "Write meteo data: %s", filename);
  if (ctl->met_type != 1)
    ERRMSG("This build writes MET_TYPE 1 (raw binary) meteo files only!");
  FILE *out;
  if (!(out = fopen(filename, "w")))
    ERRMSG("Cannot create file!");
  int version = 104, final = 999;
  FWRITE(&ctl->met_type, int, 1, out);
  FWRITE(&version, int, 1, out);
  FWRITE(&met->time, double, 1, out);
  FWRITE(&met->nx, int, 1, out);
  FWRITE(&met->ny, int, 1, out);
  FWRITE(&met->np, int, 1, out);
  FWRITE(met->lon, double, (size_t) met->nx, out);
  FWRITE(met->lat, double, (size_t) met->ny, out);
  FWRITE(met->p, double, (size_t) met->np, out);
  met_bin_body(out, 1, met);
  FWRITE(&final, int, 1, out);
  fclose(out);
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
  sprintf(filename, "%s_%d_%02d_%02d_%02d.bin", ctl->metbase, year, mon, day, hour);
}

/* -------------------------------------------------------------------------- */
/* device mirrors                                                             */
/* -------------------------------------------------------------------------- */

static void to_device_ctl(const ctl_t *c, mphip_ctl_t *d) {
  memset(d, 0, sizeof(*d));
#define CP(f) d->f = c->f
  CP(direction); CP(met_coord_type); CP(t_start); CP(t_stop); CP(dt_mod); CP(dt_met); CP(met_utm_ref_lat);
  CP(nq); CP(qnt_m); CP(qnt_vmr); CP(qnt_rp); CP(qnt_rhop); CP(qnt_ens); CP(qnt_loss_rate);
  CP(qnt_mloss_decay); CP(qnt_mloss_wet); CP(qnt_mloss_dry); CP(nens); CP(advect); CP(advect_vert_coord);
  CP(rng_type); CP(diffusion); CP(turb_pbl_scheme); CP(conv_mix_pbl);
  CP(turb_dx_pbl); CP(turb_dx_trop); CP(turb_dx_strat); CP(turb_dz_pbl); CP(turb_dz_trop); CP(turb_dz_strat);
  CP(turb_mesox); CP(turb_mesoz); CP(turb_pbl_trans); CP(conv_pbl_trans); CP(conv_cape); CP(conv_cin);
  CP(conv_dt); CP(sort_dt); CP(tdec_trop); CP(tdec_strat); CP(mixing_dt); CP(mixing_trop); CP(mixing_strat);
  CP(mixing_z0); CP(mixing_z1); CP(mixing_lon0); CP(mixing_lon1); CP(mixing_lat0); CP(mixing_lat1);
  CP(mixing_nx); CP(mixing_ny); CP(mixing_nz);
  CP(wet_depo_ic_a); CP(wet_depo_ic_b); CP(wet_depo_bc_a); CP(wet_depo_bc_b); CP(wet_depo_so2_ph);
  CP(wet_depo_ic_ret_ratio); CP(wet_depo_bc_ret_ratio); CP(dry_depo_vdep); CP(dry_depo_dp);
  CP(grid_z0); CP(grid_z1); CP(grid_lon0); CP(grid_lon1); CP(grid_lat0); CP(grid_lat1);
  CP(grid_nx); CP(grid_ny); CP(grid_nz);
#undef CP
  d->qnt_zeta = c->qnt_zeta;
  d->qnt_eta = c->qnt_eta;
  d->met_dt_out = c->met_dt_out;
  d->qnt_aoa = c->qnt_aoa;
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

/* Read-ahead of the next meteo file (HIP_MET_PREFETCH 1, forward runs): while
 * the time steps of the current interval run, a reader thread loads the file
 * after met1 into a third met_t and starts its upload on the back end's copy
 * stream (mphip_prefetch_met); at the hand-over mptrac_get_met rotates the
 * three host pointers and commits instead of reading and uploading on the
 * stepping path.  Costs one more met_t of host memory, hence off by default. */

static void *read_ahead_main(void *arg) {
  (void) arg;
  g_ahead.ok = 0;
  FILE *probe = fopen(g_ahead.filename, "r");   /* past the last file of the run: nothing to do */
  if (!probe)
    return NULL;
  fclose(probe);
  if (!mptrac_read_met(g_ahead.filename, g_ahead.ctl, g_ahead.clim, g_ahead.met, NULL))
    return NULL;
  mphip_met_t m;
  describe_met(g_ahead.met, &m);
  if (mphip_prefetch_met(g_ctx, &m) != 0) {
    WARN("Meteo read-ahead: %s", mphip_last_error(g_ctx));
    return NULL;
  }
  g_ahead.ok = 1;
  return NULL;
}

static void start_read_ahead(ctl_t *ctl, clim_t *clim, const met_t *met1) {
  if (!ctl->hip_met_prefetch || ctl->direction != 1 || g_ahead.active)
    return;
  if (!g_ahead.met)
    ALLOC(g_ahead.met, met_t, 1);
  g_ahead.ctl = ctl;
  g_ahead.clim = clim;
  get_met_filename(ctl, met1->time + 1, 1, g_ahead.filename);
  if (pthread_create(&g_ahead.thread, NULL, read_ahead_main, NULL) == 0)
    g_ahead.active = 1;
}

static void cancel_read_ahead(void) {
  if (!g_ahead.active)
    return;
  pthread_join(g_ahead.thread, NULL);
  g_ahead.active = 0;
  if (g_ahead.ok)
    HIP(mphip_discard_prefetch(g_ctx));
}

/* 1: the prefetched snapshot is the one `filename` names and now is *met1 */
static int take_read_ahead(const char *filename, met_t **met0, met_t **met1) {
  if (!g_ahead.active)
    return 0;
  pthread_join(g_ahead.thread, NULL);
  g_ahead.active = 0;
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
  if (ctl != NULL) {
    mphip_ctl_t d;
    to_device_ctl(ctl, &d);
    HIP(mphip_update_ctl(g_ctx, &d));
    g_nq = ctl->nq;
  }
  if (clim != NULL)
    HIP(mphip_update_clim(g_ctx, clim->tropo_ntime, clim->tropo_nlat, clim->tropo_time, clim->tropo_lat,
                          &clim->tropo[0][0], 73));
  if (met0 != NULL)
    upload_met(*met0, 0);
  if (met1 != NULL)
    upload_met(*met1, 1);
  if (atm != NULL) {
    const double *q[MPHIP_NQ_MAX] = { 0 };
    for (int iq = 0; iq < g_nq; iq++)
      q[iq] = atm->q[iq];
    HIP(mphip_update_atm(g_ctx, atm->np, 0, atm->np, g_nq, atm->time, atm->p, atm->lon, atm->lat, q));
  }
  if (cache != NULL) {
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
  if (atm != NULL) {
    atm_t *a = (atm_t *) atm;   /* the reference's signature is const; the data are refreshed */
    double *q[MPHIP_NQ_MAX] = { 0 };
    for (int iq = 0; iq < g_nq; iq++)
      q[iq] = a->q[iq];
    HIP(mphip_get_atm(g_ctx, a->time, a->p, a->lon, a->lat, q));
  }
  if (cache != NULL) {
    cache_t *c = (cache_t *) cache;
    HIP(mphip_get_cache(g_ctx, &c->uvwp[0][0], c->dt, NULL));
    if (g_isosurf >= 1 && g_isosurf <= 3)
      (void) mphip_get_iso(g_ctx, c->iso_var);   /* not on the device before the first time step */
  }
}

/* -------------------------------------------------------------------------- */
/* init / meteo handling / time step                                          */
/* -------------------------------------------------------------------------- */

void module_timesteps_init(ctl_t *ctl, const atm_t *atm) {
  /* host bookkeeping of mptrac.c:6046-6073 (gsl_stats_min/max -> loops) */
  double tmin = atm->time[0], tmax = atm->time[0];
  for (int ip = 1; ip < atm->np; ip++) {
    if (atm->time[ip] < tmin)
      tmin = atm->time[ip];
    if (atm->time[ip] > tmax)
      tmax = atm->time[ip];
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
    ERRMSG("Nothing to do! Check T_STOP and DIRECTION!");
  if (ctl->direction == 1)
    ctl->t_start = floor(ctl->t_start / ctl->dt_mod) * ctl->dt_mod;
  else
    ctl->t_start = ceil(ctl->t_start / ctl->dt_mod) * ctl->dt_mod;
}

void mptrac_init(ctl_t *ctl, cache_t *cache, clim_t *clim, atm_t *atm, depo_t *depo, const int ntask) {
  /* mptrac.c:6563-6584; ntask seeds the GSL generators of RNG_TYPE 0 only */
  (void) depo;
  (void) ntask;
  module_timesteps_init(ctl, atm);
  mptrac_update_device(ctl, cache, clim, NULL, NULL, atm);
}

void mptrac_get_met(ctl_t *ctl, clim_t *clim, const double t, met_t **met0, met_t **met1, dd_t *dd) {
  /* double-buffer logic of mptrac.c:6438-6559 */
  static int init;
  char filename[LEN];
  met_t *mets;

  if (t == ctl->t_start || !init) {
    init = 1;
    cancel_read_ahead();
    get_met_filename(ctl, t + (ctl->direction == -1 ? -1 : 0), -1, filename);
    if (!mptrac_read_met(filename, ctl, clim, *met0, dd))
      ERRMSG("Cannot open file!");
    get_met_filename(ctl, t + (ctl->direction == 1 ? 1 : 0), 1, filename);
    if (!mptrac_read_met(filename, ctl, clim, *met1, dd))
      ERRMSG("Cannot open file!");
    mptrac_update_device(NULL, NULL, NULL, met0, met1, NULL);
    start_read_ahead(ctl, clim, *met1);
  }
  if (t > (*met1)->time) {
    get_met_filename(ctl, t, 1, filename);
    if (!take_read_ahead(filename, met0, met1)) {
      mets = *met1;
      *met1 = *met0;
      *met0 = mets;
      if (!mptrac_read_met(filename, ctl, clim, *met1, dd))
        ERRMSG("Cannot open file!");
      map_met_slot(*met0, 0);   /* device slots follow the pointer swap */
      mptrac_update_device(NULL, NULL, NULL, NULL, met1, NULL);
    }
    start_read_ahead(ctl, clim, *met1);
  }
  if (t < (*met0)->time) {
    mets = *met1;
    *met1 = *met0;
    *met0 = mets;
    get_met_filename(ctl, t, -1, filename);
    if (!mptrac_read_met(filename, ctl, clim, *met0, dd))
      ERRMSG("Cannot open file!");
    map_met_slot(*met1, 1);
    mptrac_update_device(NULL, NULL, NULL, met0, NULL, NULL);
  }
  if ((*met0)->coord_type != (*met1)->coord_type)
    ERRMSG("Coordinate types do not match!");
  if ((*met0)->nx != 0 && (*met1)->nx != 0) {
    if ((*met0)->nx != (*met1)->nx || (*met0)->ny != (*met1)->ny || (*met0)->np != (*met1)->np)
      ERRMSG("Meteo grid dimensions do not match!");
    for (int ix = 0; ix < (*met0)->nx; ix++)
      if (fabs((*met0)->lon[ix] - (*met1)->lon[ix]) > 0.001)
        ERRMSG("Meteo grid longitudes do not match!");
    for (int iy = 0; iy < (*met0)->ny; iy++)
      if (fabs((*met0)->lat[iy] - (*met1)->lat[iy]) > 0.001)
        ERRMSG("Meteo grid latitudes do not match!");
    for (int ip = 0; ip < (*met0)->np; ip++)
      if (fabs((*met0)->p[ip] - (*met1)->p[ip]) > 0.001)
        ERRMSG("Meteo grid pressure levels do not match!");
  }
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
  /* module_isosurf_init, ISOSURF 4: read the balloon pressure time series (mptrac.c:4925-4951);
   * modes 1-3 are evaluated on the device */
  if (t == ctl->t_start && ctl->isosurf == 4) {
    LOG(1, "Read balloon pressure data: %s", ctl->balloon);
    FILE *in;
    if (!(in = fopen(ctl->balloon, "r")))
      ERRMSG("Cannot open file!");
    char line[LEN];
    while (fgets(line, LEN, in))
      if (sscanf(line, "%lg %lg", &(cache->iso_ts[cache->iso_n]), &(cache->iso_ps[cache->iso_n])) == 2)
        if ((++cache->iso_n) > NP)
          ERRMSG("Too many data points!");
    if (cache->iso_n < 1)
      ERRMSG("Could not read any data!");
    fclose(in);
    HIP(mphip_update_iso(g_ctx, NULL, cache->iso_ts, cache->iso_ps, cache->iso_n));
  }
  HIP(mphip_run_timestep(g_ctx, t));
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
static int grid_locate_irr(const double *xx, const int n, const double x) {
  int ilo = 0, ihi = n - 1, i = (ihi + ilo) >> 1;
  if (xx[i] < xx[i + 1])
    while (ihi > ilo + 1) {
      i = (ihi + ilo) >> 1;
      if (xx[i] > x)
        ihi = i;
      else
        ilo = i;
  } else
    while (ihi > ilo + 1) {
      i = (ihi + ilo) >> 1;
      if (xx[i] <= x)
        ihi = i;
      else
        ilo = i;
    }
  return ilo;
}

static double grid_temperature(const met_t *met0, const met_t *met1, const double ts, const double p,
                               const double lon, const double lat) {
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
    const met_t *m = mm[k];
    double a00 = wp * (m->t[ix][iy][ip] - m->t[ix][iy][ip + 1]) + m->t[ix][iy][ip + 1];
    const double a01 = wp * (m->t[ix][iy + 1][ip] - m->t[ix][iy + 1][ip + 1]) + m->t[ix][iy + 1][ip + 1];
    double a10 = wp * (m->t[ix + 1][iy][ip] - m->t[ix + 1][iy][ip + 1]) + m->t[ix + 1][iy][ip + 1];
    const double a11 = wp * (m->t[ix + 1][iy + 1][ip] - m->t[ix + 1][iy + 1][ip + 1]) + m->t[ix + 1][iy + 1][ip + 1];
    a00 = wy * (a00 - a01) + a01;
    a10 = wy * (a10 - a11) + a11;
    v[k] = wx * (a00 - a10) + a10;
  }
  const double wt = (met1->time - ts) / (met1->time - met0->time);
  return wt * (v[0] - v[1]) + v[1];
}

void write_grid(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm,
                const double t) {
  /* Binning and sums on the device (mphip_grid_sums; mptrac.c:13815-13872),
   * post-processing and ASCII layout as mptrac.c:13875-13918, 13954-14056;
   * the implicit volume mixing ratio (MOLMASS set, mass quantity present) uses
   * the temperature at the cell centre, interpolated on the host as in the
   * reference. */
  (void) atm;
  if (ctl->met_coord_type != 0)
    ERRMSG("Only lat/lon grid supported");
  LOG(1, "Write grid data: %s", filename);
  const size_t ncell = (size_t) ctl->grid_nx * (size_t) ctl->grid_ny * (size_t) ctl->grid_nz;
  int *np;
  double *mean, *sigma;
  ALLOC(np, int, ncell);
  ALLOC(mean, double, ncell * (size_t) (ctl->nq > 0 ? ctl->nq : 1));
  ALLOC(sigma, double, ncell * (size_t) (ctl->nq > 0 ? ctl->nq : 1));
  HIP(mphip_grid_sums(g_ctx, t, np, mean, sigma));

  const double dz = (ctl->grid_z1 - ctl->grid_z0) / ctl->grid_nz;
  const double dlon = (ctl->grid_lon1 - ctl->grid_lon0) / ctl->grid_nx;
  const double dlat = (ctl->grid_lat1 - ctl->grid_lat0) / ctl->grid_ny;
  FILE *out;
  if (!(out = fopen(filename, "w")))
    ERRMSG("Cannot create file!");
  fprintf(out, "# $1 = time [s]\n# $2 = altitude [km]\n# $3 = longitude [deg]\n# $4 = latitude [deg]\n"
          "# $5 = surface area [km^2]\n# $6 = layer depth [km]\n# $7 = column density (implicit) [kg/m^2]\n"
          "# $8 = volume mixing ratio (implicit) [ppv]\n# $9 = number of particles [1]\n");
  for (int iq = 0; iq < ctl->nq; iq++)
    fprintf(out, "# $%i = %s (mean) [%s]\n", 10 + iq, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
  if (ctl->grid_stddev)
    for (int iq = 0; iq < ctl->nq; iq++)
      fprintf(out, "# $%i = %s (stddev) [%s]\n", 10 + ctl->nq + iq, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
  fprintf(out, "\n");
  for (int ix = 0; ix < ctl->grid_nx; ix++) {
    if (ix > 0 && ctl->grid_ny > 1 && !ctl->grid_sparse)
      fprintf(out, "\n");
    for (int iy = 0; iy < ctl->grid_ny; iy++) {
      if (iy > 0 && ctl->grid_nz > 1 && !ctl->grid_sparse)
        fprintf(out, "\n");
      const double lat = ctl->grid_lat0 + dlat * (iy + 0.5);
      const double area = dlat * dlon * SQR(RE * M_PI / 180.) * cos(DEG2RAD(lat));
      for (int iz = 0; iz < ctl->grid_nz; iz++) {
        const size_t idx = (size_t) ARRAY_3D(ix, iy, ctl->grid_ny, iz, ctl->grid_nz);
        const double cd = ctl->qnt_m >= 0 ? mean[(size_t) ctl->qnt_m * ncell + idx] / (1e6 * area) : NAN;
        double vmr_impl = NAN;   /* mptrac.c:13885-13900 */
        if (ctl->qnt_m >= 0 && ctl->molmass > 0 && met0 != NULL && met1 != NULL) {
          vmr_impl = 0;
          if (mean[(size_t) ctl->qnt_m * ncell + idx] > 0) {
            const double press = P(ctl->grid_z0 + dz * (iz + 0.5));
            const double temp = grid_temperature(met0, met1, t, press, ctl->grid_lon0 + dlon * (ix + 0.5), lat);
            vmr_impl = MA / ctl->molmass * cd / (100. * press / (RA * temp) * dz * 1e3);
          }
        }
        if (ctl->grid_sparse && !(vmr_impl > 0))   /* sparse output keeps cells with vmr_impl > 0 only */
          continue;
        fprintf(out, "%.2f %g %g %g %g %g %g %g %d", t, ctl->grid_z0 + dz * (iz + 0.5),
                ctl->grid_lon0 + dlon * (ix + 0.5), lat, area, dz, cd, vmr_impl, np[idx]);
        for (int iq = 0; iq < ctl->nq; iq++) {
          const double m = np[idx] > 0 ? mean[(size_t) iq * ncell + idx] / np[idx] : NAN;
          fprintf(out, " ");
          fprintf(out, ctl->qnt_format[iq], m);
        }
        if (ctl->grid_stddev)
          for (int iq = 0; iq < ctl->nq; iq++) {
            double sd = NAN;
            if (np[idx] > 0) {
              const double m = mean[(size_t) iq * ncell + idx] / np[idx];
              const double var = sigma[(size_t) iq * ncell + idx] / np[idx] - SQR(m);
              sd = var > 0 ? sqrt(var) : 0;
            }
            fprintf(out, " ");
            fprintf(out, ctl->qnt_format[iq], sd);
          }
        fprintf(out, "\n");
      }
    }
  }
  fclose(out);
  free(np);
  free(mean);
  free(sigma);
}

void mptrac_write_output(const char *dirname, const ctl_t *ctl, met_t *met0, met_t *met1, atm_t *atm,
                         depo_t *depo, const double t) {
  /* atm and grid branches of mptrac.c:8230-8275 */
  (void) depo;
  char ext[10], filename[2 * LEN];
  double r;
  int year, mon, day, hour, min, sec;
  jsec2time(t, &year, &mon, &day, &hour, &min, &sec, &r);
  if (ctl->atm_basename[0] != '-' && (fmod(t, ctl->atm_dt_out) == 0 || t == ctl->t_stop)) {
    mptrac_update_host(NULL, NULL, NULL, NULL, NULL, atm);
    sprintf(ext, ctl->atm_type_out == 0 ? "tab" : "bin");
    sprintf(filename, "%s/%s_%04d_%02d_%02d_%02d_%02d_%02d.%s", dirname, ctl->atm_basename, year, mon, day,
            hour, min, sec, ext);
    mptrac_write_atm(filename, ctl, atm, t);
  }
  if (ctl->grid_basename[0] != '-' && fmod(t, ctl->grid_dt_out) == 0) {
    sprintf(filename, "%s/%s_%04d_%02d_%02d_%02d_%02d_%02d.tab", dirname, ctl->grid_basename, year, mon, day,
            hour, min, sec);
    write_grid(filename, ctl, met0, met1, atm, t);
  }
}
