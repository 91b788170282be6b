/*
 * output.c -- the analysis outputs of mptrac_write_output besides particle and gridded files: verification
 * against observations (CSI), ensemble means, vertical profiles, samples at observation points, station
 * crossings and VTK point clouds (reference interface: src/mptrac.h write_csi ... write_vtk; file formats as
 * documented in docs/manual and the reference's tests/trac_test goldens).
 *
 * These run on the host, on the particle arrays mptrac_write_output has just downloaded -- as in the reference,
 * whose writers are host code behind mptrac_update_host.  They are output post-processing, not part of the
 * time-step loop; nothing here is called between two device kernels of a step.
 *
 * The implementation is this repository's own: one regular lon / lat / z box grid type serves the CSI and
 * profile binning, one observation table type with a cursor serves the three observation-driven writers, and
 * the verification statistics are written from their textbook definitions (the reference calls GSL).
 */
#define _GNU_SOURCE
#include "mptrac.h"

#include <stddef.h>

/* the writers below are defined on longitude / latitude grids only (the reference refuses the others too) */
static void latlon_only(const ctl_t *ctl) {
  if (ctl->met_coord_type != 0)
    ERRMSG("Only lat/lon grid supported");
}

static FILE *create_text_file(const char *filename) {
  FILE *f = fopen(filename, "w");
  if (!f)
    ERRMSG("Cannot create file!");
  return f;
}

/* particle ip belongs to the time step around t: its time lies within half a step of t */
typedef struct {
  double t0, t1;
} step_window;

static step_window window_around(const ctl_t *ctl, double t) {
  const step_window w = { t - 0.5 * ctl->dt_mod, t + 0.5 * ctl->dt_mod };
  return w;
}

static int inside(const step_window *w, double time) {
  return !(time < w->t0 || time > w->t1);
}

/* ---------------------------------------------------------------------------------------------------------- */
/* geometry, kernels, observations                                                                            */
/* ---------------------------------------------------------------------------------------------------------- */

void geo2cart(const double z, const double lon, const double lat, double *x) {
  const double r = RE + z, phi = DEG2RAD(lat), lam = DEG2RAD(lon);
  x[0] = r * cos(phi) * cos(lam);
  x[1] = r * cos(phi) * sin(lam);
  x[2] = r * sin(phi);
}

void cart2geo(const double *x, double *z, double *lon, double *lat) {
  const double r = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  *lat = RAD2DEG(asin(x[2] / r));
  *lon = RAD2DEG(atan2(x[1], x[0]));
  *z = r - RE;
}

static double dist2(const double *a, const double *b) {
  double s = 0;
  for (int k = 0; k < 3; k++)
    s += (a[k] - b[k]) * (a[k] - b[k]);
  return s;
}

/* vertical weighting function: (height [km], weight) pairs with ascending heights, scaled so that the largest
 * weight is one */
void read_kernel(const char *filename, double kz[EP], double kw[EP], int *nk) {
  LOG(1, "Read kernel function: %s", filename);
  FILE *in = fopen(filename, "r");
  if (!in)
    ERRMSG("Cannot open file!");
  char line[LEN];
  int n = 0;
  double top = -INFINITY;
  while (fgets(line, LEN, in)) {
    double z, w;
    if (sscanf(line, "%lg %lg", &z, &w) != 2)
      continue;
    if (n > 0 && z < kz[n - 1])
      ERRMSG("Height levels must be ascending!");
    kz[n] = z;
    kw[n] = w;
    top = fmax(top, w);
    if (++n >= EP)
      ERRMSG("Too many height levels!");
  }
  fclose(in);
  if (n < 2)
    ERRMSG("Not enough height levels!");
  for (int i = 0; i < n; i++)
    kw[i] /= top;
  *nk = n;
}

/* weight at pressure p: linear in log-pressure height between the nodes, constant beyond them; no kernel: 1 */
double kernel_weight(const double kz[EP], const double kw[EP], const int nk, const double p) {
  if (nk < 2)
    return 1.0;
  const double z = Z(p);
  if (z < kz[0])
    return kw[0];
  if (z > kz[nk - 1])
    return kw[nk - 1];
  int lo = 0, hi = nk - 1;   /* kz[lo] <= z <= kz[hi] */
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if (kz[mid] > z)
      hi = mid;
    else
      lo = mid;
  }
  return kw[lo] + (kw[lo + 1] - kw[lo]) / (kz[lo + 1] - kz[lo]) * (z - kz[lo]);
}

static void log_range(const char *what, const char *unit, const double *x, int n) {
  double lo = x[0], hi = x[0];
  for (int i = 1; i < n; i++) {
    lo = fmin(lo, x[i]);
    hi = fmax(hi, x[i]);
  }
  LOG(2, "%s range: %g ... %g%s%s", what, lo, hi, unit[0] ? " " : "", unit);
}

/* observations: time [s], altitude [km], longitude, latitude [deg], value; ascending in time.
 * OBS_TYPE 0: text lines of five numbers; OBS_TYPE 1: a netCDF file with dimension nobs and the variables
 * time, alt, lon, lat, obs (classic format here) */
void read_obs(const char *filename, const ctl_t *ctl, double *rt, double *rz, double *rlon, double *rlat,
              double *robs, int *nobs) {
  LOG(1, "Read observation data: %s", filename);
  *nobs = 0;
  if (ctl->obs_type == 0) {
    FILE *in = fopen(filename, "r");
    if (!in)
      ERRMSG("Cannot open file!");
    char line[LEN];
    int n = 0;
    while (fgets(line, LEN, in))
      if (sscanf(line, "%lg %lg %lg %lg %lg", &rt[n], &rz[n], &rlon[n], &rlat[n], &robs[n]) == 5 && ++n >= NOBS)
        ERRMSG("Too many observations!");
    fclose(in);
    *nobs = n;
  } else if (ctl->obs_type == 1) {
    *nobs = mptrac_amd_read_obs_nc(filename, rt, rz, rlon, rlat, robs);
  } else
    ERRMSG("Set OBS_TYPE to 0 or 1!");
  for (int i = 1; i < *nobs; i++)
    if (rt[i] < rt[i - 1])
      ERRMSG("Time must be ascending!");
  LOG(2, "Number of observations: %d", *nobs);
  if (*nobs > 0) {
    LOG(2, "Time range: %.2f ... %.2f s", rt[0], rt[*nobs - 1]);
    log_range("Altitude", "km", rz, *nobs);
    log_range("Longitude", "deg", rlon, *nobs);
    log_range("Latitude", "deg", rlat, *nobs);
    log_range("Observation", "", robs, *nobs);
  }
}

typedef struct {
  double *t, *z, *lon, *lat, *val;
  int n;
} obs_table;

static void obs_load(obs_table *o, const char *filename, const ctl_t *ctl) {
  ALLOC(o->t, double, NOBS);
  ALLOC(o->z, double, NOBS);
  ALLOC(o->lon, double, NOBS);
  ALLOC(o->lat, double, NOBS);
  ALLOC(o->val, double, NOBS);
  read_obs(filename, ctl, o->t, o->z, o->lon, o->lat, o->val, &o->n);
}

static void obs_free(obs_table *o) {
  free(o->t);
  free(o->z);
  free(o->lon);
  free(o->lat);
  free(o->val);
  memset(o, 0, sizeof(*o));
}

/* ---------------------------------------------------------------------------------------------------------- */
/* a regular box grid (CSI, profiles)                                                                         */
/* ---------------------------------------------------------------------------------------------------------- */

typedef struct {
  double lon0, lon1, lat0, lat1, z0, z1;
  int nx, ny, nz;
  double dlon, dlat, dz;
  double *area;   /* [ny], km^2 */
} box_grid;

static void box_grid_init(box_grid *g, double lon0, double lon1, int nx, double lat0, double lat1, int ny, double z0,
                          double z1, int nz) {
  g->lon0 = lon0;
  g->lon1 = lon1;
  g->lat0 = lat0;
  g->lat1 = lat1;
  g->z0 = z0;
  g->z1 = z1;
  g->nx = nx;
  g->ny = ny;
  g->nz = nz;
  g->dlon = (lon1 - lon0) / nx;
  g->dlat = (lat1 - lat0) / ny;
  g->dz = (z1 - z0) / nz;
  ALLOC(g->area, double, ny);
  for (int iy = 0; iy < ny; iy++)
    g->area[iy] = g->dlat * g->dlon * SQR(RE * M_PI / 180.) * cos(DEG2RAD(lat0 + g->dlat * (iy + 0.5)));
}

/* index of the column of (lon, lat), of the box of (lon, lat, z); -1 outside (upper bounds exclusive) */
static long box_column(const box_grid *g, double lon, double lat) {
  if (lon < g->lon0 || lon >= g->lon1 || lat < g->lat0 || lat >= g->lat1)
    return -1;
  const int ix = (int) ((lon - g->lon0) / g->dlon), iy = (int) ((lat - g->lat0) / g->dlat);
  if (ix >= g->nx || iy >= g->ny)
    return -1;
  return (long) ARRAY_2D(ix, iy, g->ny);
}

static long box_cell(const box_grid *g, double lon, double lat, double z) {
  const long col = box_column(g, lon, lat);
  if (col < 0 || z < g->z0 || z >= g->z1)
    return -1;
  const int iz = (int) ((z - g->z0) / g->dz);
  if (iz >= g->nz)
    return -1;
  return col * g->nz + iz;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* verification statistics                                                                                    */
/* ---------------------------------------------------------------------------------------------------------- */

static double pearson(const double *x, const double *y, int n) {
  double mx = 0, my = 0;
  for (int i = 0; i < n; i++) {
    mx += x[i];
    my += y[i];
  }
  mx /= n;
  my /= n;
  double sxx = 0, syy = 0, sxy = 0;
  for (int i = 0; i < n; i++) {
    sxx += (x[i] - mx) * (x[i] - mx);
    syy += (y[i] - my) * (y[i] - my);
    sxy += (x[i] - mx) * (y[i] - my);
  }
  return sxy / (sqrt(sxx) * sqrt(syy));
}

static const double *g_rank_key;
static int by_key(const void *a, const void *b) {
  const double x = g_rank_key[*(const int *) a], y = g_rank_key[*(const int *) b];
  return (x > y) - (x < y);
}

/* ranks 1 ... n, ties share the mean of their ranks */
static void fractional_ranks(const double *x, int n, double *rank) {
  int *order = malloc((size_t) n * sizeof(int));
  if (!order)
    ERRMSG("Out of memory!");
  for (int i = 0; i < n; i++)
    order[i] = i;
  g_rank_key = x;
  qsort(order, (size_t) n, sizeof(int), by_key);
  for (int i = 0; i < n;) {
    int j = i;
    while (j + 1 < n && x[order[j + 1]] == x[order[i]])
      j++;
    for (int k = i; k <= j; k++)
      rank[order[k]] = 0.5 * (i + j) + 1.0;
    i = j + 1;
  }
  free(order);
}

static double spearman(const double *x, const double *y, int n) {
  double *rx = malloc(2 * (size_t) n * sizeof(double));
  if (!rx)
    ERRMSG("Out of memory!");
  fractional_ranks(x, n, rx);
  fractional_ranks(y, n, rx + n);
  const double r = pearson(rx, rx + n, n);
  free(rx);
  return r;
}

/* ---------------------------------------------------------------------------------------------------------- */
/* CSI                                                                                                        */
/* ---------------------------------------------------------------------------------------------------------- */

/* Model columns against gridded observations, one line per output time (and ensemble member): the contingency
 * counts of "observed above CSI_OBSMIN" vs "modelled above CSI_MODMIN" over the boxes that hold observations,
 * the scores derived from them, and error statistics of the (model, observation) pairs in which at least one
 * side is above its threshold.  Only the boxes of the call at an output time enter (the counters start from
 * zero in every call). */
void write_csi(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  static FILE *out;
  static obs_table obs;
  static box_grid grid;
  static double kz[EP], kw[EP];
  static int nk;
  /* pairs of the statistics: model, observation, observation spread.  The spread of a pair is only stored
   * when the model is above its threshold and the three arrays are cleared after an output, not per call --
   * so a pair below the threshold keeps what an earlier call without output left at its place (as the
   * reference's static arrays do, mptrac.c:13383-13389, 13440-13444) */
  static double *px, *py, *psd;

  latlon_only(ctl);
  if (ctl->qnt_m < 0)
    ERRMSG("Need quantity mass!");
  const int members = ctl->nens > 0 ? ctl->nens : 1;
  if (ctl->nens > 0 && ctl->qnt_ens < 0)
    ERRMSG("Missing ensemble IDs!");
  if (ctl->nens > NENS)
    ERRMSG("Too many ensembles!");

  if (t == ctl->t_start) {
    obs_load(&obs, ctl->csi_obsfile, ctl);
    nk = 0;
    if (ctl->csi_kernel[0] != '-')
      read_kernel(ctl->csi_kernel, kz, kw, &nk);
    LOG(1, "Write CSI%s data: %s", ctl->nens > 0 ? " ensemble" : "", filename);
    out = create_text_file(filename);
    static const char *const legend[] = { "time [s]", "ensemble ID", "number of hits (cx)", "number of misses (cy)",
      "number of false alarms (cz)", "number of observations (cx + cy)", "number of forecasts (cx + cz)", "bias (%)",
      "POD (%)", "FAR (%)", "CSI (%)", "hits by random chance", "ETS (%)", "Pearson R", "Spearman R",
      "mean error [kg/m²]", "RMSE [kg/m²]", "MAE [kg/m²]", "log-likelihood", "number of points" };
    for (int k = 0; k < 20; k++)
      fprintf(out, "# $%d = %s\n", k + 1, legend[k]);
    fputc('\n', out);
    box_grid_init(&grid, ctl->csi_lon0, ctl->csi_lon1, ctl->csi_nx, ctl->csi_lat0, ctl->csi_lat1, ctl->csi_ny,
                  ctl->csi_z0, ctl->csi_z1, ctl->csi_nz);
    ALLOC(px, double, NCSI);
    ALLOC(py, double, NCSI);
    ALLOC(psd, double, NCSI);
  }
  if (!out)
    ERRMSG("write_csi was not called at the start time of the run!");

  const step_window now = window_around(ctl, t);
  const double t0 = now.t0, t1 = now.t1;
  const size_t ncell = (size_t) grid.nx * (size_t) grid.ny * (size_t) grid.nz;
  double *model, *omean, *osq;
  int *ocount;
  ALLOC(model, double, (size_t) members * ncell);
  ALLOC(omean, double, ncell);
  ALLOC(osq, double, ncell);
  ALLOC(ocount, int, ncell);

  /* observations of this time step per box: mean and spread */
  for (int i = 0; i < obs.n; i++) {
    if (obs.t[i] < t0 || obs.t[i] >= t1 || !isfinite(obs.val[i]))
      continue;
    const long c = box_cell(&grid, obs.lon[i], obs.lat[i], obs.z[i]);
    if (c < 0)
      continue;
    omean[c] += obs.val[i];
    osq[c] += SQR(obs.val[i]);
    ocount[c]++;
  }
  for (size_t c = 0; c < ncell; c++)
    if (ocount[c]) {
      omean[c] /= ocount[c];
      osq[c] = sqrt(osq[c] / ocount[c] - SQR(omean[c]));
    }

  /* (kernel-weighted) mass per box and member -> column density [kg/m^2] */
  for (int ip = 0; ip < atm->np; ip++) {
    if (!inside(&now, atm->time[ip]))
      continue;
    const int member = ctl->nens > 0 ? (int) atm->q[ctl->qnt_ens][ip] : 0;
    if (member < 0 || member >= members)
      ERRMSG("Ensemble ID out of range!");
    const long c = box_cell(&grid, atm->lon[ip], atm->lat[ip], Z(atm->p[ip]));
    if (c >= 0)
      model[(size_t) member * ncell + (size_t) c] += kernel_weight(kz, kw, nk, atm->p[ip]) * atm->q[ctl->qnt_m][ip];
  }

  for (int e = 0; e < members; e++) {
    int boxes = 0, hits = 0, misses = 0, alarms = 0, n = 0;
    for (size_t c = 0; c < ncell; c++) {
      double *m = &model[(size_t) e * ncell + c];
      if (*m > 0)
        *m /= 1e6 * grid.area[(c / (size_t) grid.nz) % (size_t) grid.ny];
      if (!ocount[c])
        continue;
      boxes++;
      const int seen = omean[c] >= ctl->csi_obsmin, forecast = *m >= ctl->csi_modmin;
      hits += seen && forecast;
      misses += seen && !forecast;
      alarms += !seen && forecast;
      if (seen || forecast) {
        px[n] = *m;
        py[n] = omean[c];
        if (forecast)
          psd[n] = osq[c];
        if (++n >= NCSI)
          ERRMSG("Too many points for statistics!");
      }
    }
    if (fmod(t, ctl->csi_dt_out) != 0 || n == 0)
      continue;
    const int n_obs = hits + misses, n_for = hits + alarms, any = hits + misses + alarms;
    const double chance = boxes > 0 ? (1. * n_obs * n_for) / boxes : NAN;
    double sum = 0, sq = 0, absolute = 0, tss = 0;
    for (int i = 0; i < n; i++) {
      const double d = px[i] - py[i];
      sum += d;
      sq += d * d;
      absolute += fabs(d);
      if (psd[i] != 0)
        tss += SQR(d / psd[i]);
    }
    fprintf(out, "%.2f %d %d %d %d %d %d %g %g %g %g %g %g %g %g %g %g %g %g %d\n", t, ctl->nens > 0 ? e : -999, hits,
            misses, alarms, n_obs, n_for, n_obs > 0 ? 100. * n_for / n_obs : NAN, n_obs > 0 ? 100. * hits / n_obs : NAN,
            n_for > 0 ? 100. * alarms / n_for : NAN, any > 0 ? 100. * hits / any : NAN, chance,
            any - chance > 0 ? 100. * (hits - chance) / (any - chance) : NAN, pearson(px, py, n), spearman(px, py, n),
            sum / n, sqrt(sq / n), absolute / n, -0.5 * tss, n);
    memset(px, 0, (size_t) n * sizeof(double));
    memset(py, 0, (size_t) n * sizeof(double));
    memset(psd, 0, (size_t) n * sizeof(double));
  }
  free(model);
  free(omean);
  free(osq);
  free(ocount);

  if (t == ctl->t_stop) {
    fclose(out);
    out = NULL;
    obs_free(&obs);
    free(grid.area);
    free(px);
    free(py);
    free(psd);
    px = py = psd = NULL;
  }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* ensembles                                                                                                  */
/* ---------------------------------------------------------------------------------------------------------- */

/* Mean position (via Cartesian coordinates), mean altitude, mean and standard deviation of every quantity.
 * As in the reference (mptrac.c:13524-13531) the accumulators are addressed by the SLOT of the
 * quantity "ens", not by the member number a particle carries: the file holds one row, the statistics of all
 * particles of the time step.  Reproduced on purpose; the member numbers are still range-checked. */
void write_ens(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  latlon_only(ctl);
  if (ctl->qnt_ens < 0)
    ERRMSG("Missing ensemble IDs!");
  const step_window now = window_around(ctl, t);
  const int row = ctl->qnt_ens;
  double sum_x[3] = { 0, 0, 0 }, sum_z = 0, sum_q[NQ], sum_qq[NQ];
  int n = 0;
  for (int iq = 0; iq < ctl->nq; iq++)
    sum_q[iq] = sum_qq[iq] = 0;
  for (int ip = 0; ip < atm->np; ip++) {
    if (!inside(&now, atm->time[ip]))
      continue;
    if (atm->q[ctl->qnt_ens][ip] < 0 || atm->q[ctl->qnt_ens][ip] >= NENS)
      ERRMSG("Ensemble ID is out of range!");
    double x[3];
    geo2cart(0, atm->lon[ip], atm->lat[ip], x);
    for (int k = 0; k < 3; k++)
      sum_x[k] += x[k];
    sum_z += Z(atm->p[ip]);
    for (int iq = 0; iq < ctl->nq; iq++) {
      sum_q[iq] += atm->q[iq][ip];
      sum_qq[iq] += SQR(atm->q[iq][ip]);
    }
    n++;
  }

  LOG(1, "Write ensemble data: %s", filename);
  FILE *out = create_text_file(filename);
  fprintf(out, "# $1 = time [s]\n# $2 = altitude [km]\n# $3 = longitude [deg]\n# $4 = latitude [deg]\n");
  int col = 4;
  for (int pass = 0; pass < 2; pass++)
    for (int iq = 0; iq < ctl->nq; iq++)
      fprintf(out, "# $%d = %s (%s) [%s]\n", ++col, ctl->qnt_name[iq], pass ? "sigma" : "mean", ctl->qnt_unit[iq]);
  fprintf(out, "# $%d = number of members\n\n", ++col);
  if (n > 0 && row < NENS) {
    double zdummy, lon, lat;
    cart2geo(sum_x, &zdummy, &lon, &lat);
    fprintf(out, "%.2f %g %g %g", t, sum_z / n, lon, lat);
    for (int pass = 0; pass < 2; pass++)
      for (int iq = 0; iq < ctl->nq; iq++) {
        const double mean = sum_q[iq] / n, var = sum_qq[iq] / n - SQR(mean);
        fputc(' ', out);
        fprintf(out, ctl->qnt_format[iq], pass ? (var > 0 ? sqrt(var) : 0) : mean);
      }
    fprintf(out, " %d\n", n);
  }
  fclose(out);
}

/* ---------------------------------------------------------------------------------------------------------- */
/* profiles                                                                                                   */
/* ---------------------------------------------------------------------------------------------------------- */

/* For every column of the profile grid that holds observations of this time step and any particle mass: the
 * volume mixing ratio the mass of each box corresponds to, with the temperature, water vapour and ozone of the
 * meteo data at the box centres, next to the mean observation of the column. */
void write_prof(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm, const double t) {
  static FILE *out;
  static obs_table obs;
  static box_grid grid;

  latlon_only(ctl);
  if (t == ctl->t_start) {
    if (ctl->qnt_m < 0)
      ERRMSG("Need quantity mass!");
    if (ctl->molmass <= 0)
      ERRMSG("Specify molar mass!");
    obs_load(&obs, ctl->prof_obsfile, ctl);
    LOG(1, "Write profile data: %s", filename);
    out = create_text_file(filename);
    static const char *const legend[] = { "time [s]", "altitude [km]", "longitude [deg]", "latitude [deg]",
      "pressure [hPa]", "temperature [K]", "volume mixing ratio [ppv]", "H2O volume mixing ratio [ppv]",
      "O3 volume mixing ratio [ppv]", "observed BT index [K]", "number of observations" };
    for (int k = 0; k < 11; k++)
      fprintf(out, "# $%d = %s\n", k + 1, legend[k]);
    box_grid_init(&grid, ctl->prof_lon0, ctl->prof_lon1, ctl->prof_nx, ctl->prof_lat0, ctl->prof_lat1, ctl->prof_ny,
                  ctl->prof_z0, ctl->prof_z1, ctl->prof_nz);
  }
  if (!out)
    ERRMSG("write_prof was not called at the start time of the run!");

  const step_window now = window_around(ctl, t);
  const double t0 = now.t0, t1 = now.t1;
  const size_t ncol = (size_t) grid.nx * (size_t) grid.ny;
  double *mass, *osum;
  int *ocount;
  ALLOC(mass, double, ncol * (size_t) grid.nz);
  ALLOC(osum, double, ncol);
  ALLOC(ocount, int, ncol);
  for (int i = 0; i < obs.n && obs.t[i] < t1; i++) {
    if (obs.t[i] < t0 || !isfinite(obs.val[i]))
      continue;
    const long c = box_column(&grid, obs.lon[i], obs.lat[i]);
    if (c >= 0) {
      osum[c] += obs.val[i];
      ocount[c]++;
    }
  }
  for (int ip = 0; ip < atm->np; ip++) {
    if (!inside(&now, atm->time[ip]))
      continue;
    const long c = box_cell(&grid, atm->lon[ip], atm->lat[ip], Z(atm->p[ip]));
    if (c >= 0)
      mass[c] += atm->q[ctl->qnt_m][ip];
  }
  for (size_t c = 0; c < ncol; c++) {
    if (ocount[c] <= 0)
      continue;
    int loaded = 0;
    for (int iz = 0; iz < grid.nz && !loaded; iz++)
      loaded = mass[c * (size_t) grid.nz + (size_t) iz] > 0;
    if (!loaded)
      continue;
    const int ix = (int) (c / (size_t) grid.ny), iy = (int) (c % (size_t) grid.ny);
    const double lon = grid.lon0 + grid.dlon * (ix + 0.5), lat = grid.lat0 + grid.dlat * (iy + 0.5);
    fputc('\n', out);
    for (int iz = 0; iz < grid.nz; iz++) {
      const double z = grid.z0 + grid.dz * (iz + 0.5), press = P(z);
      const double temp = mptrac_amd_intpol_3d(met0, met1, offsetof(met_t, t), t, press, lon, lat);
      const double h2o = mptrac_amd_intpol_3d(met0, met1, offsetof(met_t, h2o), t, press, lon, lat);
      const double o3 = mptrac_amd_intpol_3d(met0, met1, offsetof(met_t, o3), t, press, lon, lat);
      const double air = 100. * press / (RA * temp);   /* density [kg/m^3] */
      const double vmr = MA / ctl->molmass * mass[c * (size_t) grid.nz + (size_t) iz] / (air * grid.area[iy] * grid.dz * 1e9);
      fprintf(out, "%.2f %g %g %g %g %g %g %g %g %g %d\n", t, z, lon, lat, press, temp, vmr, h2o, o3,
              osum[c] / ocount[c], ocount[c]);
    }
  }
  free(mass);
  free(osum);
  free(ocount);
  if (t == ctl->t_stop) {
    fclose(out);
    out = NULL;
    obs_free(&obs);
    free(grid.area);
  }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* samples                                                                                                    */
/* ---------------------------------------------------------------------------------------------------------- */

/* For every observation of this time step: particles and (kernel-weighted) mass inside a cylinder of radius
 * SAMPLE_DX [km] (and half depth SAMPLE_DZ [km], if positive) around it -> column density, and the volume
 * mixing ratio it stands for when a molar mass and a depth are given. */
void write_sample(const char *filename, const ctl_t *ctl, met_t *met0, met_t *met1, const atm_t *atm, const double t) {
  static FILE *out;
  static obs_table obs;
  static double kz[EP], kw[EP];
  static int nk;

  latlon_only(ctl);
  if (t == ctl->t_start) {
    obs_load(&obs, ctl->sample_obsfile, ctl);
    nk = 0;
    if (ctl->sample_kernel[0] != '-')
      read_kernel(ctl->sample_kernel, kz, kw, &nk);
    LOG(1, "Write sample data: %s", filename);
    out = create_text_file(filename);
    static const char *const legend[] = { "time [s]", "altitude [km]", "longitude [deg]", "latitude [deg]",
      "surface area [km^2]", "layer depth [km]", "number of particles [1]", "column density [kg/m^2]",
      "volume mixing ratio [ppv]", "observed BT index [K]" };
    for (int k = 0; k < 10; k++)
      fprintf(out, "# $%d = %s\n", k + 1, legend[k]);
    fputc('\n', out);
  }
  if (!out)
    ERRMSG("write_sample was not called at the start time of the run!");

  const step_window now = window_around(ctl, t);
  const double t0 = now.t0, t1 = now.t1;
  const double reach2 = SQR(ctl->sample_dx), area = M_PI * reach2;
  const double reach_lat = ctl->sample_dx * 180. / (M_PI * RE);   /* the radius in degrees of latitude */
  for (int i = 0; i < obs.n && obs.t[i] < t1; i++) {
    if (obs.t[i] < t0)
      continue;
    double centre[3];
    geo2cart(0, obs.lon[i], obs.lat[i], centre);
    const double p_obs = P(obs.z[i]), p_top = P(obs.z[i] + ctl->sample_dz), p_bottom = P(obs.z[i] - ctl->sample_dz);
    double mass = 0;
    int count = 0;
    for (int ip = 0; ip < atm->np; ip++) {
      if (atm->time[ip] < t0 || atm->time[ip] > t1 || fabs(obs.lat[i] - atm->lat[ip]) > reach_lat)
        continue;
      double x[3];
      geo2cart(0, atm->lon[ip], atm->lat[ip], x);
      if (dist2(centre, x) > reach2)
        continue;
      if (ctl->sample_dz > 0 && (atm->p[ip] > p_bottom || atm->p[ip] < p_top))
        continue;
      if (ctl->qnt_m >= 0)
        mass += kernel_weight(kz, kw, nk, atm->p[ip]) * atm->q[ctl->qnt_m][ip];
      count++;
    }
    const double cd = mass / (1e6 * area);
    double vmr = NAN;
    if (ctl->molmass > 0 && ctl->sample_dz > 0) {
      vmr = 0;
      if (mass > 0) {
        const double temp = mptrac_amd_intpol_3d(met0, met1, offsetof(met_t, t), obs.t[i], p_obs, obs.lon[i], obs.lat[i]);
        vmr = MA / ctl->molmass * cd / (100. * p_obs / (RA * temp) * ctl->sample_dz * 1e3);
      }
    }
    fprintf(out, "%.2f %g %g %g %g %g %d %g %g %g\n", obs.t[i], obs.z[i], obs.lon[i], obs.lat[i], area, ctl->sample_dz,
            count, cd, vmr, obs.val[i]);
  }
  if (t == ctl->t_stop) {
    fclose(out);
    out = NULL;
    obs_free(&obs);
  }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* station                                                                                                    */
/* ---------------------------------------------------------------------------------------------------------- */

/* Particles that come within STAT_R [km] of the station (horizontal distance) between STAT_T0 and STAT_T1 are
 * listed when they do; with a quantity "stat" each particle is listed once (the flag is set on the host copy:
 * mptrac_write_output hands the change back to the device). */
void write_station(const char *filename, const ctl_t *ctl, atm_t *atm, const double t) {
  static FILE *out;
  static double station[3];

  latlon_only(ctl);
  if (t == ctl->t_start) {
    LOG(1, "Write station data: %s", filename);
    out = create_text_file(filename);
    fprintf(out, "# $1 = time [s]\n# $2 = altitude [km]\n# $3 = longitude [deg]\n# $4 = latitude [deg]\n");
    for (int iq = 0; iq < ctl->nq; iq++)
      fprintf(out, "# $%i = %s [%s]\n", iq + 5, ctl->qnt_name[iq], ctl->qnt_unit[iq]);
    fputc('\n', out);
    geo2cart(0, ctl->stat_lon, ctl->stat_lat, station);
  }
  if (!out)
    ERRMSG("write_station was not called at the start time of the run!");
  const double t0 = t - 0.5 * ctl->dt_mod, t1 = t + 0.5 * ctl->dt_mod, reach2 = SQR(ctl->stat_r);
  for (int ip = 0; ip < atm->np; ip++) {
    const double tp = atm->time[ip];
    if (tp < t0 || tp > t1 || tp < ctl->stat_t0 || tp > ctl->stat_t1)
      continue;
    if (ctl->qnt_stat >= 0 && (int) atm->q[ctl->qnt_stat][ip])
      continue;
    double x[3];
    geo2cart(0, atm->lon[ip], atm->lat[ip], x);
    if (dist2(station, x) > reach2)
      continue;
    if (ctl->qnt_stat >= 0)
      atm->q[ctl->qnt_stat][ip] = 1;
    fprintf(out, "%.2f %g %g %g", tp, Z(atm->p[ip]), atm->lon[ip], atm->lat[ip]);
    for (int iq = 0; iq < ctl->nq; iq++) {
      fputc(' ', out);
      fprintf(out, ctl->qnt_format[iq], atm->q[iq][ip]);
    }
    fputc('\n', out);
  }
  if (t == ctl->t_stop) {
    fclose(out);
    out = NULL;
  }
}

/* ---------------------------------------------------------------------------------------------------------- */
/* VTK                                                                                                        */
/* ---------------------------------------------------------------------------------------------------------- */

/* every VTK_STRIDE-th particle of this time step as a legacy-VTK point cloud (lon / lat / scaled height, or on
 * a sphere) with the quantities as point data */
void write_vtk(const char *filename, const ctl_t *ctl, const atm_t *atm, const double t) {
  latlon_only(ctl);
  LOG(1, "Write VTK data: %s", filename);
  const step_window now = window_around(ctl, t);
  const double t0 = now.t0, t1 = now.t1;
  const int stride = ctl->vtk_stride > 0 ? ctl->vtk_stride : 1;
  int *pick, n = 0;
  ALLOC(pick, int, atm->np / stride + 1);
  for (int ip = 0; ip < atm->np; ip += stride)
    if (atm->time[ip] >= t0 && atm->time[ip] <= t1)
      pick[n++] = ip;
  FILE *out = create_text_file(filename);
  fprintf(out, "# vtk DataFile Version 3.0\nvtk output\nASCII\nDATASET POLYDATA\nPOINTS %d float\n", n);
  for (int k = 0; k < n; k++) {
    const int ip = pick[k];
    const double height = Z(atm->p[ip]) * ctl->vtk_scale + ctl->vtk_offset;
    if (ctl->vtk_sphere) {
      const double r = (RE + height) / RE, phi = DEG2RAD(atm->lat[ip]), lam = DEG2RAD(atm->lon[ip]);
      fprintf(out, "%g %g %g\n", r * cos(phi) * cos(lam), r * cos(phi) * sin(lam), r * sin(phi));
    } else
      fprintf(out, "%g %g %g\n", atm->lon[ip], atm->lat[ip], height);
  }
  fprintf(out, "POINT_DATA %d\n", n);
  for (int iq = 0; iq < ctl->nq; iq++) {
    fprintf(out, "SCALARS %s float 1\nLOOKUP_TABLE default\n", ctl->qnt_name[iq]);
    for (int k = 0; k < n; k++)
      fprintf(out, "%g\n", atm->q[iq][pick[k]]);
  }
  fclose(out);
  free(pick);
}
