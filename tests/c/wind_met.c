/* Test program (tests/test_host_driver.py): meteo files of a solid-body rotation on a global longitude / latitude
 * grid, written as netCDF (MET_TYPE 0) by mptrac_write_met -- the set-up of the reference's tests/dd_test, whose wind
 * tool produces the field (parameters of its run.sh: 360 x 181 x 60 grid, 0 ... 60 km, 50 m/s, axis tilted by 90
 * degrees, latitudes from north to south; formulae as restated in tests/refcases.py).
 *   wind_met <metbase> <first time [s]> <number of hourly files> */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  if (argc < 4)
    return 2;
  mptrac_alloc(&ctl, &cache, &clim, &met, &met1, &atm, &depo, &dd);
  char *keys[] = { argv[0], "-", "-", "-", "MET_TYPE", "0", "METBASE", argv[1] };
  mptrac_read_ctl("-", 8, keys, ctl);
  const int nx = 360, ny = 181, nz = 60;
  const double speed = 50.0, alpha = DEG2RAD(90.0);
  met->coord_type = 0;
  met->nx = nx;
  met->ny = ny;
  met->np = nz;
  for (int i = 0; i < nx; i++)
    met->lon[i] = 360.0 / nx * i;
  for (int j = 0; j < ny; j++)
    met->lat[j] = -(180.0 / (ny - 1) * j - 90.0);
  for (int k = 0; k < nz; k++)
    met->p[k] = P(60.0 / (nz - 1.0) * k);
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++) {
      const double la = DEG2RAD(met->lat[j]), lo = DEG2RAD(met->lon[i]);
      met->ps[i][j] = 1013.25f;
      met->pbl[i][j] = (float) P(1.0);
      for (int k = 0; k < nz; k++) {
        met->t[i][j][k] = 280.f;
        met->u[i][j][k] = (float) (speed * (cos(la) * cos(alpha) + sin(la) * cos(lo) * sin(alpha)));
        met->v[i][j][k] = (float) (-speed * sin(lo) * sin(alpha));
        met->w[i][j][k] = 0.f;
      }
    }
  const double t0 = atof(argv[2]);
  for (int h = 0; h < atoi(argv[3]); h++) {
    char path[2 * LEN];
    int year, mon, day, hour, min, sec;
    double r;
    met->time = t0 + 3600.0 * h;
    jsec2time(met->time, &year, &mon, &day, &hour, &min, &sec, &r);
    sprintf(path, "%s_%d_%02d_%02d_%02d.nc", argv[1], year, mon, day, hour);
    mptrac_write_met(path, ctl, met);
  }
  printf("RESULT done\n");
  return 0;
}
