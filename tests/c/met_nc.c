/* Test program (tests/test_host_output.py): a meteo snapshot written as netCDF (MET_TYPE 0) and read back.
 *   met_nc <dir> <coord_type>
 * fills a small snapshot with a formula per field, writes <dir>/met_2001_02_03_04.nc, prints what scipy should
 * find, and -- on a Cartesian grid, which the host layer's reader accepts -- reads the file back and compares
 * every field the reader fills (values pass through float scalings both ways: 1e-6 relative). */
#include "mptrac.h"

static float value(int field, int i, int j, int k) {
  return (float) (1.0 + 0.01 * field + 0.1 * i + 0.003 * j + 0.0007 * k);
}

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met, *back;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  char path[2 * LEN];
  if (argc < 3)
    return 2;
  mptrac_alloc(&ctl, &cache, &clim, &met, &back, &atm, &depo, &dd);
  char *keys[] = { argv[0], "-", "-", "-", "MET_TYPE", "0", "MET_COORD_TYPE", argv[2], "MET_UTM_REF_LAT", "50",
    "MET_UTM_REF_LON", "10", "MET_PBL", "0", "MET_CAPE", "0" };
  mptrac_read_ctl("-", 16, keys, ctl);
  met->coord_type = atoi(argv[2]);
  met->nx = 7;
  met->ny = 5;
  met->np = 4;
  time2jsec(2001, 2, 3, 4, 0, 0, 0, &met->time);
  for (int i = 0; i < met->nx; i++)
    met->lon[i] = met->coord_type ? 500000.0 + 1000.0 * i : -10.0 + 2.0 * i;
  for (int j = 0; j < met->ny; j++)
    met->lat[j] = met->coord_type ? 5500000.0 + 1000.0 * j : 40.0 + 2.0 * j;
  for (int k = 0; k < met->np; k++)
    met->p[k] = 1000.0 - 200.0 * k;
  float (*f2[])[EY] = { met->ps, met->zs, met->ts, met->us, met->vs, met->ess, met->nss, met->shf, met->lsm, met->sst,
    met->pbl, met->pt, met->tt, met->zt, met->h2ot, met->pct, met->pcb, met->cl, met->plcl, met->plfc, met->pel,
    met->cape, met->cin, met->o3c };
  float (*f3[])[EY][EP] = { met->t, met->u, met->v, met->w, met->h2o, met->o3, met->lwc, met->rwc, met->iwc, met->swc,
    met->cc };
  for (int i = 0; i < met->nx; i++)
    for (int j = 0; j < met->ny; j++) {
      for (int f = 0; f < 24; f++)
        f2[f][i][j] = value(f, i, j, 0);
      for (int f = 0; f < 11; f++)
        for (int k = 0; k < met->np; k++)
          f3[f][i][j][k] = value(30 + f, i, j, k);
    }
  sprintf(path, "%s/met_2001_02_03_04.nc", argv[1]);
  mptrac_write_met(path, ctl, met);
  printf("RESULT written %.17g %.9g %.9g %.9g\n", met->time, (double) met->ps[2][3], (double) met->w[1][2][3],
         (double) met->h2o[6][4][0]);
  if (met->coord_type != 0) {
    if (!mptrac_read_met(path, ctl, clim, back, dd))
      ERRMSG("Cannot read the file back!");
    int bad = back->nx != met->nx || back->ny != met->ny || back->np != met->np || back->time != met->time;
    float (*b2[])[EY] = { back->ps, back->zs, back->ts, back->us, back->vs, back->ess, back->nss, back->shf, back->lsm,
      back->sst, back->pbl };
    float (*b3[])[EY][EP] = { back->t, back->u, back->v, back->w, back->h2o, back->o3, back->lwc, back->rwc, back->iwc,
      back->swc, back->cc };
    for (int i = 0; i < met->nx && !bad; i++)
      for (int j = 0; j < met->ny && !bad; j++) {
        for (int f = 0; f < 11 && !bad; f++)
          bad = fabs(b2[f][i][j] / f2[f][i][j] - 1.0) > 1e-6;
        for (int f = 0; f < 11 && !bad; f++)
          for (int k = 0; k < met->np && !bad; k++)
            bad = fabs(b3[f][i][j][k] / f3[f][i][j][k] - 1.0) > 1e-6;
      }
    for (int k = 0; k < met->np; k++)
      bad = bad || fabs(back->p[k] - met->p[k]) > 1e-9;
    printf("RESULT readback %s\n", bad ? "DIFFERENT" : "same");
  }
  return 0;
}
