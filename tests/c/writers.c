/* Test program (tests/test_host_output.py): the analysis writers of the host layer on a particle file.
 *   writers <atm file> <out dir> <time> [KEY VALUE ...]
 * reads the particles (ATM_TYPE of the keys), takes <time> as start = stop = output time of a one-call run,
 * fills two meteo snapshots with constants (T 250 K, H2O 1e-5, O3 1e-6 on a 2 x 2 degree grid) for the two
 * writers that sample them, and calls every writer whose *_BASENAME key is set -- no device involved. */
#include "mptrac.h"

static void constant_met(met_t *met, double time) {
  met->time = time;
  met->nx = 181;
  met->ny = 91;
  met->np = 30;
  for (int i = 0; i < met->nx; i++)
    met->lon[i] = -180.0 + 2.0 * i;
  for (int j = 0; j < met->ny; j++)
    met->lat[j] = -90.0 + 2.0 * j;
  for (int k = 0; k < met->np; k++)
    met->p[k] = 1050.0 * exp(-0.3 * k);
  for (int i = 0; i < met->nx; i++)
    for (int j = 0; j < met->ny; j++)
      for (int k = 0; k < met->np; k++) {
        met->t[i][j][k] = 250.f;
        met->h2o[i][j][k] = 1e-5f;
        met->o3[i][j][k] = 1e-6f;
      }
}

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met0, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  char path[2 * LEN];
  if (argc < 4)
    return 2;
  mptrac_alloc(&ctl, &cache, &clim, &met0, &met1, &atm, &depo, &dd);
  mptrac_read_ctl("-", argc, argv, ctl);
  if (!mptrac_read_atm(argv[1], ctl, atm))
    ERRMSG("Cannot open file!");
  const double t = atof(argv[3]);
  ctl->t_start = ctl->t_stop = t;
  constant_met(met0, t - 1800.0);
  constant_met(met1, t + 1800.0);
  if (ctl->ens_basename[0] != '-') {
    sprintf(path, "%s/%s.tab", argv[2], ctl->ens_basename);
    write_ens(path, ctl, atm, t);
  }
  if (ctl->vtk_basename[0] != '-') {
    sprintf(path, "%s/%s.vtk", argv[2], ctl->vtk_basename);
    write_vtk(path, ctl, atm, t);
  }
  if (ctl->csi_basename[0] != '-') {
    sprintf(path, "%s/%s.tab", argv[2], ctl->csi_basename);
    write_csi(path, ctl, atm, t);
  }
  if (ctl->sample_basename[0] != '-') {
    sprintf(path, "%s/%s.tab", argv[2], ctl->sample_basename);
    write_sample(path, ctl, met0, met1, atm, t);
  }
  if (ctl->prof_basename[0] != '-') {
    sprintf(path, "%s/%s.tab", argv[2], ctl->prof_basename);
    write_prof(path, ctl, met0, met1, atm, t);
  }
  if (ctl->stat_basename[0] != '-') {
    sprintf(path, "%s/%s.tab", argv[2], ctl->stat_basename);
    write_station(path, ctl, atm, t);
  }
  if (ctl->atm_basename[0] != '-') {   /* particle file in the ATM_TYPE_OUT format, and read back as that */
    sprintf(path, "%s/%s", argv[2], ctl->atm_basename);
    mptrac_write_atm(path, ctl, atm, t);
    if (ctl->atm_type_out != 3) {
      static atm_t back;
      const int type_in = ctl->atm_type;
      ctl->atm_type = ctl->atm_type_out;
      if (!mptrac_read_atm(path, ctl, &back))
        ERRMSG("Cannot read the file back!");
      ctl->atm_type = type_in;
      int same = back.np == atm->np;
      for (int ip = 0; same && ip < atm->np; ip++) {
        same = back.time[ip] == atm->time[ip] && back.lon[ip] == atm->lon[ip] && back.lat[ip] == atm->lat[ip];
        for (int iq = 0; same && iq < ctl->nq; iq++)
          same = back.q[iq][ip] == atm->q[iq][ip] || (back.q[iq][ip] != back.q[iq][ip] && atm->q[iq][ip] != atm->q[iq][ip]);
        if (ctl->atm_type_out != 0)   /* (the text format stores the altitude, six digits) */
          same = same && back.p[ip] == atm->p[ip];
      }
      printf("RESULT roundtrip %s\n", same ? "identical" : "DIFFERENT");
    }
  }
  printf("RESULT done %d\n", atm->np);
  return 0;
}
