/* Test program (tests/test_host_output.py): the zonal-mean climatologies of mptrac_read_clim.
 *   clim_zm <out file> [KEY VALUE ...]
 * reads the control parameters and the climatologies the requested quantities need, and writes for each of the
 * five tables "name ntime np nlat" followed by its time, pressure and latitude axes and the mixing ratios in
 * index order [time][p][lat], then for each of the five trace-gas time series "name ntime", its times and its
 * mixing ratios; 17 significant digits.  No device involved. */
#include "mptrac.h"

static void dump(FILE *out, const char *name, const clim_zm_t *zm) {
  fprintf(out, "%s %d %d %d\n", name, zm->ntime, zm->np, zm->nlat);
  for (int i = 0; i < zm->ntime; i++)
    fprintf(out, "%.17g ", zm->time[i]);
  fputc('\n', out);
  for (int i = 0; i < zm->np; i++)
    fprintf(out, "%.17g ", zm->p[i]);
  fputc('\n', out);
  for (int i = 0; i < zm->nlat; i++)
    fprintf(out, "%.17g ", zm->lat[i]);
  fputc('\n', out);
  for (int it = 0; it < zm->ntime; it++)
    for (int iz = 0; iz < zm->np; iz++)
      for (int iy = 0; iy < zm->nlat; iy++)
        fprintf(out, "%.17g ", zm->vmr[it][iz][iy]);
  fputc('\n', out);
}

static void dump_ts(FILE *out, const char *name, const clim_ts_t *ts) {
  fprintf(out, "%s %d\n", name, ts->ntime);
  for (int i = 0; i < ts->ntime; i++)
    fprintf(out, "%.17g ", ts->time[i]);
  fputc('\n', out);
  for (int i = 0; i < ts->ntime; i++)
    fprintf(out, "%.17g ", ts->vmr[i]);
  fputc('\n', out);
}

int main(int argc, char *argv[]) {
  static ctl_t ctl;
  clim_t *clim;
  if (argc < 2)
    return 2;
  ALLOC(clim, clim_t, 1);
  mptrac_read_ctl("-", argc, argv, &ctl);
  mptrac_read_clim(&ctl, clim);
  FILE *out = fopen(argv[1], "w");
  if (!out)
    ERRMSG("Cannot create file!");
  dump(out, "hno3", &clim->hno3);
  dump(out, "oh", &clim->oh);
  dump(out, "h2o2", &clim->h2o2);
  dump(out, "ho2", &clim->ho2);
  dump(out, "o1d", &clim->o1d);
  dump_ts(out, "ccl4", &clim->ccl4);
  dump_ts(out, "ccl3f", &clim->ccl3f);
  dump_ts(out, "ccl2f2", &clim->ccl2f2);
  dump_ts(out, "n2o", &clim->n2o);
  dump_ts(out, "sf6", &clim->sf6);
  fclose(out);
  printf("RESULT done\n");
  free(clim);
  return 0;
}
