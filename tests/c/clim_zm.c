/* Test program (tests/test_host_output.py): the zonal-mean climatologies of mptrac_read_clim.
 *   clim_zm <out file> [KEY VALUE ...]
 * reads the control parameters and the climatologies the requested quantities need, and writes for each of the
 * five tables "name ntime np nlat" followed by its time, pressure and latitude axes and the mixing ratios in
 * index order [time][p][lat], 17 significant digits.  No device involved. */
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
  fclose(out);
  printf("RESULT done\n");
  free(clim);
  return 0;
}
