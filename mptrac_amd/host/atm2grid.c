/*
 * atm2grid.c -- gridded output of a particle file, MI355X build: the reference's atm2grid tool (src/atm2grid.c), a
 * thin wrapper over write_grid.  The binning and the sums run on the device (mphip_grid_sums), the file is written by
 * the host layer's write_grid.
 *
 *   atm2grid <ctl> <atm_in> [KEY VALUE ...]
 *
 * The output time is taken from the name of the particle file (..._YYYY_MM_DD_HH_MM_SS.<ext>), the file is called
 * <GRID_BASENAME>_YYYY_MM_DD_HH_MM_SS.tab / .nc.  As in the reference no meteo data are at hand, so the implicit
 * volume mixing ratio is not computed (nan).
 */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met0, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
      printf("\nMPTRAC atm2grid tool (MI355X build).\n\nConverts a particle file to gridded data.\n\n"
             "Usage:\n  atm2grid <ctl> <atm_in> [KEY VALUE ...]\n\n"
             "Arguments:\n  <ctl>     Control parameter file (\"-\": none).\n"
             "  <atm_in>  Particle file; its name ends in _YYYY_MM_DD_HH_MM_SS.<ext>.\n"
             "  [KEY VALUE ...]  Control parameters; GRID_BASENAME is mandatory.\n\n");
      return EXIT_SUCCESS;
    }
  if (argc < 3)
    ERRMSG("Missing or invalid command-line arguments.\n\n"
           "Usage: atm2grid <ctl> <atm_in> [KEY VALUE ...]\n\n" "Use -h for full help.");
  mptrac_alloc(&ctl, &cache, &clim, &met0, &met1, &atm, &depo, &dd);
  mptrac_read_ctl(argv[1], argc, argv, ctl);
  if (ctl->grid_basename[0] == '-')
    ERRMSG("You need to specify GRID_BASENAME!");
  if (!mptrac_read_atm(argv[2], ctl, atm))
    ERRMSG("Cannot open file!");

  /* time from the file name (time_from_filename, mptrac.c:12540-12583): the stamp starts 23 characters (text,
   * binary: 4 for the extension) or 22 (".nc") from the end */
  const int offset = ctl->atm_type < 2 ? 23 : 22, len = (int) strlen(argv[2]);
  int year, mon, day, hour, min, sec;
  if (len < offset || sscanf(argv[2] + len - offset, "%4d_%2d_%2d_%2d_%2d_%2d", &year, &mon, &day, &hour, &min, &sec) != 6
      || year < 1900 || year > 2100 || mon < 1 || mon > 12 || day < 1 || day > 31 || hour < 0 || hour > 23 || min < 0
      || min > 59)
    ERRMSG("Cannot read time from filename!");
  double t;
  time2jsec(year, mon, day, hour, min, sec, 0.0, &t);

  char filename[3 * LEN];
  sprintf(filename, "%s_%04d_%02d_%02d_%02d_%02d_%02d.%s", ctl->grid_basename, year, mon, day, hour, min, sec,
          ctl->grid_type == 0 ? "tab" : "nc");
  /* the particles go to the device.  (T_START keeps its default, as in the reference's tool: write_grid reads a
   * GRID_KERNEL only at t == T_START, so the tool never applies one -- atm2grid.c:68-92) */
  mptrac_update_device(ctl, NULL, NULL, NULL, NULL, atm);
  write_grid(filename, ctl, NULL, NULL, atm, t);
  mptrac_free(ctl, cache, clim, met0, met1, atm, depo, dd);
  return EXIT_SUCCESS;
}
