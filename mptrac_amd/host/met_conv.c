/*
 * met_conv.c -- conversion between the meteo file formats, MI355X build: the reference's met_conv tool
 * (src/met_conv.c) on the host layer's readers and writers -- MET_TYPE 0 (netCDF: classic or netCDF-4 in, classic
 * out) and 1 (the reference's binary format).  No device involved; as in mptrac_read_met of this build, netCDF input
 * is taken as stored (no meteo preprocessing).
 *
 *   met_conv <ctl> <met_in> <met_in_type> <met_out> <met_out_type> [KEY VALUE ...]
 */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
      printf("\nMPTRAC met_conv tool (MI355X build).\n\nConverts meteo files between the formats of MET_TYPE.\n\n"
             "Usage:\n  met_conv <ctl> <met_in> <met_in_type> <met_out> <met_out_type> [KEY VALUE ...]\n\n"
             "Types: 0 netCDF, 1 binary.\n\n");
      return EXIT_SUCCESS;
    }
  if (argc < 6)
    ERRMSG("Missing or invalid command-line arguments.\n\n"
           "Usage: met_conv <ctl> <met_in> <met_in_type> <met_out> <met_out_type>\n\n" "Use -h for full help.");
  mptrac_alloc(&ctl, &cache, &clim, &met, &met1, &atm, &depo, &dd);
  mptrac_read_ctl(argv[1], argc, argv, ctl);
  mptrac_read_clim(ctl, clim);
  ctl->met_type = atoi(argv[3]);
  if (!mptrac_read_met(argv[2], ctl, clim, met, dd))
    ERRMSG("Cannot open file!");
  ctl->met_type = atoi(argv[5]);
  mptrac_write_met(argv[4], ctl, met);
  return EXIT_SUCCESS;
}
