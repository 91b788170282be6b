/*
 * atm_conv.c -- conversion between the particle file formats, MI355X build: the reference's atm_conv tool
 * (src/atm_conv.c) on the host layer's readers and writers (text, binary, netCDF classic / netCDF-4 in, classic out,
 * CLaMS).  No device involved.
 *
 *   atm_conv <ctl> <atm_in> <atm_in_type> <atm_out> <atm_out_type> [KEY VALUE ...]
 */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  static ctl_t ctl;
  atm_t *atm;
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
      printf("\nMPTRAC atm_conv tool (MI355X build).\n\nConverts particle files between the formats of ATM_TYPE.\n\n"
             "Usage:\n  atm_conv <ctl> <atm_in> <atm_in_type> <atm_out> <atm_out_type> [KEY VALUE ...]\n\n"
             "Types: 0 text, 1 binary, 2 netCDF, 3 CLaMS trajectory file (output only), 4 CLaMS position file.\n\n");
      return EXIT_SUCCESS;
    }
  if (argc < 6)
    ERRMSG("Missing or invalid command-line arguments.\n\n"
           "Usage: atm_conv <ctl> <atm_in> <atm_in_type> <atm_out> <atm_out_type>\n\n" "Use -h for full help.");
  ALLOC(atm, atm_t, 1);
  mptrac_read_ctl(argv[1], argc, argv, &ctl);
  ctl.atm_type = atoi(argv[3]);
  if (!mptrac_read_atm(argv[2], &ctl, atm))
    ERRMSG("Cannot open file!");
  ctl.atm_type_out = atoi(argv[5]);
  mptrac_write_atm(argv[4], &ctl, atm, 0);
  free(atm);
  return EXIT_SUCCESS;
}
