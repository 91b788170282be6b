/* Test program (tests/test_host_output.py): particle file conversion with the host layer's readers and writers,
 * command line of the reference's atm_conv tool (tests/interoper_test/run.sh:19-20):
 *   atm_conv <ctl> <atm_in> <atm_in_type> <atm_out> <atm_out_type> [KEY VALUE ...]
 * ATM_CONV_ZETA_COORDINATE in the environment: read with ADVECT_VERT_COORD 1.  No device involved. */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  static ctl_t ctl;
  atm_t *atm;
  if (argc < 6)
    ERRMSG("Give parameters: <ctl> <atm_in> <atm_in_type> <atm_out> <atm_out_type>");
  ALLOC(atm, atm_t, 1);
  mptrac_read_ctl(argv[1], argc, argv, &ctl);
  /* the readers alone know the diabatic set-up (ZETA is the vertical coordinate of the file); a run with it is
   * refused by mptrac_read_ctl because the host layer's meteo reader has no model-level fields */
  if (getenv("ATM_CONV_ZETA_COORDINATE"))
    ctl.advect_vert_coord = 1;
  ctl.atm_type = atoi(argv[3]);
  if (!mptrac_read_atm(argv[2], &ctl, atm))
    ERRMSG("Cannot open file!");
  ctl.atm_type_out = atoi(argv[5]);
  mptrac_write_atm(argv[4], &ctl, atm, 0);
  printf("RESULT converted %d\n", atm->np);
  free(atm);
  return 0;
}
