/* Test program (tests/test_host_logic.py): mptrac_read_ctl sees a control file that was rewritten in place.
 * argv[1] = path of the control file (this program writes it). */
#include "mptrac.h"

static void put(const char *path, double dt, double stop) {
  FILE *out = fopen(path, "w");
  if (!out)
    exit(3);
  fprintf(out, "DT_MOD = %5.1f\nT_STOP = %6.1f\n", dt, stop);   /* same length every time */
  fclose(out);
}

int main(int argc, char *argv[]) {
  static const double dts[3] = { 240.0, 120.0, 120.0 }, stops[3] = { 7200.0, 3600.0, 1800.0 };
  ctl_t *ctl = calloc(1, sizeof(ctl_t));
  if (argc < 2 || !ctl)
    return 2;
  for (int k = 0; k < 3; k++) {
    put(argv[1], dts[k], stops[k]);
    mptrac_read_ctl(argv[1], 0, NULL, ctl);
    if (ctl->dt_mod != dts[k] || ctl->t_stop != stops[k]) {
      printf("RESULT stale %d %g %g\n", k, ctl->dt_mod, ctl->t_stop);
      return 1;
    }
  }
  printf("RESULT ok\n");
  return 0;
}
