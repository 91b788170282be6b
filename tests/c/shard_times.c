/* Test program (tests/test_host_logic.py): start / stop time of a run whose particles were cut into index
 * ranges -- every rank must get the range of the whole particle file.  argv[1] = number of particles; rank and
 * world size come from the environment like in the driver.  Prints: np t_start t_stop */
#include "mptrac.h"

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met0, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  if (argc < 2)
    return 2;
  mptrac_alloc(&ctl, &cache, &clim, &met0, &met1, &atm, &depo, &dd);
  mptrac_read_ctl("-", argc, argv, ctl);
  atm->np = atoi(argv[1]);
  for (int ip = 0; ip < atm->np; ip++)
    atm->time[ip] = 1000.0 + 900.0 * ip;   /* released one after the other */
  mptrac_amd_job_t job;
  mptrac_amd_job_from_env(&job);
  mptrac_amd_shard(atm, &job);
  module_timesteps_init(ctl, atm);
  printf("RESULT %d %.17g %.17g\n", atm->np, ctl->t_start, ctl->t_stop);
  return 0;
}
