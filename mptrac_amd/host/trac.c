/*
 * trac.c -- Lagrangian particle dispersion driver, MI355X build.
 *
 * Same command line and the same call sequence of the high-level interface as
 * the reference's driver (src/trac.c:43-197):
 *
 *   trac <dirlist> <ctl> <atm_in> [KEY VALUE ...]
 *
 * One process drives one GPU.  Started once, it uses the device of the control key HIP_DEVICE.  Started
 * N times by a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
 * (e.g. `python -m torch.distributed.run --nproc-per-node N --no-python trac ...`), the N processes share
 * every simulation: rank k binds to device LOCAL_RANK (the reference binds MPI ranks to devices the same way,
 * src/trac.c:70-81), keeps the k-th index range of the particles, reads the same meteo files, and the gridded
 * sums of module_mixing and of the gridded output are all-reduced by RCCL inside the back end.  Rank 0 writes
 * the gridded output; particle files are written per rank (<name>.rank<k>).  The reference's farm over work
 * directories (trac.c:83-98) is not reproduced: directories are processed one after the other.
 */
#include "mptrac.h"

#include <sys/time.h>

static double wall(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (double) tv.tv_sec + 1e-6 * (double) tv.tv_usec;
}

int main(int argc, char *argv[]) {
  ctl_t *ctl;
  cache_t *cache;
  clim_t *clim;
  met_t *met0, *met1;
  atm_t *atm;
  depo_t *depo;
  dd_t *dd;
  FILE *dirlist;
  char dirname[LEN], filename[2 * LEN];

  /* the command-line conventions of the reference's tools (USAGE, mptrac.h:2213-2222; tests/cli_test): -h or
   * --help anywhere prints the usage and succeeds, too few arguments are the standard diagnostic */
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
      printf("\nMPTRAC trac tool (MI355X build).\n\n"
             "Runs the trajectory calculations of the directories listed in <dirlist> on the GPU.\n\n"
             "Usage:\n  trac <dirlist> <ctl> <atm_in> [KEY VALUE ...]\n\n"
             "Arguments:\n"
             "  <dirlist>  File with one working directory per line.\n"
             "  <ctl>      Control parameter file, relative to each directory.\n"
             "  <atm_in>   Initial particle file, relative to each directory.\n"
             "  [KEY VALUE ...]  Control parameters that override the file.\n\n"
             "Started N times by a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /\n"
             "MASTER_PORT, the N processes share every simulation, one GPU each.\n\n");
      return EXIT_SUCCESS;
    }
  if (argc < 4)
    ERRMSG("Missing or invalid command-line arguments.\n\n"
           "Usage: trac <dirlist> <ctl> <atm_in> [KEY VALUE ...]\n\n" "Use -h for full help.");
  if (!(dirlist = fopen(argv[1], "r")))
    ERRMSG("Cannot open directory list!");
  mptrac_amd_job_t job;
  mptrac_amd_job_from_env(&job);

  while (fscanf(dirlist, "%4999s", dirname) != EOF) {

    /* ---- initialise the model run (trac.c:109-125) ---- */
    mptrac_alloc(&ctl, &cache, &clim, &met0, &met1, &atm, &depo, &dd);
    sprintf(filename, "%s/%s", dirname, argv[2]);
    mptrac_read_ctl(filename, argc, argv, ctl);
    mptrac_read_clim(ctl, clim);
    sprintf(filename, "%s/%s", dirname, argv[3]);
    if (!mptrac_read_atm(filename, ctl, atm))
      ERRMSG("Cannot open file!");
    if (job.world > 1) {
      ctl->hip_device = job.local_rank;
      mptrac_amd_shard(atm, &job);
    }
    mptrac_init(ctl, cache, clim, atm, depo, 0);
    mptrac_amd_comm_init(ctl, &job);

    /* ---- loop over time steps (trac.c:131-163) ---- */
    const double w0 = wall();
    double w_met = 0, w_step = 0, w_out = 0;   /* where the wall time of the loop goes, by call */
    long nsteps = 0;
    for (double t = ctl->t_start; ctl->direction * (t - ctl->t_stop) < ctl->dt_mod;
         t += ctl->direction * ctl->dt_mod) {
      if (ctl->direction * (t - ctl->t_stop) > 0)
        t = ctl->t_stop;
      const double a0 = wall();
      mptrac_get_met(ctl, clim, t, &met0, &met1, dd);
      const double a1 = wall();
      if (ctl->dt_mod > fabs(met0->lon[1] - met0->lon[0]) * 111132. / 150.)
        WARN("Violation of CFL criterion! Check DT_MOD!");
      mptrac_run_timestep(ctl, cache, clim, &met0, &met1, atm, depo, t, dd);
      const double a2 = wall();
      mptrac_write_output(dirname, ctl, met0, met1, atm, depo, t);
      const double a3 = wall();
      w_met += a1 - a0;
      w_step += a2 - a1;
      w_out += a3 - a2;
      nsteps++;
    }
    const double a4 = wall();
    mptrac_update_host(NULL, NULL, NULL, NULL, NULL, atm);   /* also drains the device queue */
    const double w1 = wall();

    /* ---- report (trac.c:172-188) ---- */
    LOG(1, "SIZE_NP = %d", atm->np);
    LOG(1, "SIZE_TASKS = %d", job.world);
    LOG(1, "MEMORY_ATM = %g MByte", sizeof(atm_t) / 1024. / 1024.);
    LOG(1, "MEMORY_CACHE = %g MByte", sizeof(cache_t) / 1024. / 1024.);
    LOG(1, "MEMORY_METEO = %g MByte", 2 * sizeof(met_t) / 1024. / 1024.);
    LOG(1, "TIMER_TIMESTEPS = %.3f s    (%ld calls, %.3e particle-steps/s)", w1 - w0, nsteps,
        nsteps > 1 ? (double) atm->np * (double) (nsteps - 1) / (w1 - w0) : 0.0);
    /* ... by call of the loop (the device works beside the host: time steps are queued, and whoever needs their
     * result first -- a meteo hand-over, an output, the final download -- waits for them) */
    LOG(1, "TIMER_GET_MET = %.3f s    (meteo files read, converted and handed to the device)", w_met);
    LOG(1, "TIMER_RUN_TIMESTEP = %.3f s", w_step);
    LOG(1, "TIMER_WRITE_OUTPUT = %.3f s", w_out);
    LOG(1, "TIMER_UPDATE_HOST = %.3f s    (rest of the queue + final download)", w1 - a4);
    LOG(1, "TIMER_WITHOUT_MET_AND_OUTPUT = %.3f s    (%.3e particle-steps/s)", w1 - w0 - w_met - w_out,
        nsteps > 1 && w1 - w0 - w_met - w_out > 0 ? (double) atm->np * (double) (nsteps - 1) / (w1 - w0 - w_met - w_out) : 0.0);

    mptrac_free(ctl, cache, clim, met0, met1, atm, depo, dd);
  }
  fclose(dirlist);
  return EXIT_SUCCESS;
}
