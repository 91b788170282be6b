/*
 * mptrac_hip_glue.c -- route A of INTEGRATION.md, complete: the code a maintainer of the reference adds to
 * its src/mptrac.c (at the end of the file, or as `#include "mptrac_hip_glue.c"` there) to run the
 * time-step loop on the MI355X back end while keeping the reference's own host code, structs, readers and
 * writers.  Everything is inside `#ifdef MPTRAC_HIP`; it uses the reference's types (ctl_t, met_t, atm_t,
 * cache_t, clim_t from src/mptrac.h) and only the C ABI of include/mptrac_hip.h.
 *
 * Build of the reference with it (src/Makefile):
 *     HIP ?= 0
 *     ifeq ($(HIP),1)
 *       CFLAGS  += -DMPTRAC_HIP -I$(MPTRAC_AMD)/include
 *       LDFLAGS += -L$(MPTRAC_AMD)/mptrac_amd/lib -lmptrac_hip -Wl,-rpath,$(MPTRAC_AMD)/mptrac_amd/lib
 *     (-lmptrac_hip_exact instead: the build of the same sources whose results are the CPU build's bits, INTEGRATION.md
 *     "Two libraries, one ABI")
 *     endif
 *
 * Every member of ctl_t / met_t / clim_t / cache_t / atm_t named here exists in the reference's mptrac.h:
 * integration/check_glue_fields.py verifies that against /root/reference (tests/test_abi.py runs it).  The
 * file is type-checked against the reference header by integration/typecheck_glue.py (gcc -fsyntax-only; the
 * reference itself cannot be built in this image: its header needs GSL and netCDF).
 *
 * Call sites (reference file:line -> what replaces the OpenACC region there):
 *   mptrac_alloc          mptrac.c:6336-6372   mptrac_hip_alloc(rank)            instead of `acc enter data create`
 *   mptrac_free           mptrac.c:6398-6430   mptrac_hip_free()                 instead of `acc exit data delete`
 *   mptrac_update_device  mptrac.c:8005-8057   mptrac_hip_update_device(...)     instead of `acc update device`
 *   mptrac_update_host    mptrac.c:8061-8113   mptrac_hip_update_host(...)       instead of `acc update host`
 *   mptrac_get_met        mptrac.c:6488-6491, 6513-6516  mptrac_hip_swap_met()   next to the pointer swap
 *   mptrac_run_timestep   mptrac.c:7862-8000   mptrac_hip_run_timestep(t); return;
 *   write_grid            mptrac.c:13836-13872 mptrac_hip_grid_sums(...)         instead of the binning loop
 *   mptrac_write_output   mptrac.c:8320-8323   mptrac_hip_station_flags(ctl, atm) behind write_station
 */
#ifdef MPTRAC_HIP

#include "mptrac_hip.h"

static mphip_ctx *hip_ctx;     /* process-global like rng_ctr (mptrac.c:32-40): the interface is not re-entrant */
static int hip_nq;
static int hip_ts_on[MPHIP_NTR] = { 1, 1, 1, 1, 1 };   /* CLIM_*_TIMESERIES is not "-" (mptrac.c:3858 ...) */

#define HIPCALL(x) {                                    \
    if ((x) != 0)                                       \
      ERRMSG("%s", mphip_last_error(hip_ctx));          \
  }

/* members that carry the same name in ctl_t and mphip_ctl_t */
#define HIP_CTL_SAME_NAME(X)                                                                             \
  X(direction) X(met_coord_type) X(t_start) X(t_stop) X(dt_mod) X(dt_met) X(met_utm_ref_lat) X(nq)       \
  X(qnt_m) X(qnt_vmr) X(qnt_rp) X(qnt_rhop) X(qnt_ens) X(qnt_loss_rate) X(qnt_mloss_decay)               \
  X(qnt_mloss_wet) X(qnt_mloss_dry) X(qnt_zeta) X(qnt_eta) X(qnt_aoa) X(nens) X(advect)                  \
  X(advect_vert_coord) X(rng_type) X(diffusion) X(turb_pbl_scheme) X(conv_mix_pbl)                       \
  X(turb_dx_pbl) X(turb_dx_trop) X(turb_dx_strat) X(turb_dz_pbl) X(turb_dz_trop) X(turb_dz_strat)        \
  X(turb_mesox) X(turb_mesoz) X(turb_pbl_trans) X(conv_pbl_trans) X(conv_cape) X(conv_cin) X(conv_dt)    \
  X(sort_dt) X(tdec_trop) X(tdec_strat) X(mixing_dt) X(mixing_trop) X(mixing_strat)                      \
  X(mixing_z0) X(mixing_z1) X(mixing_lon0) X(mixing_lon1) X(mixing_lat0) X(mixing_lat1)                  \
  X(mixing_nx) X(mixing_ny) X(mixing_nz)                                                                 \
  X(wet_depo_ic_a) X(wet_depo_ic_b) X(wet_depo_bc_a) X(wet_depo_bc_b) X(wet_depo_so2_ph)                 \
  X(wet_depo_ic_ret_ratio) X(wet_depo_bc_ret_ratio) X(dry_depo_vdep) X(dry_depo_dp)                      \
  X(grid_z0) X(grid_z1) X(grid_lon0) X(grid_lon1) X(grid_lat0) X(grid_lat1) X(grid_nx) X(grid_ny)        \
  X(grid_nz) X(met_dt_out) X(isosurf) X(bound_pbl) X(bound_mass) X(bound_mass_trend) X(bound_vmr)        \
  X(bound_vmr_trend) X(bound_lat0) X(bound_lat1) X(bound_p0) X(bound_p1) X(bound_dps) X(bound_dzs)       \
  X(bound_zetas) X(oh_chem_beta) X(met_utm_ref_lon)

/* module_meteo outputs: mphip_ctl_t::qnt_met[MPHIP_MQ_<X>] = ctl_t::qnt_<x> */
#define HIP_CTL_METEO_QNT(X)                                                                             \
  X(PS, ps) X(TS, ts) X(ZS, zs) X(US, us) X(VS, vs) X(ESS, ess) X(NSS, nss) X(SHF, shf) X(LSM, lsm)       \
  X(SST, sst) X(PBL, pbl) X(PT, pt) X(TT, tt) X(ZT, zt) X(H2OT, h2ot) X(ZG, zg) X(P, p) X(T, t)           \
  X(RHO, rho) X(U, u) X(V, v) X(W, w) X(H2O, h2o) X(O3, o3) X(LWC, lwc) X(RWC, rwc) X(IWC, iwc)           \
  X(SWC, swc) X(CC, cc) X(PCT, pct) X(PCB, pcb) X(CL, cl) X(PLCL, plcl) X(PLFC, plfc) X(PEL, pel)         \
  X(CAPE, cape) X(CIN, cin) X(O3C, o3c) X(VH, vh) X(VZ, vz) X(PSAT, psat) X(PSICE, psice) X(PW, pw)       \
  X(SH, sh) X(RH, rh) X(RHICE, rhice) X(THETA, theta) X(ZETA_D, zeta_d) X(TVIRT, tvirt)                  \
  X(LAPSE, lapse) X(PV, pv) X(TDEW, tdew) X(TICE, tice)                                                  \
  X(HNO3, hno3) X(OH, oh) X(H2O2, h2o2) X(HO2, ho2) X(O1D, o1d) X(TNAT, tnat) X(TSTS, tsts)

static void hip_ctl(const ctl_t *c, mphip_ctl_t *d) {
  memset(d, 0, sizeof(*d));
#define X(f) d->f = c->f;
  HIP_CTL_SAME_NAME(X)
#undef X
#define X(E, f) d->qnt_met[MPHIP_MQ_##E] = c->qnt_##f;
  HIP_CTL_METEO_QNT(X)
#undef X
  d->qnt_tracer[MPHIP_TR_CCL4] = c->qnt_Cccl4;
  d->qnt_tracer[MPHIP_TR_CCL3F] = c->qnt_Cccl3f;
  d->qnt_tracer[MPHIP_TR_CCL2F2] = c->qnt_Cccl2f2;
  d->qnt_tracer[MPHIP_TR_N2O] = c->qnt_Cn2o;
  d->qnt_tracer[MPHIP_TR_SF6] = c->qnt_Csf6;
  for (int k = 0; k < 2; k++) {
    d->wet_depo_pre[k] = c->wet_depo_pre[k];
    d->wet_depo_ic_h[k] = c->wet_depo_ic_h[k];
    d->wet_depo_bc_h[k] = c->wet_depo_bc_h[k];
  }
  /* what the device does not implement must not run silently on stale host data */
  if (c->oh_chem_reaction != 0 || c->h2o2_chem_reaction != 0 || c->kpp_chem || c->tracer_chem || c->radio_decay)
    ERRMSG("MPTRAC_HIP: chemistry and radioactive decay are not implemented on the device!");
  if (c->qnt_hno3 >= 0 || c->qnt_oh >= 0 || c->qnt_h2o2 >= 0 || c->qnt_ho2 >= 0 || c->qnt_o1d >= 0 || c->qnt_tnat >= 0
      || c->qnt_tsts >= 0)
    ERRMSG("MPTRAC_HIP: the climatology-based quantities of module_meteo are not implemented on the device!");
}

/* level fields: met_t member -> MPHIP_* slot (float [EX][EY][EP]); the model-level ones use met->npl levels */
#define HIP_MET_3D(X)                                                                                    \
  X(U, u) X(V, v) X(W, w) X(T, t) X(LWC, lwc) X(RWC, rwc) X(IWC, iwc) X(SWC, swc) X(H2O, h2o)             \
  X(Z, z) X(PV, pv) X(O3, o3) X(CC, cc)                                                                  \
  X(PL, pl) X(UL, ul) X(VL, vl) X(WL, wl) X(ZETAL, zetal) X(ZETA_DOTL, zeta_dotl)
/* surface fields (float [EX][EY]) */
#define HIP_MET_2D(X)                                                                                    \
  X(PS, ps) X(PBL, pbl) X(CAPE, cape) X(CIN, cin) X(PEL, pel) X(PCT, pct) X(PCB, pcb) X(CL, cl)           \
  X(ESS, ess) X(NSS, nss) X(SHF, shf) X(TS, ts) X(ZS, zs) X(US, us) X(VS, vs) X(LSM, lsm) X(SST, sst)     \
  X(PT, pt) X(TT, tt) X(ZT, zt) X(H2OT, h2ot) X(PLCL, plcl) X(PLFC, plfc) X(O3C, o3c)

static void hip_met_view(met_t *m, mphip_met_t *v) {
  memset(v, 0, sizeof(*v));
  v->time = m->time;
  v->coord_type = m->coord_type;
  v->nx = m->nx;
  v->ny = m->ny;
  v->np = m->np;
  v->npl = m->npl;
  v->lon = m->lon;
  v->lat = m->lat;
  v->p = m->p;
  v->sx = v->sx_ml = (long long) EY * EP;   /* float u[EX][EY][EP] */
  v->sy = v->sy_ml = EP;
  v->sx2 = EY;                              /* float ps[EX][EY] */
#define X(E, f) v->f3[MPHIP_##E] = &m->f[0][0][0];
  HIP_MET_3D(X)
#undef X
#define X(E, f) v->f2[MPHIP_##E] = &m->f[0][0];
  HIP_MET_2D(X)
#undef X
}

/* rank -> device as the reference binds MPI ranks to OpenACC devices (src/trac.c:70-81) */
void mptrac_hip_alloc(const int rank) {
  const char *per_node = getenv("MPTRAC_HIP_DEVICES");     /* GPUs per node, default 8 */
  const int ndev = per_node != NULL && atoi(per_node) > 0 ? atoi(per_node) : 8;
  if (mphip_create(&hip_ctx, rank % ndev) != 0)
    ERRMSG("MPTRAC_HIP: no usable HIP device!");
}

#ifdef MPI
/* Optional: the ranks of MPI_COMM_WORLD share ONE simulation -- rank k keeps the k-th index range of the
 * particles (the caller trims atm to it and passes ip0 / np_total to mptrac_hip_update_atm_range), and the
 * gridded sums of module_mixing and write_grid are all-reduced by RCCL inside the back end, on its stream.
 * Call once after mptrac_hip_alloc.  (The reference's own MPI mode farms work directories over the ranks,
 * src/trac.c:83-98; that needs nothing from the back end.) */
void mptrac_hip_comm_init(void) {
  int rank, size;
  unsigned char id[128];
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (rank == 0 && mphip_comm_unique_id(id) != 0)
    ERRMSG("MPTRAC_HIP: cannot create an RCCL unique id!");
  MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
  HIPCALL(mphip_comm_init(hip_ctx, size, rank, id));
}

void mptrac_hip_update_atm_range(const atm_t *atm, const long long ip0, const long long np_total) {
  const double *q[NQ];
  for (int iq = 0; iq < hip_nq; iq++)
    q[iq] = atm->q[iq];
  HIPCALL(mphip_update_atm(hip_ctx, atm->np, ip0, np_total, hip_nq, atm->time, atm->p, atm->lon, atm->lat, q));
}
#endif

void mptrac_hip_free(void) {
  mphip_destroy(hip_ctx);
  hip_ctx = NULL;
}

void mptrac_hip_update_device(const ctl_t *ctl, const cache_t *cache, const clim_t *clim, met_t **met0,
                              met_t **met1, const atm_t *atm) {
  if (ctl != NULL) {
    mphip_ctl_t d;
    hip_ctl(ctl, &d);
    HIPCALL(mphip_update_ctl(hip_ctx, &d));
    hip_nq = ctl->nq;
    hip_ts_on[MPHIP_TR_CCL4] = ctl->clim_ccl4_timeseries[0] != '-';
    hip_ts_on[MPHIP_TR_CCL3F] = ctl->clim_ccl3f_timeseries[0] != '-';
    hip_ts_on[MPHIP_TR_CCL2F2] = ctl->clim_ccl2f2_timeseries[0] != '-';
    hip_ts_on[MPHIP_TR_N2O] = ctl->clim_n2o_timeseries[0] != '-';
    hip_ts_on[MPHIP_TR_SF6] = ctl->clim_sf6_timeseries[0] != '-';
  }
  if (clim != NULL) {
    HIPCALL(mphip_update_clim(hip_ctx, clim->tropo_ntime, clim->tropo_nlat, clim->tropo_time, clim->tropo_lat,
                              &clim->tropo[0][0], 73));
    /* the zonal means module_meteo samples: clim_zm_t holds vmr[CT][CP][CY], the back end takes compact tables */
    const clim_zm_t *zm[MPHIP_NZM] = { &clim->hno3, &clim->oh, &clim->h2o2, &clim->ho2, &clim->o1d };
    for (int k = 0; k < MPHIP_NZM; k++) {
      double *v = NULL;
      if (zm[k]->ntime > 0) {
        size_t n = 0;
        ALLOC(v, double, (size_t) zm[k]->ntime * (size_t) zm[k]->np * (size_t) zm[k]->nlat);
        for (int it = 0; it < zm[k]->ntime; it++)
          for (int iz = 0; iz < zm[k]->np; iz++)
            for (int iy = 0; iy < zm[k]->nlat; iy++)
              v[n++] = zm[k]->vmr[it][iz][iy];
      }
      HIPCALL(mphip_update_clim_zm(hip_ctx, k, zm[k]->ntime, zm[k]->np, zm[k]->nlat, zm[k]->time, zm[k]->p,
                                   zm[k]->lat, v));
      free(v);
    }
    /* the surface time series of module_bound_cond's trace gases ("-" as file name: no boundary condition) */
    const clim_ts_t *ts[MPHIP_NTR] = { &clim->ccl4, &clim->ccl3f, &clim->ccl2f2, &clim->n2o, &clim->sf6 };
    for (int k = 0; k < MPHIP_NTR; k++)
      HIPCALL(mphip_update_clim_ts(hip_ctx, k, hip_ts_on[k] ? ts[k]->ntime : 0, ts[k]->time, ts[k]->vmr));
  }
  mphip_met_t v;
  if (met0 != NULL) {
    hip_met_view(*met0, &v);
    HIPCALL(mphip_update_met(hip_ctx, 0, &v));
  }
  if (met1 != NULL) {
    hip_met_view(*met1, &v);
    HIPCALL(mphip_update_met(hip_ctx, 1, &v));
  }
  if (atm != NULL) {
    const double *q[NQ];
    for (int iq = 0; iq < hip_nq; iq++)
      q[iq] = atm->q[iq];
    HIPCALL(mphip_update_atm(hip_ctx, atm->np, 0, atm->np, hip_nq, atm->time, atm->p, atm->lon, atm->lat, q));
  }
  if (cache != NULL) {
    HIPCALL(mphip_update_cache(hip_ctx, &cache->uvwp[0][0], &rng_ctr));
    HIPCALL(mphip_update_iso(hip_ctx, cache->iso_var, cache->iso_n > 0 ? cache->iso_ts : NULL,
                             cache->iso_n > 0 ? cache->iso_ps : NULL, cache->iso_n));
  }
}

void mptrac_hip_update_host(cache_t *cache, atm_t *atm) {
  if (atm != NULL) {
    double *q[NQ];
    for (int iq = 0; iq < hip_nq; iq++)
      q[iq] = atm->q[iq];
    HIPCALL(mphip_get_atm(hip_ctx, atm->time, atm->p, atm->lon, atm->lat, q));
  }
  if (cache != NULL)
    HIPCALL(mphip_get_cache(hip_ctx, &cache->uvwp[0][0], cache->dt, &rng_ctr));
}

/* next to `mets = *met1; *met1 = *met0; *met0 = mets;`: the device slots trade places, nothing is copied */
void mptrac_hip_swap_met(void) {
  HIPCALL(mphip_swap_met(hip_ctx));
}

/* the whole body of mptrac_run_timestep: module order, gating (SORT_DT, CONV_DT, MIXING_DT, MET_DT_OUT ...)
 * and the rng_ctr bookkeeping are reproduced inside the back end */
void mptrac_hip_run_timestep(const double t) {
  HIPCALL(mphip_run_timestep(hip_ctx, t));
}

/* write_grid: np[idx], mean[iq][idx], sigma[iq][idx] as flat arrays with idx = ARRAY_3D(ix, iy, ny, iz, nz) */
void mptrac_hip_grid_sums(const double t, int *np, double *mean, double *sigma) {
  HIPCALL(mphip_grid_sums(hip_ctx, t, np, mean, sigma));
}

/* write_grid with GRID_KERNEL: hand the weighting function read_kernel has just read (and normalised) to the
 * device once (mptrac.c:13779-13780); nk = 0 without a kernel file */
void mptrac_hip_grid_kernel(const int nk, const double *kz, const double *kw) {
  HIPCALL(mphip_set_grid_kernel(hip_ctx, nk, kz, kw));
}

/* write_station sets atm->q[qnt_stat] on the host copy (mptrac.c:15143-15145); on the reference's CPU path that
 * is the model state of the next step (its OpenACC build loses the flags: the device copy is never updated).
 * Hand the one quantity back instead of a whole mptrac_update_device(atm). */
void mptrac_hip_station_flags(const ctl_t *ctl, const atm_t *atm) {
  if (ctl->qnt_stat >= 0)
    HIPCALL(mphip_update_quantity(hip_ctx, ctl->qnt_stat, atm->q[ctl->qnt_stat]));
}

#endif   /* MPTRAC_HIP */
