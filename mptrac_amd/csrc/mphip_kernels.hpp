// mphip_kernels.hpp -- __global__ kernels of the MI355X back end.
//
//   step_kernel      fused per-particle time step (all per-particle modules
//                    of mptrac_run_timestep selected by a module mask)
//   pack_*           build the packed two-snapshot meteo grids
//   sort_*           32-bit LSD radix sort of the grid-cell key + fused gather
//   mix_* / grid_*   inter-parcel mixing and gridded-output sums
#pragma once

#include <type_traits>
#include "mphip_device.hpp"

namespace mphip {

// ---------------------------------------------------------------------------
// fused step
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// module_mixing (mptrac.c:5169-5347) per particle: the box a particle is in, and the relaxation towards the
// box mean (kernels box_index_kernel and mix_relax_kernel below).  Both were also tried as parts of the
// time-step launches around module_mixing: the box index at the end of the launch that moves the particles cost
// that launch 40 - 70 us (a logarithm and three true divisions per particle) against 67 us for the kernel of its
// own, the relaxation at the start of the launch of the deposition modules 150 us against 127 us -- no gain.
// ---------------------------------------------------------------------------

struct BoxGrid {
  double lon0, lon1, lat0, lat1, z0, z1;
  int nx, ny, nz;
};

constexpr int kMixMax = 8;   // mass, volume mixing ratio, five trace gases, age of air (of the reference's list, mptrac.c:5223-5230)
struct MixSet {
  double *q[kMixMax];
  int n;
};

// cell index of a particle, -1 if outside (mptrac.c:5201-5218, 13836-13855); with an ensemble the index inside
// the member's copy of the grid, ens * ngrid + cell (mptrac.c:5291-5294)
__device__ __forceinline__ int box_cell(const BoxGrid &G, double t0, double t1, double time, double lon, double lat,
                                        double p, const double *__restrict__ ens, long long i, int ngrid) {
  const double dz = (G.z1 - G.z0) / G.nz;
  const double dlon = (G.lon1 - G.lon0) / G.nx;
  const double dlat = (G.lat1 - G.lat0) / G.ny;
  const double zpart = zfromp(p);
  int c = -1;
  if (!(time < t0 || time > t1 || lon < G.lon0 || lon >= G.lon1 || lat < G.lat0 || lat >= G.lat1 || zpart < G.z0
        || zpart >= G.z1)) {
    const int ix = (int) ((lon - G.lon0) / dlon);
    const int iy = (int) ((lat - G.lat0) / dlat);
    const int iz = (int) ((zpart - G.z0) / dz);
    if (!(ix >= G.nx || iy >= G.ny || iz >= G.nz))
      c = (ix * G.ny + iy) * G.nz + iz + (ens ? (int) ens[i] * ngrid : 0);
  }
  return c;
}

// tropo_weight (mptrac.c:12748-12770, clim_tropo mptrac.c:213-237) in the reference's operation order with true
// divisions and no fused multiply-adds -- the step kernel's version multiplies by stored reciprocals.  With the
// ordered cell sums module_mixing then returns the bits of the serial code.
__device__ inline double tropo_weight_exact(const mphip_ctl_t &ctl, const DevClim &C, double time, double lat, double p) {
#pragma clang fp contract(off)
  if (ctl.met_coord_type != 0)
    lat = ctl.met_utm_ref_lat;
  double sec = fmod_trunc(time, 365.25 * 86400.);
  while (sec < 0)
    sec += 365.25 * 86400.;
  const int it = locate_irr(C.time, C.ntime, sec, 1);
  const int il = locate_reg(C.lat, C.nlat, lat);
  const double x0 = C.lat[il], x1 = C.lat[il + 1];
  const double pa = C.tropo[it][il] + (C.tropo[it][il + 1] - C.tropo[it][il]) / (x1 - x0) * (lat - x0);
  const double pb = C.tropo[it + 1][il] + (C.tropo[it + 1][il + 1] - C.tropo[it + 1][il]) / (x1 - x0) * (lat - x0);
  const double pt = pa + (pb - pa) / (C.time[it + 1] - C.time[it]) * (sec - C.time[it]);
  const double p1 = pt * 0.866877899;
  const double p0 = pt / 0.866877899;
  if (p > p0)
    return 1;
  if (p < p1)
    return 0;
  return 1.0 + (0.0 - 1.0) / (p1 - p0) * (p - p0);
}

// q += (mean - q) * mixparam for every mixed quantity of particle i in cell idx, mptrac.c:5305-5339
__device__ inline void mix_relax_one(const mphip_ctl_t &ctl, const DevClim &clim, const MixSet &mq, long long i, int idx,
                                     double time, double lat, double p, size_t ntot, const double *__restrict__ sums,
                                     const int *__restrict__ cnt) {
#pragma clang fp contract(off)
  if (idx < 0)
    return;
  double mixparam = 1.0;
  if (ctl.mixing_trop < 1 || ctl.mixing_strat < 1) {
    const double w = tropo_weight_exact(ctl, clim, time, lat, p);
    mixparam = w * ctl.mixing_trop + (1.0 - w) * ctl.mixing_strat;
  }
  const int n = cnt[idx];
  for (int k = 0; k < mq.n; k++) {
    const double sum = sums[(size_t) k * ntot + (size_t) idx];
    const double mean = n > 0 ? sum / n : sum;
    const double v = mq.q[k][i];
    mq.q[k][i] = v + (mean - v) * mixparam;
  }
}

// What the key kernel of the module_sort that runs ahead would compute in a pass of its own behind the launch that
// moves the particles -- the sort key of the next step's module_sort, its module_timesteps, and the box index of this
// step's module_mixing -- written by that launch itself, from the position it has in registers (kEmitKeys
// instantiation; keys == NULL: off).  Same functions on the same values as sort_key_kernel<true>.
struct EmitKeys {
  uint32_t *keys;         // key of module_sort (mptrac.c:5913-5920) of every slot
  double *dt_next;        // module_timesteps of the next step (mptrac.c:6016-6041)
  int *cell;              // box of module_mixing (mptrac.c:5245-5281), NULL: not wanted
  BoxGrid grid;
  double box_t0, box_t1;
  const double *ens;
  int ngrid;
  double direction, t_start, t_stop, t_next;
  // for the deposition launch behind module_mixing (depo_kernel): 1 where module_wet_depo / module_dry_depo (the bits of
  // depo_mask) have anything to do with the particle -- decided from dt, time and p, which this launch holds in
  // registers and the deposition launch would read again for every particle (NULL: not wanted)
  unsigned char *depo_busy;
  unsigned depo_mask;
};

struct StepParams {
  EmitKeys emit;
  mphip_ctl_t ctl;
  DevMet met;
  DevAtm atm;
  const DevClim *clim;
  const DevTracerSeries *tracers;   // time series of module_bound_cond's trace gases (NULL: none uploaded)
  double t;
  unsigned mask;          // MPHIP_MOD_* bits to run (when the kernel is the generic instantiation)
  int nblocks_logical;    // multiple of 8
  long long per_block;    // particles per logical block, multiple of 256
  int xcd_map;            // 1: workgroup b -> logical block (b % 8) * (n / 8) + b / 8
  const unsigned char *depo_busy;   // depo_kernel: EmitKeys::depo_busy of the launch that moved the particles (NULL: none)
  uint64_t ctr_turb, ctr_meso, ctr_conv, ctr_pbl;   // base counters of the module_rng calls
  // several consecutive time steps in one launch (kMultiStep instantiations, mphip_run_timesteps): step s runs
  // with model time t + s t_stride (accumulated as the caller's loop would) and counters + s ctr_stride
  int nsteps;
  double t_stride;
  uint64_t ctr_stride;
};

constexpr unsigned kMaskGeneric = 0xffffffffu;     // every module, module set taken from StepParams::mask
// module set from StepParams::mask too, but without the code of the rarely used modules (model-level
// advection and its init, module_diff_pbl, module_isosurf, module_bound_cond): small enough to keep
// the wind-corner cache and three waves per SIMD
constexpr unsigned kMaskGenericPL = 0xfffffffeu;
// as kMaskGenericPL, with model-level (zeta / eta) advection on the packed height fields instead of the
// pressure-level integrator (needs monotonic height columns: DevMet::ml_monotonic)
constexpr unsigned kMaskGenericML = 0xfffffffdu;
// ... and the same with StepParams::nsteps time steps per particle and launch (see kMultiStep)
constexpr unsigned kMaskGenericMLMulti = 0xfffffffbu;
template <unsigned CT>
constexpr bool kGenericML = (CT == kMaskGenericML || CT == kMaskGenericMLMulti);
template <unsigned CT>
constexpr bool kRuntimeMask = (CT == kMaskGeneric || CT == kMaskGenericPL || kGenericML<CT>);
// template mask only, on a lean instantiation: the advection is the model-level one (advect_ml_fast /
// advect_mlp_fast on the packed height fields, as in kMaskGenericML) and everything behind it the lean code of the
// pressure-level kernels -- the model-level runs of the headline module set no longer pay for the general
// versions of the diffusion, convection and sedimentation modules
constexpr unsigned kMLWinds = 1u << 27;
template <unsigned CT>
constexpr bool kLeanML = !kRuntimeMask<CT> && (CT & kMLWinds) != 0;
// template mask only, on a lean instantiation: 64-bit byte offsets into the packed wind / temperature records (and
// 64-bit column indices of the model-level fields) -- grids whose packed arrays exceed 4 GB (a 0.1 degree grid with
// 137 levels: 21 GB of wind records).  Two or three more instructions per gather; carried by the gated instantiations
// only (they serve every module set), so the kernels of the grids that fit 32 bits are what they were.
constexpr unsigned kBigGrid = 1u << 28;
template <unsigned CT>
constexpr bool kBig = !kRuntimeMask<CT> && (CT & kBigGrid) != 0;
template <unsigned CT>
constexpr bool kModelLevels = kGenericML<CT> || kLeanML<CT>;
constexpr unsigned kRareModules = MPHIP_MOD_ADVECT_INIT | MPHIP_MOD_ISOSURF_INIT | MPHIP_MOD_ISOSURF | MPHIP_MOD_DIFF_PBL
  | MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2;
constexpr unsigned kStoreDt = 1u << 30;   // write cache->dt (needed when a later launch reads it)
// template mask only: the lean instantiation integrates with two stages (ADVECT 2, the midpoint scheme -- the
// reference's default; its first stage alone is ADVECT 1) instead of the four of ADVECT 4
constexpr unsigned kTwoStage = 1u << 24;
// template mask only: these four modules are compiled in by the template mask AND switched by the run-time mask,
// so that one instantiation serves every subset of them (e.g. convection without sedimentation -- a gas
// tracer).  A flag of its own because the switches cost the headline instantiation 3 % when they were added to
// it (0.895 -> 0.92 ms: other register allocation and schedule); exact module sets keep their own kernels.
constexpr unsigned kGated = 1u << 25;
// template mask only: the particle loop runs StepParams::nsteps time steps per particle before it moves on (a run
// of steps with nothing between them -- no module_sort, mixing, output -- at small particle counts, where a step is
// shorter than a kernel launch).  The state goes through memory between the steps as it does between launches.
constexpr unsigned kMultiStep = 1u << 26;
// template mask only, on the gated pressure-level instantiations: module_diff_pbl (TURB_PBL_SCHEME 1, the closure inside
// the boundary layer) and module_isosurf are compiled in -- as function calls -- and switched by the run-time mask like
// the four optional movers.  Instantiations of their own, so that runs without them keep the kernels they had.
constexpr unsigned kPblClosure = 1u << 29;
// template mask only: the launch also writes what the key kernel of the sort ahead would (StepParams::emit)
constexpr unsigned kEmitKeys = 1u << 22;
constexpr unsigned kTemplateFlags = kTwoStage | kGated | kMultiStep | kMLWinds | kBigGrid | kPblClosure | kEmitKeys;
constexpr unsigned kOptionalModules = MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DIFF_MESO | MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI;
constexpr unsigned kTailModules = MPHIP_MOD_LOSS_ZERO | MPHIP_MOD_DECAY | MPHIP_MOD_WET_DEPO | MPHIP_MOD_DRY_DEPO;
constexpr unsigned kMovers = MPHIP_MOD_POSITION | MPHIP_MOD_ADVECT | MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DIFF_MESO | MPHIP_MOD_DIFF_PBL
  | MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI | MPHIP_MOD_ISOSURF | MPHIP_MOD_POSITION2;

// module_bound_cond, mptrac.c:3800-3879 (mass, volume mixing ratio, trace gases with a surface time series, age of air)
__device__ __forceinline__ void bound_cond(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevAtm &a,
                                           long long i, const Particle &P, const DevTracerSeries *tracers) {
  // (mptrac.c:3800-3804 as written there: the CFC-10 index is tested for "non-zero", not for "absent")
  const int *tr = ctl.qnt_tracer;
  if (ctl.qnt_m < 0 && ctl.qnt_vmr < 0 && tr[MPHIP_TR_CCL4] && tr[MPHIP_TR_CCL3F] < 0 && tr[MPHIP_TR_CCL2F2] < 0
      && tr[MPHIP_TR_N2O] < 0 && tr[MPHIP_TR_SF6] < 0 && ctl.qnt_aoa < 0)
    return;
  if (!in_boundary_region(ctl, M, A, P))
    return;
  if (ctl.qnt_m >= 0 && ctl.bound_mass >= 0)
    a.q[ctl.qnt_m][i] = ctl.bound_mass + ctl.bound_mass_trend * P.time;
  if (ctl.qnt_vmr >= 0 && ctl.bound_vmr >= 0)
    a.q[ctl.qnt_vmr][i] = ctl.bound_vmr + ctl.bound_vmr_trend * P.time;
  if (tracers)
    for (int k = 0; k < MPHIP_NTR; k++)
      if (tr[k] >= 0 && tracers->ntime[k] > 0)
        a.q[tr[k]][i] = clim_ts(*tracers, k, P.time);
  if (ctl.qnt_aoa >= 0)
    a.q[ctl.qnt_aoa][i] = P.time;
}

// common tail of decay / wet / dry deposition (mptrac.c:4251-4260, 6279-6288, 4786-4795)
__device__ __forceinline__ void apply_loss(const mphip_ctl_t &ctl, const DevAtm &a, long long i, double aux,
                                           int qnt_mloss, double rate) {
  if (ctl.qnt_m >= 0) {
    double m = a.q[ctl.qnt_m][i];
    if (qnt_mloss >= 0)
      a.q[qnt_mloss][i] += m * (1 - aux);
    a.q[ctl.qnt_m][i] = m * aux;
    if (ctl.qnt_loss_rate >= 0)
      a.q[ctl.qnt_loss_rate][i] += rate;
  }
  if (ctl.qnt_vmr >= 0)
    a.q[ctl.qnt_vmr][i] *= aux;
}

// Shortcuts of the deposition modules (DevMet::ps_skip / pct_skip).  Both modules start with an interpolation of
// a surface field and return if the particle is above it; for a time weight in [0, 1] the interpolated value
// (bilinear or nearest neighbour, then blended in time) is not below the smallest value of the two snapshots,
// so a particle below that bound (minus a guard band for the rounding) returns either way.
__device__ __forceinline__ bool between_snapshots(const DevMet &M, const Particle &P) {
  const double wt = time_weight(M, P.time);
  return wt >= 0.0 && wt <= 1.0;
}

__device__ __forceinline__ bool above_every_cloud_top(const DevMet &M, const Particle &P) {
  return P.p <= M.pct_skip && between_snapshots(M, P);
}

__device__ __forceinline__ bool above_every_surface_layer(const mphip_ctl_t &ctl, const DevMet &M, const Particle &P) {
  return P.p < M.ps_skip - ctl.dry_depo_dp && between_snapshots(M, P);
}

// module_wet_depo, mptrac.c:6170-6289
__device__ __forceinline__ void wet_depo(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevAtm &a,
                                         long long i, const Particle &P) {
  if (above_every_cloud_top(M, P))
    return;
  Stencil s = stencil_zero();
  stencil_2d(M, A, P.lon, P.lat, s);
  SurfB c2;
  load_sfb(M.sfc, M, s, c2);
  const double wt = time_weight(M, P.time);
  const double pct = sfb_time_2d(c2, s, wt, 0);
  if (!isfinite(pct) || P.p <= pct)
    return;
  const double pcb = sfb_time_2d(c2, s, wt, 1);
  const double cl = sfb_time_2d(c2, s, wt, 2);
  const double Is = libm_pow(1. / ctl.wet_depo_pre[0] * cl, 1. / ctl.wet_depo_pre[1]);
  if (Is < 0.01)
    return;
  stencil_3d(M, A, P.p, P.lon, P.lat, s);
  CloudCorners c;
  load_cloud(M, s, c);
  const double lwc = cloud_time_3d(c, s, wt, 0);
  const double rwc = cloud_time_3d(c, s, wt, 1);
  const double iwc = cloud_time_3d(c, s, wt, 2);
  const double swc = cloud_time_3d(c, s, wt, 3);
  const bool inside = (lwc > 0 || rwc > 0 || iwc > 0 || swc > 0);
  const double t = temp_time_3d(M, s, wt);

  double lambda = 0;
  if (inside) {
    double eta;
    if (t > kWdTLiquid)
      eta = 1;
    else if (t <= kWdTIce)
      eta = ctl.wet_depo_ic_ret_ratio;
    else
      eta = lin(kWdTLiquid, 1, kWdTIce, ctl.wet_depo_ic_ret_ratio, t);
    if (ctl.wet_depo_ic_a > 0)
      lambda = ctl.wet_depo_ic_a * libm_pow(Is, ctl.wet_depo_ic_b) * eta;
    else if (ctl.wet_depo_ic_h[0] > 0) {
      double h = ctl.wet_depo_ic_h[0] * libm_exp(ctl.wet_depo_ic_h[1] * (1. / t - 1. / kTRef));
      if (ctl.wet_depo_so2_ph > 0) {
        const double H_ion = libm_pow(10., -ctl.wet_depo_so2_ph);
        const double K_1 = kSO2K1Ref * libm_exp(kSO2K1Temp * (1. / t - 1. / kTRef));
        const double K_2 = kSO2K2Ref * libm_exp(kSO2K2Temp * (1. / t - 1. / kTRef));
        h *= (1. + K_1 / H_ion + K_1 * K_2 / (H_ion * H_ion));
      }
      const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
      lambda = h * kRI * t * Is / 3.6e6 / dz * eta;
    }
  } else {
    const double eta = (t > kWdTLiquidBC) ? 1 : ctl.wet_depo_bc_ret_ratio;
    if (ctl.wet_depo_bc_a > 0)
      lambda = ctl.wet_depo_bc_a * libm_pow(Is, ctl.wet_depo_bc_b) * eta;
    else if (ctl.wet_depo_bc_h[0] > 0) {
      const double h = ctl.wet_depo_bc_h[0] * libm_exp(ctl.wet_depo_bc_h[1] * (1. / t - 1. / kTRef));
      const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
      lambda = h * kRI * t * Is / 3.6e6 / dz * eta;
    }
  }
  const double aux = libm_exp(-P.dt * lambda);
  apply_loss(ctl, a, i, aux, ctl.qnt_mloss_wet, lambda);
}

// module_dry_depo, mptrac.c:4753-4796
__device__ __forceinline__ void dry_depo(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevAtm &a,
                                         long long i, const Particle &P) {
  if (above_every_surface_layer(ctl, M, P))
    return;
  Stencil s = stencil_zero();
  stencil_2d(M, A, P.lon, P.lat, s);
  SurfA c2;
  load_sfa(M, s, c2);
  const double ps = sfa_time_2d(c2, s, time_weight(M, P.time), 0);
  if (P.p < ps - ctl.dry_depo_dp)
    return;
  const double dz = 1000. * (zfromp(ps - ctl.dry_depo_dp) - zfromp(ps));
  double v_dep;
  if (ctl.qnt_rp > 0 && ctl.qnt_rhop > 0) {   // "> 0" as the reference, mptrac.c:4769
    const double t = temperature_at(M, A, P.time, P.p, P.lon, P.lat);
    v_dep = sedi(P.p, t, a.q[ctl.qnt_rp][i], a.q[ctl.qnt_rhop][i], libm_tables());
  } else
    v_dep = ctl.dry_depo_vdep;
  const double aux = libm_exp(-P.dt * v_dep / dz);
  apply_loss(ctl, a, i, aux, ctl.qnt_mloss_dry, v_dep / dz);
}

// module_wet_depo (mptrac.c:6170-6289) and module_dry_depo (mptrac.c:4753-4796) of the lean kernels as ONE function:
// the general value code on the lean stencil set-up, with the gathers of the two modules in two rounds instead of
// five dependent ones -- the surface records both start from ({pct, pcb, cl} and {ps, pbl}: one horizontal stencil
// at the final position serves both), then the level records at the particle (the cloud-water record of the wet part
// and ONE temperature for both parts: they interpolate it at the same stencil).  `wet(aux, rate)` / `dry(aux, rate)`
// receive the factor exp(-dt lambda) and the loss rate of a particle the module acts on, wet first
// (mptrac.c:7983-7993): the kernels that move the particles apply it at once (depo_pair_fast), the launch beside
// module_mixing keeps it for later.
template <bool BIG = false, class WetSink, class DrySink>
__device__ __forceinline__ void depo_pair_factor(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevAtm &a,
                                                 long long i, const Particle &P, Stencil &s, bool want_wet, bool want_dry,
                                                 WetSink &&wet, DrySink &&dry) {
  want_wet = want_wet && !above_every_cloud_top(M, P);
  want_dry = want_dry && !above_every_surface_layer(ctl, M, P);
  if (!(want_wet | want_dry))
    return;
  const double wt = time_weight(M, P.time);
  SurfB c2;
  SurfA c1;
  if (want_wet)
    load_sfb(M.sfc, M, s, c2);
  if (want_dry)
    load_pair_2d32(M.sfa, M, s, c1);
  bool wet_go = false, dry_go = false;
  double pct = 0, pcb = 0, Is = 0, dz_dry = 0;
  if (want_wet) {
    pct = sfb_time_2d(c2, s, wt, 0);
    if (isfinite(pct) && P.p > pct) {
      pcb = sfb_time_2d(c2, s, wt, 1);
      const double cl = sfb_time_2d(c2, s, wt, 2);
      Is = libm_pow(1. / ctl.wet_depo_pre[0] * cl, 1. / ctl.wet_depo_pre[1]);
      wet_go = !(Is < 0.01);
    }
  }
  if (want_dry) {
    const double ps = pair_time_2d_fast(c1, s, wt, 0);
    dry_go = !(P.p < ps - ctl.dry_depo_dp);
    if (dry_go)
      dz_dry = 1000. * (zfromp(ps - ctl.dry_depo_dp) - zfromp(ps));
  }
  if (!(wet_go | dry_go))
    return;
  const bool dry_sedi = dry_go && ctl.qnt_rp > 0 && ctl.qnt_rhop > 0;   // "> 0" as the reference, mptrac.c:4769
  double t = 0;
  CloudCorners c;
  if (wet_go | dry_sedi) {
    vert_fast(M, A, P.p, s);
    if (wet_go)
      load_cloud(M, s, c);
    t = temp_fast<BIG>(M, s, wt);
  }
  if (wet_go) {
    const double lwc = cloud_time_3d(c, s, wt, 0);
    const double rwc = cloud_time_3d(c, s, wt, 1);
    const double iwc = cloud_time_3d(c, s, wt, 2);
    const double swc = cloud_time_3d(c, s, wt, 3);
    const bool inside = (lwc > 0 || rwc > 0 || iwc > 0 || swc > 0);
    double lambda = 0;
    if (inside) {
      double eta;
      if (t > kWdTLiquid)
        eta = 1;
      else if (t <= kWdTIce)
        eta = ctl.wet_depo_ic_ret_ratio;
      else
        eta = lin(kWdTLiquid, 1, kWdTIce, ctl.wet_depo_ic_ret_ratio, t);
      if (ctl.wet_depo_ic_a > 0)
        lambda = ctl.wet_depo_ic_a * libm_pow(Is, ctl.wet_depo_ic_b) * eta;
      else if (ctl.wet_depo_ic_h[0] > 0) {
        double h = ctl.wet_depo_ic_h[0] * libm_exp(ctl.wet_depo_ic_h[1] * (1. / t - 1. / kTRef));
        if (ctl.wet_depo_so2_ph > 0) {
          const double H_ion = libm_pow(10., -ctl.wet_depo_so2_ph);
          const double K_1 = kSO2K1Ref * libm_exp(kSO2K1Temp * (1. / t - 1. / kTRef));
          const double K_2 = kSO2K2Ref * libm_exp(kSO2K2Temp * (1. / t - 1. / kTRef));
          h *= (1. + K_1 / H_ion + K_1 * K_2 / (H_ion * H_ion));
        }
        const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
        lambda = h * kRI * t * Is / 3.6e6 / dz * eta;
      }
    } else {
      const double eta = (t > kWdTLiquidBC) ? 1 : ctl.wet_depo_bc_ret_ratio;
      if (ctl.wet_depo_bc_a > 0)
        lambda = ctl.wet_depo_bc_a * libm_pow(Is, ctl.wet_depo_bc_b) * eta;
      else if (ctl.wet_depo_bc_h[0] > 0) {
        const double h = ctl.wet_depo_bc_h[0] * libm_exp(ctl.wet_depo_bc_h[1] * (1. / t - 1. / kTRef));
        const double dz = 1e3 * (zfromp(pct) - zfromp(pcb));
        lambda = h * kRI * t * Is / 3.6e6 / dz * eta;
      }
    }
    wet(libm_exp(-P.dt * lambda), lambda);
  }
  if (dry_go) {
    const double v_dep = dry_sedi ? sedi(P.p, t, a.q[ctl.qnt_rp][i], a.q[ctl.qnt_rhop][i], libm_tables()) : ctl.dry_depo_vdep;
    dry(libm_exp(-P.dt * v_dep / dz_dry), v_dep / dz_dry);
  }
}

template <bool BIG = false>
__device__ __forceinline__ void depo_pair_fast(const mphip_ctl_t &ctl, const DevMet &M, const Axes &A, const DevAtm &a,
                                               long long i, const Particle &P, Stencil &s, bool want_wet, bool want_dry) {
  depo_pair_factor<BIG>(ctl, M, A, a, i, P, s, want_wet, want_dry,
                        [&](double aux, double rate) { apply_loss(ctl, a, i, aux, ctl.qnt_mloss_wet, rate); },
                        [&](double aux, double rate) { apply_loss(ctl, a, i, aux, ctl.qnt_mloss_dry, rate); });
}

// One thread per particle, grid-stride.  Workgroup b of the launch is mapped
// to logical block (b % 8) * (n / 8) + b / 8 so that each XCD (the dispatcher
// places workgroup b on XCD b % 8) walks one contiguous eighth of the particle
// arrays: with cell-sorted particles each XCD's L2 then holds one region of
// the meteo grid instead of all of it.
// Waves per SIMD the register allocation aims at: 4 for the lean instantiations (128 VGPRs without scratch --
// reached by reducing the wind interpolation one element pair at a time, MPHIP_WIND_SERIAL, and by drawing
// the random numbers where they are used, MPHIP_RNG_EARLY 0), 3 for the general ones
#ifndef MPHIP_STEP_WAVES_PER_SIMD
#define MPHIP_STEP_WAVES_PER_SIMD 3
#endif
// ... measured per instantiation (round 3): the one with every module and the model-level one gain 8 % from a
// fourth wave although they then keep 52 / 88 bytes per lane in scratch (1.60 -> 1.48 ms, 2.89 -> 2.65 ms on the
// C3 particles); the pressure-level one (176 bytes) does not (Euler 1.11 -> 1.31 ms)
#ifndef MPHIP_GENERIC_WAVES_PER_SIMD
#define MPHIP_GENERIC_WAVES_PER_SIMD 4
#endif
#ifndef MPHIP_LEAN_WAVES_PER_SIMD
#define MPHIP_LEAN_WAVES_PER_SIMD 4
#endif
// 1: the specialised instantiations evaluate the random numbers of the
// stochastic modules between the gathers of the Runge-Kutta stages and their
// first use (rs[] is a pure function of counter and particle index)
#ifndef MPHIP_PARAMS_RELOAD
#define MPHIP_PARAMS_RELOAD 1
#endif
#ifndef MPHIP_RNG_EARLY
#define MPHIP_RNG_EARLY 0
#endif

// Particle state is streamed: every array element is read once and written once per launch, while the lines of
// the meteo records are shared by neighbouring waves.  MPHIP_STATE_NT 1 marks the state accesses of the step
// kernel non-temporal, so that they do not push meteo lines out of the L2.
#ifndef MPHIP_STATE_NT
#define MPHIP_STATE_NT 0
#endif
template <class T>
__device__ __forceinline__ T ld_state(const T *p) {
#if MPHIP_STATE_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <class T>
__device__ __forceinline__ void st_state(T *p, T v) {
#if MPHIP_STATE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// keeps a value where it was computed (the optimiser would sink the whole chain to its first use)
__device__ __forceinline__ void pin(double &x) {
  asm volatile("" : "+v"(x));
}

// 1: one Box-Muller pair behind the gathers of every Runge-Kutta stage (turb A, turb B, meso A, meso B + the
// convection uniform) instead of a whole triple behind stages 0 and 1: every wait of the integrator has
// independent arithmetic to cover it
#ifndef MPHIP_RNG_BALANCED
#define MPHIP_RNG_BALANCED 1
#endif

__device__ __forceinline__ void emit_sort_keys(const EmitKeys &E, const mphip_ctl_t &ctl, const DevMet &M, const Axes &A,
                                               long long i, const Particle &P) {
  if (E.depo_busy)      // (P.dt: the time step of THIS step, the guard of the deposition modules' particle loop)
    E.depo_busy[i] = P.dt != 0
      && (((E.depo_mask & MPHIP_MOD_WET_DEPO) && !above_every_cloud_top(M, P))
          || ((E.depo_mask & MPHIP_MOD_DRY_DEPO) && !above_every_surface_layer(ctl, M, P)));
  if (E.cell)
    E.cell[i] = box_cell(E.grid, E.box_t0, E.box_t1, P.time, P.lon, P.lat, P.p, E.ens, i, E.ngrid);
  double dt = 0.0;   // module_timesteps of the next step, mptrac.c:6016-6041
  if (E.direction * (P.time - E.t_start) >= 0 && E.direction * (P.time - E.t_stop) <= 0 && E.direction * (P.time - E.t_next) < 0)
    dt = E.t_next - P.time;
  if (M.local && (P.lon <= A.lon[0] || P.lon >= A.lon[M.nx - 1] || P.lat <= M.latmin || P.lat >= M.latmax))
    dt = 0.0;
  E.dt_next[i] = dt;
  Stencil s;
  raw_cell_fast(M, A, P.lon, P.lat, P.p, s);
  E.keys[i] = (uint32_t) ((s.ix * M.ny + s.iy) * M.np + s.ip);
}

struct RngEarly {
  unsigned mask;
  uint64_t ctr_turb, ctr_meso, ctr_conv, g;
  const double *ltab;
  double turb[3], meso[3], conv;
  double e, o;   // first pair of the triple under construction
  // pair `which` (0, 1) of the triple rs[3g .. 3g + 2] of a module_rng call with base counter c0; the second
  // pair completes out[0 .. 2] (selection as normal_triple)
  __device__ __forceinline__ void pair_of(uint64_t c0, int which, double *out) {
    const uint64_t i0 = 3 * g;
    const uint64_t y = (c0 + (i0 & ~1ull)) * kSquaresKey;
    if (which == 0) {
      normal_pair_from(ltab, y, e, o);
      pin(e);
      pin(o);
    } else {
      double eb, ob;
      normal_pair_from(ltab, y + 2 * kSquaresKey, eb, ob);
      const bool odd = (i0 & 1) != 0;
      out[0] = odd ? o : e;
      out[1] = odd ? eb : o;
      out[2] = odd ? ob : eb;
      pin(out[0]);
      pin(out[1]);
      pin(out[2]);
    }
  }
  __device__ __forceinline__ void operator()(int stage) {
#if MPHIP_RNG_BALANCED
    if (stage < 2 && (mask & MPHIP_MOD_DIFF_TURB))
      pair_of(ctr_turb, stage, turb);
    if (stage >= 2 && (mask & MPHIP_MOD_DIFF_MESO))
      pair_of(ctr_meso, stage - 2, meso);
    if (stage == 3 && (mask & MPHIP_MOD_CONVECTION)) {
      conv = uniform01(ctr_conv + g);
      pin(conv);
    }
#else
    if (stage == 0 && (mask & MPHIP_MOD_DIFF_TURB)) {
      normal_triple(ltab, ctr_turb, g, turb[0], turb[1], turb[2]);
      pin(turb[0]);
      pin(turb[1]);
      pin(turb[2]);
    }
    if (stage == 1 && (mask & MPHIP_MOD_DIFF_MESO)) {
      normal_triple(ltab, ctr_meso, g, meso[0], meso[1], meso[2]);
      pin(meso[0]);
      pin(meso[1]);
      pin(meso[2]);
    }
    if (stage == 2 && (mask & MPHIP_MOD_CONVECTION)) {
      conv = uniform01(ctr_conv + g);
      pin(conv);
    }
#endif
  }
};
// multi-step instantiations: 1 = the wind corners stay cached from one time step to the next too.  Measured (C3,
// 1e7 particles, 20 steps per launch): 0.94 ms per step against 0.78 without -- the 51 registers are then live
// through every module and the kernel keeps ~200 bytes per lane in scratch; off
#ifndef MPHIP_MULTI_KEEP_WIND
#define MPHIP_MULTI_KEEP_WIND 0
#endif
#ifndef MPHIP_MULTI_WAVES_PER_SIMD
#define MPHIP_MULTI_WAVES_PER_SIMD 4
#endif
#ifndef MPHIP_SPLITB_WAVES_PER_SIMD
#define MPHIP_SPLITB_WAVES_PER_SIMD 4
#endif
#ifndef MPHIP_ML_OFF32
#define MPHIP_ML_OFF32 1
#endif
#ifndef MPHIP_ML_WAVES_PER_SIMD
#define MPHIP_ML_WAVES_PER_SIMD 3   // the lean model-level instantiations: 148-158 VGPRs without scratch; at four waves (128 VGPRs) they spill 35-56 dwords: C3z 1.98 -> 1.81 ms per step (profiles/r04_variants.txt item 11)
#endif
#ifndef MPHIP_PBL_WAVES_PER_SIMD
#define MPHIP_PBL_WAVES_PER_SIMD 4
#endif
template <unsigned CT>
__global__ __launch_bounds__(256, kLeanML<CT> ? MPHIP_ML_WAVES_PER_SIMD : !kRuntimeMask<CT> && (CT & kPblClosure) ? MPHIP_PBL_WAVES_PER_SIMD : !kRuntimeMask<CT> ? ((CT & kMultiStep) ? MPHIP_MULTI_WAVES_PER_SIMD
                                                        : (CT & MPHIP_MOD_ADVECT) || CT == MPHIP_MOD_TIMESTEPS ? MPHIP_LEAN_WAVES_PER_SIMD : MPHIP_SPLITB_WAVES_PER_SIMD)
                                   : (CT == kMaskGenericPL ? MPHIP_STEP_WAVES_PER_SIMD : MPHIP_GENERIC_WAVES_PER_SIMD)) void step_kernel(
  const StepParams S) {
  extern __shared__ double s_axes[];
  const unsigned mask = kRuntimeMask<CT> ? S.mask : (CT & ~kTemplateFlags);
  // column indices of the model-level fields: 32 bits in the lean instantiations (launch_step's size check)
  using MLCol = std::conditional_t<kLeanML<CT> && MPHIP_ML_OFF32 && !kBig<CT>, uint32_t, size_t>;
  const DevMet &M = S.met;
  const DevAtm &a = S.atm;
  const mphip_ctl_t &ctl = S.ctl;

  const Axes A = load_axes(M, s_axes);
  // climatological tropopause table next to the axes (7.7 kB): the weights of
  // module_diff_turb / module_decay index it per lane, several times a step
  const DevClim *clim = S.clim;
  if ((mask | (kRuntimeMask<CT> ? 0u : (S.mask & kTailModules))) & (MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DECAY)) {
    double *dst = s_axes + ((axes_doubles(M) * 8 + (size_t) M.lut_size * 2 + 15) & ~(size_t) 15) / 8;
    const double *src = (const double *) S.clim;
    for (int i = threadIdx.x; i < (int) (sizeof(DevClim) / sizeof(double)); i += blockDim.x)
      dst[i] = src[i];
    clim = (const DevClim *) dst;
  }
  // ... and the tables of the C library's log and exp behind it (4 kB; with the boundary-layer closure pow's too, 7 kB)
  const double *ltab = libm_tables();   // (the general instantiations read them from device memory)
  if (!kRuntimeMask<CT> && (mask & (MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DIFF_MESO | MPHIP_MOD_SEDI))) {
    double *dst = s_axes + ((axes_doubles(M) * 8 + (size_t) M.lut_size * 2 + 15) & ~(size_t) 15) / 8
      + sizeof(DevClim) / sizeof(double);
    const double *src = libm_tables();
    for (int i = threadIdx.x; i < ((CT & kPblClosure) ? kLibmDoubles : kLibmLogExpDoubles); i += blockDim.x)
      dst[i] = src[i];
    ltab = dst;
  }
  __syncthreads();

  const int nb = S.nblocks_logical;
  const int lb = S.xcd_map ? (int) (blockIdx.x % 8) * (nb / 8) + (int) (blockIdx.x / 8) : (int) blockIdx.x;
  const long long per_block = S.per_block;
  const long long first = (long long) lb * per_block;
  long long last = first + per_block;
  if (last > a.np)
    last = a.np;

  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
#if MPHIP_PARAMS_RELOAD
    // re-read the launch parameters from the kernarg segment in every iteration (scalar loads at
    // the point of use) instead of keeping all of them live across the loop, which spills SGPRs
    auto kp = (const __attribute__((address_space(4))) StepParams *) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const StepParams &S = *(const StepParams *) kp;
    const DevMet &M = S.met;
    const DevAtm &a = S.atm;
    const mphip_ctl_t &ctl = S.ctl;
#endif
    constexpr bool multi = (!kRuntimeMask<CT> && (CT & kMultiStep) != 0) || CT == kMaskGenericMLMulti;
    const int nsteps = multi ? S.nsteps : 1;
    double t_now = S.t;
    uint64_t c_turb = S.ctr_turb, c_meso = S.ctr_meso, c_conv = S.ctr_conv, c_pbl = S.ctr_pbl;
    // multi-step instantiations: the particle, its mesoscale wind perturbations and the wind corners it used last
    // stay in registers from one step to the next (the stores of every step remain; what a step would load is what
    // the step before stored, and the meteo arrays do not change inside a launch) -- the first Runge-Kutta stage
    // of a step then gathers only in the lanes that left their grid cell since module_diff_meso of the step before
    Particle P;
    WindCache wc;
    float up = 0.f, vp = 0.f, wp = 0.f;
    for (int step = 0; step < nsteps; step++, t_now += S.t_stride, c_turb += S.ctr_stride, c_meso += S.ctr_stride,
             c_conv += S.ctr_stride, c_pbl += S.ctr_stride) {
    const bool fused_sort = a.perm != nullptr;
    if (!multi || step == 0) {
    if (fused_sort) {   // the gather of module_sort_help (mptrac.c:5944-5949) for time, p, lon, lat
      const long long src = a.perm[i];
      P.time = a.s_time[src];
      P.lon = a.s_lon[src];
      P.lat = a.s_lat[src];
      P.p = a.s_p[src];
      for (int k = 0; k < a.nq_perm; k++)   // the quantities move here too (what follows reads them at slot i)
        a.q[k][i] = a.s_q[k][src];
    } else {
      P.time = ld_state(&a.time[i]);
      P.lon = ld_state(&a.lon[i]);
      P.lat = ld_state(&a.lat[i]);
      P.p = ld_state(&a.p[i]);
    }
    if (multi) {   // (before any step can leave early: dt = 0 ...)
      if (MPHIP_MULTI_KEEP_WIND)
        wind_cache_reset(wc, true);
      if (mask & MPHIP_MOD_DIFF_MESO) {
        up = ld_state(&a.up[i]);
        vp = ld_state(&a.vp[i]);
        wp = ld_state(&a.wp[i]);
      }
    }
    }
    if (CT == kMaskGeneric && (mask & (MPHIP_MOD_ADVECT_INIT | MPHIP_MOD_ISOSURF_INIT))) {   // no dt guard (check_dt = 0)
      P.dt = 0;
      if (mask & MPHIP_MOD_ISOSURF_INIT)   // before module_advect_init, mptrac.c:7866-7870
        a.iso[i] = isosurf_value(ctl, M, A, P);
      if (mask & MPHIP_MOD_ADVECT_INIT)
        a.p[i] = pressure_from_zeta(M, A, P.time, a.q[ctl.qnt_zeta][i], P.lon, P.lat);
      continue;
    }
    if (kModelLevels<CT> && (mask & MPHIP_MOD_ADVECT_INIT)) {
      a.p[i] = pressure_from_zeta_fast(M, A, P.time, a.q[ctl.qnt_zeta][i], P.lon, P.lat);
      continue;
    }
    // module_timesteps and the store of cache->dt are run-time choices in every instantiation
    // (sort steps compute dt before the sort; launches split around module_mixing share it)
    if (S.mask & MPHIP_MOD_TIMESTEPS) {
      P.dt = timestep_of(ctl, M, A, P.time, P.lon, P.lat, t_now);
      if (S.mask & kStoreDt)
        a.dt[i] = P.dt;
    } else
      P.dt = a.dt[i];
    if (P.dt == 0) {   // guard of PARTICLE_LOOP(..., check_dt = 1), mptrac.h:1759
      if (fused_sort) {
        a.time[i] = P.time;
        a.lon[i] = P.lon;
        a.lat[i] = P.lat;
        a.p[i] = P.p;
      }
      if (CT == kMaskGeneric && (mask & MPHIP_MOD_ISOSURF))   // module_isosurf has check_dt = 0
        a.p[i] = isosurf_pressure(ctl, M, A, a, P, ctl.isosurf <= 3 ? a.iso[i] : 0.0);
      if constexpr (!kRuntimeMask<CT> && (CT & kEmitKeys) != 0)
        if (S.emit.keys)
          emit_sort_keys(S.emit, ctl, M, A, i, P);
      if constexpr (!kRuntimeMask<CT> && (CT & kPblClosure) != 0)
        if (S.mask & MPHIP_MOD_ISOSURF) {
          P.p = isosurf_call(&ctl, &M, A, &a, P.time, P.p, P.lon, P.lat, ctl.isosurf <= 3 ? a.iso[i] : 0.0);
          a.p[i] = P.p;
        }
      continue;
    }
    // random numbers belong to the external slot (rs[3 * ip + k], mptrac.c:4645)
    const uint64_t g = (uint64_t) (a.ip0 + (a.ext ? (long long) ld_state(&a.ext[i]) : i));

    // specialised instantiations = RK4 on pressure levels (launch_step): 4 stages, all hooks run
    constexpr bool early = MPHIP_RNG_EARLY && !kRuntimeMask<CT> && (CT & MPHIP_MOD_ADVECT) && !(CT & kTwoStage) && !kLeanML<CT>;   // (the hooks belong to the four-stage integrator)
    RngEarly pre;
    pre.mask = mask;
    pre.ctr_turb = c_turb;
    pre.ctr_meso = c_meso;
    pre.ctr_conv = c_conv;
    pre.g = g;
    pre.ltab = ltab;

    // the specialised instantiations run the lean versions (lat/lon grid, pressure table: launch_step)
    constexpr bool lean = !kRuntimeMask<CT>;
    if (!kModelLevels<CT> && !(multi && MPHIP_MULTI_KEEP_WIND))   // (model-level winds: the corners are first needed by module_diff_meso -- defined there,
      wind_cache_reset(wc, CT != kMaskGeneric && (lean || !MPHIP_EXACT_DIV));   //  or 48 registers would be held through the whole advection)
    // (MPHIP_EXACT_DIV: the general kernels gather with plain loads -- with the IEEE division sequences between the
    // asynchronous gathers and their wait the register allocator reuses their registers, which the machine-code
    // check of the build refuses)
    if (mask & MPHIP_MOD_POSITION) {
      if (lean)
        position_fast(M, A, P);
      else
        position(M, A, P);
    }
    if (mask & MPHIP_MOD_ADVECT) {
      // model-level advection (ADVECT_VERT_COORD 1 / 3) runs in the generic instantiation only
      if (CT == kMaskGeneric && ctl.advect_vert_coord == 2) {
        advect_mlp(ctl, M, A, P);      // pressure advection, winds from the model levels
      } else if (kModelLevels<CT> && ctl.advect_vert_coord == 2) {
        int kz = a.kz[i];
        advect_mlp_fast<MLCol>(ctl, M, A, P, kz);
        a.kz[i] = kz;
      } else if (CT == kMaskGeneric && (ctl.advect_vert_coord == 1 || ctl.advect_vert_coord == 3)) {
        const int qnt = ctl.advect_vert_coord == 1 ? ctl.qnt_zeta : ctl.qnt_eta;
        double zeta;
        advect_ml(ctl, M, A, P, zeta);
        a.q[qnt][i] = zeta;
      } else if (kModelLevels<CT>) {
        const int qnt = ctl.advect_vert_coord == 1 ? ctl.qnt_zeta : ctl.qnt_eta;
        double zeta;
        int kz = a.kz[i];     // vertical index of the last step: first guess of this step's searches
        advect_ml_fast<MLCol>(ctl, M, A, P, zeta, kz);
        a.kz[i] = kz;
        a.q[qnt][i] = zeta;
      } else if (lean && early)
        advect_rk4_fast(M, A, P, pre, wc);
      else if (lean && (CT & kTwoStage)) {
        NoHook none;
        advect_fast<2, kBig<CT>>(M, A, P, none, wc, ctl.advect == 1);
      } else if (lean) {
        NoHook none;
        advect_fast<4, kBig<CT>>(M, A, P, none, wc);
      } else
        advect(ctl, M, A, P, wc);
    }
    // (S.mask is wave-uniform: scalar tests)
    const unsigned opt = (lean && (CT & kGated)) ? (mask & S.mask & kOptionalModules) : (mask & kOptionalModules);
    if (opt & MPHIP_MOD_DIFF_TURB) {
      if (lean)
        diff_turb_fast(ctl, M, A, *clim, P, c_turb, g, early ? pre.turb : nullptr, ltab);
      else
        diff_turb(ctl, M, A, *clim, P, c_turb, g, early ? pre.turb : nullptr, ltab);
    }
#ifndef MPHIP_PBL_EMPTY
#define MPHIP_PBL_EMPTY 0     // 1: experiment -- the kPblClosure instantiations without the closure's code
#endif
    constexpr bool lean_pbl = lean && (CT & kPblClosure) != 0 && !MPHIP_PBL_EMPTY;
    if (((CT == kMaskGeneric && (mask & MPHIP_MOD_DIFF_PBL)) || (lean_pbl && (S.mask & MPHIP_MOD_DIFF_PBL)))
        && !above_every_boundary_layer(M, P)) {
      // (the perturbations live in the cache arrays; a multi-step launch with module_diff_meso holds them in registers)
      const bool held = multi && (mask & MPHIP_MOD_DIFF_MESO);
      if (!held) {
        up = ld_state(&a.up[i]);
        vp = ld_state(&a.vp[i]);
        wp = ld_state(&a.wp[i]);
      }
      if constexpr (lean_pbl) {   // a call: the kernel keeps the registers of the instantiation without the closure
        PblState st = { P.time, P.lon, P.lat, P.p, P.dt, up, vp, wp };
        st = diff_pbl_call(&M, A, st, c_pbl, g, ltab);
        P.lon = st.lon;
        P.lat = st.lat;
        P.p = st.p;
        up = st.up;
        vp = st.vp;
        wp = st.wp;
        wind_cache_reset(wc, true);   // (the cached corners do not cross the call: module_diff_meso gathers its own)
      } else
        diff_pbl<false>(M, A, P, up, vp, wp, c_pbl, g, ltab);
      st_state(&a.up[i], up);
      st_state(&a.vp[i], vp);
      st_state(&a.wp[i], wp);
    }
    if (opt & MPHIP_MOD_DIFF_MESO) {
      if (!multi) {
        up = ld_state(&a.up[i]);
        vp = ld_state(&a.vp[i]);
        wp = ld_state(&a.wp[i]);
      }
      if (kGenericML<CT>)
        wind_cache_reset(wc, true);
      if (lean)   // (without a pressure-level advection before it there are no cached corners: the streaming version)
        diff_meso_fast<!(CT & MPHIP_MOD_ADVECT) || kLeanML<CT>, kBig<CT>>(ctl, M, A, P, up, vp, wp, c_meso, g, early ? pre.meso : nullptr, wc, ltab);
      else
        diff_meso(ctl, M, A, P, up, vp, wp, c_meso, g, early ? pre.meso : nullptr, wc, ltab);
      st_state(&a.up[i], up);
      st_state(&a.vp[i], vp);
      st_state(&a.wp[i], wp);
    }
    if (lean) {
      if (opt & (MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI)) {
        const bool sedi_on = (opt & MPHIP_MOD_SEDI) != 0;
        conv_sedi_fast<kBig<CT>>(ctl, M, A, P, opt, c_conv, g, early ? &pre.conv : nullptr,
                       sedi_on ? ld_state(&a.q[ctl.qnt_rp][i]) : 0.0, sedi_on ? ld_state(&a.q[ctl.qnt_rhop][i]) : 0.0, ltab);
      }
    } else {
      if (opt & MPHIP_MOD_CONVECTION)
        convection(ctl, M, A, P, c_conv, g, early ? &pre.conv : nullptr);
      if (opt & MPHIP_MOD_SEDI)
        sedimentation(M, A, P, a.q[ctl.qnt_rp][i], a.q[ctl.qnt_rhop][i]);
    }
    if (CT == kMaskGeneric && (mask & MPHIP_MOD_ISOSURF))
      P.p = isosurf_pressure(ctl, M, A, a, P, ctl.isosurf <= 3 ? a.iso[i] : 0.0);
    if constexpr (lean && (CT & kPblClosure) != 0)
      if (S.mask & MPHIP_MOD_ISOSURF)
        P.p = isosurf_call(&ctl, &M, A, &a, P.time, P.p, P.lon, P.lat, ctl.isosurf <= 3 ? a.iso[i] : 0.0);
    if (mask & MPHIP_MOD_POSITION2) {
      if (lean)
        position_fast(M, A, P);
      else
        position(M, A, P);
    }

    if ((mask & MPHIP_MOD_ADVECT) || fused_sort)
      st_state(&a.time[i], P.time);
    if ((mask & kMovers) || fused_sort) {
      st_state(&a.lon[i], P.lon);
      st_state(&a.lat[i], P.lat);
      st_state(&a.p[i], P.p);
    }
    if constexpr (lean && (CT & kEmitKeys) != 0)
      if (S.emit.keys)
        emit_sort_keys(S.emit, ctl, M, A, i, P);

    // module_bound_cond: in the instantiation with every module, and -- switched by the run-time mask -- in the
    // gated lean instantiations and the one without movers (the launch behind module_mixing)
    constexpr bool bound_rt = lean && ((CT & kGated) || (CT & ~kTemplateFlags) == MPHIP_MOD_TIMESTEPS);
    const unsigned bmask = bound_rt ? S.mask : (CT == kMaskGeneric ? mask : 0u);
    if (bmask & MPHIP_MOD_BOUND_COND)
      bound_cond(ctl, M, A, a, i, P, S.tracers);
    // the loss / decay / deposition modules behind the movers: in the lean instantiations a run-time choice
    // too (their template mask names the movers), so that they serve every combination of these modules
    const unsigned tmask = lean ? (S.mask & kTailModules) : mask;
    if (tmask & MPHIP_MOD_LOSS_ZERO)
      a.q[ctl.qnt_loss_rate][i] = 0;
    if (tmask & MPHIP_MOD_DECAY) {   // module_decay, mptrac.c:4241-4261
      const double w = tropo_weight(ctl, *clim, P.time, P.lat, P.p);
      const double tdec = w * ctl.tdec_trop + (1 - w) * ctl.tdec_strat;
      const double aux = libm_exp(ltab, -P.dt / tdec);
      apply_loss(ctl, a, i, aux, ctl.qnt_mloss_decay, 1. / tdec);
    }
    if (lean) {
      // (particles above every cloud top / surface layer of the two snapshots need no stencil at all)
      const bool wet = (tmask & MPHIP_MOD_WET_DEPO) && !above_every_cloud_top(M, P);
      const bool dry = (tmask & MPHIP_MOD_DRY_DEPO) && !above_every_surface_layer(ctl, M, P);
      if (wet || dry) {
        Stencil sd = stencil_zero();
        horiz_fast(M, A, P.lon, P.lat, sd);
        depo_pair_fast<kBig<CT>>(ctl, M, A, a, i, P, sd, wet, dry);
      }
    } else {
      if (mask & MPHIP_MOD_WET_DEPO)
        wet_depo(ctl, M, A, a, i, P);
      if (mask & MPHIP_MOD_DRY_DEPO)
        dry_depo(ctl, M, A, a, i, P);
    }
    if (bmask & MPHIP_MOD_BOUND_COND2)
      bound_cond(ctl, M, A, a, i, P, S.tracers);
    }
  }
}

// ---------------------------------------------------------------------------
// Trajectories with the wind grid staged through an LDS tile per workgroup (north_star: "the two bracketing met_t
// pressure-level grids staged through LDS tiles per thread-block").  What it serves: runs of time steps
// that are module_timesteps + module_position + module_advect only (pure trajectories, the reference's most common
// use), handed over as one launch (mphip_run_timesteps).  A workgroup takes its particles 256 at a time; for each
// round it finds the bounding box of the particles' stencil cells (wave reductions + LDS atomics), widens it by one
// column / level on every side, stages that box of two-snapshot wind records once (cooperative, contiguous along the
// levels; at most `tile_cells` records, levels cut first) and then lets every particle take ALL its steps: a
// Runge-Kutta stage whose stencil lies inside the tile reads its eight corner records from LDS, any other one from
// global memory as the other kernels do (load_wind_cached32).  Same arithmetic on the same records: the bits of the
// kernels without a tile (test_lds_tile_trajectories_equal_the_launches_without_a_tile).  Option "lds_tile";
// measured in profiles/r05_variants.txt item 6.
// ---------------------------------------------------------------------------
#ifndef MPHIP_TILE_WAVES_PER_SIMD
#define MPHIP_TILE_WAVES_PER_SIMD 3   // 20 KB of axes + a 24 KB tile: three workgroups per CU
#endif
template <int STAGES>
__global__ __launch_bounds__(256, MPHIP_TILE_WAVES_PER_SIMD) void traj_tile_kernel(const StepParams S, int tile_cells) {
  extern __shared__ double s_axes[];
  __shared__ int s_lo[3], s_hi[3];
  const DevMet &M = S.met;
  const DevAtm &a = S.atm;
  const mphip_ctl_t &ctl = S.ctl;
  const Axes A = load_axes(M, s_axes);
  float *s_tile = (float *) (s_axes + ((axes_doubles(M) * 8 + (size_t) M.lut_size * 2 + 15) & ~(size_t) 15) / 8);
  const int nb = S.nblocks_logical;
  const int lb = S.xcd_map ? (int) (blockIdx.x % 8) * (nb / 8) + (int) (blockIdx.x / 8) : (int) blockIdx.x;
  const long long first = (long long) lb * S.per_block;
  long long last = first + S.per_block;
  if (last > a.np)
    last = a.np;
  for (long long base = first; base < last; base += 256) {   // (block-uniform: every thread meets every barrier)
    const long long i = base + threadIdx.x;
    const bool live = i < last;
    if (threadIdx.x < 3) {
      s_lo[threadIdx.x] = 1 << 30;
      s_hi[threadIdx.x] = -1;
    }
    __syncthreads();   // (also: the axes are in place / the tile of the round before is no longer read)
    Particle P;
    P.time = P.lon = P.lat = P.p = P.dt = 0;
    int lo[3] = { 1 << 30, 1 << 30, 1 << 30 }, hi[3] = { -1, -1, -1 };
    if (live) {
      P.time = ld_state(&a.time[i]);
      P.lon = ld_state(&a.lon[i]);
      P.lat = ld_state(&a.lat[i]);
      P.p = ld_state(&a.p[i]);
      Stencil s0;
      stencil_3d_fast(M, A, P.p, P.lon, P.lat, s0);   // the cell the first stage will start in
      lo[0] = hi[0] = s0.ix;
      lo[1] = hi[1] = s0.iy;
      lo[2] = hi[2] = s0.ip;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
      for (int sft = 32; sft > 0; sft >>= 1) {
        lo[d] = min(lo[d], __shfl_xor(lo[d], sft));
        hi[d] = max(hi[d], __shfl_xor(hi[d], sft));
      }
      if ((threadIdx.x & 63) == 0) {
        atomicMin(&s_lo[d], lo[d]);
        atomicMax(&s_hi[d], hi[d]);
      }
    }
    __syncthreads();
    WindTile T;
    T.rec = s_tile;
    T.x0 = max(s_lo[0] - 1, 0);
    T.y0 = max(s_lo[1] - 1, 0);
    T.z0 = max(s_lo[2] - 1, 0);
    T.nx = min(s_hi[0] + 2, M.nx - 1) - T.x0 + 1;     // (+ 1: the upper corner of a stencil, + 1: the halo)
    T.ny = min(s_hi[1] + 2, M.ny - 1) - T.y0 + 1;
    T.nz = min(s_hi[2] + 2, M.np - 1) - T.z0 + 1;
    if (T.nx * T.ny * 2 > tile_cells) {               // too many columns (a round across the date line ...): a part of them
      const int side = max(2, (int) sqrtf((float) (tile_cells / 2)));
      T.nx = min(T.nx, side);
      T.ny = min(T.ny, max(2, tile_cells / 2 / T.nx));
    }
    T.nz = min(T.nz, tile_cells / (T.nx * T.ny));
    {
      const f32x2s *src = (const f32x2s *) M.wind;
      f32x2s *dst = (f32x2s *) s_tile;
      const int per_col = T.nz * 3, total = T.nx * T.ny * per_col;
      for (int f = threadIdx.x; f < total; f += 256) {
        const int col = f / per_col, within = f - col * per_col;
        const int cx = col / T.ny, cy = col - cx * T.ny;
        dst[f] = src[((size_t) (__umul24((unsigned) (T.x0 + cx), (unsigned) M.ny) + (unsigned) (T.y0 + cy)) * (size_t) M.np
                      + (size_t) T.z0) * 3 + (size_t) within];
      }
    }
    __syncthreads();
    if (!live)
      continue;
    double t_now = S.t;
    for (int step = 0; step < S.nsteps; step++, t_now += S.t_stride) {
      P.dt = (S.mask & MPHIP_MOD_TIMESTEPS) ? timestep_of(ctl, M, A, P.time, P.lon, P.lat, t_now) : a.dt[i];
      if ((S.mask & MPHIP_MOD_TIMESTEPS) && (S.mask & kStoreDt))
        a.dt[i] = P.dt;
      if (P.dt == 0)   // guard of PARTICLE_LOOP(..., check_dt = 1), mptrac.h:1759
        continue;
      WindCache wc;
      wind_cache_reset(wc, true);
      position_fast(M, A, P);
      NoHook none;
      advect_fast<STAGES, false>(M, A, P, none, wc, STAGES == 2 && ctl.advect == 1, &T);
      position_fast(M, A, P);
      st_state(&a.time[i], P.time);
      st_state(&a.lon[i], P.lon);
      st_state(&a.lat[i], P.lat);
      st_state(&a.p[i], P.p);
    }
  }
}

// ---------------------------------------------------------------------------
// Deposition launch (module_wet_depo and / or module_dry_depo alone: what follows module_mixing in a time step).
// Most particles leave these modules at once -- above every cloud top, above the surface layer -- but in the
// stored orders every wave holds a few that do not, and a wave pays for a gather round whatever the number of
// lanes that take part.  So a workgroup first sorts out which of its particles have anything to do (two
// comparisons against the bounds of DevMet::pct_skip / ps_skip: only p, time and dt are read), packs their
// numbers into a list in LDS, and then works through the list: full waves for the stencils, the gathers and the
// pow / exp of the modules, none for the rest.  Same arithmetic per particle as
// the fused kernel (depo_pair_fast); lean configurations only (launch_step).  (module_mixing's
// relaxation inside the first pass of this kernel was measured as well: 2.36 against 2.28 ms per step of C5 --
// its dependent gathers delay the list and the barrier behind it.)
// ---------------------------------------------------------------------------

#ifndef MPHIP_DEPO_WAVES_PER_SIMD
#define MPHIP_DEPO_WAVES_PER_SIMD 4   // 112 VGPRs without scratch (132 when left to the compiler: three waves); C5 +2 %
#endif
__global__ __launch_bounds__(256, MPHIP_DEPO_WAVES_PER_SIMD) void depo_kernel(const StepParams S) {
  extern __shared__ double s_axes[];
  __shared__ int s_count;
  const DevMet &M = S.met;
  const DevAtm &a = S.atm;
  const mphip_ctl_t &ctl = S.ctl;
  const Axes A = load_axes(M, s_axes);
  // the list of the workgroup's busy particles behind the axes (per_block entries: launch_step sizes the LDS)
  int *s_list = (int *) (s_axes + ((axes_doubles(M) * 8 + (size_t) M.lut_size * 2 + 15) & ~(size_t) 15) / 8);
  if (threadIdx.x == 0)
    s_count = 0;
  __syncthreads();
  const unsigned tmask = S.mask;
  const int nb = S.nblocks_logical;
  const int lb = S.xcd_map ? (int) (blockIdx.x % 8) * (nb / 8) + (int) (blockIdx.x / 8) : (int) blockIdx.x;
  const long long first = (long long) lb * S.per_block;
  long long last = first + S.per_block;
  if (last > a.np)
    last = a.np;
  const int lane = threadIdx.x & 63;
  // which particles have anything to do: p, time and dt only
  for (long long i = first + threadIdx.x; i < first + S.per_block; i += 256) {   // (whole waves stay together)
    bool busy = false;
    if (S.depo_busy)                  // decided by the launch that moved the particles (EmitKeys::depo_busy)
      busy = i < last && S.depo_busy[i] != 0;
    else if (i < last && a.dt[i] != 0) {   // guard of PARTICLE_LOOP(..., check_dt = 1), mptrac.h:1759
      Particle P;
      P.time = a.time[i];
      P.p = a.p[i];
      busy = ((tmask & MPHIP_MOD_WET_DEPO) && !above_every_cloud_top(M, P))
        || ((tmask & MPHIP_MOD_DRY_DEPO) && !above_every_surface_layer(ctl, M, P));
    }
    const unsigned long long mine = __ballot(busy);
    int at = 0;
    if (lane == 0 && mine)
      at = atomicAdd(&s_count, __builtin_popcountll(mine));
    at = __builtin_amdgcn_readfirstlane(at);
    if (busy)
      s_list[at + __builtin_popcountll(mine & ((1ull << lane) - 1))] = (int) (i - first);
  }
  __syncthreads();
  // ... and those, in full waves (the order inside the list does not matter: particles are independent)
  const int total = s_count;
  for (int t = threadIdx.x; t < total; t += 256) {
    const long long ip = first + s_list[t];
    Particle P;
    P.time = a.time[ip];
    P.lon = a.lon[ip];
    P.lat = a.lat[ip];
    P.p = a.p[ip];
    P.dt = a.dt[ip];
    const bool wet = (tmask & MPHIP_MOD_WET_DEPO) && !above_every_cloud_top(M, P);
    const bool dry = (tmask & MPHIP_MOD_DRY_DEPO) && !above_every_surface_layer(ctl, M, P);
    Stencil sd = stencil_zero();
    horiz_fast(M, A, P.lon, P.lat, sd);
    depo_pair_fast(ctl, M, A, a, ip, P, sd, wet, dry);
  }
}

// ---------------------------------------------------------------------------
// packing of the two bracketing snapshots
// ---------------------------------------------------------------------------

// One kernel builds every packed record from the per-slot staging copies
// (NULL source = field not uploaded -> zeros).
struct PackArgs {
  const float *f3[2][MPHIP_N3D];   // [snapshot][field]
  const float *f2[2][MPHIP_N2D];
  float *wind, *temp;
  f32x4 *cloud, *sfa, *sfb, *sfc, *sfd;
  f32x4 *cp2;                      // {cape,pel}0 {cape,pel}1 per column
  f32x4 *mx;                       // [2][cell] {z,pv}01, {o3,cc}01 (NULL: none)
  f32x4 *mx2;                      // [7][col] surface pairs of module_meteo (NULL: none)
  float *h2o;                      // {h2o}0 {h2o}1 (NULL: none)
  float *mlw;                      // model-level {ul,vl,zeta_dot} records (NULL: none)
  int mlw_third;                   // MPHIP_ZETA_DOTL, or MPHIP_WL for ADVECT_VERT_COORD 2 ({ul,vl,wl})
  float *zl2, *pl2;                // model-level {zetal0,zetal1}, {pl0,pl1} pairs (NULL: none)
  int *ml_mono;                    // cleared to 0 by a thread that finds a non-monotonic height column
  int nml;                         // model levels
  size_t ncell, ncol, ncell_ml;
};

__global__ void pack_kernel(PackArgs a) {
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < a.ncell; i += stride) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int k = 0; k < 3; k++)
        a.wind[6 * i + (k < 2 ? 2 * t + k : 4 + t)] = a.f3[t][MPHIP_U + k] ? a.f3[t][MPHIP_U + k][i] : 0.f;   // {u0,v0,u1,v1,w0,w1}
      a.temp[2 * i + t] = a.f3[t][MPHIP_T] ? a.f3[t][MPHIP_T][i] : 0.f;
      if (a.h2o)
        a.h2o[2 * i + t] = a.f3[t][MPHIP_H2O] ? a.f3[t][MPHIP_H2O][i] : 0.f;
      if (a.cloud) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; k++)
          v[k] = a.f3[t][MPHIP_LWC + k] ? a.f3[t][MPHIP_LWC + k][i] : 0.f;
        a.cloud[2 * i + t] = v;
      }
    }
    if (a.mx) {
#pragma unroll
      for (int pr = 0; pr < 2; pr++) {
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int f = 0; f < 2; f++)
            v[2 * t + f] = a.f3[t][MPHIP_Z + 2 * pr + f] ? a.f3[t][MPHIP_Z + 2 * pr + f][i] : 0.f;
        a.mx[(size_t) pr * a.ncell + i] = v;
      }
    }
  }
  if (a.mlw)
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < a.ncell_ml; i += stride)
#pragma unroll
      for (int t = 0; t < 2; t++) {
        a.mlw[6 * i + 3 * t + 0] = a.f3[t][MPHIP_UL] ? a.f3[t][MPHIP_UL][i] : 0.f;
        a.mlw[6 * i + 3 * t + 1] = a.f3[t][MPHIP_VL] ? a.f3[t][MPHIP_VL][i] : 0.f;
        a.mlw[6 * i + 3 * t + 2] = a.f3[t][a.mlw_third] ? a.f3[t][a.mlw_third][i] : 0.f;
      }
  if (a.zl2) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < a.ncell_ml; i += stride)
#pragma unroll
      for (int t = 0; t < 2; t++) {
        a.zl2[2 * i + t] = a.f3[t][MPHIP_ZETAL] ? a.f3[t][MPHIP_ZETAL][i] : 0.f;
        a.pl2[2 * i + t] = a.f3[t][MPHIP_PL] ? a.f3[t][MPHIP_PL][i] : 0.f;
      }
    // strict monotonicity of every height column (the fast model-level path relies on it)
    const size_t ncol_ml = a.ncell_ml / (size_t) a.nml;
    for (size_t c = blockIdx.x * (size_t) blockDim.x + threadIdx.x; c < 4 * ncol_ml; c += stride) {
      const int which = (int) (c / ncol_ml);        // 0, 1: zetal of met0 / met1; 2, 3: pl
      const float *h = a.f3[which & 1][which < 2 ? MPHIP_ZETAL : MPHIP_PL];
      if (!h)
        continue;
      h += (c % ncol_ml) * (size_t) a.nml;
      const bool asc = h[0] < h[1];
      bool ok = true;
      for (int k = 0; k + 1 < a.nml; k++)
        ok = ok && (asc ? h[k] < h[k + 1] : h[k] > h[k + 1]);
      if (!ok)
        *a.ml_mono = 0;
    }
  }
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < a.ncol; i += stride) {
    f32x4 va, vp;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      va[2 * t] = a.f2[t][MPHIP_PS] ? a.f2[t][MPHIP_PS][i] : 0.f;
      va[2 * t + 1] = a.f2[t][MPHIP_PBL] ? a.f2[t][MPHIP_PBL][i] : 0.f;
      f32x4 vb, vc;
      vb[0] = a.f2[t][MPHIP_CAPE] ? a.f2[t][MPHIP_CAPE][i] : 0.f;
      vb[1] = a.f2[t][MPHIP_CIN] ? a.f2[t][MPHIP_CIN][i] : 0.f;
      vb[2] = a.f2[t][MPHIP_PEL] ? a.f2[t][MPHIP_PEL][i] : 0.f;
      vb[3] = 0.f;
      vc[0] = a.f2[t][MPHIP_PCT] ? a.f2[t][MPHIP_PCT][i] : 0.f;
      vc[1] = a.f2[t][MPHIP_PCB] ? a.f2[t][MPHIP_PCB][i] : 0.f;
      vc[2] = a.f2[t][MPHIP_CL] ? a.f2[t][MPHIP_CL][i] : 0.f;
      vc[3] = 0.f;
      a.sfb[2 * i + t] = vb;
      vp[2 * t] = vb[0];
      vp[2 * t + 1] = vb[2];
      a.sfc[2 * i + t] = vc;
      if (a.sfd) {
        f32x4 vd;
        vd[0] = a.f2[t][MPHIP_ESS] ? a.f2[t][MPHIP_ESS][i] : 0.f;
        vd[1] = a.f2[t][MPHIP_NSS] ? a.f2[t][MPHIP_NSS][i] : 0.f;
        vd[2] = a.f2[t][MPHIP_SHF] ? a.f2[t][MPHIP_SHF][i] : 0.f;
        vd[3] = 0.f;
        a.sfd[2 * i + t] = vd;
      }
    }
    a.sfa[i] = va;
    a.cp2[i] = vp;
    if (a.mx2) {
#pragma unroll
      for (int pr = 0; pr < 7; pr++) {
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int f = 0; f < 2; f++) {
            const int fld = MPHIP_TS + 2 * pr + f;
            v[2 * t + f] = (fld < MPHIP_N2D && a.f2[t][fld]) ? a.f2[t][fld][i] : 0.f;
          }
        a.mx2[(size_t) pr * a.ncol + i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// module_sort: key, radix sort, gather (mptrac.c:5887-5995)
// ---------------------------------------------------------------------------

// Cell key of every particle.  tile = 0: the reference's module_sort key on the
// raw coordinates (mptrac.c:5913-5919).  tile = T > 0: key of the internal
// locality order (never observable): computed on the longitude/latitude the
// interpolation would use (intpol_check_lon_lat), and ordered as (T x T
// horizontal tile, level, column within the tile) so that the 64 particles of a
// wavefront sit in a few adjacent columns AND in a few adjacent levels --
// level-dependent branches (convection below the equilibrium level, vertical
// diffusion above the tropopause, surface reflection) then go one way per
// wavefront.
// With a TimestepArgs the kernel is module_timesteps as well (sort steps of
// mphip_run_timestep: dt is computed per slot before the sort, mptrac.c:7877-7881).
struct TimestepArgs {
  double direction, t_start, t_stop, t;
};

// ... and, when module_mixing of the current step is still to come, its box index on the way (the same four
// arrays are read): cell == NULL = off
struct BoxArgs {
  int *cell;
  BoxGrid grid;
  double t0, t1;
  const double *ens;
  int ngrid;
};

// Z-order number of tile (tx, ty): the low `m` bits of both interleaved (x above y), what is left of the longer
// coordinate on top.  Tiles that are neighbours in either direction are then close in the stored order, so
// the columns two tiles share are still in the L2 when the second one needs them.
__host__ __device__ __forceinline__ uint32_t tile_z_order(uint32_t tx, uint32_t ty, int m) {
  uint32_t z = 0;
  for (int b = 0; b < m; b++)
    z |= ((ty >> b) & 1u) << (2 * b) | ((tx >> b) & 1u) << (2 * b + 1);
  return z | ((tx >> m) | (ty >> m)) << (2 * m);   // (only one of the two has bits left)
}

// LEAN: lat/lon grid with the pressure look-up table (launch_step's condition): the indices come from
// raw_cell_fast -- first guess, verified, general search for the rare miss -- instead of three bisections
template <bool LEAN>
__global__ void sort_key_kernel(DevMet M, DevAtm a, int tile, int zbits, uint32_t *__restrict__ keys,
                                int *__restrict__ idx, const TimestepArgs ts, double *__restrict__ dt_out,
                                const BoxArgs box) {
  const int wrapped = tile > 0;
  const int nty = tile > 0 ? (M.ny + tile - 1) / tile : 0;
  extern __shared__ double s_axes[];
  const Axes A = load_axes(M, s_axes);
  __syncthreads();
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < a.np;
       i += (long long) gridDim.x * blockDim.x) {
    double lon = a.lon[i], lat = a.lat[i];
    if (box.cell)
      box.cell[i] = box_cell(box.grid, box.t0, box.t1, a.time[i], lon, lat, a.p[i], box.ens, i, box.ngrid);
    if (dt_out) {   // module_timesteps, mptrac.c:6016-6041
      const double time = a.time[i];
      double dt = 0.0;
      if (ts.direction * (time - ts.t_start) >= 0 && ts.direction * (time - ts.t_stop) <= 0
          && ts.direction * (time - ts.t) < 0)
        dt = ts.t - time;
      if (M.local && (lon <= A.lon[0] || lon >= A.lon[M.nx - 1] || lat <= M.latmin || lat >= M.latmax))
        dt = 0.0;
      dt_out[i] = dt;
    }
    if (wrapped) {
      double lon2, lat2;
      check_horizontal(M, A, lon, lat, lon2, lat2);
      lon = lon2;
      lat = lat2;
    }
    int ix, iy, iz;
    if (LEAN) {
      Stencil s;
      raw_cell_fast(M, A, lon, lat, a.p[i], s);
      ix = s.ix;
      iy = s.iy;
      iz = s.ip;
    } else {
      ix = locate_reg(A.lon, M.nx, lon);
      iy = locate_lat(M, A, lat);
      iz = locate_p(M, A, a.p[i]);
    }
    if (tile == 0)
      keys[i] = (uint32_t) ((ix * M.ny + iy) * M.np + iz);
    else {
      const uint32_t t = zbits >= 0 ? tile_z_order((uint32_t) (ix / tile), (uint32_t) (iy / tile), zbits)
                                    : (uint32_t) ((ix / tile) * nty + iy / tile);
      keys[i] = (uint32_t) ((t * M.np + iz) * (tile * tile) + (ix % tile) * tile + iy % tile);
    }
    if (idx)
      idx[i] = (int) i;
  }
}

constexpr int kSortThreads = 256;
constexpr int kSortRounds = 16;                            // keys per thread
constexpr int kSortTile = kSortThreads * kSortRounds;      // keys per workgroup
// Digit width BITS (8, 9 or 10) is chosen per sort so that the key needs the fewest passes: the 26-bit cell
// key of a 721 x 361 x 137 grid sorts in three 9-bit passes instead of four 8-bit ones.
constexpr int kRadixMaxBits = 10;

// per-tile digit histogram -> counts[digit * ntiles + tile]
template <int BITS>
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const uint32_t *__restrict__ keys, long long n,
                                                                 int shift, int ntiles, uint32_t *__restrict__ counts,
                                                                 const uint32_t *__restrict__ n_dev = nullptr,
                                                                 int in_stride = 1) {   // 2: interleaved (key, value) pairs
  constexpr int kRadix = 1 << BITS;
  if (n_dev) {   // the number of pairs is only known on the device: the launch covers an upper bound, the
    n = (long long) *n_dev;   // counter layout follows the actual number of tiles
    ntiles = (int) ((n + kSortTile - 1) / kSortTile);
    if ((int) blockIdx.x >= ntiles)
      return;
  }
  __shared__ uint32_t h[kRadix];
  for (int d = threadIdx.x; d < kRadix; d += kSortThreads)
    h[d] = 0;
  __syncthreads();
  const long long base = (long long) blockIdx.x * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; r++) {
    const long long i = base + r * kSortThreads + threadIdx.x;
    if (i < n)
      atomicAdd(&h[(keys[i * in_stride] >> shift) & (kRadix - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kSortThreads)
    counts[(size_t) d * ntiles + blockIdx.x] = h[d];
}

// Exclusive prefix sum over the m counters in two levels: every workgroup
// scans its chunk of 1024 x PER counters in place and publishes the chunk
// total; a single workgroup then scans the chunk totals (<= 1024 of them);
// the scatter kernel adds the two.  PER = 4 while that covers m (more, shorter
// chunks: 10 instead of 19 us for the 1.25e6 counters of 1e7 keys), 16 beyond
// (up to 2.6e8 keys per context).
constexpr int kScanThreads = 1024;
constexpr int kScanPerSmall = 4, kScanPerLarge = 16;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wsum, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= d)
      x += y;
  }
  if (lane == 63)
    wsum[wave] = x;
  __syncthreads();
  uint32_t off = 0, tot = 0;
  for (int w = 0; w < nwaves; w++) {
    const uint32_t ws = wsum[w];
    if (w < wave)
      off += ws;
    tot += ws;
  }
  *total = tot;
  return off + x - v;
}

template <int PER>
__global__ __launch_bounds__(kScanThreads) void sort_scan_local_kernel(uint32_t *__restrict__ counts, size_t m,
                                                                       uint32_t *__restrict__ chunk_sums,
                                                                       const uint32_t *__restrict__ n_dev = nullptr,
                                                                       int radix = 0) {
  constexpr int kScanChunk = kScanThreads * PER;
  __shared__ uint32_t wsum[kScanThreads / 64];
  if (n_dev) {   // (see sort_hist_kernel) radix counters for each of the actual tiles
    m = (size_t) radix * (((size_t) *n_dev + kSortTile - 1) / kSortTile);
    if ((size_t) blockIdx.x * kScanChunk >= m)
      return;
  }
  const size_t base = (size_t) blockIdx.x * kScanChunk + (size_t) threadIdx.x * PER;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    v[k] = base + k < m ? counts[base + k] : 0;
    sum += v[k];
  }
  uint32_t total;
  uint32_t off = block_exclusive_scan(sum, wsum, &total);
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if (base + k < m)
      counts[base + k] = off;
    off += v[k];
  }
  if (threadIdx.x == 0)
    chunk_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanThreads) void sort_scan_chunks_kernel(uint32_t *__restrict__ chunk_sums, int nchunks) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  const uint32_t v = (int) threadIdx.x < nchunks ? chunk_sums[threadIdx.x] : 0;
  uint32_t total;
  const uint32_t off = block_exclusive_scan(v, wsum, &total);
  if ((int) threadIdx.x < nchunks)
    chunk_sums[threadIdx.x] = off;
}

// Stable scatter of one digit pass.  A workgroup ranks its whole tile first,
// re-orders it by digit in LDS and only then writes: the keys of one digit
// leave as one contiguous run (kSortTile / kRadix = 16 pairs on average, whole
// 64-byte segments) instead of 4-byte stores to 256 different places per round.
//   ranking: wave w owns keys [1024 w, 1024 (w + 1)) of the tile and walks them
//   in rounds of 64 -- (wave, round, lane) is the input order, so ranks are
//   stable; the lanes of a round that hold the same digit are found with eight
//   ballots, and a per-wave counter row in LDS carries the running count from
//   round to round (no workgroup barrier inside the loop).
template <int BITS>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                                    const int *__restrict__ vals_in,
                                                                    uint32_t *__restrict__ keys_out,
                                                                    int *__restrict__ vals_out, long long n, int shift,
                                                                    int ntiles, const uint32_t *__restrict__ offsets,
                                                                    const uint32_t *__restrict__ chunk_offsets,
                                                                    int chunk_shift,   // log2 of the scan's chunk length
                                                                    const uint32_t *__restrict__ n_dev = nullptr,
                                                                    int in_stride = 1) {   // 2: the input is one array of (key, value) pairs
  constexpr int kRadix = 1 << BITS;
  if (n_dev) {
    n = (long long) *n_dev;
    ntiles = (int) ((n + kSortTile - 1) / kSortTile);
    if ((int) blockIdx.x >= ntiles)
      return;
  }
  constexpr int kDigitsPerThread = kRadix / kSortThreads;   // 1, 2 or 4 consecutive digits per thread
  constexpr int kWaves = kSortThreads / 64;
  constexpr int kPerWave = kSortTile / kWaves;
  __shared__ uint32_t s_key[kSortTile];
  __shared__ int s_val[kSortTile];
  __shared__ uint32_t s_cnt[kWaves][kRadix];   // per-wave digit counts, then exclusive prefix over the waves
  __shared__ uint32_t s_dbase[kRadix];         // first tile-local position of a digit
  __shared__ uint32_t s_gdelta[kRadix];        // global position - tile-local position
  __shared__ uint32_t s_wsum[kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int d = threadIdx.x; d < kRadix; d += kSortThreads)
      s_cnt[w][d] = 0;
  __syncthreads();

  const long long base = (long long) blockIdx.x * kSortTile + (long long) wave * kPerWave + lane;
  uint32_t key[kSortRounds];
  int val[kSortRounds];
  uint32_t off[kSortRounds];
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const long long i = base + r * 64;
    const bool valid = i < n;
    key[r] = valid ? keys_in[i * in_stride] : 0xffffffffu;
    val[r] = valid ? (vals_in ? vals_in[i * in_stride] : (int) i) : 0;   // (no value array: the position itself, first pass of an index sort)
  }
  volatile uint32_t *cnt = s_cnt[wave];
#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    const bool valid = base + r * 64 < n;
    const uint32_t d = (key[r] >> shift) & (kRadix - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
      const unsigned long long bal = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? bal : ~bal;
    }
    const unsigned long long below = peers & ((1ull << lane) - 1ull);
    const uint32_t prev = cnt[d];              // every lane of the wave reads before the leaders write
    off[r] = prev + __popcll(below);
    if (valid && below == 0)
      cnt[d] = prev + __popcll(peers);
  }
  __syncthreads();

  // a thread takes kDigitsPerThread consecutive digits: prefix over the waves, then over the digits
  {
    uint32_t tot[kDigitsPerThread], sum = 0;
#pragma unroll
    for (int k = 0; k < kDigitsPerThread; k++) {
      const int d = threadIdx.x * kDigitsPerThread + k;
      uint32_t t = 0;
#pragma unroll
      for (int w = 0; w < kWaves; w++) {
        const uint32_t c = s_cnt[w][d];
        s_cnt[w][d] = t;
        t += c;
      }
      tot[k] = t;
      sum += t;
    }
    uint32_t all;
    uint32_t dbase = block_exclusive_scan(sum, s_wsum, &all);
#pragma unroll
    for (int k = 0; k < kDigitsPerThread; k++) {
      const int d = threadIdx.x * kDigitsPerThread + k;
      const size_t slot = (size_t) d * ntiles + blockIdx.x;
      s_dbase[d] = dbase;
      s_gdelta[d] = offsets[slot] + chunk_offsets[slot >> chunk_shift] - dbase;
      dbase += tot[k];
    }
  }
  __syncthreads();

#pragma unroll
  for (int r = 0; r < kSortRounds; r++) {
    if (base + r * 64 < n) {
      const uint32_t d = (key[r] >> shift) & (kRadix - 1);
      const uint32_t pos = s_dbase[d] + s_cnt[wave][d] + off[r];
      s_key[pos] = key[r];
      s_val[pos] = val[r];
    }
  }
  __syncthreads();

  const long long left = n - (long long) blockIdx.x * kSortTile;
  const int count = left < kSortTile ? (int) left : kSortTile;
#pragma unroll 4
  for (int j = 0; j < kSortRounds; j++) {
    const int L = j * kSortThreads + threadIdx.x;
    if (L < count) {
      const uint32_t k = s_key[L];
      const uint32_t pos = (uint32_t) L + s_gdelta[(k >> shift) & (kRadix - 1)];
      keys_out[pos] = k;
      vals_out[pos] = s_val[L];
    }
  }
}

// ---------------------------------------------------------------------------
// module_sort as a REPAIR of the previous order (mptrac.c:5887-5957 sorts from scratch every SORT_DT).
// When module_sort runs in every step the particles are stored in the order of the previous sort, key_prev[j] (the
// sorted keys of that sort) is non-decreasing along the slots j, and most particles are still in the cell they were
// in: a particle whose new key equals key_prev at its slot is a "stayer".  The stayers, taken in slot order, are a
// sorted sequence already (same keys as before, ties in slot order = the stable order); the others ("movers", ~15 % per
// step on workload C5) are compacted in slot order, sorted by the stable radix sort -- a few hundred thousand pairs
// instead of 10^7 -- and the two sorted sequences are merged by (key, slot).  The result is exactly the stable sort of
// all (key, slot) pairs: same permutation, same keys as the full sort (tests compare the bits with it and the oracle).
//   repair_count_kernel   movers of every tile of 1024 slots
//   repair_scan_kernel    exclusive prefix of the tile counts (one workgroup), the number of movers
//   repair_split_kernel   movers -> (mk, mi), stayers -> (sk, si), both in slot order
//   (radix passes over the movers, number of pairs on the device)
//   repair_merge_kernel   merge path: every workgroup produces 2048 consecutive pairs of the result from the two
//                         sub-ranges a diagonal search assigns to it (coalesced reads and writes, the merge in LDS)
// ---------------------------------------------------------------------------
constexpr int kRepairTile = 1024;     // slots per workgroup of the count / split kernels (256 threads x 4 consecutive slots)
constexpr int kMergeTile = 2048;      // pairs per workgroup of the merge kernel (256 threads x 8)

__global__ __launch_bounds__(256) void repair_count_kernel(const uint32_t *__restrict__ key_new, const uint32_t *__restrict__ key_prev,
                                                           long long n, uint32_t *__restrict__ tile_count) {
  __shared__ uint32_t wsum[4];
  const long long base = (long long) blockIdx.x * kRepairTile + 4 * threadIdx.x;
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (base + k < n)
      c += key_new[base + k] != key_prev[base + k];
  uint32_t total;
  (void) block_exclusive_scan(c, wsum, &total);
  if (threadIdx.x == 0)
    tile_count[blockIdx.x] = total;
}

// counts[0 .. ntiles) -> exclusive prefix in place; nm[0] = their sum (the movers), nm[1] = n - sum (the stayers)
__global__ __launch_bounds__(1024) void repair_scan_kernel(uint32_t *__restrict__ counts, int ntiles, long long n,
                                                           uint32_t *__restrict__ nm) {
  __shared__ uint32_t wsum[16];
  const int per = (ntiles + 1023) / 1024;
  const int first = threadIdx.x * per;
  uint32_t sum = 0;
  for (int k = first; k < first + per && k < ntiles; k++)
    sum += counts[k];
  uint32_t total;
  uint32_t off = block_exclusive_scan(sum, wsum, &total);
  for (int k = first; k < first + per && k < ntiles; k++) {
    const uint32_t c = counts[k];
    counts[k] = off;
    off += c;
  }
  if (threadIdx.x == 0) {
    nm[0] = total;
    nm[1] = (uint32_t) (n - (long long) total);
  }
}

__global__ __launch_bounds__(256) void repair_split_kernel(const uint32_t *__restrict__ key_new, const uint32_t *__restrict__ key_prev,
                                                           long long n, const uint32_t *__restrict__ tile_offset,
                                                           uint32_t *__restrict__ mk, int *__restrict__ mi,
                                                           uint32_t *__restrict__ sk, int *__restrict__ si) {
  __shared__ uint32_t wsum[4];
  const long long tile0 = (long long) blockIdx.x * kRepairTile;
  const long long base = tile0 + 4 * threadIdx.x;
  uint32_t key[4];
  bool mover[4];
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    key[k] = base + k < n ? key_new[base + k] : 0u;
    mover[k] = base + k < n && key[k] != key_prev[base + k];
    c += mover[k];
  }
  uint32_t total;
  const uint32_t before = block_exclusive_scan(c, wsum, &total);      // movers of this tile in front of the thread's slots
  uint32_t m = tile_offset[blockIdx.x] + before;                       // ... of the whole sequence
  uint32_t st = (uint32_t) (base - (long long) m);                     // stayers in front = slots in front - movers in front
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k >= n)
      break;
    if (mover[k]) {
      mk[m] = key[k];
      mi[m] = (int) (base + k);
      m++;
    } else {
      sk[st] = key[k];
      si[st] = (int) (base + k);
      st++;
    }
  }
}

__device__ __forceinline__ uint64_t key_slot(const uint32_t *__restrict__ k, const int *__restrict__ i, uint32_t at) {
  return ((uint64_t) k[at] << 32) | (uint32_t) i[at];
}

// number of elements taken from A (the stayers) among the first d of the merged sequence
__device__ __forceinline__ uint32_t merge_diagonal(const uint32_t *__restrict__ ak, const int *__restrict__ ai, uint32_t na,
                                                   const uint32_t *__restrict__ bk, const int *__restrict__ bi, uint32_t nb,
                                                   uint32_t d) {
  uint32_t lo = d > nb ? d - nb : 0u, hi = d < na ? d : na;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (key_slot(ak, ai, mid) < key_slot(bk, bi, d - 1 - mid))
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void repair_merge_kernel(const uint32_t *__restrict__ sk, const int *__restrict__ si,
                                                           const uint32_t *__restrict__ mk, const int *__restrict__ mi,
                                                           const uint32_t *__restrict__ nm, long long n,
                                                           uint32_t *__restrict__ keys_out, int *__restrict__ vals_out) {
  __shared__ uint64_t s_pair[kMergeTile];
  __shared__ uint32_t s_split[2];
  const uint32_t nb = nm[0], na = nm[1];
  const uint32_t d0 = (uint32_t) blockIdx.x * kMergeTile;
  if ((long long) d0 >= n)
    return;
  const uint32_t d1 = (uint32_t) ((long long) d0 + kMergeTile < n ? d0 + kMergeTile : n);
  if (threadIdx.x < 2)
    s_split[threadIdx.x] = merge_diagonal(sk, si, na, mk, mi, nb, threadIdx.x == 0 ? d0 : d1);
  __syncthreads();
  const uint32_t a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
  const uint32_t la = a1 - a0, lb = b1 - b0;          // la + lb = d1 - d0 <= kMergeTile
  // the two sub-ranges side by side in LDS: [0, la) from the stayers, [la, la + lb) from the movers
  for (uint32_t j = threadIdx.x; j < la; j += 256)
    s_pair[j] = key_slot(sk, si, a0 + j);
  for (uint32_t j = threadIdx.x; j < lb; j += 256)
    s_pair[la + j] = key_slot(mk, mi, b0 + j);
  __syncthreads();
  // every thread merges eight consecutive results: its own diagonal inside the tile, then a serial merge
  constexpr int kPer = kMergeTile / 256;
  const uint32_t t0 = (uint32_t) threadIdx.x * kPer;
  if (t0 >= la + lb)
    return;
  uint32_t lo = t0 > lb ? t0 - lb : 0u, hi = t0 < la ? t0 : la;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (s_pair[mid] < s_pair[la + (t0 - 1 - mid)])
      lo = mid + 1;
    else
      hi = mid;
  }
  uint32_t ia = lo, ib = t0 - lo;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    if (t0 + k >= la + lb)
      break;
    const bool take_a = ib >= lb || (ia < la && s_pair[ia] < s_pair[la + ib]);
    const uint64_t v = take_a ? s_pair[ia] : s_pair[la + ib];
    ia += take_a;
    ib += !take_a;
    keys_out[d0 + t0 + k] = (uint32_t) (v >> 32);
    vals_out[d0 + t0 + k] = (int) (uint32_t) v;
  }
}

// Fused re-ordering of every per-particle array in one pass.
//   gather : out[i] = in[perm[i]]     (module_sort_help, mptrac.c:5944-5949,
//                                      and the internal locality order)
//   scatter: out[ext[i]] = in[i]      (back to the external slot order)
struct PermArgs {
  const double *in8[4 + MPHIP_NQ_MAX + 2];
  double *out8[4 + MPHIP_NQ_MAX + 2];
  const float *in4[4];   // uvwp + the model-level index hint (an int array moved as 4-byte words)
  float *out4[4];
  const int *ext_in;     // gather only: slot ids travel with the particles
  int *ext_out;          // (ext_in == NULL means identity)
  int n8, n4;
};

// Both kernels walk contiguous runs per workgroup, and workgroup b takes run (b % 8) * (n / 8) + b / 8:
// the dispatcher places workgroup b on XCD b % 8, so each XCD (own L2) works on one contiguous eighth of
// the index range and the 8-byte elements of a cache line are consumed through one L2 instead of eight.
struct PermGeom {
  int nblocks;            // multiple of 8
  long long per_block;    // multiple of 256
};

__device__ __forceinline__ void perm_range(const PermGeom &pg, long long n, long long &first, long long &last) {
  const int lb = (int) (blockIdx.x % 8) * (pg.nblocks / 8) + (int) (blockIdx.x / 8);
  first = (long long) lb * pg.per_block;
  last = first + pg.per_block < n ? first + pg.per_block : n;
}

__global__ void perm_gather_kernel(PermArgs g, const int *__restrict__ perm, long long n, PermGeom pg) {
  long long first, last;
  perm_range(pg, n, first, last);
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    const int src = perm[i];
    for (int a = 0; a < g.n8; a++)
      g.out8[a][i] = g.in8[a][src];
    for (int a = 0; a < g.n4; a++)
      g.out4[a][i] = g.in4[a][src];
    if (g.ext_out)
      g.ext_out[i] = g.ext_in ? g.ext_in[src] : src;
  }
}

__global__ void perm_scatter_kernel(PermArgs g, const int *__restrict__ ext, long long n, PermGeom pg) {
  long long first, last;
  perm_range(pg, n, first, last);
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    const int dst = ext[i];
    for (int a = 0; a < g.n8; a++)
      g.out8[a][dst] = g.in8[a][i];
    for (int a = 0; a < g.n4; a++)
      g.out4[a][dst] = g.in4[a][i];
  }
}

// Random permutations (the first sort out of the caller's order, the way back at a download) through records.
// A structure-of-arrays element moved to a random place costs a memory transaction of its own whatever its size --
// ten of them per particle in the kernels above, 1.3-2.1 ms for 1e7 particles of C3.  Two passes instead:
//   pack    the arrays -> one record per particle (8-byte fields, then 4-byte ones, padded to 16 bytes), written in
//           16-byte pieces either where the particle is (gather) or where it goes (scatter: dst = ext[i]);
//   unpack  records -> arrays, every store coalesced; the gather reads record perm[i] here.
// One random transaction of a record (one or two lines) per particle instead of ten.
struct RecordGeom {
  int n8, n4;        // fields (the slot ids of a gather travel as one more 4-byte field)
  int chunks;        // 16-byte pieces per record
  int inv;           // p / chunks = (p * inv) >> 16 for p < 64 * chunks
};

__device__ __forceinline__ void record_fields(const PermArgs &g, const RecordGeom &rg, int c, long long src, bool ids,
                                              f32x4u &v) {
  // piece c: doubles 2c, 2c + 1 while they last, then four 4-byte fields per piece
  const int d8 = (rg.n8 + 1) / 2;
  if (c < d8) {
    const double a = g.in8[2 * c][src];
    const double b = 2 * c + 1 < rg.n8 ? g.in8[2 * c + 1][src] : 0.0;
    v[0] = __uint_as_float((uint32_t) __double2loint(a));
    v[1] = __uint_as_float((uint32_t) __double2hiint(a));
    v[2] = __uint_as_float((uint32_t) __double2loint(b));
    v[3] = __uint_as_float((uint32_t) __double2hiint(b));
  } else {
    const int f0 = 4 * (c - d8);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int f = f0 + k;
      float x = 0.f;
      if (f < g.n4)
        x = g.in4[f][src];
      else if (f == g.n4 && ids)
        x = __int_as_float(g.ext_in ? g.ext_in[src] : (int) src);
      v[k] = x;
    }
  }
}

// Both kernels move the records between memory and LDS as a wave: piece p of the wave's 64 records is handled by
// lane p % 64 (record p / chunks, piece p % chunks), so the `chunks` lanes of one record touch its bytes side by side --
// one or two memory transactions per record instead of one per 16-byte piece and lane -- and the sequential side
// (pack of a gather, unpack of a scatter) is fully coalesced.  Every lane then reads / has written its own record in
// LDS.  (p / chunks by a multiplication: RecordGeom::inv, exact for p < 64 * chunks, checked by the host.)
constexpr int kRecordWaves = 4;

// records[dst] <- particle i; dst = i (gather: the unpack pass follows the permutation) or ext[i] (scatter)
__global__ __launch_bounds__(64 * kRecordWaves) void perm_pack_kernel(PermArgs g, RecordGeom rg, const int *__restrict__ dst_of,
                                                                      f32x4u *__restrict__ rec, long long n, PermGeom pg) {
  extern __shared__ f32x4u s_rec[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4u *tile = s_rec + (size_t) wave * 64 * (size_t) rg.chunks;
  long long first, last;
  perm_range(pg, n, first, last);
  const bool ids = g.ext_out != nullptr;
  for (long long base = first + 64 * wave; base < last; base += 64 * kRecordWaves) {
    const long long i = base + lane;
    const bool live = i < last;
    const int dst = live ? (dst_of ? dst_of[i] : (int) i) : 0;
    if (live)
      for (int c = 0; c < rg.chunks; c++) {
        f32x4u v;
        record_fields(g, rg, c, i, ids, v);
        tile[lane * rg.chunks + c] = v;
      }
    __builtin_amdgcn_wave_barrier();
    const int nrec = last - base < 64 ? (int) (last - base) : 64;
    for (int p0 = 0; p0 < nrec * rg.chunks; p0 += 64) {   // (every lane takes part in the shuffle)
      const int p = p0 + lane;
      const int r = (int) (((unsigned) p * (unsigned) rg.inv) >> 16) & 63, c = p - r * rg.chunks;
      const int d = __shfl(dst, r);
      if (p < nrec * rg.chunks)
        rec[(size_t) d * (size_t) rg.chunks + (size_t) c] = tile[p];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// arrays[i] <- records[src]; src = src_of[i] (gather) or i (scatter)
__global__ __launch_bounds__(64 * kRecordWaves) void perm_unpack_kernel(PermArgs g, RecordGeom rg, const int *__restrict__ src_of,
                                                                        const f32x4u *__restrict__ rec, long long n, PermGeom pg) {
  extern __shared__ f32x4u s_rec[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4u *tile = s_rec + (size_t) wave * 64 * (size_t) rg.chunks;
  long long first, last;
  perm_range(pg, n, first, last);
  const int d8 = (rg.n8 + 1) / 2;
  for (long long base = first + 64 * wave; base < last; base += 64 * kRecordWaves) {
    const long long i = base + lane;
    const bool live = i < last;
    const int src = live ? (src_of ? src_of[i] : (int) i) : 0;
    const int nrec = last - base < 64 ? (int) (last - base) : 64;
    for (int p0 = 0; p0 < nrec * rg.chunks; p0 += 64) {   // (every lane takes part in the shuffle)
      const int p = p0 + lane;
      const int r = (int) (((unsigned) p * (unsigned) rg.inv) >> 16) & 63, c = p - r * rg.chunks;
      const int sr = __shfl(src, r);
      if (p < nrec * rg.chunks)
        tile[p] = rec[(size_t) sr * (size_t) rg.chunks + (size_t) c];
    }
    __builtin_amdgcn_wave_barrier();
    if (live)
      for (int c = 0; c < rg.chunks; c++) {
        const f32x4u v = tile[lane * rg.chunks + c];
        if (c < d8) {
          g.out8[2 * c][i] = __hiloint2double((int) __float_as_uint(v[1]), (int) __float_as_uint(v[0]));
          if (2 * c + 1 < rg.n8)
            g.out8[2 * c + 1][i] = __hiloint2double((int) __float_as_uint(v[3]), (int) __float_as_uint(v[2]));
        } else {
          const int f0 = 4 * (c - d8);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int f = f0 + k;
            if (f < g.n4)
              g.out4[f][i] = v[k];
            else if (f == g.n4 && g.ext_out)
              g.ext_out[i] = __float_as_int(v[k]);
          }
        }
      }
    __builtin_amdgcn_wave_barrier();
  }
}

// out[i] = in[ext[i]]: one array handed over in the caller's order into the stored order
__global__ void gather_by_ext_kernel(const double *__restrict__ in, const int *__restrict__ ext, double *__restrict__ out,
                                     long long n) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
    out[i] = in[ext[i]];
}

__global__ void keys_to_double_kernel(const uint32_t *__restrict__ k, double *__restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
    out[i] = (double) k[i];
}

// ---------------------------------------------------------------------------
// module_mixing (mptrac.c:5169-5347) and write_grid sums (mptrac.c:13815-13872)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void box_index_kernel(DevAtm a, BoxGrid G, double t0, double t1, int *__restrict__ cell,
                                                        const double *__restrict__ ens, int ngrid) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < a.np;
       i += (long long) gridDim.x * blockDim.x)
    cell[i] = box_cell(G, t0, t1, a.time[i], a.lon[i], a.lat[i], a.p[i], ens, i, ngrid);
}

__global__ void mix_relax_kernel(mphip_ctl_t ctl, const DevClim *clim, DevAtm a, const int *__restrict__ cell, MixSet mq,
                                 size_t ntot, const double *__restrict__ sums, const int *__restrict__ cnt) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < a.np;
       i += (long long) gridDim.x * blockDim.x)
    mix_relax_one(ctl, *clim, mq, i, cell[i], a.time[i], a.lat[i], a.p[i], ntot, sums, cnt);
}

// Block-private accumulation table in LDS.  The particles are stored in the
// locality order and every block walks one contiguous run of it, so a block
// sees few distinct output cells: sums are collected per cell in an LDS hash
// table (key = cell, nv doubles per entry, LDS fp64 atomics) and only the
// table is flushed with global atomics -- two orders of magnitude fewer of
// them, and no long same-address queues in L2.  A particle whose cell finds no
// free slot within kProbes tries adds to global memory directly (unsorted
// input, very fine output grids).
#ifndef MPHIP_TABLE_GROUP_LOG2
#define MPHIP_TABLE_GROUP_LOG2 4
#endif
constexpr int kProbes = 8;

struct LdsTable {
  int *keys;        // [T], -1 = free
  double *vals;     // [nv][T]
  int T, shift, nv;

  __device__ __forceinline__ void init(void *smem, int T_, int nv_) {
    T = T_;
    nv = nv_;
    shift = 32 - (31 - __builtin_clz((unsigned) T_));
    vals = (double *) smem;
    keys = (int *) (vals + (size_t) nv_ * T_);
    for (int i = threadIdx.x; i < T_; i += blockDim.x)
      keys[i] = -1;
    for (int i = threadIdx.x; i < nv_ * T_; i += blockDim.x)
      vals[i] = 0.0;
    __syncthreads();
  }
  // slot of `key`, claiming a free one if needed; -1 if the probe sequence is full.
  // 2^G consecutive cells (one 128-byte line of the output array) hash to 2^G consecutive
  // slots, so that the flush sends the atomics of a line from neighbouring lanes: with a
  // hash per cell the flush of module_mixing took 300 of the kernel's 360 us, now 45.
  __device__ __forceinline__ int slot_for(int key) const {
    constexpr int G = MPHIP_TABLE_GROUP_LOG2;
    unsigned h = ((((unsigned) key >> G) * 2654435761u) >> (shift + G)) << G | ((unsigned) key & ((1u << G) - 1u));
    for (int k = 0; k < kProbes; k++) {
      const int old = atomicCAS(&keys[h], -1, key);
      if (old == -1 || old == key)
        return (int) h;
      h = (h + (1u << G)) & (unsigned) (T - 1);
    }
    return -1;
  }
  __device__ __forceinline__ void add(int slot, int v, double x) const {
    __hip_atomic_fetch_add(&vals[(size_t) v * T + slot], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // buf[v * stride + key] += table
  __device__ __forceinline__ void flush(double *__restrict__ buf, size_t stride) const {
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
      const int key = keys[i];
      if (key >= 0)
        for (int v = 0; v < nv; v++)
          unsafeAtomicAdd(&buf[(size_t) v * stride + (size_t) key], vals[(size_t) v * T + i]);
    }
  }
};

// module_mixing's cell sums for every mixed quantity in one pass (mptrac.c:5223-5230, 5289-5303):
// sums[k * ntot + idx] += q_k, cnt[idx] += 1 with idx = ens * ngrid + cell.  The particle count is the
// same for all quantities and is kept once, as 32-bit integers (a third fewer bytes through the all-reduce
// than [sum | count] pairs of doubles per quantity).
__global__ __launch_bounds__(256) void mix_accumulate_kernel(DevAtm a, const int *__restrict__ cell, MixSet mq,
                                                             size_t ntot, double *__restrict__ sums,
                                                             int *__restrict__ cnt, int T, long long per_block) {
  extern __shared__ double s_tab[];
  LdsTable tab;
  tab.init(s_tab, T, mq.n + 1);   // value mq.n of an entry: the count
  const long long first = blockIdx.x * per_block;
  const long long last = first + per_block < a.np ? first + per_block : a.np;
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    const int idx = cell[i];
    if (idx >= 0) {
      const int slot = tab.slot_for(idx);
      for (int k = 0; k < mq.n; k++) {
        const double v = mq.q[k][i];
        if (slot >= 0)
          tab.add(slot, k, v);
        else
          unsafeAtomicAdd(&sums[(size_t) k * ntot + (size_t) idx], v);
      }
      if (slot >= 0)
        tab.add(slot, mq.n, 1.0);
      else
        atomicAdd(&cnt[idx], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tab.T; i += blockDim.x) {
    const int key = tab.keys[i];
    if (key >= 0) {
      for (int k = 0; k < mq.n; k++)
        unsafeAtomicAdd(&sums[(size_t) k * ntot + (size_t) key], tab.vals[(size_t) k * tab.T + i]);
      atomicAdd(&cnt[key], (int) tab.vals[(size_t) mq.n * tab.T + i]);
    }
  }
}

// ---------------------------------------------------------------------------
// Cell sums in the reference's order.  The reference accumulates serially over the particle index
// (module_mixing mptrac.c:5289-5303, write_grid mptrac.c:13862-13872): cell sum = ((q_a + q_b) + q_c) ... in
// ascending ip.  Floating-point atomics add in whatever order the hardware serves them, so their sums differ
// from that in the last bits and from run to run.  Here every cell is owned by one wave, which adds the cell's
// summands one after the other in ascending external index -- bit-identical to the serial loop:
//   1. the cells of the particles as a sequence in external order (the stored order after module_sort; else
//      scattered through the permutation, cell_pairs_kernel);
//   2. the cells are taken in groups of G consecutive indices (whole vertical columns of the grid, <= 128 cells), and
//      the sequence is cut into runs of equal group (run_heads_count / run_offsets / run_compact: an
//      order-preserving compaction).  The stored orders follow the meteo grid, so the particles of a column
//      are neighbours: ~100 times fewer runs than particles after module_sort;
//   3. the runs are sorted by group with the stable radix sort of module_sort: the runs of one group end up side
//      by side, in ascending external index;
//   4. one wave per group streams the group's particles in that order, 64 at a time, into a table of the G cells
//      in LDS.  Lanes that hit the same cell take turns in lane order (ds_min on a claim word decides whose
//      turn it is; cells with many of the 64 are added up from the lanes one after the other), so every cell
//      adds in exactly the serial order; the table is then written out (cell_sum_groups_kernel).  No global
//      atomics at all.
// ---------------------------------------------------------------------------

constexpr int kGroupMax = 128;
constexpr int kRunRounds = 16;
constexpr int kRunTile = 256 * kRunRounds;   // sequence positions per workgroup of the compaction

// values a cell sums up: count() values per particle (the kernel handles B of them per pass)
struct MixVals {   // module_mixing: the mixed quantities
  MixSet mq;
  __device__ __forceinline__ int count() const { return mq.n; }
  __device__ __forceinline__ double get(int k, long long i) const { return mq.q[k][i]; }
};

// vertical weighting function of the gridded output (GRID_KERNEL; kernel_weight, mptrac.c:3298-3320): nk nodes
// (height [km], weight), linear in between, constant beyond; nk < 2: weight one
struct GridKernel {
  const double *kz, *kw;   // device arrays
  const double *p;         // pressure of every stored particle
  int nk;
  __device__ __forceinline__ double weight(long long i) const {
    if (nk < 2)
      return 1.0;
    const double z = zfromp(p[i]);
    if (z < kz[0])
      return kw[0];
    if (z > kz[nk - 1])
      return kw[nk - 1];
    const int idx = locate_irr(kz, nk, z, 1);
    return lin(kz[idx], kw[idx], kz[idx + 1], kw[idx + 1], z);
  }
};

struct GridVals {  // write_grid: kernel * q and its square for every quantity (mptrac.c:13862-13872)
  const double *q[MPHIP_NQ_MAX];
  // the same values as one record of nq doubles per stored particle (grid_records_kernel), or NULL: the ordered
  // sums gather a particle's values one cell list entry at a time, and a gather pulls a whole 128-byte line --
  // one or two lines per particle from the records instead of one per quantity from the arrays
  const double *rec;
  int nq;
  GridKernel kern;
  __device__ __forceinline__ int count() const { return 2 * nq; }
  __device__ __forceinline__ double get(int k, long long i) const {
    const int kq = k < nq ? k : k - nq;
    const double v = kern.weight(i) * (rec ? rec[(size_t) i * (size_t) nq + (size_t) kq] : q[kq][i]);
    return k < nq ? v : v * v;
  }
};

// rec[i * nq + iq] = q[iq][i] (coalesced on both sides: a wave writes 64 consecutive records)
__global__ void grid_records_kernel(GridVals g, long long n, double *__restrict__ rec) {
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
    for (int iq = 0; iq < g.nq; iq++)
      rec[(size_t) i * (size_t) g.nq + (size_t) iq] = g.q[iq][i];
}

// sequence in external order: seq_cell[ip] = cell, seq_slot[ip] = where the particle is stored
// (`gate`: the kernels of the general pass do nothing unless *gate is set -- see ordered_cell_sums)
__global__ void cell_pairs_kernel(const int *__restrict__ cell, const int *__restrict__ ext, long long n,
                                  int *__restrict__ seq_cell, int *__restrict__ seq_slot, const int *__restrict__ gate) {
  if (gate && *gate == 0)
    return;
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
    const int at = ext[i];
    seq_cell[at] = cell[i];
    seq_slot[at] = (int) i;
  }
}

__device__ __forceinline__ int group_of(int cell, int G) {
  return cell >= 0 ? cell / G : -1;
}

// A run starts where the group changes.  A workgroup looks at kRunTile consecutive positions: wave w at the w-th
// quarter, 64 neighbouring positions per round (coalesced), one ballot per round = the heads of the round.
struct RunHeads {
  unsigned long long heads[kRunRounds];   // per round: lanes that start a run
  int group[kRunRounds];                  // this lane's group per round
  uint32_t count;                         // runs that start in this wave's part
};

__device__ __forceinline__ RunHeads run_heads_of_wave(const int *__restrict__ seq, long long n, int G) {
  RunHeads H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long first = (long long) blockIdx.x * kRunTile + (long long) wave * (kRunTile / 4);
  int carry = first > 0 && first <= n ? group_of(seq[first - 1], G) : 0;   // group of the position before the round
  H.count = 0;
#pragma unroll
  for (int r = 0; r < kRunRounds; r++) {
    const long long i = first + r * 64 + lane;
    const int g = i < n ? group_of(seq[i], G) : 0;
    int prev = __shfl_up(g, 1);
    if (lane == 0)
      prev = carry;
    H.heads[r] = __ballot(i < n && (i == 0 || g != prev));
    H.group[r] = g;
    H.count += (uint32_t) __builtin_popcountll(H.heads[r]);
    carry = __builtin_amdgcn_readlane(g, 63);
  }
  return H;
}

__global__ __launch_bounds__(256) void run_heads_count_kernel(const int *__restrict__ seq, long long n, int G,
                                                              uint32_t *__restrict__ tile_runs,
                                                              const int *__restrict__ gate) {
  __shared__ uint32_t wsum[4];
  if (gate && *gate == 0)
    return;
  const RunHeads H = run_heads_of_wave(seq, n, G);
  if ((threadIdx.x & 63) == 0)
    wsum[threadIdx.x >> 6] = H.count;
  __syncthreads();
  if (threadIdx.x == 0)
    tile_runs[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the per-tile run counts in place (one workgroup, any length); tile_runs[ntiles] = number of runs
// (gate closed: no runs)
__global__ __launch_bounds__(kScanThreads) void run_offsets_kernel(uint32_t *__restrict__ tile_runs, int ntiles,
                                                                   const int *__restrict__ gate) {
  __shared__ uint32_t wsum[kScanThreads / 64];
  if (gate && *gate == 0) {
    if (threadIdx.x == 0)
      tile_runs[ntiles] = 0;
    return;
  }
  uint32_t carry = 0;
  for (int base = 0; base < ntiles; base += kScanThreads) {
    const int i = base + (int) threadIdx.x;
    const uint32_t v = i < ntiles ? tile_runs[i] : 0;
    uint32_t total;
    const uint32_t off = block_exclusive_scan(v, wsum, &total);
    if (i < ntiles)
      tile_runs[i] = carry + off;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0)
    tile_runs[ntiles] = carry;
}

// run r: key = its group (`outside` for particles that are not in the grid), id = r, start = first position;
// run_start[number of runs] = n closes the last run
__global__ __launch_bounds__(256) void run_compact_kernel(const int *__restrict__ seq, long long n, int G,
                                                          const uint32_t *__restrict__ tile_offset, int ntiles,
                                                          uint32_t outside, uint32_t *__restrict__ run_key,
                                                          int *__restrict__ run_id, uint32_t *__restrict__ run_start,
                                                          const int *__restrict__ gate) {
  __shared__ uint32_t wsum[4];
  if (gate && *gate == 0)
    return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const RunHeads H = run_heads_of_wave(seq, n, G);
  if (lane == 0)
    wsum[wave] = H.count;
  __syncthreads();
  uint32_t r0 = tile_offset[blockIdx.x];
  for (int w = 0; w < wave; w++)
    r0 += wsum[w];
  const long long first = (long long) blockIdx.x * kRunTile + (long long) wave * (kRunTile / 4);
#pragma unroll
  for (int k = 0; k < kRunRounds; k++) {
    const unsigned long long heads = H.heads[k];
    if ((heads >> lane) & 1) {
      const uint32_t r = r0 + (uint32_t) __builtin_popcountll(heads & ((1ull << lane) - 1));
      run_key[r] = H.group[k] >= 0 ? (uint32_t) H.group[k] : outside;
      if (run_id)
        run_id[r] = (int) r;
      run_start[r] = (uint32_t) (first + k * 64 + lane);
    }
    r0 += (uint32_t) __builtin_popcountll(heads);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    run_start[tile_offset[ntiles]] = (uint32_t) n;
}

// x of lane l (l uniform in the wave)
__device__ __forceinline__ double lane_value(double x, int l) {
  const unsigned long long u = (unsigned long long) __double_as_longlong(x);
  const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) u, l);
  const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (u >> 32), l);
  return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}

// LDS of one wave of cell_sum_groups_kernel
template <int B>
struct GroupTable {
  double sum[B][kGroupMax];
  uint32_t cnt[kGroupMax], claim[kGroupMax];
  uint32_t excl[64], first[64];   // the current window of 64 runs: exclusive prefix of the lengths, first positions
};

// the 64 lanes add x[0..B) to the table entries sl (lanes with live == false do nothing), entry by entry in lane order
template <int B>
__device__ __forceinline__ void group_table_add(GroupTable<B> &T, int lane, bool live, int sl, const double (&x)[B]) {
  // rounds in which the lowest pending lane of every cell adds its value (a few rounds when the 64 particles
  // spread over many cells) ...
  bool pending = live;
  for (int round = 0;; round++) {
    const int left = __builtin_popcountll(__ballot(pending));
    if (left == 0 || (round >= 2 && left > 24))
      break;
    if (pending)
      atomicMin(&T.claim[sl], (uint32_t) lane);
    __builtin_amdgcn_wave_barrier();
    const bool turn = pending && T.claim[sl] == (uint32_t) lane;
    __builtin_amdgcn_wave_barrier();
    if (turn) {
#pragma unroll
      for (int b = 0; b < B; b++)
        T.sum[b][sl] += x[b];
      T.cnt[sl] += 1;
      T.claim[sl] = ~0u;
      pending = false;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ... then cell by cell: the values of the cell's lanes are read from the lanes in ascending order and added
  // to the table entry by every lane alike (few cells with many particles each)
  unsigned long long rest = __ballot(pending);
  while (rest) {
    const int l0 = __builtin_ctzll(rest);
    const int s0 = __builtin_amdgcn_readlane(sl, l0);
    unsigned long long same = __ballot(pending && sl == s0);
    rest &= ~same;
    double acc[B];
#pragma unroll
    for (int b = 0; b < B; b++)
      acc[b] = T.sum[b][s0];
    const uint32_t more = (uint32_t) __builtin_popcountll(same);
    while (same) {
      const int l = __builtin_ctzll(same);
      same &= same - 1;
#pragma unroll
      for (int b = 0; b < B; b++)
        acc[b] += lane_value(x[b], l);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == l0) {
#pragma unroll
      for (int b = 0; b < B; b++)
        T.sum[b][s0] = acc[b];
      T.cnt[s0] += more;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

constexpr int kSortCap = 1024;   // particles of a group that one wave can put into index order in LDS

// bitonic sort of m (a power of two >= 64) 64-bit keys in LDS by one wave, ascending
__device__ __forceinline__ void wave_sort(unsigned long long *__restrict__ a, int m, int lane) {
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < m / 2; t += 64) {
        const int i = 2 * t - (t & (j - 1));   // lower element of pair t (bit log2(j) is zero)
        const unsigned long long lo = a[i], hi = a[i + j];
        const bool up = (i & k) == 0;
        if ((lo > hi) == up) {
          a[i] = hi;
          a[i + j] = lo;
        }
      }
    }
  __builtin_amdgcn_wave_barrier();
}

// Sums over the sorted runs (keys[j] = group, ids[j] = run; the runs of a group are neighbours; *nruns_dev of
// them).  Wave w looks at runs 64 w ... 64 w + 63 and does the groups that begin there.
//   BYINDEX false: the sequence the runs were cut from is in external order already (seq_slot: where particle
//     number p of the sequence is stored, NULL = at p): the group's particles are streamed as they come.
//   BYINDEX true: the sequence is the stored order (ext[p] = external index of the particle stored at p): the wave
//     collects the group's particles, sorts them by external index in LDS (<= kSortCap of them; a larger group
//     sets *overflow and is left to the general pass that follows) and streams them in that order.
// sums[v * ntot + cell], cnt[cell] / cnt_as_double[cell] (either may be NULL) must be zero on entry: groups
// without particles are not touched.
template <class VALS, int B, bool BYINDEX>
__global__ __launch_bounds__(256) void cell_sum_groups_kernel(VALS vals, const uint32_t *__restrict__ keys,
                                                              const int *__restrict__ ids,
                                                              const uint32_t *__restrict__ nruns_dev, uint32_t outside,
                                                              const uint32_t *__restrict__ run_start,
                                                              const int *__restrict__ seq_cell,
                                                              const int *__restrict__ seq_slot,
                                                              const int *__restrict__ ext, int *__restrict__ overflow,
                                                              int G, size_t ntot, double *__restrict__ sums,
                                                              int *__restrict__ cnt, double *__restrict__ cnt_as_double) {
  __shared__ GroupTable<B> s_table[4];
  __shared__ unsigned long long s_sort[BYINDEX ? 4 : 1][BYINDEX ? kSortCap : 1];
  const long long nruns = (long long) *nruns_dev;
  const int nv = vals.count();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GroupTable<B> &T = s_table[wave];
  const long long nwaves = (long long) gridDim.x * 4;
  for (long long jbase = ((long long) blockIdx.x * 4 + wave) * 64; jbase < nruns; jbase += nwaves * 64) {
    const long long j = jbase + lane;
    const uint32_t mine = j < nruns ? keys[j] : outside;
    const uint32_t before = j > 0 && j < nruns ? keys[j - 1] : outside;
    unsigned long long heads = __ballot(mine != outside && (j == 0 || before != mine));
    while (heads) {
      const int src = __builtin_ctzll(heads);
      heads &= heads - 1;
      const long long j0 = jbase + src;
      const uint32_t g = (uint32_t) __builtin_amdgcn_readlane((int) mine, src);
      // end of the group's runs: the first position of another key (found 64 positions at a time)
      long long j1 = j0;
      for (;;) {
        const long long k = j1 + lane;
        const unsigned long long same = __ballot(k < nruns && keys[k] == g);
        const int run = same == ~0ull ? 64 : __builtin_ctzll(~same);
        j1 += run;
        if (run < 64)
          break;
      }
      const long long cell0 = (long long) g * G;
      // BYINDEX: the group's particles as (external index, position) keys, in index order
      uint32_t nsorted = 0;
      if (BYINDEX) {
        for (long long jw = j0; jw < j1; jw += 64) {
          const int r = jw + lane < j1 ? ids[jw + lane] : -1;
          const uint32_t first = r >= 0 ? run_start[r] : 0;
          const uint32_t len = r >= 0 ? run_start[r + 1] - first : 0;
          uint32_t incl = len;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = (uint32_t) __shfl_up((int) incl, d);
            if (lane >= d)
              incl += y;
          }
          const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
          if (nsorted + total > (uint32_t) kSortCap) {
            nsorted = ~0u;
            break;
          }
          for (uint32_t q = 0; q < len; q++) {   // (runs are a few particles long)
            const uint32_t p = first + q;
            s_sort[wave][nsorted + (incl - len) + q] = ((unsigned long long) (uint32_t) ext[p] << 32) | p;
          }
          nsorted += total;
        }
        if (nsorted == ~0u) {
          if (lane == 0)
            *overflow = 1;
          continue;
        }
        int m = 64;
        while ((uint32_t) m < nsorted)
          m <<= 1;
        for (int t = (int) nsorted + lane; t < m; t += 64)
          s_sort[wave][t] = ~0ull;
        wave_sort(s_sort[wave], m, lane);
      }
      for (int v0 = 0; v0 < (nv > 0 ? nv : 1); v0 += B) {   // (no values: one pass for the counts)
        for (int sl = lane; sl < G; sl += 64) {
#pragma unroll
          for (int b = 0; b < B; b++)
            T.sum[b][sl] = 0.0;
          T.cnt[sl] = 0;
          T.claim[sl] = ~0u;
        }
        __builtin_amdgcn_wave_barrier();
        if (BYINDEX) {
          for (uint32_t e0 = 0; e0 < nsorted; e0 += 64) {
            const bool live = e0 + lane < nsorted;
            const long long slot = live ? (long long) (uint32_t) s_sort[wave][e0 + lane] : 0;
            const int sl = live ? (int) ((long long) seq_cell[slot] - cell0) : 0;
            double x[B];
#pragma unroll
            for (int b = 0; b < B; b++)
              x[b] = live && v0 + b < nv ? vals.get(v0 + b, slot) : 0.0;
            group_table_add<B>(T, lane, live, sl, x);
          }
        } else {
          // windows of 64 runs: lane l holds run jw + l; the window's particles are numbered through
          for (long long jw = j0; jw < j1; jw += 64) {
            const int r = jw + lane < j1 ? ids[jw + lane] : -1;
            const uint32_t first = r >= 0 ? run_start[r] : 0;
            const uint32_t len = r >= 0 ? run_start[r + 1] - first : 0;
            uint32_t incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
              const uint32_t y = (uint32_t) __shfl_up((int) incl, d);
              if (lane >= d)
                incl += y;
            }
            const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
            __builtin_amdgcn_wave_barrier();
            T.excl[lane] = incl - len;
            T.first[lane] = first;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t e0 = 0; e0 < total; e0 += 64) {
              const uint32_t e = e0 + lane;
              const bool live = e < total;
              // the run of particle e: the last lane whose exclusive prefix is <= e
              int w = 0;
#pragma unroll
              for (int step = 32; step > 0; step >>= 1)
                if (T.excl[w + step] <= e)
                  w += step;
              const uint32_t p = live ? T.first[w] + (e - T.excl[w]) : 0;
              const int sl = live ? (int) ((long long) seq_cell[p] - cell0) : 0;
              const long long slot = live ? (seq_slot ? (long long) seq_slot[p] : (long long) p) : 0;
              double x[B];
#pragma unroll
              for (int b = 0; b < B; b++)
                x[b] = live && v0 + b < nv ? vals.get(v0 + b, slot) : 0.0;
              group_table_add<B>(T, lane, live, sl, x);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        for (int sl = lane; sl < G; sl += 64) {
          const long long c = cell0 + sl;
          if (c < (long long) ntot) {
#pragma unroll
            for (int b = 0; b < B; b++)
              if (v0 + b < nv)
                sums[(size_t) (v0 + b) * ntot + (size_t) c] = T.sum[b][sl];
            if (v0 == 0) {
              if (cnt)
                cnt[c] = (int) T.cnt[sl];
              if (cnt_as_double)
                cnt_as_double[c] = (double) T.cnt[sl];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ---- crowded cells: one lane per (cell, value) -------------------------------------------------------------
// When the cells hold many particles each (a 2-D output grid: 10^7 particles on 360 x 180 cells), the whole
// (cell, slot) list is sorted by cell -- laid out in external order first, so that the stable sort leaves every
// cell's particles in ascending external index -- and lane (cell, value) walks its cell's list and adds: the
// chains are long, there are cells x values of them, and the lanes of a cell share the list reads.

// (box != NULL: the cell of every particle is computed here -- write_grid's box index, one pass over the
//  particle arrays less than box_index_kernel + this kernel)
struct GridBoxArgs {
  BoxGrid G;
  double t0, t1;
  const double *time, *lon, *lat, *p;
};

__global__ void cell_slot_pairs_kernel(const int *__restrict__ cell, const int *__restrict__ ext, long long n,
                                       uint32_t outside, uint2 *__restrict__ pairs, const GridBoxArgs box, int use_box) {
  // one 8-byte store per particle: the stores go all over the array (ext is a random permutation of the stored
  // order), and a store costs a memory transaction whatever its width -- half as many as with two arrays
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
    const long long at = ext ? (long long) ext[i] : i;
    const int c = use_box ? box_cell(box.G, box.t0, box.t1, box.time[i], box.lon[i], box.lat[i], box.p[i], nullptr, i, 0)
                          : cell[i];
    pairs[at] = make_uint2(c >= 0 ? (uint32_t) c : outside, (uint32_t) i);
  }
}

#ifndef MPHIP_CHAIN_LOADS
#define MPHIP_CHAIN_LOADS 16   // (gridded output of C3: 267 us with 4, 249 with 8, 225 with 16)
#endif
template <class VALS>
__global__ __launch_bounds__(256) void cell_sum_chains_kernel(VALS vals, const int *__restrict__ slots,
                                                              const uint32_t *__restrict__ sorted_keys, long long nlist,
                                                              size_t ntot,
                                                              double *__restrict__ sums, int *__restrict__ cnt,
                                                              double *__restrict__ cnt_as_double) {
  const int nv = vals.count();
  const int width = nv < 1 ? 1 : nv < 64 ? nv : 64;   // lanes per cell (one for the counts when there are no values)
  const int per_wave = 64 / width;            // cells per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / width, v0 = lane % width;
  // Every wave walks a contiguous run of cells, the runs of the workgroups of one XCD next to each other
  // (workgroup b runs on XCD b % 8): neighbouring cells read neighbouring lines of the quantity arrays, since the
  // stored order follows the meteo grid.  (Every gather still misses the L2 -- 2.5e7 misses, 3.1 GB per output of
  // C3 for 0.24 GB of values, profiles/r03_gridsums_counters.txt -- whatever the number of waves in flight.)
  const size_t nblocks = gridDim.x, lb = (blockIdx.x % 8) * (nblocks / 8) + blockIdx.x / 8;
  const size_t wave = lb * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (nblocks * blockDim.x) >> 6;
  if (sub >= per_wave)
    return;
  const size_t groups = (ntot + per_wave - 1) / per_wave, per = (groups + nwaves - 1) / nwaves;
  for (size_t gidx = wave * per; gidx < (wave + 1) * per && gidx < groups; gidx++) {
    const size_t c = gidx * per_wave + sub;
    if (c >= ntot)
      break;
    // the cell's range [b, e) of the list sorted by cell: two searches in the sorted keys (2 x log2(n) probes per
    // cell; a pass over the whole list that marks the boundaries, and a clearing pass before it, cost more)
    uint32_t b, e;
    {
      uint32_t lo = 0, hi = (uint32_t) nlist;   // first position with key >= c
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (sorted_keys[mid] < (uint32_t) c)
          lo = mid + 1;
        else
          hi = mid;
      }
      b = lo;
      hi = (uint32_t) nlist;                    // first position with key > c
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (sorted_keys[mid] <= (uint32_t) c)
          lo = mid + 1;
        else
          hi = mid;
      }
      e = lo;
    }
    for (int v = v0; v < nv; v += width) {
      double sum = 0.0;
      uint32_t k = b;
      constexpr int kInFlight = MPHIP_CHAIN_LOADS;   // the walk is a chain of dependent loads: its speed is the
      for (; k + kInFlight <= e; k += kInFlight) {   // number of them in flight; the additions stay in list order
        int sl[kInFlight];
        double x[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; u++)
          sl[u] = slots[k + u];
#pragma unroll
        for (int u = 0; u < kInFlight; u++)
          x[u] = vals.get(v, (long long) sl[u]);
#pragma unroll
        for (int u = 0; u < kInFlight; u++)
          sum += x[u];
      }
      for (; k < e; k++)
        sum += vals.get(v, (long long) slots[k]);
      sums[(size_t) v * ntot + c] = sum;
    }
    if (v0 == 0) {
      if (cnt)
        cnt[c] = (int) (e - b);
      if (cnt_as_double)
        cnt_as_double[c] = (double) (e - b);
    }
  }
}

// counts <-> doubles around an all-reduce hook that only knows doubles (tests, staged host collectives)
// ---- the exchange of module_mixing's cell sums between ranks, restricted to the occupied levels ----------------------
// Boxes are indexed (column, level) with the level fastest; the particles of a run occupy a band of levels (0.5-30 km of
// a grid that spans -5 ... 85 km: a third of it), the same band on every rank (index-range shards are spread over the
// globe alike).  level_occupancy_kernel marks the levels that hold a particle on this rank (summed over the ranks: on
// any rank); pack / unpack move the band [lo, hi] of every column into a dense buffer and back, so that the all-reduce
// carries (hi - lo + 1) / nz of the bytes.
__global__ void level_occupancy_kernel(const int *__restrict__ cnt, size_t ntot, int nz, double *__restrict__ occ) {
  __shared__ unsigned s_occ[256];
  for (int l = threadIdx.x; l < 256; l += blockDim.x)
    s_occ[l] = 0;
  __syncthreads();
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < ntot; i += (size_t) gridDim.x * blockDim.x)
    if (cnt[i] > 0)
      s_occ[(int) (i % (size_t) nz)] = 1;      // (a benign race: everybody writes the same value)
  __syncthreads();
  for (int l = threadIdx.x; l < nz; l += blockDim.x)
    if (s_occ[l])
      occ[l] = 1.0;                            // (likewise)
}

// dense[(q * ncol + c) * nl + (l - lo)] <-> full[q * ntot + c * nz + l] for lo <= l <= hi (nl = hi - lo + 1); T = double / int
template <class T>
__global__ void pack_levels_kernel(const T *__restrict__ full, T *__restrict__ dense, size_t ncol, int nz, int lo, int nl,
                                   int nq, bool unpack, T *__restrict__ full_out) {
  const size_t per = ncol * (size_t) nl, total = per * (size_t) nq;
  for (size_t j = blockIdx.x * (size_t) blockDim.x + threadIdx.x; j < total; j += (size_t) gridDim.x * blockDim.x) {
    const size_t q = j / per, r = j - q * per, c = r / (size_t) nl;
    const int l = (int) (r - c * (size_t) nl);
    const size_t at = q * ncol * (size_t) nz + c * (size_t) nz + (size_t) (lo + l);
    if (unpack)
      full_out[at] = dense[j];
    else
      dense[j] = full[at];
  }
}

__global__ void int_to_double_kernel(const int *__restrict__ in, double *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    out[i] = (double) in[i];
}

__global__ void double_to_int_kernel(const double *__restrict__ in, int *__restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    out[i] = (int) in[i];
}

// buf[0 .. ncell) += 1, buf[(1 + iq) ncell ..] += q, buf[(1 + nq + iq) ncell ..] += q^2
__global__ __launch_bounds__(256) void grid_accumulate_kernel(DevAtm a, const int *__restrict__ cell, int nq,
                                                              size_t ncell, double *__restrict__ buf, int T,
                                                              long long per_block, GridKernel kern) {
  extern __shared__ double s_tab[];
  LdsTable tab;
  tab.init(s_tab, T, 1 + 2 * nq);
  const long long first = blockIdx.x * per_block;
  const long long last = first + per_block < a.np ? first + per_block : a.np;
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    const int c = cell[i];
    if (c >= 0) {
      const int slot = tab.slot_for(c);
      if (slot >= 0)
        tab.add(slot, 0, 1.0);
      else
        unsafeAtomicAdd(&buf[c], 1.0);
      const double kernel = kern.weight(i);
      for (int iq = 0; iq < nq; iq++) {
        const double v = kernel * a.q[iq][i];
        if (slot >= 0) {
          tab.add(slot, 1 + iq, v);
          tab.add(slot, 1 + nq + iq, v * v);
        } else {
          unsafeAtomicAdd(&buf[(size_t) (1 + iq) * ncell + c], v);
          unsafeAtomicAdd(&buf[(size_t) (1 + nq + iq) * ncell + c], v * v);
        }
      }
    }
  }
  tab.flush(buf, ncell);
}

// ---------------------------------------------------------------------------
// module_meteo (mptrac.c:5062-5165): sample the meteo fields at every particle
// (no dt guard) into the requested quantities.  Only the fields a requested
// quantity depends on are read (need3 / need2, set by the host): level fields
// from the packed two-snapshot records, so that a particle's 2 x 2 x 2 x 2 corner
// values of up to four fields come with 16 loads and neighbouring particles of
// the locality order share cache lines; the values are
// those INTPOL_TIME_ALL (mptrac.h:1278-1318) produces -- one index / weight
// set from the first 3-D call, re-used by all other 3-D and 2-D calls.
// ---------------------------------------------------------------------------

struct MeteoArgs {
  mphip_ctl_t ctl;
  DevMet met;
  DevAtm atm;
  unsigned need3, need2;
  DevZm zm[MPHIP_NZM];             // zonal-mean climatologies (tables of the requested quantities only)
  int nblocks_logical;             // as StepParams: contiguous runs of the locality order per block
  long long per_block;
  int xcd_map;
};

#ifndef MPHIP_METEO_WAVES_PER_SIMD
#define MPHIP_METEO_WAVES_PER_SIMD 3
#endif
__global__ __launch_bounds__(256, MPHIP_METEO_WAVES_PER_SIMD) void meteo_kernel(const MeteoArgs G) {
  extern __shared__ double s_axes[];
  const DevMet &M = G.met;
  const DevAtm &a = G.atm;
  const int *qm = G.ctl.qnt_met;
  const Axes A = load_axes(M, s_axes);
  __syncthreads();
  const unsigned n3 = G.need3, n2 = G.need2;
  const int nb = G.nblocks_logical;
  const int lb = G.xcd_map ? (int) (blockIdx.x % 8) * (nb / 8) + (int) (blockIdx.x / 8) : (int) blockIdx.x;
  const long long first = (long long) lb * G.per_block;
  long long last = first + G.per_block;
  if (last > a.np)
    last = a.np;
  auto bits = [](int lo, int hi) { return ((2u << hi) - 1u) & ~((1u << lo) - 1u); };   // bits lo ... hi
  const bool want_wind = n3 & bits(MPHIP_U, MPHIP_W);
  const bool want_cloud = n3 & bits(MPHIP_LWC, MPHIP_SWC);
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    const double tm = a.time[i], p = a.p[i], lon = a.lon[i], lat = a.lat[i];
    Stencil s = stencil_zero();
    stencil_3d(M, A, p, lon, lat, s);
    const double wt = time_weight(M, tm);
#define SETQ(k, val)                                                                          \
  if (qm[k] >= 0)                                                                             \
    a.q[qm[k]][i] = (val)
    // every group stores its quantities right after its loads (the order of the
    // stores is immaterial; short live ranges keep the kernel at 3+ waves / SIMD)
    if (want_wind) {
      WindCorners c;
      load_wind(M, s, c);
      const double u = wind_time_3d(c, s, wt, 0);
      const double v = wind_time_3d(c, s, wt, 1);
      const double w = wind_time_3d(c, s, wt, 2);
      SETQ(MPHIP_MQ_U, u);
      SETQ(MPHIP_MQ_V, v);
      SETQ(MPHIP_MQ_W, w);
      SETQ(MPHIP_MQ_VH, sqrt(u * u + v * v));
      SETQ(MPHIP_MQ_VZ, -1e3 * kH0 / p * w);
    } else {   // the reference's zero-initialised fields
      SETQ(MPHIP_MQ_U, 0.0);
      SETQ(MPHIP_MQ_V, 0.0);
      SETQ(MPHIP_MQ_W, 0.0);
      SETQ(MPHIP_MQ_VH, 0.0);
      SETQ(MPHIP_MQ_VZ, -1e3 * kH0 / p * 0.0);
    }
    if (want_cloud) {
      CloudCorners c;
      load_quad(M.cloud, M, s, c);
      SETQ(MPHIP_MQ_LWC, cloud_time_3d(c, s, wt, 0));
      SETQ(MPHIP_MQ_RWC, cloud_time_3d(c, s, wt, 1));
      SETQ(MPHIP_MQ_IWC, cloud_time_3d(c, s, wt, 2));
      SETQ(MPHIP_MQ_SWC, cloud_time_3d(c, s, wt, 3));
    }
    if (n3 & bits(MPHIP_Z, MPHIP_PV)) {
      PairCorners c;
      load_pair_3d(M.mx, M, s, c);
      SETQ(MPHIP_MQ_ZG, pair_field_time_3d(c, s, wt, 0));
      SETQ(MPHIP_MQ_PV, pair_field_time_3d(c, s, wt, 1));
    }
    if (n3 & bits(MPHIP_O3, MPHIP_CC)) {
      PairCorners c;
      load_pair_3d(M.mx + (size_t) M.nx * (size_t) M.ny * (size_t) M.np, M, s, c);
      SETQ(MPHIP_MQ_O3, pair_field_time_3d(c, s, wt, 0));
      SETQ(MPHIP_MQ_CC, pair_field_time_3d(c, s, wt, 1));
    }
    // surface fields: sfa {ps,pbl}, sfb {cape,cin,pel}, sfc {pct,pcb,cl}, sfd {ess,nss,shf}, then the pairs of mx2
    double ps = 0.0;
    if (n2 & bits(MPHIP_PS, MPHIP_PBL)) {
      SurfA c;
      load_sfa(M, s, c);
      ps = sfa_time_2d(c, s, wt, 0);
      SETQ(MPHIP_MQ_PS, ps);
      SETQ(MPHIP_MQ_PBL, sfa_time_2d(c, s, wt, 1));
    }
    if (n2 & bits(MPHIP_CAPE, MPHIP_PEL)) {
      SurfB c;
      load_sfb(M.sfb, M, s, c);
      SETQ(MPHIP_MQ_CAPE, sfb_time_2d(c, s, wt, 0));
      SETQ(MPHIP_MQ_CIN, sfb_time_2d(c, s, wt, 1));
      SETQ(MPHIP_MQ_PEL, sfb_time_2d(c, s, wt, 2));
    }
    if (n2 & bits(MPHIP_PCT, MPHIP_CL)) {
      SurfB c;
      load_sfb(M.sfc, M, s, c);
      SETQ(MPHIP_MQ_PCT, sfb_time_2d(c, s, wt, 0));
      SETQ(MPHIP_MQ_PCB, sfb_time_2d(c, s, wt, 1));
      SETQ(MPHIP_MQ_CL, sfb_time_2d(c, s, wt, 2));
    }
    if (n2 & bits(MPHIP_ESS, MPHIP_SHF)) {
      SurfB c;
      load_sfb(M.sfd, M, s, c);
      SETQ(MPHIP_MQ_ESS, sfb_time_2d(c, s, wt, 0));
      SETQ(MPHIP_MQ_NSS, sfb_time_2d(c, s, wt, 1));
      SETQ(MPHIP_MQ_SHF, sfb_time_2d(c, s, wt, 2));
    }
#define PAIR2(pr, qa, qb)                                                                     \
  if (n2 & bits(MPHIP_TS + 2 * (pr), MPHIP_TS + 2 * (pr) + 1)) {                              \
    SurfA c;                                                                                  \
    load_pair_2d(M.mx2 + (size_t) (pr) * (size_t) M.nx * (size_t) M.ny, M, s, c);             \
    SETQ(qa, sfa_time_2d(c, s, wt, 0));                                                       \
    if ((qb) >= 0)                                                                            \
      SETQ((qb) >= 0 ? (qb) : 0, sfa_time_2d(c, s, wt, 1));                                   \
  }
    PAIR2(0, MPHIP_MQ_TS, MPHIP_MQ_ZS)
    PAIR2(1, MPHIP_MQ_US, MPHIP_MQ_VS)
    PAIR2(2, MPHIP_MQ_LSM, MPHIP_MQ_SST)
    PAIR2(3, MPHIP_MQ_PT, MPHIP_MQ_TT)
    PAIR2(4, MPHIP_MQ_ZT, MPHIP_MQ_H2OT)
    PAIR2(5, MPHIP_MQ_PLCL, MPHIP_MQ_PLFC)
    PAIR2(6, MPHIP_MQ_O3C, -1)
#undef PAIR2
    SETQ(MPHIP_MQ_P, p);
    // temperature, water vapour, surface pressure and what derives from them
    const double t = ((n3 >> MPHIP_T) & 1u) ? temp_time_3d(M, s, wt) : 0.0;
    const double h2o = ((n3 >> MPHIP_H2O) & 1u) ? pair_time_3d(M.h2o, M, s, wt) : 0.0;
    SETQ(MPHIP_MQ_T, t);
    SETQ(MPHIP_MQ_H2O, h2o);
    SETQ(MPHIP_MQ_RHO, rho_air(p, t));
    SETQ(MPHIP_MQ_PSAT, psat_of(t));
    SETQ(MPHIP_MQ_PSICE, psice_of(t));
    SETQ(MPHIP_MQ_PW, pw_of(p, h2o));
    SETQ(MPHIP_MQ_SH, sh_of(h2o));
    SETQ(MPHIP_MQ_RH, pw_of(p, h2o) / psat_of(t) * 100.);      // RH, mptrac.h:1906
    SETQ(MPHIP_MQ_RHICE, pw_of(p, h2o) / psice_of(t) * 100.);  // RHICE, mptrac.h:1936
    SETQ(MPHIP_MQ_THETA, theta_of(p, t));
    SETQ(MPHIP_MQ_ZETA_D, zeta_of(ps, p, t));
    SETQ(MPHIP_MQ_TVIRT, tvirt(t, h2o));
    SETQ(MPHIP_MQ_LAPSE, lapse_rate(t, h2o));
    SETQ(MPHIP_MQ_TDEW, tdew_of(p, h2o));
    const double tice = tice_of(p, h2o);
    SETQ(MPHIP_MQ_TICE, tice);
    // the climatology part of the list (mptrac.c:5129-5141, 5158-5163)
    if (qm[MPHIP_MQ_HNO3] >= 0 || qm[MPHIP_MQ_OH] >= 0 || qm[MPHIP_MQ_H2O2] >= 0 || qm[MPHIP_MQ_HO2] >= 0
        || qm[MPHIP_MQ_O1D] >= 0 || qm[MPHIP_MQ_TNAT] >= 0) {
      const double lat_ref = G.ctl.met_coord_type == 0 ? lat : G.ctl.met_utm_ref_lat;
      if (qm[MPHIP_MQ_HNO3] >= 0)
        a.q[qm[MPHIP_MQ_HNO3]][i] = clim_zm(G.zm[MPHIP_ZM_HNO3], tm, lat_ref, p);
      if (qm[MPHIP_MQ_OH] >= 0)
        a.q[qm[MPHIP_MQ_OH]][i] = clim_oh(G.ctl, G.zm[MPHIP_ZM_OH], tm, lon, lat, p);
      if (qm[MPHIP_MQ_H2O2] >= 0)
        a.q[qm[MPHIP_MQ_H2O2]][i] = clim_zm(G.zm[MPHIP_ZM_H2O2], tm, lat_ref, p);
      if (qm[MPHIP_MQ_HO2] >= 0)
        a.q[qm[MPHIP_MQ_HO2]][i] = clim_zm(G.zm[MPHIP_ZM_HO2], tm, lat_ref, p);
      if (qm[MPHIP_MQ_O1D] >= 0)
        a.q[qm[MPHIP_MQ_O1D]][i] = clim_zm(G.zm[MPHIP_ZM_O1D], tm, lat_ref, p);
      if (qm[MPHIP_MQ_TNAT] >= 0) {   // (at the particle's own latitude, Cartesian grid or not: mptrac.c:5160)
        const double tnat = nat_temperature(p, h2o, clim_zm(G.zm[MPHIP_ZM_HNO3], tm, lat, p));
        a.q[qm[MPHIP_MQ_TNAT]][i] = tnat;
        SETQ(MPHIP_MQ_TSTS, 0.5 * (tice + tnat));   // of the two quantities just stored (both requested: check_meteo)
      }
    }
#undef SETQ
  }
}

// ---------------------------------------------------------------------------
// self-test kernels
// ---------------------------------------------------------------------------

__global__ void test_sincosf_kernel(uint32_t first, uint32_t count, float *__restrict__ c, float *__restrict__ s) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const float x = __uint_as_float(first + i);
    c[i] = libm_sincosf(x, 1);
    s[i] = libm_sincosf(x, 0);
  }
}

// Cost of the building blocks of the step kernel on the resident particles and grids (profiling aid:
// tools/piece_cost.py runs every PIECE under rocprofv3 --pmc and subtracts piece 0).  Each thread loads
// its particle like the step kernel, runs the piece `reps` times on slightly different inputs and adds
// the result to out[i] so that nothing is optimised away.
template <int PIECE>
__global__ __launch_bounds__(256, MPHIP_STEP_WAVES_PER_SIMD) void piece_kernel(const StepParams S, int reps,
                                                                               double *__restrict__ out) {
  extern __shared__ double s_axes[];
  const DevMet &M = S.met;
  const DevAtm &a = S.atm;
  const mphip_ctl_t &ctl = S.ctl;
  const Axes A = load_axes(M, s_axes);
  double *dst = s_axes + ((axes_doubles(M) * 8 + (size_t) M.lut_size * 2 + 15) & ~(size_t) 15) / 8;
  const double *src = (const double *) S.clim;
  for (int i = threadIdx.x; i < (int) (sizeof(DevClim) / sizeof(double)); i += blockDim.x)
    dst[i] = src[i];
  const DevClim &clim = *(const DevClim *) dst;
  double *ltab = dst + sizeof(DevClim) / sizeof(double);
  for (int i = threadIdx.x; i < kLibmLogExpDoubles; i += blockDim.x)
    ltab[i] = libm_tables()[i];
  __syncthreads();
  const int nb = S.nblocks_logical;
  const int lb = S.xcd_map ? (int) (blockIdx.x % 8) * (nb / 8) + (int) (blockIdx.x / 8) : (int) blockIdx.x;
  const long long first = (long long) lb * S.per_block;
  long long last = first + S.per_block;
  if (last > a.np)
    last = a.np;
  for (long long i = first + threadIdx.x; i < last; i += blockDim.x) {
    Particle P;
    P.time = a.time[i];
    P.lon = a.lon[i];
    P.lat = a.lat[i];
    P.p = a.p[i];
    P.dt = 180.0;
    const uint64_t g = (uint64_t) (a.ip0 + (a.ext ? (long long) a.ext[i] : i));
    double acc = 0.0;
    for (int r = 0; r < reps; r++) {
      const double lon = P.lon + 1e-3 * r, lat = P.lat - 1e-3 * r, p = P.p * (1.0 + 1e-4 * r);
      if (PIECE == 1) {
        Stencil s;
        stencil_3d(M, A, p, lon, lat, s);
        acc += s.wp + s.wx + s.wy + (double) (s.ip + s.ix + s.iy);
      } else if (PIECE == 2) {
        Stencil s = stencil_zero();
        stencil_2d(M, A, lon, lat, s);
        acc += s.wx + s.wy + (double) (s.ix + s.iy);
      } else if (PIECE == 3) {   // one Runge-Kutta stage's interpolation: stencil + 12 loads + 3 components
        Stencil s;
        stencil_3d(M, A, p, lon, lat, s);
        WindCorners c;
        load_wind(M, s, c);
        const double wt = time_weight(M, P.time + r);
        acc += wind_time_3d(c, s, wt, 0) + wind_time_3d(c, s, wt, 1) + wind_time_3d(c, s, wt, 2);
      } else if (PIECE == 4) {
        double r0, r1, r2;
        normal_triple(libm_tables(), S.ctr_turb + (uint64_t) r, g, r0, r1, r2);   // (table in device memory)
        acc += r0 + r1 + r2;
      } else if (PIECE == 5) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        position(M, A, Q);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 6) {   // {ps, pbl} at the particle: what module_diff_turb / module_convection start with
        Stencil s = stencil_zero();
        stencil_2d(M, A, lon, lat, s);
        SurfA c;
        load_sfa(M, s, c);
        const double wt = time_weight(M, P.time + r);
        acc += sfa_time_2d(c, s, wt, 0) + sfa_time_2d(c, s, wt, 1);
      } else if (PIECE == 7) {
        acc += temperature_at(M, A, P.time + r, p, lon, lat);
      } else if (PIECE == 8) {
        acc += dx2coord(0, 100.0 + r, lat) + dy2coord(0, 50.0 + r);
      } else if (PIECE == 9) {
        acc += tropo_weight(ctl, clim, P.time + r, lat, p);
      } else if (PIECE == 10) {
        acc += sedi(p, 220.0 + r, 1.0, 1000.0, ltab);
      } else if (PIECE == 11) {
        acc += uniform01(S.ctr_conv + g + (uint64_t) r);
      } else if (PIECE == 12) {   // module_diff_turb as a whole
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        diff_turb(ctl, M, A, clim, Q, S.ctr_turb + (uint64_t) r, g, nullptr, ltab);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 13) {   // module_convection + module_sedi
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        convection(ctl, M, A, Q, S.ctr_conv + (uint64_t) r, g);
        sedimentation(M, A, Q, 1.0, 1000.0);
        acc += Q.p;
      } else if (PIECE == 14) {   // module_diff_meso without the particle-array traffic
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        float up = 0.1f * r, vp = 0.2f, wp = 1e-4f;
        WindCache wc;
        wind_cache_reset(wc, false);
        diff_meso(ctl, M, A, Q, up, vp, wp, S.ctr_meso + (uint64_t) r, g, nullptr, wc, ltab);
        acc += Q.lon + Q.lat + Q.p + (double) (up + vp + wp);
      } else if (PIECE == 15) {   // module_advect (RK4) without the wind-corner cache
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        WindCache wc;
        wind_cache_reset(wc, false);
        NoHook none;
        advect_n<4>(M, A, Q, none, wc);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 16) {
        Stencil s;
        stencil_3d_fast(M, A, p, lon, lat, s);
        acc += s.wp + s.wx + s.wy + (double) (s.ip + s.ix + s.iy);
      } else if (PIECE == 17) {
        Stencil s = stencil_zero();
        horiz_fast(M, A, lon, lat, s);
        acc += s.wx + s.wy + (double) (s.ix + s.iy);
      } else if (PIECE == 18) {   // one Runge-Kutta stage's interpolation, lean
        Stencil s;
        stencil_3d_fast(M, A, p, lon, lat, s);
        WindCorners c;
        load_wind(M, s, c);
        double u, v, w;
        wind_uvw_fast(c, s, time_weight(M, P.time + r), u, v, w);
        acc += u + v + w;
      } else if (PIECE == 19) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        position_fast(M, A, Q);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 20) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        diff_turb_fast(ctl, M, A, clim, Q, S.ctr_turb + (uint64_t) r, g, nullptr, ltab);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 21) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        conv_sedi_fast(ctl, M, A, Q, MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI, S.ctr_conv + (uint64_t) r, g, nullptr, 1.0, 1000.0, ltab);
        acc += Q.p;
      } else if (PIECE == 22) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        float up = 0.1f * r, vp = 0.2f, wp = 1e-4f;
        WindCache wc;
        wind_cache_reset(wc, true);
        diff_meso_fast(ctl, M, A, Q, up, vp, wp, S.ctr_meso + (uint64_t) r, g, nullptr, wc, ltab);
        acc += Q.lon + Q.lat + Q.p + (double) (up + vp + wp);
      } else if (PIECE == 23) {
        Particle Q = P;
        Q.lon = lon;
        Q.lat = lat;
        Q.p = p;
        WindCache wc;
        wind_cache_reset(wc, true);
        NoHook none;
        advect_rk4_fast(M, A, Q, none, wc);
        acc += Q.lon + Q.lat + Q.p;
      } else if (PIECE == 24) {
        double rr = 0.0, ee = 0.0;
        normal_pair_from(ltab, (S.ctr_turb + g + (uint64_t) r) * kSquaresKey, ee, rr);
        acc += rr + ee;
      } else if (PIECE == 25) {
        acc += libm_log(ltab, u64_to_double(squares(S.ctr_turb + g + (uint64_t) r) | 1) * 0x1p-64);
      } else if (PIECE == 26) {
        float sv, cv;
        libm_sincosf_both((float) (6.28 * uniform01(S.ctr_turb + g + (uint64_t) r)), sv, cv);
        acc += (double) sv + (double) cv;
      } else {
        acc += lon + lat + p;
      }
    }
    out[i] = acc;
  }
}

// out[i] = exp(x[i]) / log(x[i]) / pow(x[i], y[i]) / sqrt(x[i]) / cos(x[i]) / sin(x[i]) (op 0 .. 5) as the kernels
// evaluate them: the C library's functions of mphip_libm.h and the square root of the Box-Muller radius; op + 16: exp /
// log / pow tables copied to LDS first
__global__ __launch_bounds__(256) void test_libm_kernel(int op, const double *__restrict__ x, const double *__restrict__ y,
                                                         long long n, double *__restrict__ out) {
  __shared__ double s_tab[kLibmDoubles];
  const double *lt = libm_tables();
  if (op & 16) {
    for (int i = threadIdx.x; i < kLibmDoubles; i += blockDim.x)
      s_tab[i] = lt[i];
    __syncthreads();
    lt = s_tab;
  }
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
    switch (op & 15) {
    case 0: out[i] = libm_exp(lt, x[i]); break;
    case 1: out[i] = libm_log(lt, x[i]); break;
    case 2: out[i] = libm_pow(lt, x[i], y[i]); break;
    case 4: out[i] = libm_cos(x[i]); break;
    case 5: out[i] = libm_sin(x[i]); break;
    default: out[i] = sqrt_rn(x[i]); break;
    }
  }
}

// out[i], i < n: what module_rng(ctl, rs, n, method) leaves in rs[i]
__global__ void test_rng_kernel(uint64_t ctr, long long n, int method, double *__restrict__ out) {
  const double *ltab = libm_tables();
  for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
    if (method == 0)
      out[i] = uniform01(ctr + (uint64_t) i);
    else if (method == 1) {
      double e, o;
      normal_pair(ltab, ctr, (uint64_t) i & ~1ull, e, o);
      out[i] = (i & 1) ? o : e;
    } else {      // method 2: the same normals the way the modules take them -- rs[3 g .. 3 g + 2] of particle g at once
      double r[3];
      normal_triple(ltab, ctr, (uint64_t) (i / 3), r[0], r[1], r[2]);
      out[i] = r[i % 3];
    }
  }
}

}   // namespace mphip
