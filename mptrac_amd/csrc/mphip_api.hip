// mphip_api.hip -- C ABI of the MI355X back end (include/mptrac_hip.h):
// context, device mirrors of the reference structs, the module scheduler of
// mptrac_run_timestep and the kernel launches.  Built for gfx950 only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "mphip_kernels.hpp"

using namespace mphip;

namespace {

constexpr unsigned kAdv = MPHIP_MOD_TIMESTEPS | MPHIP_MOD_POSITION | MPHIP_MOD_ADVECT | MPHIP_MOD_POSITION2;
constexpr unsigned kAdvTurb = kAdv | MPHIP_MOD_DIFF_TURB;
constexpr unsigned kAdvDiff = kAdvTurb | MPHIP_MOD_DIFF_MESO;
constexpr unsigned kAdvTurbConvSedi = kAdvTurb | MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI;
constexpr unsigned kAdvDiffConvSedi = kAdvDiff | MPHIP_MOD_CONVECTION | MPHIP_MOD_SEDI;
// every mover behind module_advect, dt from memory (single-module sequences of a caller)
constexpr unsigned kDiffConvSediOnly = MPHIP_MOD_TIMESTEPS | MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DIFF_MESO | MPHIP_MOD_CONVECTION
  | MPHIP_MOD_SEDI | MPHIP_MOD_POSITION2;
constexpr unsigned kTailOnly = MPHIP_MOD_TIMESTEPS;   // no mover: a launch of loss / decay / deposition modules only
constexpr unsigned kParticleBits = 0x3fffu | MPHIP_MOD_ISOSURF | MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2
  | MPHIP_MOD_ISOSURF_INIT;

struct MetSlot {
  bool valid = false;
  double time = 0;
  float *f3[MPHIP_N3D] = {};
  float *f2[MPHIP_N2D] = {};
  bool has3[MPHIP_N3D] = {};
  bool has2[MPHIP_N2D] = {};
  std::vector<double> lon, lat, p;    // the snapshot's own axes: the reference interpolates on those of the current met0
  float ps11 = 0.f;                   // ps at grid node [1][1] (module_position reflects there)
  // smallest surface pressure of the snapshot (-inf if a value is not finite: no shortcut then) and smallest
  // finite cloud-top pressure: lower bounds of what the deposition modules interpolate (DevMet::ps_skip, pct_skip)
  double ps_min = -HUGE_VAL, pct_min = -HUGE_VAL;
  // extremes of ps / pbl (all values finite, else unknown = the defaults) and the smallest pel that is not a NaN:
  // bounds of what module_diff_turb and module_convection interpolate (DevMet::turb_skip, conv_skip)
  double ps_max = HUGE_VAL, pbl_min = -HUGE_VAL, pbl_max = HUGE_VAL, pel_min = -HUGE_VAL;
};

}   // namespace

struct mphip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::mutex err_lock;

  mphip_ctl_t ctl;
  bool have_ctl = false;
  DevClim *d_clim = nullptr;
  bool have_clim = false;
  double *d_zm[MPHIP_NZM] = {};      // zonal-mean climatologies: [time | p | lat | vmr] each
  DevZm zm[MPHIP_NZM] = {};
  double *d_ts[MPHIP_NTR] = {};      // trace-gas time series: [time | vmr] each
  DevTracerSeries h_tracers = {};    // host copy of *d_tracers
  DevTracerSeries *d_tracers = nullptr;

  // meteo
  MetSlot slot[2];
  int flip = 0;                       // logical slot s lives in slot[s ^ flip]
  MetSlot next;                       // snapshot being uploaded ahead of time (mphip_prefetch_met)
  bool next_pending = false;
  hipStream_t copy_stream = nullptr;  // uploads of `next` run here, beside the kernels on `stream`
  hipEvent_t next_ready = nullptr;
  hipEvent_t main_mark = nullptr;     // copies into `next` start after the kernels queued so far (they may read its arrays)
  // module_meteo only writes quantities nothing on the device reads: its launch is deferred until someone
  // can see the result (a download, gridded output, a single-module call) or its inputs change; a pending
  // launch that the next time step would overwrite unseen is dropped
  bool fuse_sort = true;              // module_sort's gather of time, p, lon, lat inside the following step launch
  const int *fused_perm = nullptr;    // set by do_sort, consumed by the next launch_step
  bool fuse_quantities = true;        // ... which then moves the quantity arrays too (option fuse_sort_quantities)
  bool fused_quantities = false;
  bool lazy_meteo = true;
  bool meteo_pending = false;
  bool pin_host_atm = false;          // page-lock the caller's particle arrays (persistent atm_t of a C caller only)
  // One page-locked span of caller memory (option pin_host_atm: the particle arrays of a persistent atm_t,
  // which lie next to each other in one allocation); released by mphip_destroy.
  std::vector<std::pair<uintptr_t, uintptr_t>> pinned;
  std::mutex pinned_lock;
  std::thread uploader;               // mphip_prefetch_met: issues the (blocking, pageable) copies beside the stepping thread
  int uploader_rc = 0;
  std::atomic<bool> uploader_done{ false };
  std::string uploader_err;
  int nx = 0, ny = 0, npl = 0, coord_type = 0;   // npl: pressure levels (met_t::np)
  int nml = 0;                                    // model levels (met_t::npl), 0 = none uploaded
  // Packed two-snapshot grids (layouts: mphip_device.hpp).  `pk` is the set the kernels read; `pk_next` is
  // built beside the time steps from (met1, prefetched snapshot) on the copy stream and trades places with it
  // in mphip_commit_met, so that a hand-over costs the stepping stream nothing.
  struct PackedGrids {
    float *wind = nullptr, *temp = nullptr, *h2o = nullptr;
    float *mlw = nullptr, *zl2 = nullptr, *pl2 = nullptr;   // model levels: {ul,vl,zeta_dot|wl}, {zetal}, {pl} pairs
    f32x4 *mx = nullptr, *mx2 = nullptr;                      // level / surface pair records of module_meteo
    f32x4 *cloud = nullptr, *sfa = nullptr, *sfb = nullptr, *sfc = nullptr, *sfd = nullptr;
    f32x4 *cp2 = nullptr;                                     // {cape,pel} pair records (module_convection)
    int *ml_mono = nullptr;                                   // cleared by the pack kernel if a height column is not monotonic
    bool ml_monotonic = false;
    void release() {
      for (void *q : { (void *) wind, (void *) temp, (void *) h2o, (void *) mlw, (void *) zl2, (void *) pl2, (void *) mx,
                       (void *) mx2, (void *) cloud, (void *) sfa, (void *) sfb, (void *) sfc, (void *) sfd, (void *) cp2,
                       (void *) ml_mono })
        if (q)
          (void) hipFree(q);
      *this = PackedGrids();
    }
  };
  PackedGrids pk, pk_next;
  bool pk_next_valid = false;         // pk_next holds (met1, prefetched snapshot), packed with the control parameters of then
  std::vector<double> h_lon, h_lat, h_p;
  double *d_axes = nullptr;           // axes blob (layout: DevMet::axes)
  int lut_base = 0, lut_size = 0;
  size_t axes_bytes = 0;
  bool packed_dirty = true;

  // particles
  long long np = 0, ip0 = 0, np_total = 0, cap = 0;
  int nq = 0;
  double *d_arr[4 + MPHIP_NQ_MAX] = {};   // time, p, lon, lat, q[*]
  double *d_alt[4 + MPHIP_NQ_MAX] = {};   // gather targets (ping-pong)
  float *d_uvwp[3] = {};
  float *d_uvwp_alt[3] = {};
  double *d_dt = nullptr;
  double *d_dt_alt = nullptr;
  uint64_t rng_ctr = 0;
  double *d_iso = nullptr, *d_iso_alt = nullptr;   // cache->iso_var (allocated on first use)
  int *d_kz = nullptr, *d_kz_alt = nullptr;        // model-level search hint per particle (allocated on first use)
  double *d_iso_ts = nullptr, *d_iso_ps = nullptr; // balloon time series (ISOSURF 4)
  int iso_n = 0;

  // internal locality order: particles are stored sorted by meteo grid cell;
  // d_ext[i] is the external slot (the reference's ip) of stored particle i
  int *d_ext = nullptr, *d_ext_alt = nullptr;
  bool ext_identity = true;
  int locality_interval = 60;         // re-sort every this many steps (0 = keep the caller's order)
  double *d_rec = nullptr;            // mphip_grid_sums: the quantities as one record per particle (GridVals::rec)
  size_t rec_cap = 0;
  bool grid_records = true;
  int chain_blocks = 0;               // workgroups of the chain walk of the ordered sums (0: default; tuning)
  double *h_sums = nullptr;           // page-locked staging of mphip_grid_sums
  size_t h_sums_cap = 0;
  double *d_grid_kernel = nullptr;    // GRID_KERNEL: kz[nk] | kw[nk] (mphip_set_grid_kernel)
  int grid_nk = 0;
  bool locality_zorder = false;       // tiles of the locality key numbered along a Z-order curve instead of row by row
  int locality_tile = 0;              // horizontal tile edge of the locality key (columns); 0 = 4, or 8 with model-level winds
  int step_blocks = 8192;             // upper bound of the step kernel's grid
  int step_blocks_multi = 32768;       // ... of a multi-step launch (mphip_run_timesteps)
  int xcd_map = 1;
  int big_grid = 0;                   // option "big_grid": take the 64-bit-offset (kBigGrid) instantiations on a grid that fits 32 bits too (tests)
  int perm_records = 1;               // random permutations of the particle arrays through records (permute_random)
  void *d_prec = nullptr;             // ... their buffer
  size_t prec_cap = 0;
  int multi_step = 64;                // mphip_run_timesteps: most time steps per launch (0 = always one by one)
  bool force_generic = false;
  bool compact_depo = true;           // deposition-only launches through depo_kernel (0: the fused kernel's tail)
  int sort_bits = 0;                  // digit width of the radix sort (0 = fewest passes; 8, 9, 10: tuning / tests)
  int steps_since_resort = 1 << 30;

  // sort
  uint32_t *d_keys[2] = {};
  int *d_vals[2] = {};
  uint32_t *d_counts = nullptr;
  size_t counts_cap = 0;
  int sorted_buf = -1;                // which d_keys/d_vals pair holds the last result
  // module_sort as a repair of the previous order (repair_* kernels): valid while the particles are stored in the order
  // of the last module_sort (stored_is_sorted) -- its sorted keys are then non-decreasing along the slots
  // option "lds_tile": cells of the LDS wind tile of traj_tile_kernel (0: off); runs of pure trajectory steps only
  int lds_tile = 0;
  bool emit_keys = true;              // option "emit_keys": the launch that moves the particles writes the keys of the sort ahead
  unsigned char *d_depo_busy = nullptr;   // EmitKeys::depo_busy (np bytes) ...
  long long depo_busy_cap = 0;
  bool depo_busy_valid = false;       // ... written by the last launch and not yet used by the deposition launch of its step
  bool sort_repair = true;            // option "sort_repair"
  bool ahead_priority = false;        // option "ahead_priority": the stream of the sort ahead at the highest priority (C5: no difference, profiles/r05_variants.txt item 3)
  bool stored_is_sorted = false;
  long long sorted_n = 0;
  uint32_t *rep_sk = nullptr, *rep_mk[2] = {}, *rep_fk = nullptr, *rep_tiles = nullptr, *rep_nm = nullptr;
  int *rep_si = nullptr, *rep_mi[2] = {}, *rep_fv = nullptr;
  long long rep_cap = 0;
  // module_sort of the NEXT time step, started beside the rest of this one (option "sort_ahead"): once the launch
  // that moves the particles has run, their positions are final for this step -- module_mixing and the deposition
  // modules only change quantities -- so the keys and the radix sort of the next step's module_sort (and its
  // module_timesteps) can run on a second stream, into buffers of their own, while the main stream does the
  // rest of the step.  The next mphip_run_timestep takes the result if it is called with the expected time and
  // nothing touched the particles, the grids or the control parameters in between; otherwise it is dropped.
  bool sort_ahead = true;
  bool ahead_box = true;              // the key kernel of the sort ahead also writes module_mixing's box index (0: own kernel; 2.26 vs 2.29 ms on C5)
  hipStream_t ahead_stream = nullptr;
  hipEvent_t ahead_mark = nullptr, ahead_done = nullptr;
  uint32_t *ahead_keys[2] = {};
  int *ahead_vals[2] = {};
  uint32_t *ahead_counts = nullptr;
  size_t ahead_counts_cap = 0;
  double *ahead_dt = nullptr;
  long long ahead_cap = 0;
  bool ahead_valid = false;
  double ahead_t = 0;
  int ahead_cur = 0;

  // mixing / grid sums
  int *d_cell = nullptr;
  double *d_sums = nullptr;
  size_t sums_cap = 0;

  int *d_cnt = nullptr;               // particles per mixing cell (32-bit: the reference's `int count[]`)
  // exchange of the occupied levels only (exchange_occupied_levels): per-level occupancy, the dense band, what was found
  // option "mix_exchange_levels" (default 0): measured with a one-rank communicator, the band costs 0.14 ms per step (the
  // host reads the occupancy: the one synchronisation in the step path; pack / unpack) against a MODELLED saving of
  // 0.15-0.2 ms on eight GPUs -- undecidable without xGMI, so the whole-grid all-reduce stays the default
  bool mix_exchange_levels = false;
  double *d_occ = nullptr, *h_occ = nullptr, *d_band = nullptr;
  int *d_band_cnt = nullptr;
  size_t band_cap = 0;
  int mix_band_lo = -1, mix_band_hi = -1;
  size_t cnt_cap = 0;
  int deterministic_sums = 1;         // cell sums in the reference's serial order (0: floating-point atomics)
  int sum_path = 0;                   // ordered sums: 0 = by crowding, 1 = groups of cells per wave, 2 = a lane per (cell, value)
  unsigned long long *d_lists = nullptr;   // work space of the ordered sums (sequence, runs, sort buffers)
  size_t lists_cap = 0;

  mphip_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  void *comm = nullptr;               // RCCL communicator (ncclComm_t) of mphip_comm_init, NULL = single rank / hook
  int comm_ranks = 1, comm_rank = 0;

  // profiling of the fused step kernel
  bool prof = false;
  std::vector<hipEvent_t> ev;         // start/stop pairs
  size_t ev_used = 0;
};

namespace {

int fail(mphip_ctx *ctx, const std::string &msg) {
  if (ctx) {   // (the uploader thread of mphip_prefetch_met reports through here too)
    std::lock_guard<std::mutex> guard(ctx->err_lock);
    ctx->err = msg;
  }
  return 1;
}

#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                   \
  } while (0)

template <typename T>
int dev_alloc(mphip_ctx *ctx, T **p, size_t n) {
  if (*p) {
    HIPCHK(hipFree(*p));
    *p = nullptr;
  }
  if (n) {
    HIPCHK(hipMalloc((void **) p, n * sizeof(T)));
  }
  return 0;
}

void dev_free(void *p) {
  if (p)
    (void) hipFree(p);
}

int grid_for(long long n, int block = 256, int maxb = 8192) {
  long long b = (n + block - 1) / block;
  if (b < 1)
    b = 1;
  if (b > maxb)
    b = maxb;
  return (int) b;
}

DevAtm dev_atm(const mphip_ctx *c) {
  DevAtm a;
  a.time = c->d_arr[0];
  a.p = c->d_arr[1];
  a.lon = c->d_arr[2];
  a.lat = c->d_arr[3];
  for (int iq = 0; iq < MPHIP_NQ_MAX; iq++)
    a.q[iq] = iq < c->nq ? c->d_arr[4 + iq] : nullptr;
  a.up = c->d_uvwp[0];
  a.vp = c->d_uvwp[1];
  a.wp = c->d_uvwp[2];
  a.dt = c->d_dt;
  a.ext = c->ext_identity ? nullptr : c->d_ext;
  a.iso = c->d_iso;
  a.kz = c->d_kz;
  a.iso_ts = c->d_iso_ts;
  a.iso_ps = c->d_iso_ps;
  a.iso_n = c->iso_n;
  a.perm = c->fused_perm;
  a.s_time = c->d_alt[0];
  a.s_p = c->d_alt[1];
  a.s_lon = c->d_alt[2];
  a.s_lat = c->d_alt[3];
  a.nq_perm = c->fused_perm && c->fused_quantities ? c->nq : 0;
  for (int iq = 0; iq < MPHIP_NQ_MAX; iq++)
    a.s_q[iq] = iq < c->nq ? c->d_alt[4 + iq] : nullptr;
  a.np = c->np;
  a.ip0 = c->ip0;
  a.np_total = c->np_total;
  return a;
}

DevMet dev_met(const mphip_ctx *c) {
  DevMet M;
  M.wind = c->pk.wind;
  M.temp = c->pk.temp;
  M.cloud = c->pk.cloud;
  M.mx = c->pk.mx;
  M.mx2 = c->pk.mx2;
  M.sfa = c->pk.sfa;
  M.sfb = c->pk.sfb;
  M.cp2 = c->pk.cp2;
  M.sfc = c->pk.sfc;
  M.sfd = c->pk.sfd;
  M.h2o = c->pk.h2o;
  M.mlw = c->pk.mlw;
  M.zl2 = c->pk.zl2;
  M.pl2 = c->pk.pl2;
  M.ml_monotonic = c->pk.ml_monotonic ? 1 : 0;
  for (int t = 0; t < 2; t++) {
    M.zl[t] = c->slot[t ^ c->flip].f3[MPHIP_ZETAL];
    M.pll[t] = c->slot[t ^ c->flip].f3[MPHIP_PL];
  }
  M.npl = c->nml;
  M.axes = c->d_axes;
  M.lut_base = c->lut_base;
  M.lut_size = c->lut_size;
  M.lat_x0 = c->h_lat[0];
  M.lat_inv_dx = (c->ny - 1) / (c->h_lat[c->ny - 1] - c->h_lat[0]);
  M.nx = c->nx;
  M.ny = c->ny;
  M.np = c->npl;
  M.coord_type = c->coord_type;
  M.time0 = c->slot[0 ^ c->flip].time;
  M.time1 = c->slot[1 ^ c->flip].time;
  M.inv_dtime = 1.0 / (M.time1 - M.time0);
  double latmin = c->h_lat[0], latmax = c->h_lat[0];
  for (double v : c->h_lat) {
    latmin = std::min(latmin, v);
    latmax = std::max(latmax, v);
  }
  M.latmin = latmin;
  M.latmax = latmax;
  M.local = (std::fabs(c->h_lon[c->nx - 1] - c->h_lon[0] - 360.0) >= 0.01);
  // direction test of locate_irr at its first midpoint (mptrac.c:3502-3504)
  const int my = (c->ny - 1) >> 1, mp = (c->npl - 1) >> 1;
  M.lat_ascending = c->h_lat[my] < c->h_lat[my + 1];
  M.p_ascending = c->h_p[mp] < c->h_p[mp + 1];
  M.lon_first = c->h_lon[0];
  M.lon_last = c->h_lon[c->nx - 1];
  M.inv_dlon0 = 1.0 / (c->h_lon[1] - c->h_lon[0]);
  M.lat_search_max = std::nextafter(latmax, -HUGE_VAL);
  M.p_min = *std::min_element(c->h_p.begin(), c->h_p.end());
  M.p_search_max = std::nextafter(*std::max_element(c->h_p.begin(), c->h_p.end()), -HUGE_VAL);
  M.p_cmp_off = M.p_ascending ? 1 : 0;
  M.p_step = M.p_ascending ? 1 : -1;
  M.ps11[0] = c->slot[0 ^ c->flip].ps11;
  M.ps11[1] = c->slot[1 ^ c->flip].ps11;
  // a little below the smallest value of either snapshot: what lies below that is below every interpolated value
  M.ps_skip = std::min(c->slot[0].ps_min, c->slot[1].ps_min) - 1e-6;
  M.pct_skip = std::min(c->slot[0].pct_min, c->slot[1].pct_min) - 1e-6;
  // module_diff_turb / module_convection: bounds below which the boundary layer / the convective column cannot
  // reach, from the extremes of ps and pbl over both snapshots (p1 = pbl - trans (ps - pbl) is linear in both:
  // its smallest value is at a corner of the box of extremes) and the smallest pel
  {
    const double ps_lo = std::min(c->slot[0].ps_min, c->slot[1].ps_min), ps_hi = std::max(c->slot[0].ps_max, c->slot[1].ps_max);
    const double pbl_lo = std::min(c->slot[0].pbl_min, c->slot[1].pbl_min),
                 pbl_hi = std::max(c->slot[0].pbl_max, c->slot[1].pbl_max);
    auto p1_min = [&](double trans) {
      double lo = HUGE_VAL;
      for (double pbl : { pbl_lo, pbl_hi })
        for (double ps : { ps_lo, ps_hi })
          lo = std::min(lo, pbl - trans * (ps - pbl));
      return lo;
    };
    const bool known = std::isfinite(ps_lo) && std::isfinite(ps_hi) && std::isfinite(pbl_lo) && std::isfinite(pbl_hi);
    M.turb_skip = -HUGE_VAL;
    M.conv_skip = -HUGE_VAL;
    if (known && c->have_ctl) {
      const double b = std::min({ ps_lo, pbl_lo, p1_min(c->ctl.turb_pbl_trans) });
      M.turb_skip = b - 1e-6 * std::fabs(b) - 1e-6;
      double lo = ps_lo;
      if (c->ctl.conv_mix_pbl)
        lo = std::min(lo, p1_min(c->ctl.conv_pbl_trans));
      if (c->ctl.conv_cape >= 0)
        lo = std::min(lo, std::min(c->slot[0].pel_min, c->slot[1].pel_min));
      if (lo == lo)
        M.conv_skip = lo - 1e-6 * std::fabs(lo) - 1e-6;
    }
  }
  M.meso_dt = std::fabs(c->ctl.dt_mod);
  M.meso_r = 1 - 2 * M.meso_dt / c->ctl.dt_met;
  M.meso_r2 = std::sqrt(1 - M.meso_r * M.meso_r);
  if (!c->have_ctl || !(M.meso_dt > 0) || !std::isfinite(M.meso_r2))
    M.meso_dt = -1;    // never equal to |dt|: the kernels compute the coefficients themselves
  return M;
}

size_t axes_lds_bytes(const mphip_ctx *c) {
  return c->axes_bytes;
}

// the reference's bisection (mptrac.c:3495-3521), host copy for the look-up table
int host_locate_irr(const std::vector<double> &xx, double x) {
  const int n = (int) xx.size();
  int lo = 0, hi = n - 1;
  int mid = (hi + lo) >> 1;
  if (xx[mid] < xx[mid + 1]) {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] > x)
        hi = mid;
      else
        lo = mid;
    }
  } else {
    while (hi > lo + 1) {
      mid = (hi + lo) >> 1;
      if (xx[mid] <= x)
        hi = mid;
      else
        lo = mid;
    }
  }
  return lo;
}

// axes, reciprocal interval widths and the pressure look-up table in one blob
int upload_axes(mphip_ctx *ctx) {
  const int nx = ctx->nx, ny = ctx->ny, np = ctx->npl;
  const size_t nd = 2 * (size_t) (nx + ny + np);
  std::vector<int16_t> lut;
  ctx->lut_base = ctx->lut_size = 0;
  const double pmin = *std::min_element(ctx->h_p.begin(), ctx->h_p.end());
  const double pmax = *std::max_element(ctx->h_p.begin(), ctx->h_p.end());
  if (pmin > 0 && std::isfinite(pmax) && np < 32000) {
    long long kmin, kmax;
    memcpy(&kmin, &pmin, 8);
    memcpy(&kmax, &pmax, 8);
    kmin >>= 45;
    kmax >>= 45;
    if (kmax - kmin + 1 <= 8192) {
      ctx->lut_base = (int) kmin;
      ctx->lut_size = (int) (kmax - kmin + 1);
      lut.resize((size_t) ctx->lut_size);
      for (int j = 0; j < ctx->lut_size; j++) {
        const long long bits = (kmin + j) << 45;
        double edge;
        memcpy(&edge, &bits, 8);
        lut[(size_t) j] = (int16_t) host_locate_irr(ctx->h_p, edge);
      }
    }
  }
  const size_t bytes = ((nd * sizeof(double) + lut.size() * sizeof(int16_t)) + 15) & ~(size_t) 15;
  std::vector<double> blob(bytes / sizeof(double), 0.0);
  double *b = blob.data();
  std::copy(ctx->h_lon.begin(), ctx->h_lon.end(), b);
  std::copy(ctx->h_lat.begin(), ctx->h_lat.end(), b + nx);
  std::copy(ctx->h_p.begin(), ctx->h_p.end(), b + nx + ny);
  double *inv = b + nx + ny + np;
  for (int i = 0; i + 1 < nx; i++)
    inv[i] = 1.0 / (ctx->h_lon[i + 1] - ctx->h_lon[i]);
  for (int i = 0; i + 1 < ny; i++)
    inv[nx + i] = 1.0 / (ctx->h_lat[i + 1] - ctx->h_lat[i]);
  for (int i = 0; i + 1 < np; i++)
    inv[nx + ny + i] = 1.0 / (ctx->h_p[i + 1] - ctx->h_p[i]);
  if (!lut.empty())
    memcpy(b + nd, lut.data(), lut.size() * sizeof(int16_t));
  if (dev_alloc(ctx, &ctx->d_axes, blob.size()))
    return 1;
  HIPCHK(hipMemcpyAsync(ctx->d_axes, blob.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->axes_bytes = bytes;
  return 0;
}

// arrays of a packed set for the fields the two snapshots carry (allocated on first need; hipMalloc, so
// called from the stepping thread only)
struct PackPlan {
  bool cloud = false, ml = false, pbl = false, mx = false, mx2 = false, ml_heights = false, coord2 = false;
};

PackPlan pack_plan(const mphip_ctx *ctx, const MetSlot &s0, const MetSlot &s1) {
  PackPlan P;
  const MetSlot *ss[2] = { &s0, &s1 };
  for (int t = 0; t < 2; t++) {
    for (int f = MPHIP_LWC; f <= MPHIP_SWC; f++)
      P.cloud = P.cloud || ss[t]->has3[f];
    for (int f = MPHIP_Z; f <= MPHIP_CC; f++)
      P.mx = P.mx || ss[t]->has3[f];
    for (int f = MPHIP_TS; f <= MPHIP_O3C; f++)
      P.mx2 = P.mx2 || ss[t]->has2[f];
    P.ml = P.ml || ss[t]->has3[MPHIP_UL] || ss[t]->has3[MPHIP_VL] || ss[t]->has3[MPHIP_ZETA_DOTL] || ss[t]->has3[MPHIP_WL];
    P.pbl = P.pbl || ss[t]->has3[MPHIP_H2O] || ss[t]->has2[MPHIP_ESS] || ss[t]->has2[MPHIP_NSS] || ss[t]->has2[MPHIP_SHF];
  }
  P.ml = P.ml && (size_t) ctx->nx * ctx->ny * ctx->nml > 0;
  // packed height pairs: {zetal}, {pl}; ADVECT_VERT_COORD 2 searches in pl only
  P.coord2 = ctx->have_ctl && ctx->ctl.advect_vert_coord == 2;
  P.ml_heights = P.ml && s0.has3[MPHIP_PL] && s1.has3[MPHIP_PL]
    && (P.coord2 || (s0.has3[MPHIP_ZETAL] && s1.has3[MPHIP_ZETAL]));
  return P;
}

int pack_alloc(mphip_ctx *ctx, mphip_ctx::PackedGrids &K, const PackPlan &P) {
  const size_t ncell = (size_t) ctx->nx * ctx->ny * ctx->npl, ncol = (size_t) ctx->nx * ctx->ny;
  const size_t ncell_ml = (size_t) ctx->nx * ctx->ny * ctx->nml;
  if (P.ml && !K.mlw
      && (dev_alloc(ctx, &K.mlw, 6 * ncell_ml) || dev_alloc(ctx, &K.zl2, 2 * ncell_ml) || dev_alloc(ctx, &K.pl2, 2 * ncell_ml)))
    return 1;
  if (!K.ml_mono && dev_alloc(ctx, &K.ml_mono, 1))
    return 1;
  if (!K.wind && (dev_alloc(ctx, &K.wind, 6 * ncell) || dev_alloc(ctx, &K.temp, 2 * ncell)))
    return 1;
  if (!K.cp2 && dev_alloc(ctx, &K.cp2, ncol))
    return 1;
  if (!K.sfa && (dev_alloc(ctx, &K.sfa, ncol) || dev_alloc(ctx, &K.sfb, 2 * ncol) || dev_alloc(ctx, &K.sfc, 2 * ncol)))
    return 1;
  if (P.cloud && !K.cloud && dev_alloc(ctx, &K.cloud, 2 * ncell))
    return 1;
  if (P.mx && !K.mx && dev_alloc(ctx, &K.mx, 2 * ncell))
    return 1;
  if (P.mx2 && !K.mx2 && dev_alloc(ctx, &K.mx2, 7 * ncol))
    return 1;
  if (P.pbl && !K.sfd && (dev_alloc(ctx, &K.sfd, 2 * ncol) || dev_alloc(ctx, &K.h2o, 2 * ncell)))
    return 1;
  return 0;
}

// build the packed set K from the staging copies of two snapshots on `stream` (arrays allocated: pack_alloc).
// No hipMalloc, no use of ctx->err: also called by the uploader thread.  *mono_pending: the monotonicity flag of
// the height columns has to be read back after the kernel (pack_finish).
int pack_launch(mphip_ctx *ctx, mphip_ctx::PackedGrids &K, const PackPlan &P, const MetSlot &s0, const MetSlot &s1,
                hipStream_t stream) {
  const size_t ncell = (size_t) ctx->nx * ctx->ny * ctx->npl, ncol = (size_t) ctx->nx * ctx->ny;
  PackArgs a;
  const MetSlot *ss[2] = { &s0, &s1 };
  for (int t = 0; t < 2; t++) {
    for (int f = 0; f < MPHIP_N3D; f++)
      a.f3[t][f] = ss[t]->has3[f] ? ss[t]->f3[f] : nullptr;
    for (int f = 0; f < MPHIP_N2D; f++)
      a.f2[t][f] = ss[t]->has2[f] ? ss[t]->f2[f] : nullptr;
  }
  if (P.ml_heights) {
    const int one = 1;
    if (hipMemcpyAsync(K.ml_mono, &one, sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess)
      return 1;
  }
  a.wind = K.wind;
  a.temp = K.temp;
  a.cloud = P.cloud ? K.cloud : nullptr;
  a.mx = P.mx ? K.mx : nullptr;
  a.mx2 = P.mx2 ? K.mx2 : nullptr;
  a.sfa = K.sfa;
  a.sfb = K.sfb;
  a.cp2 = K.cp2;
  a.sfc = K.sfc;
  a.sfd = P.pbl ? K.sfd : nullptr;
  a.h2o = P.pbl ? K.h2o : nullptr;
  a.mlw = P.ml ? K.mlw : nullptr;
  a.mlw_third = P.coord2 ? MPHIP_WL : MPHIP_ZETA_DOTL;
  a.zl2 = P.ml_heights ? K.zl2 : nullptr;
  a.pl2 = P.ml_heights ? K.pl2 : nullptr;
  a.ml_mono = K.ml_mono;
  a.nml = ctx->nml;
  a.ncell = ncell;
  a.ncol = ncol;
  a.ncell_ml = (size_t) ctx->nx * ctx->ny * ctx->nml;
  hipLaunchKernelGGL(pack_kernel, dim3(grid_for((long long) ncell)), dim3(256), 0, stream, a);
  if (hipGetLastError() != hipSuccess)
    return 1;
  K.ml_monotonic = false;
  if (P.ml_heights) {   // (model levels only) the fast kernel relies on monotonic height columns
    int mono = 0;
    if (hipMemcpyAsync(&mono, K.ml_mono, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess
        || hipStreamSynchronize(stream) != hipSuccess)
      return 1;
    K.ml_monotonic = mono != 0;
  }
  return 0;
}

// the device axes follow the current met0: after a hand-over (mphip_swap_met / mphip_commit_met) the new met0 is
// the old met1, and the reference accepts axes that differ by up to 1e-3 between files (mptrac.c:6543-6556)
int axes_follow_met0(mphip_ctx *ctx) {
  const MetSlot &s0 = ctx->slot[0 ^ ctx->flip];
  if (!s0.lon.empty() && (s0.lon != ctx->h_lon || s0.lat != ctx->h_lat || s0.p != ctx->h_p)) {
    ctx->h_lon = s0.lon;
    ctx->h_lat = s0.lat;
    ctx->h_p = s0.p;
    if (upload_axes(ctx))
      return 1;
  }
  return 0;
}

// (re)build the packed two-snapshot grids from the per-slot staging copies
int ensure_packed(mphip_ctx *ctx) {
  if (!ctx->packed_dirty)
    return 0;
  const MetSlot &s0 = ctx->slot[0 ^ ctx->flip], &s1 = ctx->slot[1 ^ ctx->flip];
  if (!s0.valid || !s1.valid)
    return fail(ctx, "meteo data for both met0 and met1 must be uploaded before stepping");
  if (axes_follow_met0(ctx))
    return 1;
  const PackPlan P = pack_plan(ctx, s0, s1);
  if (pack_alloc(ctx, ctx->pk, P))
    return 1;
  if (pack_launch(ctx, ctx->pk, P, s0, s1, ctx->stream))
    return fail(ctx, "packing the meteo grids failed");
  ctx->packed_dirty = false;
  return 0;
}

bool both_have3(const mphip_ctx *c, int f) {
  return c->slot[0].has3[f] && c->slot[1].has3[f];
}

bool both_have2(const mphip_ctx *c, int f) {
  return c->slot[0].has2[f] && c->slot[1].has2[f];
}

// cache->iso_var lives on the device only while an isosurface mode needs it
int ensure_iso(mphip_ctx *ctx) {
  if (ctx->d_iso)
    return 0;
  const size_t n = (size_t) std::max<long long>(ctx->np, 1);
  if (dev_alloc(ctx, &ctx->d_iso, n) || dev_alloc(ctx, &ctx->d_iso_alt, n))
    return 1;
  HIPCHK(hipMemsetAsync(ctx->d_iso, 0, n * sizeof(double), ctx->stream));   // calloc'ed cache_t
  return 0;
}

// every field a module mask reads must have been uploaded for both snapshots
int check_fields(mphip_ctx *ctx, unsigned mask) {
  const mphip_ctl_t &c = ctx->ctl;
  auto need3 = [&](int f, const char *who) {
    return both_have3(ctx, f) ? 0 : fail(ctx, std::string(who) + ": a required 3-D meteo field was not uploaded");
  };
  auto need2 = [&](int f, const char *who) {
    return both_have2(ctx, f) ? 0 : fail(ctx, std::string(who) + ": a required 2-D meteo field was not uploaded");
  };
  if (mask & (MPHIP_MOD_POSITION | MPHIP_MOD_POSITION2))
    if (need2(MPHIP_PS, "module_position"))
      return 1;
  const bool model_levels = (c.advect_vert_coord == 1 || c.advect_vert_coord == 3);
  if ((mask & MPHIP_MOD_ADVECT) && c.advect_vert_coord == 2)
    if (need3(MPHIP_PL, "module_advect") || need3(MPHIP_UL, "module_advect") || need3(MPHIP_VL, "module_advect")
        || need3(MPHIP_WL, "module_advect") || ctx->nml < 2)
      return 1;
  if ((mask & MPHIP_MOD_DIFF_MESO) || ((mask & MPHIP_MOD_ADVECT) && c.advect_vert_coord == 0))
    if (need3(MPHIP_U, "module_advect") || need3(MPHIP_V, "module_advect") || need3(MPHIP_W, "module_advect"))
      return 1;
  if (((mask & MPHIP_MOD_ADVECT) && model_levels) || (mask & MPHIP_MOD_ADVECT_INIT)) {
    if (need3(MPHIP_PL, "module_advect") || need3(MPHIP_ZETAL, "module_advect")
        || need3(MPHIP_UL, "module_advect") || need3(MPHIP_VL, "module_advect")
        || need3(MPHIP_ZETA_DOTL, "module_advect") || ctx->nml < 2)
      return 1;
    if ((c.advect_vert_coord == 3 ? c.qnt_eta : c.qnt_zeta) < 0)
      return fail(ctx, "model-level advection needs quantity zeta (ADVECT_VERT_COORD 1) or eta (3)");
  }
  if (mask & MPHIP_MOD_DIFF_TURB)
    if (need2(MPHIP_PS, "module_diff_turb") || need2(MPHIP_PBL, "module_diff_turb"))
      return 1;
  if (mask & MPHIP_MOD_DIFF_PBL)
    if (need2(MPHIP_PS, "module_diff_pbl") || need2(MPHIP_PBL, "module_diff_pbl")
        || need2(MPHIP_ESS, "module_diff_pbl") || need2(MPHIP_NSS, "module_diff_pbl")
        || need2(MPHIP_SHF, "module_diff_pbl") || need3(MPHIP_T, "module_diff_pbl")
        || need3(MPHIP_H2O, "module_diff_pbl"))
      return 1;
  if (mask & MPHIP_MOD_CONVECTION) {
    if (need2(MPHIP_PS, "module_convection") || need3(MPHIP_T, "module_convection"))
      return 1;
    if (c.conv_mix_pbl && need2(MPHIP_PBL, "module_convection"))
      return 1;
    if (c.conv_cape >= 0
        && (need2(MPHIP_CAPE, "module_convection") || need2(MPHIP_CIN, "module_convection")
            || need2(MPHIP_PEL, "module_convection")))
      return 1;
  }
  if (mask & MPHIP_MOD_SEDI) {
    if (need3(MPHIP_T, "module_sedi"))
      return 1;
    if (c.qnt_rp < 0 || c.qnt_rhop < 0)
      return fail(ctx, "module_sedi needs quantities rp and rhop");
  }
  if (mask & (MPHIP_MOD_DECAY | MPHIP_MOD_WET_DEPO | MPHIP_MOD_DRY_DEPO))
    if (c.qnt_m < 0 && c.qnt_vmr < 0)
      return fail(ctx, "Module needs quantity mass or volume mixing ratio!");
  if (mask & MPHIP_MOD_WET_DEPO) {
    if (need2(MPHIP_PCT, "module_wet_depo") || need2(MPHIP_PCB, "module_wet_depo") || need2(MPHIP_CL, "module_wet_depo")
        || need3(MPHIP_T, "module_wet_depo") || need3(MPHIP_LWC, "module_wet_depo")
        || need3(MPHIP_RWC, "module_wet_depo") || need3(MPHIP_IWC, "module_wet_depo")
        || need3(MPHIP_SWC, "module_wet_depo"))
      return fail(ctx, "module_wet_depo: cloud fields (pct, pcb, cl, lwc, rwc, iwc, swc, t) were not uploaded");
  }
  if (mask & MPHIP_MOD_DRY_DEPO)
    if (need2(MPHIP_PS, "module_dry_depo"))
      return 1;
  if (mask & (MPHIP_MOD_ISOSURF | MPHIP_MOD_ISOSURF_INIT)) {
    if (c.isosurf < 1 || c.isosurf > 4)
      return fail(ctx, "module_isosurf: ISOSURF must be 1 ... 4");
    if ((c.isosurf == 2 || c.isosurf == 3) && need3(MPHIP_T, "module_isosurf"))
      return 1;
    if (c.isosurf == 4 && (mask & MPHIP_MOD_ISOSURF) && ctx->iso_n < 1)
      return fail(ctx, "module_isosurf: the balloon pressure time series was not uploaded");
    if (c.isosurf <= 3 && ensure_iso(ctx))
      return 1;
  }
  if (mask & (MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2)) {
    if ((c.bound_dps > 0 || c.bound_dzs > 0 || c.bound_zetas > 0 || c.bound_pbl) && need2(MPHIP_PS, "module_bound_cond"))
      return 1;
    if (c.bound_zetas > 0 && need3(MPHIP_T, "module_bound_cond"))
      return 1;
    if (c.bound_pbl && need2(MPHIP_PBL, "module_bound_cond"))
      return 1;
  }
  if ((mask & (MPHIP_MOD_DIFF_TURB | MPHIP_MOD_DECAY)) && !ctx->have_clim)
    return fail(ctx, "climatological tropopause data were not uploaded");
  return 0;
}

// nsteps > 1: that many consecutive time steps in one launch (kMultiStep instantiations; the caller has checked that
// one exists for this module set, multi_step_mask)
// The lean instantiations need a lat/lon grid with the pressure look-up table; cell indices are built with 24-bit
// multiplies (cell32: columns below 2^24, cells below 2^32).  lean32_ok: 32-bit byte offsets into the packed grids
// (the largest record has 24 bytes); lean64_ok: a grid beyond that -- or the option "big_grid" -- takes the kBigGrid
// instantiations with 64-bit offsets.
// With winds from the model levels (ADVECT_VERT_COORD 1..3) the lean kernels index the model-level records -- nml
// levels per column, which need not be the np pressure levels of the same file (met_t::npl vs met_t::np,
// mptrac.h:3862) -- so every size test takes the larger of the two level counts.
static unsigned long long guard_levels(const mphip_ctx *ctx) {
  const bool ml = ctx->ctl.advect_vert_coord >= 1 && ctx->ctl.advect_vert_coord <= 3;
  return (unsigned long long) (ml ? std::max(ctx->npl, ctx->nml) : ctx->npl);
}
static bool lean_grid(const mphip_ctx *ctx) {
  const unsigned long long cols = (unsigned long long) ctx->nx * ctx->ny;
  return ctx->coord_type == 0 && ctx->lut_size > 0 && cols < (1ull << 24)
    && cols * guard_levels(ctx) < (1ull << 32);
}
static bool fits32(const mphip_ctx *ctx) {
  return (unsigned long long) ctx->nx * ctx->ny * guard_levels(ctx) * 24ull < (1ull << 32);
}
static bool lean32_ok(const mphip_ctx *ctx) { return lean_grid(ctx) && fits32(ctx) && !ctx->big_grid; }
static bool lean64_ok(const mphip_ctx *ctx) { return lean_grid(ctx) && (!fits32(ctx) || ctx->big_grid); }

int launch_step(mphip_ctx *ctx, unsigned mask, double t, uint64_t ctr_turb, uint64_t ctr_meso, uint64_t ctr_conv,
                uint64_t ctr_pbl = 0, int nsteps = 1, double t_stride = 0, uint64_t ctr_stride = 0,
                const EmitKeys *emit = nullptr, bool *emitted = nullptr) {
  if (ctx->np == 0)
    return 0;
  if (ensure_packed(ctx) || check_fields(ctx, mask))
    return 1;
  if (ctx->ctl.advect_vert_coord >= 1 && ctx->ctl.advect_vert_coord <= 3 && !ctx->d_kz) {
    const size_t n = (size_t) std::max<long long>(ctx->np, 1);
    if (dev_alloc(ctx, &ctx->d_kz, n) || dev_alloc(ctx, &ctx->d_kz_alt, n))
      return 1;
    std::vector<int> mid(n, std::max(ctx->nml / 2, 0));     // any level: the hint only shortens the search
    HIPCHK(hipMemcpyAsync(ctx->d_kz, mid.data(), n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  StepParams S;
  memset(&S.emit, 0, sizeof(S.emit));
  S.depo_busy = nullptr;
  if (emitted)
    *emitted = false;
  S.ctl = ctx->ctl;
  S.met = dev_met(ctx);
  S.atm = dev_atm(ctx);
  S.clim = ctx->d_clim;
  S.tracers = ctx->d_tracers;
  S.t = t;
  S.mask = mask;
  // every block walks a whole number of 256-particle rounds (no half-empty last round)
  // (multi-step launches: more, shorter blocks -- a thread that takes 20 steps per particle should not also walk
  // five particles; measured on C3, alternating in one process: 0.744-0.781 / 0.738-0.757 / 0.729-0.747 ms per step
  // with 8 192 / 16 384 / 32 768 logical blocks)
  const int blocks = nsteps > 1 ? ctx->step_blocks_multi : ctx->step_blocks;
  long long per_block = (ctx->np + blocks - 1) / blocks;
  per_block = std::max<long long>(256, (per_block + 255) / 256 * 256);
  int nb = (int) ((ctx->np + per_block - 1) / per_block);
  nb = (nb + 7) & ~7;
  S.nblocks_logical = nb;
  S.per_block = per_block;
  S.xcd_map = ctx->xcd_map;
  S.ctr_turb = ctr_turb;
  S.ctr_meso = ctr_meso;
  S.ctr_conv = ctr_conv;
  S.ctr_pbl = ctr_pbl;
  S.nsteps = nsteps;
  S.t_stride = t_stride;
  S.ctr_stride = ctr_stride;
  // axes, climatological tropopause, the tables of log and exp (+ pow's for the instantiations with the closure, below)
  size_t lds = axes_lds_bytes(ctx) + sizeof(DevClim) + (size_t) kLibmLogExpDoubles * sizeof(double);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->prof) {
    if (ctx->ev_used + 2 > ctx->ev.size()) {
      for (int k = 0; k < 2; k++) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ctx->ev.push_back(e);
      }
    }
    e0 = ctx->ev[ctx->ev_used++];
    e1 = ctx->ev[ctx->ev_used++];
    HIPCHK(hipEventRecord(e0, ctx->stream));
  }
  const bool ml_ = ctx->ctl.advect_vert_coord >= 1 && ctx->ctl.advect_vert_coord <= 3;   // winds from the model levels
  // model levels: the fast path needs monotonic height columns and none of the rarely used modules
  const bool ml_fast = ml_ && ctx->pk.ml_monotonic && ctx->nml <= kLockstepMaxLevels
    && !(mask & (kRareModules & ~MPHIP_MOD_ADVECT_INIT)) && !ctx->force_generic;
  // ... module_bound_cond is no obstacle where the gated lean model-level instantiation can run (it switches the module
  // at run time, as every gated instantiation does)
  constexpr unsigned kBoundBits = MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2;
  const bool ml_fast_bound = ml_ && ctx->pk.ml_monotonic && ctx->nml <= kLockstepMaxLevels
    && !(mask & (kRareModules & ~MPHIP_MOD_ADVECT_INIT & ~kBoundBits)) && !ctx->force_generic;
  const unsigned rare_bits = mask & kRareModules & ~(ml_fast ? MPHIP_MOD_ADVECT_INIT : 0u);
  const bool rare = (ml_ && !ml_fast) || rare_bits;
  // the specialised instantiations take module_timesteps / the dt store from the run-time mask
  // ... and run the lean code: lat/lon grid with a pressure look-up table
  const bool lean_ok = lean32_ok(ctx), big_ok = lean64_ok(ctx);
  // (the lean instantiations are keyed on the movers; loss / decay / deposition are run-time bits in all of them)
  // (ADVECT 2 and 1 -- midpoint, the reference's default, and Euler -- share the two-stage instantiations)
  const unsigned scheme = (mask & MPHIP_MOD_ADVECT) && ctx->ctl.advect != 4 ? kTwoStage : 0u;
  // An exact module set has its own lean instantiation.  Any other subset of {turbulent, mesoscale diffusion,
  // convection, sedimentation} on top of the time step's movers, and every set with module_bound_cond, runs the
  // largest one with those switched at run time (kGated; module_bound_cond also in the instantiation without
  // movers).  Everything else (single-module calls, the other rarely used modules) takes a general instantiation.
  constexpr unsigned kBound = MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2;
  unsigned sel = kMaskGeneric;
  if (!(rare_bits & ~kBound) && (ml_fast || (ml_fast_bound && (mask & kBound))) && !ctx->force_generic && (lean_ok || big_ok)) {
    // model-level winds: the headline module set has lean instantiations (its subsets the gated one); the rest
    // stays with the general model-level kernels below
    const unsigned req = (mask | MPHIP_MOD_TIMESTEPS) & ~(kStoreDt | kTailModules | kBound);
    if (req == kAdvDiffConvSedi && !(mask & kBound) && lean_ok)
      sel = req | kMLWinds | (nsteps > 1 ? kMultiStep : 0u);
    else if ((req & ~kOptionalModules) == kAdv)   // (subsets, every set with module_bound_cond, and every set of a big grid)
      sel = kAdvDiffConvSedi | kGated | kMLWinds | (big_ok ? kBigGrid : 0u) | (nsteps > 1 ? kMultiStep : 0u);
  } else if (!(rare_bits & ~(kBound | MPHIP_MOD_DIFF_PBL | MPHIP_MOD_ISOSURF)) && (mask & (MPHIP_MOD_DIFF_PBL | MPHIP_MOD_ISOSURF)) && !ml_
             && !ctx->force_generic && lean_ok) {
    // the closure inside the boundary layer (TURB_PBL_SCHEME 1) and module_isosurf: gated instantiations of their own
    const unsigned req = (mask | MPHIP_MOD_TIMESTEPS) & ~(kStoreDt | kTailModules | kBound | MPHIP_MOD_DIFF_PBL | MPHIP_MOD_ISOSURF);
    if ((req & ~kOptionalModules) == kAdv)
      sel = kAdvDiffConvSedi | kGated | kPblClosure | scheme | (nsteps > 1 ? kMultiStep : 0u);
  } else if (!(rare_bits & ~kBound) && !ml_ && !ctx->force_generic && lean_ok) {
    const unsigned req = (mask | MPHIP_MOD_TIMESTEPS) & ~(kStoreDt | kTailModules | kBound);
    // (kAdvTurbConvSedi has a kernel of its own for single steps only: several steps per launch take the gated one)
    const bool exact = req == kAdv || req == kAdvTurb || req == kAdvDiff || req == kAdvDiffConvSedi
      || (req == kAdvTurbConvSedi && nsteps == 1);
    if (req == kTailOnly)
      sel = kTailOnly;
    else if (req == kDiffConvSediOnly && !(mask & kBound))
      sel = kDiffConvSediOnly;
    else if (exact && !(mask & kBound))
      sel = req | scheme | (nsteps > 1 ? kMultiStep : 0u);
    else if ((req & ~kOptionalModules) == kAdv)
      sel = kAdvDiffConvSedi | kGated | scheme | (nsteps > 1 ? kMultiStep : 0u);
  }
  if (sel == kMaskGeneric && !(rare_bits & ~kBound) && !ml_ && !ctx->force_generic && big_ok) {
    // a grid beyond 32-bit offsets: the gated instantiation with 64-bit ones serves every set of the time step's movers
    const unsigned req = (mask | MPHIP_MOD_TIMESTEPS) & ~(kStoreDt | kTailModules | kBound);
    if ((req & ~kOptionalModules) == kAdv)
      sel = kAdvDiffConvSedi | kGated | kBigGrid | scheme | (nsteps > 1 ? kMultiStep : 0u);
  }
  // module_wet_depo / module_dry_depo alone (the launch behind module_mixing): the kernel that packs the few
  // particles with anything to do into full waves
  constexpr unsigned kDepo = MPHIP_MOD_WET_DEPO | MPHIP_MOD_DRY_DEPO;
  const size_t depo_lds = ((axes_lds_bytes(ctx) + 15) & ~(size_t) 15) + (size_t) per_block * sizeof(int);
  if (sel == kTailOnly && (mask & kDepo) && !(mask & ~kDepo) && ctx->compact_depo && !ctx->fused_perm
      && depo_lds <= 64 * 1024) {
    S.depo_busy = ctx->depo_busy_valid ? ctx->d_depo_busy : nullptr;
    ctx->depo_busy_valid = false;
    hipLaunchKernelGGL(depo_kernel, dim3(nb), dim3(256), depo_lds, ctx->stream, S);
    HIPCHK(hipGetLastError());
    ctx->fused_perm = nullptr;
    if (ctx->prof)
      HIPCHK(hipEventRecord(e1, ctx->stream));
    return 0;
  }
  // the headline module set, one step per launch, and the caller wants the keys of the sort ahead from this launch
  ctx->depo_busy_valid = false;     // (any other launch: positions may move, the flags of an earlier one are void)
  if (emit && emit->keys && sel == kAdvDiffConvSedi && nsteps == 1) {
    sel |= kEmitKeys;
    S.emit = *emit;
    if (emitted)
      *emitted = true;
    ctx->depo_busy_valid = emit->depo_busy != nullptr;
  }
  // pure trajectories, several steps per launch: the kernel that stages the wind grid through an LDS tile (option lds_tile)
  if (ctx->lds_tile > 0 && nsteps > 1 && (sel == (kAdv | kMultiStep) || sel == (kAdv | kTwoStage | kMultiStep))
      && !(mask & (kTailModules | kBound)) && !ctx->fused_perm) {
    const size_t tile_lds = ((axes_lds_bytes(ctx) + 15) & ~(size_t) 15) + (size_t) ctx->lds_tile * 24;
    if (tile_lds <= 64 * 1024) {
      if (sel & kTwoStage)
        hipLaunchKernelGGL(traj_tile_kernel<2>, dim3(nb), dim3(256), tile_lds, ctx->stream, S, ctx->lds_tile);
      else
        hipLaunchKernelGGL(traj_tile_kernel<4>, dim3(nb), dim3(256), tile_lds, ctx->stream, S, ctx->lds_tile);
      HIPCHK(hipGetLastError());
      if (ctx->prof)
        HIPCHK(hipEventRecord(e1, ctx->stream));
      return 0;
    }
  }
  if (sel < kMaskGenericMLMulti && (sel & kPblClosure))
    lds += (size_t) (kLibmDoubles - kLibmLogExpDoubles) * sizeof(double);
  // (-DMPHIP_QUICK: a development build with the five instantiations the headline workloads launch; every other
  // module set then runs a general kernel)
  switch (sel) {
#define STEP_CASE(M)                                                                                  \
  case M:                                                                                             \
    hipLaunchKernelGGL(step_kernel<M>, dim3(nb), dim3(256), lds, ctx->stream, S);                     \
    break;
#ifdef MPHIP_QUICK
#define STEP_CASE_REST(M)
#else
#define STEP_CASE_REST(M) STEP_CASE(M)
#endif
    STEP_CASE_REST(kAdv)
    STEP_CASE_REST(kAdvTurb)
    STEP_CASE_REST(kAdvDiff)
    STEP_CASE_REST(kAdvTurbConvSedi)
    STEP_CASE(kAdvDiffConvSedi)
    STEP_CASE_REST(kAdv | kTwoStage)
    STEP_CASE_REST(kAdvTurb | kTwoStage)
    STEP_CASE_REST(kAdvDiff | kTwoStage)
    STEP_CASE_REST(kAdvTurbConvSedi | kTwoStage)
    STEP_CASE_REST(kAdvDiffConvSedi | kTwoStage)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kTwoStage)
    STEP_CASE_REST(kTailOnly)
    STEP_CASE_REST(kDiffConvSediOnly)
    STEP_CASE_REST(kAdv | kMultiStep)
    STEP_CASE_REST(kAdvTurb | kMultiStep)
    STEP_CASE_REST(kAdvDiff | kMultiStep)
    STEP_CASE(kAdvDiffConvSedi | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kMLWinds)
    STEP_CASE(kAdvDiffConvSedi | kMLWinds | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kMLWinds)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kMLWinds | kMultiStep)
    STEP_CASE_REST(kAdv | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvTurb | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvDiff | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kTwoStage | kMultiStep)
    STEP_CASE(kAdvDiffConvSedi | kEmitKeys)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kPblClosure)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kPblClosure | kTwoStage)
    STEP_CASE(kAdvDiffConvSedi | kGated | kPblClosure | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kPblClosure | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid | kTwoStage)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid | kTwoStage | kMultiStep)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid | kMLWinds)
    STEP_CASE_REST(kAdvDiffConvSedi | kGated | kBigGrid | kMLWinds | kMultiStep)
#undef STEP_CASE_REST
#undef STEP_CASE
  default:
    if (nsteps > 1 && !(ml_fast && !rare && !ctx->force_generic))
      return fail(ctx, "internal: no multi-step instantiation for this module set");
    if (rare || ctx->force_generic)
      hipLaunchKernelGGL(step_kernel<kMaskGeneric>, dim3(nb), dim3(256), lds, ctx->stream, S);
    else if (ml_fast && nsteps > 1)
      hipLaunchKernelGGL(step_kernel<kMaskGenericMLMulti>, dim3(nb), dim3(256), lds, ctx->stream, S);
    else if (ml_fast)
      hipLaunchKernelGGL(step_kernel<kMaskGenericML>, dim3(nb), dim3(256), lds, ctx->stream, S);
    else
      hipLaunchKernelGGL(step_kernel<kMaskGenericPL>, dim3(nb), dim3(256), lds, ctx->stream, S);
  }
  HIPCHK(hipGetLastError());
  ctx->fused_perm = nullptr;   // module_sort's gather, if one was pending, has happened in this launch
  if (ctx->prof)
    HIPCHK(hipEventRecord(e1, ctx->stream));
  return 0;
}

// ---- module_meteo -------------------------------------------------------------

// meteo fields each module_meteo quantity is computed from (mptrac.c:5091-5157)
struct MeteoDeps {
  unsigned need3 = 0, need2 = 0;
};

MeteoDeps meteo_deps(const mphip_ctl_t &c) {
  MeteoDeps d;
  auto q = [&](int k) { return c.qnt_met[k] >= 0; };
  auto f3 = [&](int k, int f) { if (q(k)) d.need3 |= 1u << f; };
  auto f2 = [&](int k, int f) { if (q(k)) d.need2 |= 1u << f; };
  f2(MPHIP_MQ_PS, MPHIP_PS);     f2(MPHIP_MQ_TS, MPHIP_TS);     f2(MPHIP_MQ_ZS, MPHIP_ZS);
  f2(MPHIP_MQ_US, MPHIP_US);     f2(MPHIP_MQ_VS, MPHIP_VS);     f2(MPHIP_MQ_ESS, MPHIP_ESS);
  f2(MPHIP_MQ_NSS, MPHIP_NSS);   f2(MPHIP_MQ_SHF, MPHIP_SHF);   f2(MPHIP_MQ_LSM, MPHIP_LSM);
  f2(MPHIP_MQ_SST, MPHIP_SST);   f2(MPHIP_MQ_PBL, MPHIP_PBL);   f2(MPHIP_MQ_PT, MPHIP_PT);
  f2(MPHIP_MQ_TT, MPHIP_TT);     f2(MPHIP_MQ_ZT, MPHIP_ZT);     f2(MPHIP_MQ_H2OT, MPHIP_H2OT);
  f2(MPHIP_MQ_PCT, MPHIP_PCT);   f2(MPHIP_MQ_PCB, MPHIP_PCB);   f2(MPHIP_MQ_CL, MPHIP_CL);
  f2(MPHIP_MQ_PLCL, MPHIP_PLCL); f2(MPHIP_MQ_PLFC, MPHIP_PLFC); f2(MPHIP_MQ_PEL, MPHIP_PEL);
  f2(MPHIP_MQ_CAPE, MPHIP_CAPE); f2(MPHIP_MQ_CIN, MPHIP_CIN);   f2(MPHIP_MQ_O3C, MPHIP_O3C);
  f3(MPHIP_MQ_ZG, MPHIP_Z);      f3(MPHIP_MQ_T, MPHIP_T);       f3(MPHIP_MQ_U, MPHIP_U);
  f3(MPHIP_MQ_V, MPHIP_V);       f3(MPHIP_MQ_W, MPHIP_W);       f3(MPHIP_MQ_H2O, MPHIP_H2O);
  f3(MPHIP_MQ_O3, MPHIP_O3);     f3(MPHIP_MQ_LWC, MPHIP_LWC);   f3(MPHIP_MQ_RWC, MPHIP_RWC);
  f3(MPHIP_MQ_IWC, MPHIP_IWC);   f3(MPHIP_MQ_SWC, MPHIP_SWC);   f3(MPHIP_MQ_CC, MPHIP_CC);
  f3(MPHIP_MQ_PV, MPHIP_PV);
  // derived quantities
  f3(MPHIP_MQ_RHO, MPHIP_T);     f3(MPHIP_MQ_VH, MPHIP_U);      f3(MPHIP_MQ_VH, MPHIP_V);
  f3(MPHIP_MQ_VZ, MPHIP_W);      f3(MPHIP_MQ_PSAT, MPHIP_T);    f3(MPHIP_MQ_PSICE, MPHIP_T);
  f3(MPHIP_MQ_PW, MPHIP_H2O);    f3(MPHIP_MQ_SH, MPHIP_H2O);    f3(MPHIP_MQ_RH, MPHIP_T);
  f3(MPHIP_MQ_RH, MPHIP_H2O);    f3(MPHIP_MQ_RHICE, MPHIP_T);   f3(MPHIP_MQ_RHICE, MPHIP_H2O);
  f3(MPHIP_MQ_THETA, MPHIP_T);   f3(MPHIP_MQ_ZETA_D, MPHIP_T);  f2(MPHIP_MQ_ZETA_D, MPHIP_PS);
  f3(MPHIP_MQ_TVIRT, MPHIP_T);   f3(MPHIP_MQ_TVIRT, MPHIP_H2O); f3(MPHIP_MQ_LAPSE, MPHIP_T);
  f3(MPHIP_MQ_LAPSE, MPHIP_H2O); f3(MPHIP_MQ_TDEW, MPHIP_H2O);  f3(MPHIP_MQ_TICE, MPHIP_H2O);
  f3(MPHIP_MQ_TNAT, MPHIP_H2O);
  return d;
}

bool meteo_requested(const mphip_ctl_t &c) {
  for (int k = 0; k < MPHIP_NMQ; k++)
    if (c.qnt_met[k] >= 0)
      return true;
  return false;
}

// checks of a module_meteo call (quantity indices, uploaded fields); 0 = fine
int check_meteo(mphip_ctx *ctx) {
  const mphip_ctl_t &c = ctx->ctl;
  if (ctx->np == 0 || !meteo_requested(c))
    return 0;
  for (int k = 0; k < MPHIP_NMQ; k++)
    if (c.qnt_met[k] >= c.nq)
      return fail(ctx, "module_meteo: quantity index out of range");
  const MetSlot &s0 = ctx->slot[0 ^ ctx->flip], &s1 = ctx->slot[1 ^ ctx->flip];
  if (!s0.valid || !s1.valid)
    return fail(ctx, "meteo data for both met0 and met1 must be uploaded before stepping");
  static const char *const n3[MPHIP_N3D] = { "u", "v", "w", "t", "lwc", "rwc", "iwc", "swc", "pl", "ul", "vl", "zetal",
                                             "zeta_dotl", "h2o", "z", "pv", "o3", "cc", "wl" };
  static const char *const n2[MPHIP_N2D] = { "ps", "pbl", "cape", "cin", "pel", "pct", "pcb", "cl", "ess", "nss", "shf",
                                             "ts", "zs", "us", "vs", "lsm", "sst", "pt", "tt", "zt", "h2ot", "plcl",
                                             "plfc", "o3c" };
  // the climatology quantities: tables, and the rule of mptrac.c:5074-5076
  static const struct { int q, zm; const char *name; } zmq[] = {
    { MPHIP_MQ_HNO3, MPHIP_ZM_HNO3, "HNO3" }, { MPHIP_MQ_TNAT, MPHIP_ZM_HNO3, "HNO3" }, { MPHIP_MQ_OH, MPHIP_ZM_OH, "OH" },
    { MPHIP_MQ_H2O2, MPHIP_ZM_H2O2, "H2O2" }, { MPHIP_MQ_HO2, MPHIP_ZM_HO2, "HO2" }, { MPHIP_MQ_O1D, MPHIP_ZM_O1D, "O1D" } };
  for (const auto &e : zmq)
    if (c.qnt_met[e.q] >= 0 && !ctx->d_zm[e.zm])
      return fail(ctx, std::string("module_meteo: the ") + e.name + " climatology was not uploaded");
  if (c.qnt_met[MPHIP_MQ_TSTS] >= 0 && (c.qnt_met[MPHIP_MQ_TICE] < 0 || c.qnt_met[MPHIP_MQ_TNAT] < 0))
    return fail(ctx, "Need T_ice and T_NAT to calculate T_STS!");
  const MeteoDeps d = meteo_deps(c);
  for (int f = 0; f < MPHIP_N3D; f++)
    if (((d.need3 >> f) & 1u) && (!s0.has3[f] || !s1.has3[f]))
      return fail(ctx, std::string("module_meteo: meteo field ") + n3[f] + " was not uploaded");
  for (int f = 0; f < MPHIP_N2D; f++)
    if (((d.need2 >> f) & 1u) && (!s0.has2[f] || !s1.has2[f]))
      return fail(ctx, std::string("module_meteo: meteo field ") + n2[f] + " was not uploaded");
  return 0;
}

int launch_meteo(mphip_ctx *ctx) {
  const mphip_ctl_t &c = ctx->ctl;
  if (ctx->np == 0 || !meteo_requested(c))
    return 0;   // nothing to set: every SET_ATM of the reference is a no-op
  if (check_meteo(ctx) || ensure_packed(ctx))
    return 1;
  MeteoArgs G;
  memset(&G, 0, sizeof(G));
  const MeteoDeps d = meteo_deps(c);
  G.ctl = c;
  G.met = dev_met(ctx);
  G.atm = dev_atm(ctx);
  G.need3 = d.need3;
  G.need2 = d.need2;
  for (int k = 0; k < MPHIP_NZM; k++)
    G.zm[k] = ctx->zm[k];
  long long per_block = (ctx->np + ctx->step_blocks - 1) / ctx->step_blocks;
  per_block = std::max<long long>(256, (per_block + 255) / 256 * 256);
  int nb = (int) ((ctx->np + per_block - 1) / per_block);
  nb = (nb + 7) & ~7;
  G.nblocks_logical = nb;
  G.per_block = per_block;
  G.xcd_map = ctx->xcd_map;
  hipLaunchKernelGGL(meteo_kernel, dim3(nb), dim3(256), axes_lds_bytes(ctx), ctx->stream, G);
  HIPCHK(hipGetLastError());
  return 0;
}

// module_meteo of mphip_run_timestep: checked now, run when its result can be seen (lazy_meteo)
int schedule_meteo(mphip_ctx *ctx) {
  if (!ctx->lazy_meteo)
    return launch_meteo(ctx);
  if (check_meteo(ctx))
    return 1;
  ctx->meteo_pending = true;
  return 0;
}

int flush_meteo(mphip_ctx *ctx) {
  if (!ctx->meteo_pending)
    return 0;
  ctx->meteo_pending = false;
  return launch_meteo(ctx);
}

PermGeom perm_geom(long long n) {
  PermGeom pg;
  long long per_block = (n + 8191) / 8192;
  per_block = std::max<long long>(256, (per_block + 255) / 256 * 256);
  int nb = (int) ((n + per_block - 1) / per_block);
  pg.nblocks = std::max(8, (nb + 7) & ~7);
  pg.per_block = per_block;
  return pg;
}

PermArgs perm_args(mphip_ctx *ctx, bool with_cache) {
  PermArgs g;
  memset(&g, 0, sizeof(g));
  g.n8 = 4 + ctx->nq;
  for (int k = 0; k < g.n8; k++) {
    g.in8[k] = ctx->d_arr[k];
    g.out8[k] = ctx->d_alt[k];
  }
  if (with_cache) {
    g.in8[g.n8] = ctx->d_dt;
    g.out8[g.n8] = ctx->d_dt_alt;
    g.n8++;
    if (ctx->d_iso) {
      g.in8[g.n8] = ctx->d_iso;
      g.out8[g.n8] = ctx->d_iso_alt;
      g.n8++;
    }
    g.n4 = 3;
    for (int k = 0; k < 3; k++) {
      g.in4[k] = ctx->d_uvwp[k];
      g.out4[k] = ctx->d_uvwp_alt[k];
    }
    if (ctx->d_kz) {
      g.in4[3] = (const float *) ctx->d_kz;
      g.out4[3] = (float *) ctx->d_kz_alt;
      g.n4 = 4;
    }
  }
  return g;
}

void perm_swap(mphip_ctx *ctx, bool with_cache) {
  for (int k = 0; k < 4 + ctx->nq; k++)
    std::swap(ctx->d_arr[k], ctx->d_alt[k]);
  if (with_cache) {
    std::swap(ctx->d_dt, ctx->d_dt_alt);
    std::swap(ctx->d_iso, ctx->d_iso_alt);
    std::swap(ctx->d_kz, ctx->d_kz_alt);
    for (int k = 0; k < 3; k++)
      std::swap(ctx->d_uvwp[k], ctx->d_uvwp_alt[k]);
  }
}

// stable LSD radix sort of n (key, value) pairs that starts in keys[0] / vals[0] and ping-pongs between the two
// buffer pairs; *cur_out = the pair that holds the result
// (n_dev: the number of pairs lives on the device and n is only its upper bound)
// (index_sort: the values are the positions 0, 1, 2 ...; vals[0] is not read, the first pass generates them)
// (packed_first: the input is keys[0] read as n interleaved (key, value) pairs -- 2 n words that may overlap
//  vals[0], which is then not read; the passes write to keys[1] / vals[1] first)
int radix_passes(mphip_ctx *ctx, uint32_t *const keys[2], int *const vals[2], long long n, int key_bits, int *cur_out,
                 const uint32_t *n_dev = nullptr, bool index_sort = false, bool packed_first = false) {
  *cur_out = 0;
  if (n <= 0)
    return 0;
  const int ntiles = (int) ((n + kSortTile - 1) / kSortTile);
  // digit width with the fewest passes (8 bits if it is a tie)
  int bits = ctx->sort_bits;
  if (bits == 0) {
    bits = 8;
    for (int b = 9; b <= kRadixMaxBits; b++)
      if ((key_bits + b - 1) / b < (key_bits + bits - 1) / bits)
        bits = b;
  }
  const int passes = (key_bits + bits - 1) / bits;
  const size_t m = ((size_t) 1 << bits) * ntiles;
  // chunk length of the two-level scan: 4096 counters while <= 1024 chunks cover m, else 16384
  const bool small = (m + kScanThreads * kScanPerSmall - 1) / (kScanThreads * kScanPerSmall) <= (size_t) kScanThreads;
  const int chunk_shift = small ? 12 : 14;
  static_assert(kScanThreads * kScanPerSmall == 1 << 12 && kScanThreads * kScanPerLarge == 1 << 14, "chunk lengths");
  const int nchunks = (int) ((m + ((size_t) 1 << chunk_shift) - 1) >> chunk_shift);
  if (nchunks > kScanThreads)
    return fail(ctx, "too many particles for the two-level scan of the radix sort");
  if (m + kScanThreads > ctx->counts_cap) {
    if (dev_alloc(ctx, &ctx->d_counts, m + kScanThreads))   // counters + chunk totals
      return 1;
    ctx->counts_cap = m + kScanThreads;
  }
  uint32_t *d_chunks = ctx->d_counts + m;
  int cur = 0;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = bits * pass;
    const int stride = packed_first && pass == 0 ? 2 : 1;
#define SORT_PASS(B)                                                                                                   \
  hipLaunchKernelGGL(sort_hist_kernel<B>, dim3(ntiles), dim3(kSortThreads), 0, ctx->stream, keys[cur], n, shift,       \
                     ntiles, ctx->d_counts, n_dev, stride);                                                            \
  if (small)                                                                                                           \
    hipLaunchKernelGGL(sort_scan_local_kernel<kScanPerSmall>, dim3(nchunks), dim3(kScanThreads), 0, ctx->stream,       \
                       ctx->d_counts, m, d_chunks, n_dev, 1 << B);                                                     \
  else                                                                                                                 \
    hipLaunchKernelGGL(sort_scan_local_kernel<kScanPerLarge>, dim3(nchunks), dim3(kScanThreads), 0, ctx->stream,       \
                       ctx->d_counts, m, d_chunks, n_dev, 1 << B);                                                     \
  hipLaunchKernelGGL(sort_scan_chunks_kernel, dim3(1), dim3(kScanThreads), 0, ctx->stream, d_chunks, nchunks);         \
  hipLaunchKernelGGL(sort_scatter_kernel<B>, dim3(ntiles), dim3(kSortThreads), 0, ctx->stream, keys[cur],              \
                     index_sort && pass == 0 ? (const int *) nullptr                                                   \
                                             : (stride == 2 ? (const int *) keys[cur] + 1 : vals[cur]),                \
                     keys[cur ^ 1], vals[cur ^ 1], n, shift, ntiles, ctx->d_counts, d_chunks, chunk_shift, n_dev, stride)
    if (bits == 8) {
      SORT_PASS(8);
    } else if (bits == 9) {
      SORT_PASS(9);
    } else {
      SORT_PASS(10);
    }
#undef SORT_PASS
    cur ^= 1;
  }
  HIPCHK(hipGetLastError());
  *cur_out = cur;
  return 0;
}

int bits_for(unsigned long long kmax) {   // number of key bits that can be non-zero for keys <= kmax
  int key_bits = 1;
  while (key_bits < 32 && (kmax >> key_bits) != 0)
    key_bits++;
  return key_bits;
}

// module_sort keys + sort of (key, index); returns the buffer holding the result
// largest sort key + 1 of the (tiled) cell order; 0 = does not fit 32 bits
int bits_of(unsigned long long v) {   // bits needed for the numbers 0 ... v
  int b = 0;
  while (v >> b)
    b++;
  return b;
}

// tiles of the locality key in Z-order: number of interleaved bit pairs (-1: row-major tile numbers)
int tile_zbits(const mphip_ctx *ctx, int tile) {
  if (tile <= 0 || !ctx->locality_zorder)
    return -1;
  const unsigned long long ntx = (ctx->nx + tile - 1) / tile, nty = (ctx->ny + tile - 1) / tile;
  return std::min(bits_of(ntx - 1), bits_of(nty - 1));
}

unsigned long long sort_key_range(const mphip_ctx *ctx, int tile) {
  unsigned long long kmax = (unsigned long long) ctx->nx * ctx->ny * ctx->npl;
  if (tile > 0) {
    const unsigned long long ntx = (ctx->nx + tile - 1) / tile, nty = (ctx->ny + tile - 1) / tile;
    unsigned long long ntiles = ntx * nty;
    const int m = tile_zbits(ctx, tile);
    if (m >= 0)   // upper bound: every pattern of the interleaved bits under the largest value of the bits on top
      ntiles = ((std::max(ntx - 1, nty - 1) >> m) + 1) << (2 * m);
    kmax = ntiles * ctx->npl * tile * tile;
  }
  return kmax > 0xffffffffULL ? 0 : kmax;
}

// keys of module_sort (and module_timesteps with timestep_t, and module_mixing's box index with box) -> d_keys[0]
int sort_keys(mphip_ctx *ctx, int tile, const double *timestep_t, const BoxArgs *box) {
  if (!sort_key_range(ctx, tile))
    return fail(ctx, "meteo grid too large for the 32-bit sort key");
  const DevMet M = dev_met(ctx);
  const DevAtm a = dev_atm(ctx);
  TimestepArgs ts = { (double) ctx->ctl.direction, ctx->ctl.t_start, ctx->ctl.t_stop, timestep_t ? *timestep_t : 0.0 };
  BoxArgs none;
  memset(&none, 0, sizeof(none));
  const bool lean = ctx->coord_type == 0 && ctx->lut_size > 0 && !ctx->force_generic;
  if (lean)
    hipLaunchKernelGGL(sort_key_kernel<true>, dim3(grid_for(ctx->np)), dim3(256), axes_lds_bytes(ctx), ctx->stream, M, a, tile,
                       tile_zbits(ctx, tile), ctx->d_keys[0], (int *) nullptr, ts, timestep_t ? ctx->d_dt : nullptr,
                       box ? *box : none);
  else
    hipLaunchKernelGGL(sort_key_kernel<false>, dim3(grid_for(ctx->np)), dim3(256), axes_lds_bytes(ctx), ctx->stream, M, a, tile,
                       tile_zbits(ctx, tile), ctx->d_keys[0], (int *) nullptr, ts, timestep_t ? ctx->d_dt : nullptr,
                       box ? *box : none);
  HIPCHK(hipGetLastError());
  return 0;
}

int sort_pairs(mphip_ctx *ctx, int tile, int *result_buf, const double *timestep_t) {
  if (sort_keys(ctx, tile, timestep_t, nullptr))
    return 1;
  return radix_passes(ctx, ctx->d_keys, ctx->d_vals, ctx->np, bits_for(sort_key_range(ctx, tile)), result_buf, nullptr,
                      true);
}

// ---- module_sort as a repair of the previous order ---------------------------------------------------------------
// key_prev: the sorted keys of the module_sort whose order the particles are stored in; the new keys are in
// ctx->d_keys[0] (sort_keys).  Leaves the sorted (key, slot) pairs in ctx->d_keys[0] / d_vals[0].
int repair_sort(mphip_ctx *ctx, const uint32_t *key_prev, long long n, int key_bits, int *cur_out) {
  if (n > ctx->rep_cap) {
    const size_t m = (size_t) n;
    if (dev_alloc(ctx, &ctx->rep_sk, m) || dev_alloc(ctx, &ctx->rep_si, m) || dev_alloc(ctx, &ctx->rep_fk, m)
        || dev_alloc(ctx, &ctx->rep_fv, m) || dev_alloc(ctx, &ctx->rep_mk[0], m) || dev_alloc(ctx, &ctx->rep_mk[1], m)
        || dev_alloc(ctx, &ctx->rep_mi[0], m) || dev_alloc(ctx, &ctx->rep_mi[1], m)
        || dev_alloc(ctx, &ctx->rep_tiles, (m + kRepairTile - 1) / kRepairTile) || dev_alloc(ctx, &ctx->rep_nm, (size_t) 2))
      return 1;
    ctx->rep_cap = n;
  }
  const int ntiles = (int) ((n + kRepairTile - 1) / kRepairTile);
  const uint32_t *key_new = ctx->d_keys[0];
  hipLaunchKernelGGL(repair_count_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, key_new, key_prev, n, ctx->rep_tiles);
  hipLaunchKernelGGL(repair_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->rep_tiles, ntiles, n, ctx->rep_nm);
  hipLaunchKernelGGL(repair_split_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, key_new, key_prev, n,
                     (const uint32_t *) ctx->rep_tiles, ctx->rep_mk[0], ctx->rep_mi[0], ctx->rep_sk, ctx->rep_si);
  HIPCHK(hipGetLastError());
  int mc = 0;     // the movers: the stable radix sort over as many pairs as there are (the count stays on the device)
  if (radix_passes(ctx, ctx->rep_mk, ctx->rep_mi, n, key_bits, &mc, ctx->rep_nm))
    return 1;
  const int nmerge = (int) ((n + kMergeTile - 1) / kMergeTile);
  hipLaunchKernelGGL(repair_merge_kernel, dim3(nmerge), dim3(256), 0, ctx->stream, (const uint32_t *) ctx->rep_sk,
                     (const int *) ctx->rep_si, (const uint32_t *) ctx->rep_mk[mc], (const int *) ctx->rep_mi[mc],
                     (const uint32_t *) ctx->rep_nm, n, ctx->rep_fk, ctx->rep_fv);
  HIPCHK(hipGetLastError());
  std::swap(ctx->d_keys[0], ctx->rep_fk);
  std::swap(ctx->d_vals[0], ctx->rep_fv);
  *cur_out = 0;
  return 0;
}

// ---- module_sort ahead of time (mphip_ctx::sort_ahead) -----------------------------------------------------

// the sort buffers of the context <-> the buffers of the sort that runs ahead
void ahead_swap_buffers(mphip_ctx *ctx) {
  for (int k = 0; k < 2; k++) {
    std::swap(ctx->d_keys[k], ctx->ahead_keys[k]);
    std::swap(ctx->d_vals[k], ctx->ahead_vals[k]);
  }
  std::swap(ctx->d_counts, ctx->ahead_counts);
  std::swap(ctx->counts_cap, ctx->ahead_counts_cap);
  std::swap(ctx->d_dt, ctx->ahead_dt);
}

// forget a sort that runs ahead (something it depends on is about to change); waits for its kernels, which may
// still be reading the particle arrays
int ahead_drop(mphip_ctx *ctx) {
  if (!ctx->ahead_valid)
    return 0;
  ctx->ahead_valid = false;
  HIPCHK(hipStreamSynchronize(ctx->ahead_stream));
  return 0;
}

// keys, module_timesteps and radix sort of the module_sort call that mphip_run_timestep(t_next) will make, on
// the second stream, behind the kernels queued on the main stream so far
// (box: module_mixing of the current step is still to come -- the key kernel then runs on the main stream and
// leaves the box index of every particle in box->cell on the way, one pass over the particle arrays less)
// stream, events and buffers of the sort ahead
int ahead_ensure(mphip_ctx *ctx) {
  const long long n = ctx->np;
  if (!ctx->ahead_stream) {
    // (with the order repair the sort ahead is little work in few workgroups: at a higher priority they do not queue
    //  behind the thousands of workgroups of the deposition launch)
    int prio_low = 0, prio_high = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (ctx->ahead_priority && prio_high != prio_low) {
      HIPCHK(hipStreamCreateWithPriority(&ctx->ahead_stream, hipStreamNonBlocking, prio_high));
    } else
      HIPCHK(hipStreamCreateWithFlags(&ctx->ahead_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ahead_mark, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ahead_done, hipEventDisableTiming));
  }
  if (n > ctx->ahead_cap) {
    for (int k = 0; k < 2; k++)
      if (dev_alloc(ctx, &ctx->ahead_keys[k], (size_t) n) || dev_alloc(ctx, &ctx->ahead_vals[k], (size_t) n))
        return 1;
    if (dev_alloc(ctx, &ctx->ahead_dt, (size_t) n))
      return 1;
    ctx->ahead_cap = n;
  }
  return 0;
}

// (keys_ready: the launch that moved the particles has written keys, dt and box index already -- EmitKeys)
int ahead_launch(mphip_ctx *ctx, double t_next, const BoxArgs *box = nullptr, bool keys_ready = false) {
  const long long n = ctx->np;
  if (ahead_ensure(ctx))
    return 1;
  // the particles are stored in the order of the last module_sort: its sorted keys (in the buffers of the context, which
  // the sort ahead does not touch) let this one repair that order instead of sorting from scratch
  const uint32_t *key_prev = ctx->sort_repair && ctx->stored_is_sorted && ctx->sorted_buf >= 0 && ctx->sorted_n == n
    ? ctx->d_keys[ctx->sorted_buf] : nullptr;
  // the sort code runs as it is, on the other stream and the other buffers
  hipStream_t main_stream = ctx->stream;
  ahead_swap_buffers(ctx);
  int cur = 0, rc = 0;
  if (box && !keys_ready)
    rc = sort_keys(ctx, 0, &t_next, box);
  if (!rc && (hipEventRecord(ctx->ahead_mark, main_stream) != hipSuccess
              || hipStreamWaitEvent(ctx->ahead_stream, ctx->ahead_mark, 0) != hipSuccess))
    rc = fail(ctx, "stream hand-over of the sort ahead of time failed");
  ctx->stream = ctx->ahead_stream;
  if (!rc && !box && !keys_ready)
    rc = sort_keys(ctx, 0, &t_next, nullptr);
  if (!rc)
    rc = key_prev ? repair_sort(ctx, key_prev, n, bits_for(sort_key_range(ctx, 0)), &cur)
                  : radix_passes(ctx, ctx->d_keys, ctx->d_vals, n, bits_for(sort_key_range(ctx, 0)), &cur, nullptr, true);
  ctx->stream = main_stream;
  ahead_swap_buffers(ctx);
  if (rc)
    return 1;
  HIPCHK(hipEventRecord(ctx->ahead_done, ctx->ahead_stream));
  ctx->ahead_valid = true;
  ctx->ahead_t = t_next;
  ctx->ahead_cur = cur;
  return 0;
}

// put every per-particle array back into the external slot order
// The particle arrays of `g` moved by a RANDOM permutation: out[i] = in[index[i]] (gather) or out[index[i]] = in[i]
// (scatter), through one record per particle (perm_pack_kernel / perm_unpack_kernel).  Falls back to the
// array-by-array kernels if the record buffer cannot be had (option perm_records 0 does so always).
int permute_random(mphip_ctx *ctx, const PermArgs &g, const int *index, long long n, bool scatter) {
  const PermGeom pg = perm_geom(n);
  RecordGeom rg;
  rg.n8 = g.n8;
  rg.n4 = g.n4;
  const int words4 = g.n4 + (g.ext_out ? 1 : 0);
  rg.chunks = (g.n8 + 1) / 2 + (words4 + 3) / 4;
  rg.inv = 65536 / rg.chunks + 1;
  for (int p = 0; p < 64 * rg.chunks; p++)
    if ((int) (((unsigned) p * (unsigned) rg.inv) >> 16) != p / rg.chunks)
      return fail(ctx, "internal: record piece index");
  const size_t need = (size_t) n * (size_t) rg.chunks * 16;
  bool records = ctx->perm_records && n >= 65536;
  if (records && need > ctx->prec_cap) {
    if (ctx->d_prec)
      (void) hipFree(ctx->d_prec);
    ctx->d_prec = nullptr;
    ctx->prec_cap = 0;
    if (hipMalloc(&ctx->d_prec, need) == hipSuccess)
      ctx->prec_cap = need;
    else {
      (void) hipGetLastError();
      records = false;
    }
  }
  if (!records) {
    if (scatter)
      hipLaunchKernelGGL(perm_scatter_kernel, dim3(pg.nblocks), dim3(256), 0, ctx->stream, g, index, n, pg);
    else
      hipLaunchKernelGGL(perm_gather_kernel, dim3(pg.nblocks), dim3(256), 0, ctx->stream, g, index, n, pg);
    HIPCHK(hipGetLastError());
    return 0;
  }
  f32x4u *rec = (f32x4u *) ctx->d_prec;
  const size_t lds = (size_t) kRecordWaves * 64 * (size_t) rg.chunks * 16;
  hipLaunchKernelGGL(perm_pack_kernel, dim3(pg.nblocks), dim3(64 * kRecordWaves), lds, ctx->stream, g, rg,
                     scatter ? index : (const int *) nullptr, rec, n, pg);
  hipLaunchKernelGGL(perm_unpack_kernel, dim3(pg.nblocks), dim3(64 * kRecordWaves), lds, ctx->stream, g, rg,
                     scatter ? (const int *) nullptr : index, (const f32x4u *) rec, n, pg);
  HIPCHK(hipGetLastError());
  return 0;
}

int restore_external_order(mphip_ctx *ctx) {
  if (ctx->ext_identity || ctx->np == 0) {
    ctx->ext_identity = true;
    return 0;
  }
  if (ahead_drop(ctx))
    return 1;
  PermArgs g = perm_args(ctx, true);
  if (permute_random(ctx, g, ctx->d_ext, ctx->np, true))
    return 1;
  perm_swap(ctx, true);
  ctx->stored_is_sorted = false;
  ctx->ext_identity = true;
  ctx->steps_since_resort = 1 << 30;
  return 0;
}

// internal locality order: store the particles sorted by the grid cell their
// interpolation stencil starts in.  Nothing observable changes: random numbers
// follow the external slot (d_ext) and downloads restore the external order.
int locality_sort(mphip_ctx *ctx) {
  if (ctx->np == 0)
    return 0;
  if (ahead_drop(ctx) || ensure_packed(ctx))
    return 1;
  int cur = 0;
  // measured (tools/gpu_ablate.py tiles): 4 x 4 columns for the pressure-level kernels, 8 x 8 for the model-level ones
  const bool ml_winds = ctx->have_ctl && ctx->ctl.advect_vert_coord >= 1 && ctx->ctl.advect_vert_coord <= 3;
  const int tile = ctx->locality_tile > 0 ? ctx->locality_tile : (ml_winds ? 8 : 4);
  if (sort_pairs(ctx, tile, &cur, nullptr))
    return 1;
  PermArgs g = perm_args(ctx, true);
  g.ext_in = ctx->ext_identity ? nullptr : ctx->d_ext;
  g.ext_out = ctx->d_ext_alt;
  if (ctx->ext_identity) {   // out of the caller's order: a random permutation
    if (permute_random(ctx, g, ctx->d_vals[cur], ctx->np, false))
      return 1;
  } else {                   // a re-sort: most particles stay where they are, the gathers hit the caches
    const PermGeom pg = perm_geom(ctx->np);
    hipLaunchKernelGGL(perm_gather_kernel, dim3(pg.nblocks), dim3(256), 0, ctx->stream, g, ctx->d_vals[cur], ctx->np, pg);
    HIPCHK(hipGetLastError());
  }
  perm_swap(ctx, true);
  std::swap(ctx->d_ext, ctx->d_ext_alt);
  ctx->stored_is_sorted = false;
  ctx->ext_identity = false;
  ctx->steps_since_resort = 0;
  return 0;
}

// module_sort, mptrac.c:5887-5957: observable re-ordering of atm (time, p, lon,
// lat, q[*]); cache->uvwp and cache->dt stay with their slots as in the
// reference.
int do_sort(mphip_ctx *ctx, const double *timestep_t = nullptr) {
  const long long n = ctx->np;
  if (n == 0)
    return 0;
  int cur = 0;
  if (timestep_t && ctx->ahead_valid && ctx->ahead_t == *timestep_t && ctx->ext_identity) {
    // keys, dt and the sorted permutation were computed beside the previous time step: take them over
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ahead_done, 0));
    ahead_swap_buffers(ctx);
    ctx->ahead_valid = false;
    cur = ctx->ahead_cur;
  } else {
    if (ahead_drop(ctx) || ensure_packed(ctx) || restore_external_order(ctx))
      return 1;
    if (sort_pairs(ctx, 0, &cur, timestep_t))   // with timestep_t: module_timesteps in the key kernel
      return 1;
  }
  ctx->sorted_buf = cur;
  ctx->stored_is_sorted = true;      // (once the gather below -- or the step launch that carries it -- has run)
  ctx->sorted_n = n;
  const PermGeom pg = perm_geom(n);
  if (timestep_t && ctx->fuse_sort) {
    // inside mphip_run_timestep the step launch that follows reads time, p, lon, lat through the
    // permutation and writes them in the new order (DevAtm::perm); only the quantity arrays move here
    PermArgs g;
    memset(&g, 0, sizeof(g));
    g.n8 = ctx->nq;
    for (int k = 0; k < ctx->nq; k++) {
      g.in8[k] = ctx->d_arr[4 + k];
      g.out8[k] = ctx->d_alt[4 + k];
    }
    // ... or not even those: option fuse_sort_quantities (default) lets the step launch move them as well
    ctx->fused_quantities = ctx->fuse_quantities;
    if (ctx->nq > 0 && !ctx->fused_quantities) {
      hipLaunchKernelGGL(perm_gather_kernel, dim3(pg.nblocks), dim3(256), 0, ctx->stream, g, ctx->d_vals[cur], n, pg);
      HIPCHK(hipGetLastError());
    }
    for (int k = 0; k < 4 + ctx->nq; k++)
      std::swap(ctx->d_arr[k], ctx->d_alt[k]);   // d_alt[0..3] now hold the pre-sort time, p, lon, lat
    ctx->fused_perm = ctx->d_vals[cur];
  } else {
    PermArgs g = perm_args(ctx, false);
    hipLaunchKernelGGL(perm_gather_kernel, dim3(pg.nblocks), dim3(256), 0, ctx->stream, g, ctx->d_vals[cur], n, pg);
    HIPCHK(hipGetLastError());
    perm_swap(ctx, false);
  }
  ctx->steps_since_resort = 0;   // the observable order is a locality order already
  return 0;
}

int ensure_sums(mphip_ctx *ctx, size_t n) {
  if (n > ctx->sums_cap) {
    if (dev_alloc(ctx, &ctx->d_sums, n))
      return 1;
    ctx->sums_cap = n;
  }
  return 0;
}

// ---- RCCL (loaded on first use: single-GPU callers need no librccl) ------------
// Only the handful of entry points the gridded reductions need; types restated from rccl.h (stable NCCL ABI).
struct Rccl {
  typedef struct { char internal[128]; } UniqueId;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(void *, int *) = nullptr;
  int (*CommUserRank)(void *, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
constexpr int kNcclInt32 = 2, kNcclFloat64 = 8, kNcclSum = 0;

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
        break;
    if (!h) {
      r.why = std::string("cannot load librccl.so: ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(h, n);
      if (!p)
        r.why = std::string("librccl.so lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId)) sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank)) sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy)) sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce)) sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart)) sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd)) sym("ncclGroupEnd");
    r.CommCount = (decltype(r.CommCount)) sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank)) sym("ncclCommUserRank");
    r.GetErrorString = (decltype(r.GetErrorString)) sym("ncclGetErrorString");
    r.ok = r.why.empty();
  });
  return r;
}

// RCCL writes a version banner to stdout when it initialises; callers of this library own stdout (bench.py
// prints exactly one JSON line there), so the banner goes to stderr instead
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0)
      dup2(2, 1);
  }
  ~StdoutToStderr() {
    fflush(stdout);
    if (saved >= 0) {
      dup2(saved, 1);
      close(saved);
    }
  }
};

#define RCCLCHK(call)                                                                        \
  do {                                                                                       \
    const int e_ = (call);                                                                   \
    if (e_ != 0)                                                                             \
      return fail(ctx, std::string(#call) + ": " + rccl().GetErrorString(e_));               \
  } while (0)

// Sum over the ranks, in place: `count` doubles at dbuf and (optionally) `icount` 32-bit integers at ibuf.
// With a communicator (mphip_comm_init) both go to RCCL as one group on the context's stream and the host does
// not wait; with an all-reduce hook (tests, staged host collectives) the stream is drained and the integers
// travel as doubles in `scratch` (icount doubles).
int run_allreduce(mphip_ctx *ctx, double *dbuf, size_t count, int *ibuf = nullptr, size_t icount = 0,
                  double *scratch = nullptr) {
  if (ctx->comm) {
    Rccl &R = rccl();
    RCCLCHK(R.GroupStart());
    if (count)
      RCCLCHK(R.AllReduce(dbuf, dbuf, count, kNcclFloat64, kNcclSum, ctx->comm, ctx->stream));
    if (icount)
      RCCLCHK(R.AllReduce(ibuf, ibuf, icount, kNcclInt32, kNcclSum, ctx->comm, ctx->stream));
    RCCLCHK(R.GroupEnd());
    return 0;
  }
  if (!ctx->allreduce)
    return 0;
  if (icount) {
    hipLaunchKernelGGL(int_to_double_kernel, dim3(grid_for((long long) icount)), dim3(256), 0, ctx->stream, ibuf, scratch,
                       icount);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (count && ctx->allreduce(dbuf, count, ctx->allreduce_user))
    return fail(ctx, "all-reduce hook reported an error");
  if (icount) {
    if (ctx->allreduce(scratch, icount, ctx->allreduce_user))
      return fail(ctx, "all-reduce hook reported an error");
    hipLaunchKernelGGL(double_to_int_kernel, dim3(grid_for((long long) icount)), dim3(256), 0, ctx->stream, scratch, ibuf,
                       icount);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// module_mixing, mptrac.c:5169-5347
// launch geometry of the LDS-table accumulation kernels: table entries (power of two) that fit in
// 64 kB of LDS for nv doubles per entry, and contiguous runs of the stored order per block
struct AccumGeom {
  int T, nblocks;
  long long per_block;
  size_t lds;
};

AccumGeom accum_geom(const mphip_ctx *ctx, int nv) {
  AccumGeom g;
  g.T = 2048;
  while (g.T > 64 && (size_t) g.T * (8 * (size_t) nv + 4) > 64 * 1024)
    g.T >>= 1;
  g.lds = (size_t) g.T * (8 * (size_t) nv + 4);
  long long per_block = (ctx->np + 4095) / 4096;
  per_block = std::max<long long>(256, (per_block + 255) / 256 * 256);
  g.per_block = per_block;
  g.nblocks = (int) std::max<long long>(1, (ctx->np + per_block - 1) / per_block);
  return g;
}

// Sums of VALS per cell in the reference's serial order (steps 1-4 of "Cell sums" in mphip_kernels.hpp).
// ctx->d_cell holds the cell of every stored particle; writes sums[v * ntot + cell] (nv values) and the counts
// as integers and / or doubles (either may be NULL).
// `every_cell`: cells of groups without particles must read zero afterwards (an all-reduce or the host looks at
// all of them); module_mixing on a single rank only ever reads the cells its particles are in, which the
// group kernel has written, and skips the clearing passes
// (box: the cells are write_grid's box indices and have not been computed yet -- the crowded-cell path does that on
//  its way; the other path calls box_index_kernel first)
template <class VALS>
int ordered_cell_sums(mphip_ctx *ctx, const VALS &vals, int nv, int column, size_t ntot, double *sums, int *cnt,
                      double *cnt_as_double, bool every_cell = true, const GridBoxArgs *box = nullptr) {
  const long long n = ctx->np;
  // cells per group: whole vertical columns of the grid, as many as fit the table in LDS while there are
  // still >= 2^14 groups to spread over the waves
  int G = kGroupMax;
  if (column >= 1 && column <= kGroupMax) {
    const size_t ncol = std::max<size_t>(1, std::min<size_t>(kGroupMax / column, ntot / column / 16384));
    G = column * (int) ncol;
  }
  const size_t ngroups = (ntot + G - 1) / G;
  if (ntot >= 0x7fffffffULL)
    return fail(ctx, "too many grid cells for 32-bit cell indices");
  // crowded cells (a 2-D output grid, a dense plume on a coarse grid): sort the whole particle list by cell and
  // give every (cell, value) chain a lane -- "crowded cells" in mphip_kernels.hpp
  const bool chains = ctx->sum_path == 2 || (ctx->sum_path == 0 && (double) n >= 16.0 * (double) ntot);
  if (chains && n > 0) {
    const size_t words32 = 4 * (size_t) n;
    if ((words32 + 1) / 2 > ctx->lists_cap) {
      if (dev_alloc(ctx, &ctx->d_lists, (words32 + 1) / 2))
        return 1;
      ctx->lists_cap = (words32 + 1) / 2;
    }
    uint32_t *base = (uint32_t *) ctx->d_lists;
    uint32_t *keys[2] = { base, base + 2 * n };
    int *slots[2] = { (int *) (base + n), (int *) (base + 3 * n) };
    // (cell, slot) pairs in external order, interleaved in the first half of the buffer; the first pass of the
    // sort reads them from there and writes the second half
    GridBoxArgs gb;
    memset(&gb, 0, sizeof(gb));
    if (box)
      gb = *box;
    hipLaunchKernelGGL(cell_slot_pairs_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_cell,
                       ctx->ext_identity ? (const int *) nullptr : ctx->d_ext, n, (uint32_t) ntot, (uint2 *) base, gb,
                       box ? 1 : 0);
    int cur = 0;
    if (radix_passes(ctx, keys, slots, n, bits_for(ntot), &cur, nullptr, false, true))
      return 1;
    const int width = nv < 1 ? 1 : nv < 64 ? nv : 64;   // nv == 0: counts only (gridded output without quantities)
    const long long waves = ((long long) ntot + 64 / width - 1) / (64 / width);
    // (measured on C3's output, 1e7 particles on 360 x 180 cells: 1.10 ms per output with every wave resident,
    // 1.21 / 1.36 / 1.86 ms with 512 / 256 / 64 workgroups -- the gathers are bound by the 128-byte lines they
    // pull for 8 bytes each, 3.1 GB per output, and fewer waves in flight do not make the lines live longer)
    int chain_blocks = ctx->chain_blocks > 0 ? ctx->chain_blocks : 8192;
    // (a multiple of 8 workgroups: the kernel maps them to contiguous runs of cells per XCD)
    chain_blocks = std::max(8, std::min(chain_blocks, (grid_for(waves * 64) + 7) & ~7) & ~7);
    // every cell finds its range of the sorted list itself (a pass over the list and a clearing pass less)
    hipLaunchKernelGGL(cell_sum_chains_kernel<VALS>, dim3(chain_blocks), dim3(256), 0, ctx->stream, vals,
                       slots[cur], keys[cur], n, ntot, sums, cnt, cnt_as_double);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (box && n > 0)   // the other path works on ctx->d_cell
    hipLaunchKernelGGL(box_index_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, dev_atm(ctx), box->G, box->t0,
                       box->t1, ctx->d_cell, (const double *) nullptr, 0);
  if (every_cell) {
    HIPCHK(hipMemsetAsync(sums, 0, (size_t) nv * ntot * sizeof(double), ctx->stream));
    if (cnt)
      HIPCHK(hipMemsetAsync(cnt, 0, ntot * sizeof(int), ctx->stream));
    if (cnt_as_double)
      HIPCHK(hipMemsetAsync(cnt_as_double, 0, ntot * sizeof(double), ctx->stream));
  }
  if (n == 0)
    return 0;
  // one allocation, 32-bit words: [sequence cell | sequence slot | keys 0 | ids 0 | keys 1 | ids 1 | run starts
  // (n + 1) | runs per tile (ntiles + 1) | overflow flag]
  const int ntiles = (int) ((n + kRunTile - 1) / kRunTile);
  const size_t words32 = 7 * (size_t) n + 1 + (size_t) ntiles + 1 + 1;
  if ((words32 + 1) / 2 > ctx->lists_cap) {
    if (dev_alloc(ctx, &ctx->d_lists, (words32 + 1) / 2))
      return 1;
    ctx->lists_cap = (words32 + 1) / 2;
  }
  uint32_t *base = (uint32_t *) ctx->d_lists;
  int *seq_cell = (int *) base, *seq_slot = (int *) (base + n);
  uint32_t *keys[2] = { base + 2 * n, base + 4 * n };
  int *ids[2] = { (int *) (base + 3 * n), (int *) (base + 5 * n) };
  uint32_t *run_start = base + 6 * n, *tile_runs = base + 7 * n + 1;
  int *overflow = (int *) (tile_runs + ntiles + 1);
  const uint32_t *nruns_dev = tile_runs + ntiles;   // stays on the device: the launches cover the upper bound n
  const int nblocks = (int) std::min<long long>((n + 255) / 256, 16384);
  const int key_bits = bits_for(ngroups);
  // runs of `seq`, sorted by group -> keys[cur], ids[cur] (kernels do nothing while *gate == 0)
  auto sorted_runs = [&](const int *seq, const int *gate, int *cur) -> int {
    hipLaunchKernelGGL(run_heads_count_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, seq, n, G, tile_runs, gate);
    hipLaunchKernelGGL(run_offsets_kernel, dim3(1), dim3(kScanThreads), 0, ctx->stream, tile_runs, ntiles, gate);
    hipLaunchKernelGGL(run_compact_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, seq, n, G, tile_runs, ntiles,
                       (uint32_t) ngroups, keys[0], (int *) nullptr, run_start, gate);
    return radix_passes(ctx, keys, ids, n, key_bits, cur, nruns_dev, true);
  };
#define GROUPS(B, BYINDEX, SEQ, SLOT)                                                                                  \
  hipLaunchKernelGGL((cell_sum_groups_kernel<VALS, B, BYINDEX>), dim3(nblocks), dim3(256), 0, ctx->stream, vals,       \
                     keys[cur], ids[cur], nruns_dev, (uint32_t) ngroups, run_start, SEQ, SLOT, ctx->d_ext, overflow, G, \
                     ntot, sums, cnt, cnt_as_double)
#define GROUPS_BY_WIDTH(BYINDEX, SEQ, SLOT)                                                                            \
  if (nv <= 1) {   /* values per pass: as many as there are, up to four */                                             \
    GROUPS(1, BYINDEX, SEQ, SLOT);                                                                                     \
  } else if (nv == 2) {                                                                                                \
    GROUPS(2, BYINDEX, SEQ, SLOT);                                                                                     \
  } else if (nv == 3) {                                                                                                \
    GROUPS(3, BYINDEX, SEQ, SLOT);                                                                                     \
  } else {                                                                                                             \
    GROUPS(4, BYINDEX, SEQ, SLOT);                                                                                     \
  }
  int cur = 0;
  if (ctx->ext_identity) {
    // the stored order is the external order (after module_sort): stream the groups as they come
    if (sorted_runs(ctx->d_cell, nullptr, &cur))
      return 1;
    GROUPS_BY_WIDTH(false, ctx->d_cell, (const int *) nullptr)
  } else {
    // internal locality order: the runs of the STORED order are long; every wave puts its group into index
    // order in LDS.  A group that does not fit there raises the flag ...  (With hundreds of particles per
    // group -- a 2-D output grid -- the sort in LDS costs more than the general pass: 2.3 against 1.5 ms for
    // 1e7 particles on 360 x 180 cells; that case goes to the general pass directly.)
    const bool in_lds = (double) n / (double) ngroups <= 256.0;
    HIPCHK(hipMemsetAsync(overflow, in_lds ? 0 : 1, sizeof(int), ctx->stream));
    if (in_lds) {
      if (sorted_runs(ctx->d_cell, nullptr, &cur))
        return 1;
      GROUPS_BY_WIDTH(true, ctx->d_cell, (const int *) nullptr)
    }
    // ... and the general pass runs: the whole sequence laid out in external order (one run per particle, the
    // permutation is random in the index), same kernels.  Without the flag its launches return at once.
    hipLaunchKernelGGL(cell_pairs_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, ctx->d_cell, ctx->d_ext, n,
                       seq_cell, seq_slot, overflow);
    if (sorted_runs(seq_cell, overflow, &cur))
      return 1;
    GROUPS_BY_WIDTH(false, seq_cell, seq_slot)
  }
#undef GROUPS_BY_WIDTH
#undef GROUPS
  HIPCHK(hipGetLastError());
  return 0;
}

int ensure_cell_counts(mphip_ctx *ctx, size_t ntot) {
  if (ntot > ctx->cnt_cap) {
    if (dev_alloc(ctx, &ctx->d_cnt, ntot))
      return 1;
    ctx->cnt_cap = ntot;
  }
  return 0;
}

// module_mixing, mptrac.c:5169-5347:
//   mixing_plan   what is mixed, the grid, the buffers
//   mixing_cells  the box of every particle
//   mixing_sums   sums and counts per box, all-reduce over the ranks
//   mixing_relax  every particle towards the mean of its box
struct MixGrid {
  BoxGrid grid;
  double t0, t1;
  const double *ens;
  int ngrid;
};

struct MixPlan {
  MixGrid box;      // the grid and the time window
  MixSet mq;        // what is mixed
  size_t ntot;      // boxes (times ensemble members)
};

// false in *active: nothing to mix
int mixing_plan(mphip_ctx *ctx, double t, MixPlan *P, bool *active) {
  const mphip_ctl_t &c = ctx->ctl;
  *active = false;
  if (!ctx->have_clim)
    return fail(ctx, "climatological tropopause data were not uploaded");
  const int ngrid = c.mixing_nx * c.mixing_ny * c.mixing_nz;
  const int nens = c.nens > 0 ? c.nens : 1;
  const size_t ntot = (size_t) ngrid * nens;
  const DevAtm a = dev_atm(ctx);
  memset(P, 0, sizeof(*P));
  // the mixed quantities (hot-path subset of mptrac.c:5223-5230), all in one pass
  MixSet &mq = P->mq;
  for (int iq : { c.qnt_m, c.qnt_vmr, c.qnt_tracer[MPHIP_TR_CCL4], c.qnt_tracer[MPHIP_TR_CCL3F], c.qnt_tracer[MPHIP_TR_CCL2F2],
                  c.qnt_tracer[MPHIP_TR_N2O], c.qnt_tracer[MPHIP_TR_SF6], c.qnt_aoa })
    if (iq >= 0)
      mq.q[mq.n++] = a.q[iq];
  if (mq.n == 0)
    return 0;
  // [sums of quantity 0 | 1 | 2 | scratch for a doubles-only all-reduce hook]
  const bool hook = !ctx->comm && ctx->allreduce;
  if (ensure_sums(ctx, ((size_t) mq.n + (hook ? 1 : 0)) * ntot) || ensure_cell_counts(ctx, ntot))
    return 1;
  P->ntot = ntot;
  P->box.grid = BoxGrid{ c.mixing_lon0, c.mixing_lon1, c.mixing_lat0, c.mixing_lat1, c.mixing_z0, c.mixing_z1,
                          c.mixing_nx, c.mixing_ny, c.mixing_nz };
  P->box.t0 = t - 0.5 * c.dt_mod;
  P->box.t1 = t + 0.5 * c.dt_mod;
  P->box.ens = (c.nens > 0 && c.qnt_ens >= 0) ? a.q[c.qnt_ens] : nullptr;
  P->box.ngrid = ngrid;
  *active = true;
  return 0;
}

int mixing_cells(mphip_ctx *ctx, const MixPlan &P) {
  if (ctx->np == 0)
    return 0;
  const MixGrid &h = P.box;
  hipLaunchKernelGGL(box_index_kernel, dim3(grid_for(ctx->np)), dim3(256), 0, ctx->stream, dev_atm(ctx), h.grid, h.t0, h.t1,
                     ctx->d_cell, h.ens, h.ngrid);
  HIPCHK(hipGetLastError());
  return 0;
}

// The exchange of a mixing step restricted to the band of levels that holds particles on any rank (SURVEY 8e (2):
// 5.83e6 boxes x 12 B = 70 MB per mixed quantity and step for the default grid; the particles of BASELINE configs[4]
// occupy 31 of its 90 levels).  A small all-reduce of the per-level occupancy first (nz doubles), its result read by
// the host -- the one synchronisation of the step path, taken only by runs of several ranks --, then pack, all-reduce
// of the band, unpack.  The boxes outside the band hold no particle on any rank: nothing to add there, and every box
// inside adds the same partial sums in the same order as the exchange of the whole grid (identical bits:
// test_mixing_exchange_of_the_occupied_levels_equals_the_whole_grid).  Falls back to the whole grid when the band
// covers four fifths of it.
int exchange_occupied_levels(mphip_ctx *ctx, int nq, size_t ntot, int nz) {
  const size_t ncol = ntot / (size_t) nz;
  if (!ctx->d_occ || !ctx->h_occ) {      // (both: a failed host allocation must not leave the pair half made)
    if (ctx->h_occ) {
      (void) hipHostFree(ctx->h_occ);
      ctx->h_occ = nullptr;
    }
    if (dev_alloc(ctx, &ctx->d_occ, (size_t) 256))
      return 1;
    HIPCHK(hipHostMalloc((void **) &ctx->h_occ, 256 * sizeof(double), hipHostMallocDefault));
  }
  HIPCHK(hipMemsetAsync(ctx->d_occ, 0, (size_t) nz * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(level_occupancy_kernel, dim3(std::min<size_t>(1024, (ntot + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const int *) ctx->d_cnt, ntot, nz, ctx->d_occ);
  HIPCHK(hipGetLastError());
  if (run_allreduce(ctx, ctx->d_occ, (size_t) nz))
    return 1;
  HIPCHK(hipMemcpyAsync(ctx->h_occ, ctx->d_occ, (size_t) nz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int lo = nz, hi = -1;
  for (int l = 0; l < nz; l++)
    if (ctx->h_occ[l] > 0) {
      lo = std::min(lo, l);
      hi = std::max(hi, l);
    }
  ctx->mix_band_lo = lo;
  ctx->mix_band_hi = hi;
  if (hi < lo)        // no particle in any box on any rank: nothing to exchange
    return 0;
  const int nl = hi - lo + 1;
  if (5 * nl >= 4 * nz)
    return run_allreduce(ctx, ctx->d_sums, (size_t) nq * ntot, ctx->d_cnt, ntot, ctx->d_sums + (size_t) nq * ntot);
  const size_t per = ncol * (size_t) nl;
  // dense buffers: [sums of the quantities | scratch for a doubles-only hook] and the counts
  if (((size_t) nq + 1) * per > ctx->band_cap) {
    if (dev_alloc(ctx, &ctx->d_band, ((size_t) nq + 1) * per) || dev_alloc(ctx, &ctx->d_band_cnt, per))
      return 1;
    ctx->band_cap = ((size_t) nq + 1) * per;
  }
  const int blocks = (int) std::min<size_t>(8192, ((size_t) nq * per + 255) / 256);
  hipLaunchKernelGGL(pack_levels_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double *) ctx->d_sums,
                     ctx->d_band, ncol, nz, lo, nl, nq, false, (double *) nullptr);
  hipLaunchKernelGGL(pack_levels_kernel<int>, dim3(blocks), dim3(256), 0, ctx->stream, (const int *) ctx->d_cnt,
                     ctx->d_band_cnt, ncol, nz, lo, nl, 1, false, (int *) nullptr);
  HIPCHK(hipGetLastError());
  if (run_allreduce(ctx, ctx->d_band, (size_t) nq * per, ctx->d_band_cnt, per, ctx->d_band + (size_t) nq * per))
    return 1;
  hipLaunchKernelGGL(pack_levels_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double *) nullptr,
                     ctx->d_band, ncol, nz, lo, nl, nq, true, ctx->d_sums);
  hipLaunchKernelGGL(pack_levels_kernel<int>, dim3(blocks), dim3(256), 0, ctx->stream, (const int *) nullptr,
                     ctx->d_band_cnt, ncol, nz, lo, nl, 1, true, ctx->d_cnt);
  HIPCHK(hipGetLastError());
  return 0;
}

int mixing_sums(mphip_ctx *ctx, const MixPlan &P) {
  const MixSet &mq = P.mq;
  const size_t ntot = P.ntot;
  if (ctx->deterministic_sums != 0) {
    MixVals vals = { mq };
    const bool exchanged = ctx->comm != nullptr || ctx->allreduce != nullptr;
    if (ordered_cell_sums(ctx, vals, mq.n, ctx->ctl.mixing_nz, ntot, ctx->d_sums, ctx->d_cnt, (double *) nullptr, exchanged))
      return 1;
  } else {
    HIPCHK(hipMemsetAsync(ctx->d_sums, 0, (size_t) mq.n * ntot * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_cnt, 0, ntot * sizeof(int), ctx->stream));
    if (ctx->np) {
      const AccumGeom g = accum_geom(ctx, mq.n + 1);
      hipLaunchKernelGGL(mix_accumulate_kernel, dim3(g.nblocks), dim3(256), g.lds, ctx->stream, dev_atm(ctx), ctx->d_cell,
                         mq, ntot, ctx->d_sums, ctx->d_cnt, g.T, g.per_block);
      HIPCHK(hipGetLastError());
    }
  }
  // one exchange per mixing step: the sums of every mixed quantity and the cell counts
  const bool exchange = ctx->comm != nullptr || ctx->allreduce != nullptr;
  const int nz = ctx->ctl.mixing_nz;
  if (exchange && ctx->mix_exchange_levels && nz >= 8 && nz <= 256 && ntot >= ((size_t) 1 << 18))
    return exchange_occupied_levels(ctx, mq.n, ntot, nz);
  return run_allreduce(ctx, ctx->d_sums, (size_t) mq.n * ntot, ctx->d_cnt, ntot, ctx->d_sums + (size_t) mq.n * ntot);
}

int mixing_relax(mphip_ctx *ctx, const MixPlan &P) {
  if (ctx->np == 0)
    return 0;
  hipLaunchKernelGGL(mix_relax_kernel, dim3(grid_for(ctx->np)), dim3(256), 0, ctx->stream, ctx->ctl, ctx->d_clim,
                     dev_atm(ctx), ctx->d_cell, P.mq, P.ntot, ctx->d_sums, ctx->d_cnt);
  HIPCHK(hipGetLastError());
  return 0;
}

// (cells_ready: the box indices are in d_cell already, left there by the key kernel of a sort ahead of time)
int do_mixing(mphip_ctx *ctx, double t, bool cells_ready = false) {
  MixPlan P;
  bool active;
  if (mixing_plan(ctx, t, &P, &active))
    return 1;
  if (!active)
    return 0;
  return (!cells_ready && mixing_cells(ctx, P)) || mixing_sums(ctx, P) || mixing_relax(ctx, P);
}

void unpin_all(mphip_ctx *ctx, std::vector<std::pair<uintptr_t, uintptr_t>> &list) {
  std::lock_guard<std::mutex> guard(ctx->pinned_lock);
  for (auto &r : list)
    if (hipHostUnregister((void *) r.first) != hipSuccess)
      (void) hipGetLastError();
  list.clear();
}

// host -> device copies of one snapshot's fields into the staging arrays of slot S (asynchronous on `stream`)
int upload_fields(mphip_ctx *ctx, MetSlot &S, const mphip_met_t *met, bool new_grid, hipStream_t stream) {
  const int nml = met->npl > 0 ? met->npl : 0;
  const size_t ncell = (size_t) met->nx * met->ny * met->np, ncol = (size_t) met->nx * met->ny;
  for (int f = 0; f < MPHIP_N3D; f++) {
    const bool is_ml = (f >= MPHIP_PL && f <= MPHIP_ZETA_DOTL) || f == MPHIP_WL;
    const long long nlev = is_ml ? nml : met->np;
    const long long sx = is_ml ? met->sx_ml : met->sx, sy = is_ml ? met->sy_ml : met->sy;
    S.has3[f] = met->f3[f] != nullptr && nlev > 0;
    if (!S.has3[f])
      continue;
    const size_t n3 = (size_t) met->nx * met->ny * (size_t) nlev;
    if (new_grid || !S.f3[f])
      if (dev_alloc(ctx, &S.f3[f], n3))
        return 1;
    if (sy == nlev && sx == (long long) met->ny * nlev) {
      HIPCHK(hipMemcpyAsync(S.f3[f], met->f3[f], n3 * sizeof(float), hipMemcpyHostToDevice, stream));
    } else {
      if (sy < nlev || sx < sy * met->ny || sx % sy != 0)
        return fail(ctx, "unsupported 3-D meteo strides");
      hipMemcpy3DParms p;
      memset(&p, 0, sizeof(p));
      p.srcPtr = make_hipPitchedPtr((void *) met->f3[f], (size_t) sy * sizeof(float), (size_t) nlev,
                                    (size_t) (sx / sy));
      p.dstPtr = make_hipPitchedPtr(S.f3[f], (size_t) nlev * sizeof(float), (size_t) nlev, (size_t) met->ny);
      p.extent = make_hipExtent((size_t) nlev * sizeof(float), (size_t) met->ny, (size_t) met->nx);
      p.kind = hipMemcpyHostToDevice;
      HIPCHK(hipMemcpy3DAsync(&p, stream));
    }
  }
  S.ps11 = met->f2[MPHIP_PS] ? met->f2[MPHIP_PS][(size_t) met->sx2 + 1] : 0.f;
  S.ps_min = S.pct_min = S.pbl_min = S.pel_min = -HUGE_VAL;
  S.ps_max = S.pbl_max = HUGE_VAL;
  // smallest and largest value of a surface field if every value is finite
  auto extremes = [&](int f, double &lo_out, double &hi_out) {
    if (!met->f2[f])
      return;
    double lo = HUGE_VAL, hi = -HUGE_VAL;
    for (int ix = 0; ix < met->nx; ix++)
      for (int iy = 0; iy < met->ny; iy++) {
        const float v = met->f2[f][(size_t) ix * (size_t) met->sx2 + iy];
        if (!std::isfinite(v))
          return;
        lo = std::min(lo, (double) v);
        hi = std::max(hi, (double) v);
      }
    lo_out = lo;
    hi_out = hi;
  };
  extremes(MPHIP_PS, S.ps_min, S.ps_max);
  extremes(MPHIP_PBL, S.pbl_min, S.pbl_max);
  if (met->f2[MPHIP_PEL]) {
    double lo = HUGE_VAL;
    for (int ix = 0; ix < met->nx; ix++)
      for (int iy = 0; iy < met->ny; iy++) {
        const float v = met->f2[MPHIP_PEL][(size_t) ix * (size_t) met->sx2 + iy];
        if (v == v)   // (a NaN never starts convection: dmin(ptop, NaN) is a NaN and p >= NaN is false)
          lo = std::min(lo, (double) v);
      }
    S.pel_min = lo;
  }
  if (met->f2[MPHIP_PCT]) {
    double lo = HUGE_VAL;
    for (int ix = 0; ix < met->nx; ix++)
      for (int iy = 0; iy < met->ny; iy++) {
        const float v = met->f2[MPHIP_PCT][(size_t) ix * (size_t) met->sx2 + iy];
        if (std::isfinite(v))
          lo = std::min(lo, (double) v);
      }
    if (lo < HUGE_VAL)
      S.pct_min = lo;
  }
  for (int f = 0; f < MPHIP_N2D; f++) {
    S.has2[f] = met->f2[f] != nullptr;
    if (!S.has2[f])
      continue;
    if (new_grid || !S.f2[f])
      if (dev_alloc(ctx, &S.f2[f], ncol))
        return 1;
    if (met->sx2 == met->ny) {
      HIPCHK(hipMemcpyAsync(S.f2[f], met->f2[f], ncol * sizeof(float), hipMemcpyHostToDevice, stream));
    } else {
      if (met->sx2 < met->ny)
        return fail(ctx, "unsupported 2-D meteo stride");
      HIPCHK(hipMemcpy2DAsync(S.f2[f], (size_t) met->ny * sizeof(float), met->f2[f], (size_t) met->sx2 * sizeof(float),
                              (size_t) met->ny * sizeof(float), (size_t) met->nx, hipMemcpyHostToDevice, stream));
    }
  }
  return 0;
}

}   // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" {

size_t mphip_sizeof_ctl(void) {
  return sizeof(mphip_ctl_t);
}

size_t mphip_sizeof_met(void) {
  return sizeof(mphip_met_t);
}

const char *mphip_version(void) {
#if MPHIP_EXACT_DIV
  return "mptrac_amd 0.1 (gfx950, reference rounding)";   // libmptrac_hip_exact.so (mptrac_amd/build.py:EXACT_FLAGS)
#else
  return "mptrac_amd 0.1 (gfx950)";
#endif
}

int mphip_create(mphip_ctx **out, int device) {
  if (!out)
    return 1;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "mptrac_hip: no HIP device available (this back end has no CPU fallback)\n");
    return 2;
  }
  if (device < 0 || device >= ndev) {
    fprintf(stderr, "mptrac_hip: device %d out of range (%d devices)\n", device, ndev);
    return 3;
  }
  mphip_ctx *ctx = new mphip_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    fprintf(stderr, "mptrac_hip: cannot initialise device %d\n", device);
    delete ctx;
    return 4;
  }
  memset(&ctx->ctl, 0, sizeof(ctx->ctl));
  *out = ctx;
  return 0;
}

void mphip_destroy(mphip_ctx *ctx) {
  if (!ctx)
    return;
  (void) hipSetDevice(ctx->device);
  (void) hipStreamSynchronize(ctx->stream);
  if (ctx->comm)
    (void) rccl().CommDestroy(ctx->comm);
  if (ctx->uploader.joinable())
    ctx->uploader.join();
  if (ctx->copy_stream) {
    (void) hipStreamSynchronize(ctx->copy_stream);
    (void) hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->next_ready)
    (void) hipEventDestroy(ctx->next_ready);
  if (ctx->main_mark)
    (void) hipEventDestroy(ctx->main_mark);
  unpin_all(ctx, ctx->pinned);
  for (MetSlot *s : { &ctx->slot[0], &ctx->slot[1], &ctx->next }) {
    for (auto p : s->f3)
      dev_free(p);
    for (auto p : s->f2)
      dev_free(p);
  }
  dev_free(ctx->d_clim);
  for (auto p : ctx->d_zm)
    dev_free(p);
  for (auto p : ctx->d_ts)
    dev_free(p);
  dev_free(ctx->d_tracers);
  dev_free(ctx->d_axes);
  ctx->pk.release();
  ctx->pk_next.release();
  for (void *q : { (void *) ctx->rep_sk, (void *) ctx->rep_si, (void *) ctx->rep_fk, (void *) ctx->rep_fv, (void *) ctx->rep_mk[0],
                   (void *) ctx->rep_mk[1], (void *) ctx->rep_mi[0], (void *) ctx->rep_mi[1], (void *) ctx->rep_tiles, (void *) ctx->rep_nm })
    dev_free(q);
  if (ctx->ahead_stream) {
    (void) hipStreamSynchronize(ctx->ahead_stream);
    (void) hipStreamDestroy(ctx->ahead_stream);
    (void) hipEventDestroy(ctx->ahead_mark);
    (void) hipEventDestroy(ctx->ahead_done);
  }
  for (int k = 0; k < 2; k++) {
    dev_free(ctx->ahead_keys[k]);
    dev_free(ctx->ahead_vals[k]);
  }
  dev_free(ctx->ahead_counts);
  dev_free(ctx->ahead_dt);
  for (auto p : ctx->d_arr)
    dev_free(p);
  for (auto p : ctx->d_alt)
    dev_free(p);
  for (auto p : ctx->d_uvwp)
    dev_free(p);
  for (auto p : ctx->d_uvwp_alt)
    dev_free(p);
  dev_free(ctx->d_dt);
  dev_free(ctx->d_iso);
  dev_free(ctx->d_iso_alt);
  dev_free(ctx->d_kz);
  dev_free(ctx->d_kz_alt);
  dev_free(ctx->d_iso_ts);
  dev_free(ctx->d_iso_ps);
  dev_free(ctx->d_dt_alt);
  dev_free(ctx->d_ext);
  dev_free(ctx->d_ext_alt);
  for (int k = 0; k < 2; k++) {
    dev_free(ctx->d_keys[k]);
    dev_free(ctx->d_vals[k]);
  }
  dev_free(ctx->d_counts);
  dev_free(ctx->d_prec);
  dev_free(ctx->d_cell);
  dev_free(ctx->d_sums);
  dev_free(ctx->d_cnt);
  dev_free(ctx->d_depo_busy);
  dev_free(ctx->d_occ);
  dev_free(ctx->d_band);
  dev_free(ctx->d_band_cnt);
  if (ctx->h_occ)
    (void) hipHostFree(ctx->h_occ);
  dev_free(ctx->d_lists);
  dev_free(ctx->d_grid_kernel);
  dev_free(ctx->d_rec);
  if (ctx->h_sums)
    (void) hipHostFree(ctx->h_sums);
  for (auto e : ctx->ev)
    (void) hipEventDestroy(e);
  (void) hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *mphip_last_error(const mphip_ctx *ctx) {
  return ctx ? ctx->err.c_str() : "no context";
}

int mphip_update_ctl(mphip_ctx *ctx, const mphip_ctl_t *ctl) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx || !ctl)
    return fail(ctx, "null argument");
  if (ctl->nq < 0 || ctl->nq > MPHIP_NQ_MAX)
    return fail(ctx, "nq out of range");
  if (ctl->rng_type != 1)
    return fail(ctx, "only RNG_TYPE 1 (Squares) is implemented on the device");
  if (ctl->advect_vert_coord < 0 || ctl->advect_vert_coord > 3)
    return fail(ctx, "Set ADVECT_VERT_COORD to 0, 1, 2, or 3!");
  if (!(ctl->advect == 0 || ctl->advect == 1 || ctl->advect == 2 || ctl->advect == 4))
    return fail(ctx, "Set ADVECT to 1, 2, or 4!");
  // module_meteo runs after the loss / mixing / deposition modules here and is evaluated lazily: none of the
  // quantities it fills may be one those modules read or write
  for (int k = 0; k < MPHIP_NMQ; k++)
    if (ctl->qnt_met[k] >= 0)
      for (int other : { ctl->qnt_m, ctl->qnt_vmr, ctl->qnt_aoa, ctl->qnt_loss_rate, ctl->qnt_mloss_decay,
                         ctl->qnt_mloss_wet, ctl->qnt_mloss_dry, ctl->qnt_rp, ctl->qnt_rhop, ctl->qnt_ens,
                         ctl->qnt_zeta, ctl->qnt_eta, ctl->qnt_tracer[0], ctl->qnt_tracer[1], ctl->qnt_tracer[2],
                         ctl->qnt_tracer[3], ctl->qnt_tracer[4] })
        if (ctl->qnt_met[k] == other)
          return fail(ctx, "a module_meteo quantity shares its index with a mass / mixing-ratio / loss / particle quantity");
  if (ctx->have_ctl && flush_meteo(ctx))   // a deferred module_meteo belongs to the old parameters
    return 1;
  if (!ctx->have_ctl || ctx->ctl.advect_vert_coord != ctl->advect_vert_coord)
    ctx->packed_dirty = true;               // the model-level wind records carry zeta_dot or omega
  ctx->ctl = *ctl;
  ctx->have_ctl = true;
  return 0;
}

int mphip_update_clim(mphip_ctx *ctx, int ntime, int nlat, const double *tropo_time, const double *tropo_lat,
                      const double *tropo, int ld) {
  if (!ctx || !tropo_time || !tropo_lat || !tropo)
    return fail(ctx, "null argument");
  if (ntime < 2 || ntime > 12 || nlat < 2 || nlat > 73 || ld < nlat)
    return fail(ctx, "tropopause climatology dimensions out of range");
  HIPCHK(hipSetDevice(ctx->device));
  DevClim h;
  memset(&h, 0, sizeof(h));
  h.ntime = ntime;
  h.nlat = nlat;
  for (int i = 0; i < ntime; i++)
    h.time[i] = tropo_time[i];
  for (int j = 0; j < nlat; j++)
    h.lat[j] = tropo_lat[j];
  for (int i = 0; i < ntime; i++)
    for (int j = 0; j < nlat; j++)
      h.tropo[i][j] = tropo[(size_t) i * ld + j];
  for (int i = 0; i + 1 < ntime; i++)
    h.inv_dtime[i] = 1.0 / (h.time[i + 1] - h.time[i]);
  for (int j = 0; j + 1 < nlat; j++)
    h.inv_dlat[j] = 1.0 / (h.lat[j + 1] - h.lat[j]);
  if (!ctx->d_clim)
    HIPCHK(hipMalloc((void **) &ctx->d_clim, sizeof(DevClim)));
  HIPCHK(hipMemcpyAsync(ctx->d_clim, &h, sizeof(DevClim), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->have_clim = true;
  return 0;
}

int mphip_update_clim_zm(mphip_ctx *ctx, int which, int ntime, int np, int nlat, const double *time, const double *p,
                         const double *lat, const double *vmr) {
  if (!ctx || which < 0 || which >= MPHIP_NZM)
    return fail(ctx, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))   // (a deferred module_meteo reads the tables of its own step)
    return 1;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  dev_free(ctx->d_zm[which]);
  ctx->d_zm[which] = nullptr;
  memset(&ctx->zm[which], 0, sizeof(DevZm));
  if (ntime == 0)
    return 0;
  if (!time || !p || !lat || !vmr)
    return fail(ctx, "null argument");
  if (ntime < 2 || np < 2 || nlat < 2 || ntime > 4096 || np > 4096 || nlat > 4096)
    return fail(ctx, "climatology dimensions out of range");
  if (!(p[0] > p[1]))
    return fail(ctx, "Pressure data are not descending!");      // messages of read_clim_zm, mptrac.c:8768, 8774
  if (!(lat[0] < lat[1]))
    return fail(ctx, "Latitude data are not ascending!");
  const size_t nv = (size_t) ntime * (size_t) np * (size_t) nlat, n = (size_t) ntime + np + nlat + nv;
  double *d = nullptr;
  if (dev_alloc(ctx, &d, n))
    return 1;
  ctx->d_zm[which] = d;
  HIPCHK(hipMemcpy(d, time, (size_t) ntime * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + ntime, p, (size_t) np * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + ntime + np, lat, (size_t) nlat * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + ntime + np + nlat, vmr, nv * sizeof(double), hipMemcpyHostToDevice));
  DevZm &z = ctx->zm[which];
  z.time = d;
  z.p = d + ntime;
  z.lat = d + ntime + np;
  z.vmr = d + ntime + np + nlat;
  z.ntime = ntime;
  z.np = np;
  z.nlat = nlat;
  return 0;
}

int mphip_update_clim_ts(mphip_ctx *ctx, int which, int ntime, const double *time, const double *vmr) {
  if (!ctx || which < 0 || which >= MPHIP_NTR)
    return fail(ctx, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (ahead_drop(ctx))
    return 1;
  HIPCHK(hipStreamSynchronize(ctx->stream));   // kernels in flight may read the old series
  dev_free(ctx->d_ts[which]);
  ctx->d_ts[which] = nullptr;
  DevTracerSeries &h = ctx->h_tracers;
  h.time[which] = h.vmr[which] = nullptr;
  h.ntime[which] = 0;
  if (ntime != 0) {
    if (!time || !vmr)
      return fail(ctx, "null argument");
    if (ntime < 2)
      return fail(ctx, "Not enough data points!");                 // messages of read_clim_ts, mptrac.c:8712-8730
    for (int i = 1; i < ntime; i++)
      if (!(time[i] > time[i - 1]))
        return fail(ctx, "Time series must be ascending!");
    double *d = nullptr;
    if (dev_alloc(ctx, &d, 2 * (size_t) ntime))
      return 1;
    ctx->d_ts[which] = d;
    HIPCHK(hipMemcpy(d, time, (size_t) ntime * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d + ntime, vmr, (size_t) ntime * sizeof(double), hipMemcpyHostToDevice));
    h.time[which] = d;
    h.vmr[which] = d + ntime;
    h.ntime[which] = ntime;
  }
  if (!ctx->d_tracers)
    HIPCHK(hipMalloc((void **) &ctx->d_tracers, sizeof(DevTracerSeries)));
  HIPCHK(hipMemcpy(ctx->d_tracers, &h, sizeof(DevTracerSeries), hipMemcpyHostToDevice));
  return 0;
}

int mphip_update_met(mphip_ctx *ctx, int slot, const mphip_met_t *met) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx || !met || slot < 0 || slot > 1)
    return fail(ctx, "bad argument");
  if (met->nx < 2 || met->ny < 2 || met->np < 2 || !met->lon || !met->lat || !met->p)
    return fail(ctx, "meteo grid dimensions out of range");
  // index arithmetic of the kernels (cell_of / col_of): 24-bit factors, 31-bit cell index
  if ((long long) met->nx * met->ny >= (1LL << 24) || (long long) met->nx * met->ny * std::max(met->np, met->npl) >= (1LL << 31)
      || met->np >= (1 << 24))
    return fail(ctx, "meteo grid too large for the device index arithmetic");
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))   // a deferred module_meteo samples the snapshots it was scheduled with
    return 1;
  const int nml = met->npl > 0 ? met->npl : 0;
  const bool new_grid = (met->nx != ctx->nx || met->ny != ctx->ny || met->np != ctx->npl || nml != ctx->nml);
  if (new_grid) {
    if (ctx->slot[0].valid || ctx->slot[1].valid) {
      // mptrac_get_met: "Meteo grid dimensions do not match!" (mptrac.c:6543-6546)
      if (ctx->slot[(slot ^ 1) ^ ctx->flip].valid)
        return fail(ctx, "Meteo grid dimensions do not match!");
    }
    ctx->nx = met->nx;
    ctx->ny = met->ny;
    ctx->npl = met->np;
    ctx->nml = nml;
    if (ctx->uploader.joinable())
      ctx->uploader.join();
    if (ctx->copy_stream)
      HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    for (auto &q : ctx->next.f3) {
      dev_free(q);
      q = nullptr;
    }
    for (auto &q : ctx->next.f2) {
      dev_free(q);
      q = nullptr;
    }
    ctx->next.valid = false;
    ctx->next_pending = false;
    ctx->pk.release();
    ctx->pk_next.release();
    ctx->pk_next_valid = false;
  }
  if (ctx->next_pending) {   // a snapshot uploaded now changes what the prefetched pair would have been packed from
    if (ctx->uploader.joinable())
      ctx->uploader.join();
    HIPCHK(hipStreamSynchronize(ctx->copy_stream));
    ctx->pk_next_valid = false;
  }
  ctx->coord_type = met->coord_type;
  MetSlot &S = ctx->slot[slot ^ ctx->flip];
  S.lon.assign(met->lon, met->lon + met->nx);
  S.lat.assign(met->lat, met->lat + met->ny);
  S.p.assign(met->p, met->p + met->np);
  // the reference interpolates on met0's axes (mptrac.c:3010-3020); slot 0 defines them
  if (slot == 0 || new_grid || ctx->h_lon.empty()) {
    ctx->h_lon = S.lon;
    ctx->h_lat = S.lat;
    ctx->h_p = S.p;
    if (upload_axes(ctx))
      return 1;
  }
  if (upload_fields(ctx, S, met, new_grid, ctx->stream))
    return 1;
  HIPCHK(hipStreamSynchronize(ctx->stream));   // the host arrays may be reused by the caller
  S.time = met->time;
  S.valid = true;
  ctx->packed_dirty = true;
  return 0;
}

int mphip_swap_met(mphip_ctx *ctx) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx)
    return 1;
  if (flush_meteo(ctx))
    return 1;
  ctx->flip ^= 1;
  ctx->packed_dirty = true;
  return 0;
}

// Page-lock the caller's particle arrays (option pin_host_atm) with ONE registration spanning all of them:
// a copy must lie inside a single registration, and neighbouring arrays of one allocation share pages, so
// per-array registrations are not an option.  Only done when the arrays lie close together (the members of
// an atm_t); failure is not an error -- the copies then go through the runtime's staging buffers.
static void pin_host_span(mphip_ctx *ctx, const void *const *ptrs, int n, size_t bytes_each) {
  if (!ctx->pin_host_atm || !bytes_each)
    return;
  uintptr_t lo = ~(uintptr_t) 0, hi = 0;
  size_t total = 0;
  for (int k = 0; k < n; k++)
    if (ptrs[k]) {
      lo = std::min(lo, (uintptr_t) ptrs[k]);
      hi = std::max(hi, (uintptr_t) ptrs[k] + bytes_each);
      total += bytes_each;
    }
  if (!total || hi - lo > total + total / 2 + (1u << 20))
    return;   // scattered arrays
  std::lock_guard<std::mutex> guard(ctx->pinned_lock);
  for (auto &r : ctx->pinned)
    if (lo >= r.first && hi <= r.second)
      return;
    else if (lo < r.second && hi > r.first)
      return;   // partly covered by an earlier span: leave it alone
  if (hipHostRegister((void *) lo, hi - lo, hipHostRegisterDefault) == hipSuccess)
    ctx->pinned.emplace_back(lo, hi);
  else
    (void) hipGetLastError();
}

int mphip_prefetch_met(mphip_ctx *ctx, const mphip_met_t *met) {
  if (!ctx || !met)
    return fail(ctx, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  const int nml = met->npl > 0 ? met->npl : 0;
  if (!ctx->slot[0].valid || !ctx->slot[1].valid)
    return fail(ctx, "mphip_prefetch_met needs both snapshots on the device (mphip_update_met first)");
  if (met->nx != ctx->nx || met->ny != ctx->ny || met->np != ctx->npl || nml != ctx->nml)
    return fail(ctx, "Meteo grid dimensions do not match!");   // mptrac.c:6543-6546
  if (ctx->next_pending)
    return fail(ctx, "a prefetched snapshot is waiting for mphip_commit_met");
  if (!ctx->copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->next_ready, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->main_mark, hipEventDisableTiming));
  }
  // the staging arrays of `next` were met0 until the last commit: kernels queued before it may still read them
  HIPCHK(hipEventRecord(ctx->main_mark, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->main_mark, 0));
  // Copies from ordinary (pageable) host memory keep the calling thread busy until they are done, and
  // page-locking arrays the caller owns is fragile (a copy has to lie inside one registration, arrays of
  // one allocation share pages, the caller may free them): an uploader thread of the library issues them
  // instead.  mphip_commit_met / mphip_discard_prefetch join it.
  const mphip_met_t desc = *met;
  ctx->uploader_rc = 0;
  ctx->uploader_done = false;
  ctx->uploader_err.clear();
  ctx->next_pending = true;
  ctx->next.time = met->time;
  ctx->next.lon.assign(met->lon, met->lon + met->nx);
  ctx->next.lat.assign(met->lat, met->lat + met->ny);
  ctx->next.p.assign(met->p, met->p + met->np);
  // The packed grids of the step after the hand-over -- (today's met1, the new snapshot) -- are built on the copy
  // stream as well, into the second set of packed arrays: the hand-over then costs the stepping stream nothing.
  // Which fields the new snapshot carries is known from its description; the arrays are allocated here, on the
  // calling thread, the uploader thread only launches.  (Pressure-level configurations; the model-level records
  // depend on a monotonicity check that is read back, and on ADVECT_VERT_COORD: they are packed at the commit.)
  ctx->pk_next_valid = false;
  bool pack_ahead = ctx->nml == 0;
  PackPlan plan;
  if (pack_ahead) {
    MetSlot probe = ctx->next;   // (presence flags only)
    for (int f = 0; f < MPHIP_N3D; f++)
      probe.has3[f] = met->f3[f] != nullptr;
    for (int f = 0; f < MPHIP_N2D; f++)
      probe.has2[f] = met->f2[f] != nullptr;
    plan = pack_plan(ctx, ctx->slot[1 ^ ctx->flip], probe);
    if (pack_alloc(ctx, ctx->pk_next, plan))
      return 1;
  }
  ctx->uploader = std::thread([ctx, desc, pack_ahead, plan]() {
    int rc = hipSetDevice(ctx->device) == hipSuccess ? 0 : 1;
    if (!rc)
      rc = upload_fields(ctx, ctx->next, &desc, false, ctx->copy_stream);
    if (!rc && pack_ahead) {
      // met1 of today is met0 after the commit; its staging arrays are not written while it is a current slot
      rc = pack_launch(ctx, ctx->pk_next, plan, ctx->slot[1 ^ ctx->flip], ctx->next, ctx->copy_stream);
      ctx->pk_next_valid = rc == 0;
    }
    if (!rc && hipEventRecord(ctx->next_ready, ctx->copy_stream) != hipSuccess)
      rc = 1;
    ctx->uploader_rc = rc;
    ctx->uploader_done = true;
  });
  return 0;
}

// waits for the uploader thread; 0 = the snapshot is on its way (next_ready recorded)
static int join_uploader(mphip_ctx *ctx) {
  if (ctx->uploader.joinable())
    ctx->uploader.join();
  if (ctx->uploader_rc) {
    ctx->next_pending = false;
    ctx->next.valid = false;
    return fail(ctx, "upload of the prefetched meteo snapshot failed: " + ctx->err);
  }
  return 0;
}

int mphip_commit_met(mphip_ctx *ctx) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx)
    return 1;
  if (!ctx->next_pending)
    return fail(ctx, "no prefetched snapshot to commit");
  HIPCHK(hipSetDevice(ctx->device));
  if (join_uploader(ctx) || flush_meteo(ctx))
    return 1;
  // kernels launched from here on wait for the upload; the host does not
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->next_ready, 0));
  ctx->next.valid = true;
  std::swap(ctx->slot[0 ^ ctx->flip], ctx->next);   // the old met0 arrays become the next staging slot
  ctx->flip ^= 1;                                    // old met1 -> met0, prefetched -> met1
  ctx->next.valid = false;
  ctx->next_pending = false;
  if (ctx->pk_next_valid) {
    // the packed grids of (new met0, new met1) are ready on the copy stream (the stepping stream waits for
    // next_ready above): the two sets trade places
    std::swap(ctx->pk, ctx->pk_next);
    ctx->pk_next_valid = false;
    ctx->packed_dirty = false;
    if (axes_follow_met0(ctx))
      return 1;
  } else {
    ctx->packed_dirty = true;
  }
  return 0;
}

int mphip_discard_prefetch(mphip_ctx *ctx) {
  if (!ctx)
    return 1;
  if (!ctx->next_pending)
    return 0;
  HIPCHK(hipSetDevice(ctx->device));
  (void) join_uploader(ctx);
  HIPCHK(hipStreamSynchronize(ctx->copy_stream));
  ctx->next.valid = false;
  ctx->next_pending = false;
  ctx->pk_next_valid = false;
  return 0;
}

int mphip_prefetch_done(mphip_ctx *ctx) {
  if (!ctx || !ctx->next_pending)
    return 1;
  // the uploader thread has handed its last copy to the copy stream and that copy has finished
  return ctx->uploader_done && hipEventQuery(ctx->next_ready) == hipSuccess ? 1 : 0;
}

int mphip_update_atm(mphip_ctx *ctx, long long np, long long ip0, long long np_total, int nq, const double *time,
                     const double *p, const double *lon, const double *lat, const double *const *q) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx || np < 0 || nq < 0 || nq > MPHIP_NQ_MAX || ip0 < 0 || np_total < ip0 + np)
    return fail(ctx, "bad particle counts");
  if (np > 2147483647LL)
    return fail(ctx, "too many particles for one device context");
  if (np && (!time || !p || !lon || !lat || (nq && !q)))
    return fail(ctx, "null particle array");
  HIPCHK(hipSetDevice(ctx->device));
  const bool fresh = (np != ctx->np || nq != ctx->nq || !ctx->d_arr[0]);
  if (fresh) {
    const size_t n = (size_t) std::max<long long>(np, 1);
    for (int k = 0; k < 4 + MPHIP_NQ_MAX; k++) {
      const size_t want = k < 4 + nq ? n : 0;
      if (dev_alloc(ctx, &ctx->d_arr[k], want) || dev_alloc(ctx, &ctx->d_alt[k], want))
        return 1;
    }
    for (int k = 0; k < 3; k++) {
      if (dev_alloc(ctx, &ctx->d_uvwp[k], n) || dev_alloc(ctx, &ctx->d_uvwp_alt[k], n))
        return 1;
      HIPCHK(hipMemsetAsync(ctx->d_uvwp[k], 0, n * sizeof(float), ctx->stream));   // calloc'ed cache_t
    }
    if (dev_alloc(ctx, &ctx->d_dt, n) || dev_alloc(ctx, &ctx->d_dt_alt, n))
      return 1;
    HIPCHK(hipMemsetAsync(ctx->d_dt, 0, n * sizeof(double), ctx->stream));
    if (dev_alloc(ctx, &ctx->d_ext, n) || dev_alloc(ctx, &ctx->d_ext_alt, n))
      return 1;
    for (int k = 0; k < 2; k++)
      if (dev_alloc(ctx, &ctx->d_keys[k], n) || dev_alloc(ctx, &ctx->d_vals[k], n))
        return 1;
    if (dev_alloc(ctx, &ctx->d_cell, n))
      return 1;
    // the buffers of the sort that runs ahead trade places with d_keys / d_vals / d_dt when a sort is adopted,
    // so after that they have whatever size those had: one capacity cannot describe them -- start over
    for (int k = 0; k < 2; k++)
      if (dev_alloc(ctx, &ctx->ahead_keys[k], 0) || dev_alloc(ctx, &ctx->ahead_vals[k], 0))
        return 1;
    if (dev_alloc(ctx, &ctx->ahead_dt, 0))
      return 1;
    ctx->ahead_cap = 0;
    // ... and so do two buffers of the order repair (repair_sort swaps rep_fk / rep_fv with d_keys[0] / d_vals[0])
    if (dev_alloc(ctx, &ctx->rep_sk, 0) || dev_alloc(ctx, &ctx->rep_si, 0) || dev_alloc(ctx, &ctx->rep_fk, 0)
        || dev_alloc(ctx, &ctx->rep_fv, 0) || dev_alloc(ctx, &ctx->rep_mk[0], 0) || dev_alloc(ctx, &ctx->rep_mk[1], 0)
        || dev_alloc(ctx, &ctx->rep_mi[0], 0) || dev_alloc(ctx, &ctx->rep_mi[1], 0))
      return 1;
    ctx->rep_cap = 0;
    dev_free(ctx->d_iso);
    dev_free(ctx->d_iso_alt);
    ctx->d_iso = ctx->d_iso_alt = nullptr;
    dev_free(ctx->d_kz);
    dev_free(ctx->d_kz_alt);
    ctx->d_kz = ctx->d_kz_alt = nullptr;
    ctx->sorted_buf = -1;
    ctx->stored_is_sorted = false;
  }
  ctx->depo_busy_valid = false;
  ctx->meteo_pending = false;   // the uploaded quantity arrays replace whatever module_meteo would have written
  if (!fresh && restore_external_order(ctx))   // keep cache->uvwp with its slot across a re-upload
    return 1;
  ctx->ext_identity = true;
  ctx->stored_is_sorted = false;     // (the caller's particles: in no order this library knows of)
  ctx->steps_since_resort = 1 << 30;
  ctx->np = np;
  ctx->nq = nq;
  ctx->ip0 = ip0;
  ctx->np_total = np_total;
  const double *src[4] = { time, p, lon, lat };
  if (np) {
    const void *all[4 + MPHIP_NQ_MAX] = { time, p, lon, lat };
    for (int iq = 0; iq < nq; iq++)
      all[4 + iq] = q[iq];
    pin_host_span(ctx, all, 4 + nq, (size_t) np * sizeof(double));
  }
  for (int k = 0; k < 4 && np; k++)
    HIPCHK(hipMemcpyAsync(ctx->d_arr[k], src[k], (size_t) np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  for (int iq = 0; iq < nq && np; iq++) {
    if (!q[iq])
      return fail(ctx, "null quantity array");
    HIPCHK(hipMemcpyAsync(ctx->d_arr[4 + iq], q[iq], (size_t) np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_update_quantity(mphip_ctx *ctx, int iq, const double *q) {
  if (!ctx || !q)
    return fail(ctx, "null argument");
  if (iq < 0 || iq >= ctx->nq)
    return fail(ctx, "no such quantity");
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))   // (a pending module_meteo may be about to write this array)
    return 1;
  if (ctx->np == 0)
    return 0;
  const size_t bytes = (size_t) ctx->np * sizeof(double);
  if (ctx->ext_identity) {
    HIPCHK(hipMemcpyAsync(ctx->d_arr[4 + iq], q, bytes, hipMemcpyHostToDevice, ctx->stream));
  } else {   // through the alternate buffer (free between two sorts), then into the stored order
    HIPCHK(hipMemcpyAsync(ctx->d_alt[4 + iq], q, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(gather_by_ext_kernel, dim3(grid_for(ctx->np)), dim3(256), 0, ctx->stream, ctx->d_alt[4 + iq],
                       ctx->d_ext, ctx->d_arr[4 + iq], ctx->np);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_get_atm(mphip_ctx *ctx, double *time, double *p, double *lon, double *lat, double *const *q) {
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))
    return 1;
  // The caller's order is produced in the alternate buffers (free between two sorts); the resident
  // arrays keep their internal order, so an output costs one scatter pass and no re-sort.
  double *const *srcs = ctx->d_arr;
  if (!ctx->ext_identity && ctx->np) {
    PermArgs g = perm_args(ctx, false);
    if (permute_random(ctx, g, ctx->d_ext, ctx->np, true))
      return 1;
    srcs = ctx->d_alt;
  }
  double *dst[4] = { time, p, lon, lat };
  if (ctx->np) {
    const void *all[4 + MPHIP_NQ_MAX] = { time, p, lon, lat };
    for (int iq = 0; iq < ctx->nq; iq++)
      all[4 + iq] = q ? q[iq] : nullptr;
    pin_host_span(ctx, all, 4 + ctx->nq, (size_t) ctx->np * sizeof(double));
  }
  for (int k = 0; k < 4; k++)
    if (dst[k] && ctx->np)
      HIPCHK(hipMemcpyAsync(dst[k], srcs[k], (size_t) ctx->np * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  for (int iq = 0; iq < ctx->nq; iq++)
    if (q && q[iq] && ctx->np)
      HIPCHK(hipMemcpyAsync(q[iq], srcs[4 + iq], (size_t) ctx->np * sizeof(double), hipMemcpyDeviceToHost,
                            ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_update_cache(mphip_ctx *ctx, const float *uvwp, const uint64_t *rng_ctr) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  if (rng_ctr)
    ctx->rng_ctr = *rng_ctr;
  if (uvwp && restore_external_order(ctx))
    return 1;
  if (uvwp && ctx->np) {
    std::vector<float> tmp((size_t) ctx->np);
    for (int k = 0; k < 3; k++) {
      for (long long i = 0; i < ctx->np; i++)
        tmp[(size_t) i] = uvwp[3 * (size_t) i + k];
      HIPCHK(hipMemcpy(ctx->d_uvwp[k], tmp.data(), tmp.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  return 0;
}

int mphip_get_cache(mphip_ctx *ctx, float *uvwp, double *dt, uint64_t *rng_ctr) {
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  if ((uvwp || dt) && restore_external_order(ctx))
    return 1;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (rng_ctr)
    *rng_ctr = ctx->rng_ctr;
  if (uvwp && ctx->np) {
    std::vector<float> tmp((size_t) ctx->np);
    for (int k = 0; k < 3; k++) {
      HIPCHK(hipMemcpy(tmp.data(), ctx->d_uvwp[k], tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
      for (long long i = 0; i < ctx->np; i++)
        uvwp[3 * (size_t) i + k] = tmp[(size_t) i];
    }
  }
  if (dt && ctx->np)
    HIPCHK(hipMemcpy(dt, ctx->d_dt, (size_t) ctx->np * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int mphip_update_iso(mphip_ctx *ctx, const double *iso_var, const double *iso_ts, const double *iso_ps, int iso_n) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  if (iso_var && ctx->np) {
    if (restore_external_order(ctx) || ensure_iso(ctx))
      return 1;
    HIPCHK(hipMemcpyAsync(ctx->d_iso, iso_var, (size_t) ctx->np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  if (iso_ts && iso_ps) {
    if (iso_n < 1)
      return fail(ctx, "Could not read any data!");   // module_isosurf_init, mptrac.c:4943-4944
    if (dev_alloc(ctx, &ctx->d_iso_ts, (size_t) iso_n) || dev_alloc(ctx, &ctx->d_iso_ps, (size_t) iso_n))
      return 1;
    HIPCHK(hipMemcpyAsync(ctx->d_iso_ts, iso_ts, (size_t) iso_n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_iso_ps, iso_ps, (size_t) iso_n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ctx->iso_n = iso_n;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_get_iso(mphip_ctx *ctx, double *iso_var) {
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  if (iso_var && ctx->np) {
    if (!ctx->d_iso)
      return fail(ctx, "cache->iso_var is not on the device (no isosurface mode has run)");
    if (restore_external_order(ctx))
      return 1;
    HIPCHK(hipMemcpyAsync(iso_var, ctx->d_iso, (size_t) ctx->np * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return 0;
}

int mphip_run_timestep(mphip_ctx *ctx, double t) {
  if (!ctx)
    return 1;
  if (!ctx->have_ctl)
    return fail(ctx, "control parameters were not uploaded");
  HIPCHK(hipSetDevice(ctx->device));
  const mphip_ctl_t &c = ctx->ctl;
  const uint64_t n = (uint64_t) ctx->np_total;
  unsigned mask = MPHIP_MOD_TIMESTEPS;
  // a deferred module_meteo of the previous step: dropped if this step runs module_meteo again (nobody
  // saw its values), evaluated now -- before the particles move -- otherwise
  {
    const bool again = c.met_dt_out > 0 && (c.met_dt_out < c.dt_mod || fmod(t, c.met_dt_out) == 0);
    if (again)
      ctx->meteo_pending = false;
    else if (flush_meteo(ctx))
      return 1;
  }

  // module_isosurf_init and module_advect_init at the first call (mptrac.c:7863-7870)
  if (t == c.t_start) {
    unsigned init = 0;
    if (c.isosurf >= 1 && c.isosurf <= 3)
      init |= MPHIP_MOD_ISOSURF_INIT;
    if (c.advect_vert_coord == 1)
      init |= MPHIP_MOD_ADVECT_INIT;
    if (init && launch_step(ctx, init, t, 0, 0, 0))
      return 1;
  }

  // module_timesteps + module_sort (mptrac.c:7877-7881).  The reference
  // permutes atm but not cache->dt, so on sort steps dt is computed per slot
  // before the sort and read back per slot afterwards.
  if (c.sort_dt > 0 && fmod(t, c.sort_dt) == 0) {
    if (restore_external_order(ctx) || do_sort(ctx, &t))
      return 1;
    mask = 0;
  } else if (ctx->locality_interval > 0 && ctx->steps_since_resort >= ctx->locality_interval) {
    if (locality_sort(ctx))
      return 1;
  }
  if (ctx->steps_since_resort < (1 << 29))
    ctx->steps_since_resort++;
  mask |= MPHIP_MOD_POSITION;
  if (c.advect > 0)
    mask |= MPHIP_MOD_ADVECT;
  uint64_t ctr_turb = 0, ctr_meso = 0, ctr_conv = 0;
  if (c.diffusion
      && (c.turb_dx_pbl > 0 || c.turb_dz_pbl > 0 || c.turb_dx_trop > 0 || c.turb_dz_trop > 0 || c.turb_dx_strat > 0
          || c.turb_dz_strat > 0)) {
    mask |= MPHIP_MOD_DIFF_TURB;
    ctr_turb = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;   // module_rng(..., 3 * np, 1), mptrac.c:4600, 5812
  }
  uint64_t ctr_pbl = 0;
  if (c.diffusion && c.turb_pbl_scheme == 1) {
    mask |= MPHIP_MOD_DIFF_PBL;
    ctr_pbl = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;   // module_rng(..., 3 * np, 1), mptrac.c:4354
  }
  if (c.diffusion && (c.turb_mesox > 0 || c.turb_mesoz > 0)) {
    mask |= MPHIP_MOD_DIFF_MESO;
    ctr_meso = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;
  }
  if ((c.conv_mix_pbl || c.conv_cape >= 0) && (c.conv_dt <= 0 || fmod(t, c.conv_dt) == 0)) {
    mask |= MPHIP_MOD_CONVECTION;
    ctr_conv = ctx->rng_ctr;
    ctx->rng_ctr += n + 1;       // module_rng(..., np, 0), mptrac.c:4113
  }
  if (c.qnt_rp >= 0 && c.qnt_rhop >= 0)
    mask |= MPHIP_MOD_SEDI;
  if (c.isosurf >= 1 && c.isosurf <= 4)
    mask |= MPHIP_MOD_ISOSURF;
  mask |= MPHIP_MOD_POSITION2;
  const bool bound = c.bound_lat0 < c.bound_lat1 && c.bound_p0 > c.bound_p1;
  if (bound)
    mask |= MPHIP_MOD_BOUND_COND;
  if (c.qnt_loss_rate >= 0)
    mask |= MPHIP_MOD_LOSS_ZERO;
  if (c.tdec_trop > 0 && c.tdec_strat > 0)
    mask |= MPHIP_MOD_DECAY;
  unsigned tail = 0;
  if ((c.wet_depo_ic_a > 0 || c.wet_depo_ic_h[0] > 0) && (c.wet_depo_bc_a > 0 || c.wet_depo_bc_h[0] > 0))
    tail |= MPHIP_MOD_WET_DEPO;
  if (c.dry_depo_vdep > 0)
    tail |= MPHIP_MOD_DRY_DEPO;
  if (bound)
    tail |= MPHIP_MOD_BOUND_COND2;
  // module_meteo (mptrac.c:7921-7924) sits between the final module_position and the loss / decay /
  // mixing / deposition modules; those neither move particles nor touch a quantity it sets, so it
  // runs after them here (own kernel, every particle)
  const bool meteo_now = c.met_dt_out > 0 && (c.met_dt_out < c.dt_mod || fmod(t, c.met_dt_out) == 0);
  const bool mixing_now = c.mixing_trop >= 0 && c.mixing_strat >= 0 && (c.mixing_dt <= 0 || fmod(t, c.mixing_dt) == 0);
  // module_sort of the next step can start as soon as this step's particles have moved (sort_ahead)
  const double t_next = t + c.direction * c.dt_mod;
  const bool sort_next = ctx->sort_ahead && ctx->np > 0 && ctx->ext_identity && c.sort_dt > 0
    && fmod(t_next, c.sort_dt) == 0 && c.direction * (t_next - c.t_stop) <= 0;
  if (!mixing_now) {
    if (launch_step(ctx, mask | tail, t, ctr_turb, ctr_meso, ctr_conv, ctr_pbl))
      return 1;
    if (sort_next && ahead_launch(ctx, t_next))
      return 1;
    return meteo_now ? schedule_meteo(ctx) : 0;
  }
  if (tail && (mask & MPHIP_MOD_TIMESTEPS))
    mask |= kStoreDt;
  // the keys of the sort ahead, its module_timesteps and module_mixing's box index from the launch that moves the
  // particles (EmitKeys) instead of a kernel of their own behind it, where an instantiation for it exists
  EmitKeys ek;
  memset(&ek, 0, sizeof(ek));
  bool emitted = false;
  if (sort_next && ctx->ahead_box && ctx->emit_keys) {
    MixPlan plan;
    bool act = false;
    if (mixing_plan(ctx, t, &plan, &act) || ahead_ensure(ctx))
      return 1;
    if (act) {
      ek.keys = ctx->ahead_keys[0];
      ek.dt_next = ctx->ahead_dt;
      ek.cell = ctx->d_cell;
      ek.grid = plan.box.grid;
      ek.box_t0 = plan.box.t0;
      ek.box_t1 = plan.box.t1;
      ek.ens = plan.box.ens;
      ek.ngrid = plan.box.ngrid;
      ek.direction = (double) c.direction;
      ek.t_start = c.t_start;
      ek.t_stop = c.t_stop;
      ek.t_next = t_next;
      constexpr unsigned kDepo = MPHIP_MOD_WET_DEPO | MPHIP_MOD_DRY_DEPO;
#ifndef MPHIP_DEPO_FLAGS
#define MPHIP_DEPO_FLAGS 1      // 0: experiment -- the deposition launch decides from dt, time, p itself (profiles/r06_ab_depo_flags.txt)
#endif
      if (MPHIP_DEPO_FLAGS && (tail & kDepo) && !(tail & ~kDepo) && ctx->compact_depo) {   // the deposition launch behind module_mixing will be depo_kernel
        if (ctx->np > ctx->depo_busy_cap) {
          if (dev_alloc(ctx, &ctx->d_depo_busy, (size_t) ctx->np))
            return 1;
          ctx->depo_busy_cap = ctx->np;
        }
        ek.depo_busy = ctx->d_depo_busy;
        ek.depo_mask = tail;
      }
    }
  }
  if (launch_step(ctx, mask, t, ctr_turb, ctr_meso, ctr_conv, ctr_pbl, 1, 0, 0, &ek, &emitted))
    return 1;
  bool cells_ready = false;
  if (sort_next) {   // ... beside module_mixing and the deposition launch
    MixPlan plan;
    if (mixing_plan(ctx, t, &plan, &cells_ready))
      return 1;
    const BoxArgs box = { ctx->d_cell, plan.box.grid, plan.box.t0, plan.box.t1, plan.box.ens, plan.box.ngrid };
    if (!ctx->ahead_box)
      cells_ready = false;
    if (ahead_launch(ctx, t_next, cells_ready ? &box : nullptr, emitted && cells_ready))
      return 1;
  }
  if (do_mixing(ctx, t, cells_ready))
    return 1;
  if (tail && launch_step(ctx, tail, t, 0, 0, 0, 0))
    return 1;
  return meteo_now ? schedule_meteo(ctx) : 0;
}

// mphip_run_timesteps: n calls of mphip_run_timestep at t_first, t_first + stride, ... (stride = direction * DT_MOD,
// accumulated as the loop of trac.c:208 accumulates it).  Runs of steps with nothing between them -- no module_sort,
// no mixing, no module_meteo, no rarely used module, convection (if any) in every step, the internal re-sort not
// due -- and a lean instantiation of their module set go to the device as ONE launch in which every particle takes
// its steps one after the other: at small particle counts a step is shorter than a kernel launch.  Everything else
// takes the steps one by one; the results are the same either way (tests/test_gpu_parity.py).
int mphip_run_timesteps(mphip_ctx *ctx, double t_first, int nsteps) {
  if (!ctx)
    return 1;
  if (!ctx->have_ctl)
    return fail(ctx, "control parameters were not uploaded");
  if (nsteps < 0)
    return fail(ctx, "mphip_run_timesteps: negative step count");
  const mphip_ctl_t &c = ctx->ctl;
  const double stride = c.direction * c.dt_mod;
  double t = t_first;
  int done = 0;
  while (done < nsteps) {
    // how many steps from here on can share a launch?
    int batch = 0;
    // module_meteo (lazy: evaluated when its values can be seen, dropped when the next step schedules it again --
    // mphip_run_timestep) lets a batch run on, except that it must be evaluated between a step that schedules it and a
    // step that does not: a batch ends behind such a step.  (module_meteo without a quantity to fill does nothing.)
    const bool meteo = meteo_requested(c) && c.met_dt_out > 0;
    auto meteo_at = [&](double tt) { return meteo && (c.met_dt_out < c.dt_mod || fmod(tt, c.met_dt_out) == 0); };
    // module_sort and module_mixing run at multiples of SORT_DT / MIXING_DT: such a step takes the single-step path
    // (which sorts, or splits its launch around the mixing), the steps between two of them can share launches
    // (module_convection with CONV_DT > 0 likewise: due steps on their own, the steps between without it)
    const bool conv_on = c.conv_mix_pbl || c.conv_cape >= 0;
    auto scheduled = [&](double tt) {
      return (c.sort_dt > 0 && fmod(tt, c.sort_dt) == 0)
        || (c.mixing_trop >= 0 && c.mixing_strat >= 0 && (c.mixing_dt <= 0 || fmod(tt, c.mixing_dt) == 0))
        || (conv_on && c.conv_dt > 0 && fmod(tt, c.conv_dt) == 0);
    };
    const bool quiet = ctx->multi_step && ctx->np > 0 && t != c.t_start && !scheduled(t)
      && c.advect > 0   // (every integrator has its multi-step instantiations; without module_advect: single steps)
      && !ctx->fused_perm
      && !ctx->force_generic;
    if (quiet) {
      batch = nsteps - done;
      if (ctx->locality_interval > 0)
        batch = std::min(batch, ctx->locality_interval - ctx->steps_since_resort);
      batch = std::min(batch, ctx->multi_step);
      double tt = t;
      for (int j = 0; j < batch; j++) {
        if (j > 0 && scheduled(tt)) {   // the batch ends before this step
          batch = j;
          break;
        }
        const double tn = tt + stride;
        if (meteo_at(tt) && (!ctx->lazy_meteo || !meteo_at(tn))) {   // ... behind this one
          batch = j + 1;
          break;
        }
        tt = tn;
      }
    }
    if (batch < 2) {
      if (mphip_run_timestep(ctx, t))
        return 1;
      t += stride;
      done++;
      continue;
    }
    // the module set and the counters of mphip_run_timestep, for `batch` steps at once
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t) ctx->np_total;
    unsigned mask = MPHIP_MOD_TIMESTEPS | MPHIP_MOD_POSITION | MPHIP_MOD_ADVECT | MPHIP_MOD_POSITION2;
    uint64_t per_step = 0, off_turb = 0, off_meso = 0, off_conv = 0, off_pbl = 0;
    if (c.diffusion
        && (c.turb_dx_pbl > 0 || c.turb_dz_pbl > 0 || c.turb_dx_trop > 0 || c.turb_dz_trop > 0 || c.turb_dx_strat > 0
            || c.turb_dz_strat > 0)) {
      mask |= MPHIP_MOD_DIFF_TURB;
      off_turb = per_step;
      per_step += 3 * n + 1;
    }
    const bool pbl_closure = c.diffusion && c.turb_pbl_scheme == 1;
    if (pbl_closure) {     // (between module_diff_turb and module_diff_meso, mptrac.c:7893-7901)
      off_pbl = per_step;
      per_step += 3 * n + 1;
    }
    if (c.diffusion && (c.turb_mesox > 0 || c.turb_mesoz > 0)) {
      mask |= MPHIP_MOD_DIFF_MESO;
      off_meso = per_step;
      per_step += 3 * n + 1;
    }
    if (conv_on && !(c.conv_dt > 0)) {   // (CONV_DT > 0: the steps of a batch are the ones without convection)
      mask |= MPHIP_MOD_CONVECTION;
      off_conv = per_step;
      per_step += n + 1;
    }
    if (c.qnt_rp >= 0 && c.qnt_rhop >= 0)
      mask |= MPHIP_MOD_SEDI;
    const unsigned movers = mask;
    // winds from the model levels: the model-level instantiation (any of these module sets), if the height columns
    // are monotonic; pressure levels: the exact lean instantiations
    const bool ml_winds = c.advect_vert_coord >= 1 && c.advect_vert_coord <= 3;
    if (ml_winds && ensure_packed(ctx))
      return 1;
    const bool lean_ok = lean32_ok(ctx) || lean64_ok(ctx);   // (launch_step's conditions; 64-bit offsets: the gated instantiations)
    // module_bound_cond (per particle: its own time, the tracer series on the device) is switched at run time in the
    // gated instantiations (pressure and model levels)
    const bool bound = c.bound_lat0 < c.bound_lat1 && c.bound_p0 > c.bound_p1;
    const bool exact = ml_winds ? ctx->pk.ml_monotonic && ctx->nml <= kLockstepMaxLevels && ctx->d_kz != nullptr && (!bound || lean_ok)
                                : lean_ok && (movers & ~kOptionalModules) == kAdv;   // (exact sets: their own kernels; subsets: the gated one)
    // the closure inside the boundary layer has lean instantiations for pressure-level winds on grids within 32-bit
    // offsets; anything else with it takes the general kernel, one step per launch
    const bool isosurf = c.isosurf >= 1 && c.isosurf <= 4;     // (module_isosurf: the same instantiations)
    const bool closure_ok = !(pbl_closure || isosurf) || (!ml_winds && lean32_ok(ctx));
    if (pbl_closure)
      mask |= MPHIP_MOD_DIFF_PBL;
    if (isosurf)
      mask |= MPHIP_MOD_ISOSURF;
    if (bound)
      mask |= MPHIP_MOD_BOUND_COND | MPHIP_MOD_BOUND_COND2;
    if (c.qnt_loss_rate >= 0)
      mask |= MPHIP_MOD_LOSS_ZERO;
    if (c.tdec_trop > 0 && c.tdec_strat > 0)
      mask |= MPHIP_MOD_DECAY;
    if ((c.wet_depo_ic_a > 0 || c.wet_depo_ic_h[0] > 0) && (c.wet_depo_bc_a > 0 || c.wet_depo_bc_h[0] > 0))
      mask |= MPHIP_MOD_WET_DEPO;
    if (c.dry_depo_vdep > 0)
      mask |= MPHIP_MOD_DRY_DEPO;
    if (!exact || !closure_ok) {     // no multi-step instantiation of this module set
      if (mphip_run_timestep(ctx, t))
        return 1;
      t += stride;
      done++;
      continue;
    }
    // a deferred module_meteo of the step before: dropped if the first step of the batch schedules it again,
    // evaluated now -- before the particles move -- otherwise (as mphip_run_timestep)
    if (meteo_at(t))
      ctx->meteo_pending = false;
    else if (flush_meteo(ctx))
      return 1;
    if (launch_step(ctx, mask, t, ctx->rng_ctr + off_turb, ctx->rng_ctr + off_meso, ctx->rng_ctr + off_conv,
                    ctx->rng_ctr + off_pbl, batch, stride, per_step))
      return 1;
    ctx->rng_ctr += per_step * (uint64_t) batch;
    if (ctx->steps_since_resort < (1 << 29))
      ctx->steps_since_resort += batch;
    double t_last = t;
    for (int k = 0; k < batch; k++) {
      t_last = t;
      t += stride;
    }
    done += batch;
    if (meteo_at(t_last) && schedule_meteo(ctx))   // (only the last step of a batch can leave one to be seen)
      return 1;
  }
  return 0;
}

int mphip_module(mphip_ctx *ctx, unsigned modules, double t) {
  if (ctx && ahead_drop(ctx))
    return 1;
  if (!ctx)
    return 1;
  if (!ctx->have_ctl)
    return fail(ctx, "control parameters were not uploaded");
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))
    return 1;
  if (modules == MPHIP_MOD_SORT)
    return do_sort(ctx);
  if (modules == MPHIP_MOD_MIXING)
    return do_mixing(ctx, t);
  if (modules == MPHIP_MOD_METEO)
    return launch_meteo(ctx);
  if (modules & ~kParticleBits)
    return fail(ctx, "module_sort / module_mixing / module_meteo must be called on their own");
  const uint64_t n = (uint64_t) ctx->np_total;
  uint64_t ctr_turb = 0, ctr_meso = 0, ctr_conv = 0, ctr_pbl = 0;
  if (modules & MPHIP_MOD_DIFF_TURB) {
    ctr_turb = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;
  }
  if (modules & MPHIP_MOD_DIFF_PBL) {
    ctr_pbl = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;
  }
  if (modules & MPHIP_MOD_DIFF_MESO) {
    ctr_meso = ctx->rng_ctr;
    ctx->rng_ctr += 3 * n + 1;
  }
  if (modules & MPHIP_MOD_CONVECTION) {
    ctr_conv = ctx->rng_ctr;
    ctx->rng_ctr += n + 1;
  }
  if (modules & MPHIP_MOD_TIMESTEPS)
    modules |= kStoreDt;
  return launch_step(ctx, modules, t, ctr_turb, ctr_meso, ctr_conv, ctr_pbl);
}

int mphip_get_sort(mphip_ctx *ctx, double *keys, int *perm) {
  if (!ctx || ctx->sorted_buf < 0)
    return fail(ctx, "module_sort has not been run");
  HIPCHK(hipSetDevice(ctx->device));
  const long long n = ctx->np;
  if (perm)
    HIPCHK(hipMemcpyAsync(perm, ctx->d_vals[ctx->sorted_buf], (size_t) n * sizeof(int), hipMemcpyDeviceToHost,
                          ctx->stream));
  if (keys) {
    double *d_tmp = nullptr;
    HIPCHK(hipMalloc((void **) &d_tmp, (size_t) n * sizeof(double)));
    hipLaunchKernelGGL(keys_to_double_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream,
                       ctx->d_keys[ctx->sorted_buf], d_tmp, n);
    HIPCHK(hipMemcpyAsync(keys, d_tmp, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_tmp));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_grid_sums(mphip_ctx *ctx, double t, int *cnt, double *mean, double *sigma) {
  if (!ctx || !cnt || !mean || !sigma)
    return fail(ctx, "null argument");
  if (!ctx->have_ctl)
    return fail(ctx, "control parameters were not uploaded");
  HIPCHK(hipSetDevice(ctx->device));
  if (flush_meteo(ctx))
    return 1;
  const mphip_ctl_t &c = ctx->ctl;
  const size_t ncell = (size_t) c.grid_nx * c.grid_ny * c.grid_nz;
  const size_t total = ncell * (size_t) (1 + 2 * ctx->nq);
  if (ensure_sums(ctx, total))
    return 1;
  const bool ordered = ctx->deterministic_sums != 0;
  const DevAtm a = dev_atm(ctx);
  const GridKernel kern = { ctx->d_grid_kernel, ctx->d_grid_kernel ? ctx->d_grid_kernel + ctx->grid_nk : nullptr, a.p,
                            ctx->grid_nk };
  const BoxGrid G = { c.grid_lon0, c.grid_lon1, c.grid_lat0, c.grid_lat1, c.grid_z0, c.grid_z1, c.grid_nx, c.grid_ny,
                      c.grid_nz };
  const GridBoxArgs gbox = { G, t - 0.5 * c.dt_mod, t + 0.5 * c.dt_mod, a.time, a.lon, a.lat, a.p };
  if (ctx->np && !ordered)   // (the ordered sums compute the box index on their way)
    hipLaunchKernelGGL(box_index_kernel, dim3(grid_for(ctx->np)), dim3(256), 0, ctx->stream, a, G, gbox.t0, gbox.t1,
                       ctx->d_cell, (const double *) nullptr, 0);
  if (ordered) {
    GridVals vals;
    for (int iq = 0; iq < ctx->nq; iq++)
      vals.q[iq] = a.q[iq];
    vals.nq = ctx->nq;
    vals.kern = kern;
    vals.rec = nullptr;
    // the particles' values as records (see GridVals): worth the extra pass from two quantities on; without the
    // memory for it the sums gather from the arrays
    if (ctx->nq >= 2 && ctx->np > 0 && ctx->grid_records) {
      const size_t need = (size_t) ctx->np * (size_t) ctx->nq;
      if (need > ctx->rec_cap) {
        dev_free(ctx->d_rec);
        ctx->d_rec = nullptr;
        ctx->rec_cap = 0;
        if (hipMalloc((void **) &ctx->d_rec, need * sizeof(double)) == hipSuccess)
          ctx->rec_cap = need;
        else
          (void) hipGetLastError();
      }
      if (ctx->d_rec) {
        hipLaunchKernelGGL(grid_records_kernel, dim3(grid_for(ctx->np)), dim3(256), 0, ctx->stream, vals, ctx->np, ctx->d_rec);
        vals.rec = ctx->d_rec;
      }
    }
    // [counts | sums of q | sums of q^2], as grid_accumulate_kernel
    if (ordered_cell_sums(ctx, vals, 2 * ctx->nq, c.grid_nz, ncell, ctx->d_sums + ncell, (int *) nullptr, ctx->d_sums, true,
                          &gbox))
      return 1;
  } else {
    HIPCHK(hipMemsetAsync(ctx->d_sums, 0, total * sizeof(double), ctx->stream));
    if (ctx->np) {
      const AccumGeom g = accum_geom(ctx, 1 + 2 * ctx->nq);
      hipLaunchKernelGGL(grid_accumulate_kernel, dim3(g.nblocks), dim3(256), g.lds, ctx->stream, a, ctx->d_cell, ctx->nq,
                         ncell, ctx->d_sums, g.T, g.per_block, kern);
    }
  }
  HIPCHK(hipGetLastError());
  if (run_allreduce(ctx, ctx->d_sums, total))
    return 1;
  // through a page-locked staging buffer of the context: from pageable memory the 3.6 MB of a 360 x 180 grid
  // with three quantities took 0.3 ms of the output's 0.9
  if (total > ctx->h_sums_cap) {
    if (ctx->h_sums)
      (void) hipHostFree(ctx->h_sums);
    ctx->h_sums = nullptr;
    ctx->h_sums_cap = 0;
    HIPCHK(hipHostMalloc((void **) &ctx->h_sums, total * sizeof(double), hipHostMallocDefault));
    ctx->h_sums_cap = total;
  }
  double *h = ctx->h_sums;
  HIPCHK(hipMemcpyAsync(h, ctx->d_sums, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < ncell; i++)
    cnt[i] = (int) h[i];
  memcpy(mean, h + ncell, ncell * (size_t) ctx->nq * sizeof(double));
  memcpy(sigma, h + ncell * (size_t) (1 + ctx->nq), ncell * (size_t) ctx->nq * sizeof(double));
  return 0;
}

int mphip_set_grid_kernel(mphip_ctx *ctx, int nk, const double *kz, const double *kw) {
  if (!ctx)
    return 1;
  if (nk < 0 || nk > 1024 || (nk >= 2 && (!kz || !kw)))
    return fail(ctx, "bad kernel function");
  for (int k = 1; k < nk; k++)
    if (kz[k] < kz[k - 1])
      return fail(ctx, "height levels of the kernel function must be ascending");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (dev_alloc(ctx, &ctx->d_grid_kernel, nk >= 2 ? 2 * (size_t) nk : 0))
    return 1;
  ctx->grid_nk = nk >= 2 ? nk : 0;
  if (ctx->grid_nk) {
    HIPCHK(hipMemcpy(ctx->d_grid_kernel, kz, (size_t) nk * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_grid_kernel + nk, kw, (size_t) nk * sizeof(double), hipMemcpyHostToDevice));
  }
  return 0;
}

int mphip_set_allreduce(mphip_ctx *ctx, mphip_allreduce_fn fn, void *user) {
  if (!ctx)
    return 1;
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  return 0;
}

int mphip_comm_unique_id(void *id128) {
  if (!id128)
    return 1;
  Rccl &R = rccl();
  if (!R.ok) {
    fprintf(stderr, "mptrac_hip: %s\n", R.why.c_str());
    return 2;
  }
  Rccl::UniqueId id;
  StdoutToStderr quiet;
  if (R.GetUniqueId(&id) != 0)
    return 3;
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int mphip_comm_init(mphip_ctx *ctx, int nranks, int rank, const void *id128) {
  if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(ctx, "bad communicator arguments");
  Rccl &R = rccl();
  if (!R.ok)
    return fail(ctx, R.why);
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->comm) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    RCCLCHK(R.CommDestroy(ctx->comm));
    ctx->comm = nullptr;
  }
  Rccl::UniqueId id;
  memcpy(&id, id128, sizeof(id));
  StdoutToStderr quiet;
  RCCLCHK(R.CommInitRank(&ctx->comm, nranks, id, rank));
  ctx->comm_ranks = nranks;
  ctx->comm_rank = rank;
  return 0;
}

int mphip_comm_destroy(mphip_ctx *ctx) {
  if (!ctx)
    return 1;
  if (ctx->comm) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    RCCLCHK(rccl().CommDestroy(ctx->comm));
    ctx->comm = nullptr;
  }
  ctx->comm_ranks = 1;
  ctx->comm_rank = 0;
  return 0;
}

int mphip_comm_query(mphip_ctx *ctx, int *nranks, int *rank) {
  if (!ctx || !nranks || !rank)
    return 1;
  *nranks = 0;
  *rank = 0;
  if (!ctx->comm)
    return 0;
  Rccl &R = rccl();
  RCCLCHK(R.CommCount(ctx->comm, nranks));
  RCCLCHK(R.CommUserRank(ctx->comm, rank));
  return 0;
}

int mphip_set_option(mphip_ctx *ctx, const char *name, double value) {
  if (!ctx || !name)
    return 1;
  if (strcmp(name, "locality_sort_interval") == 0) {
    if (value < 0)
      return fail(ctx, "locality_sort_interval must be >= 0");
    ctx->locality_interval = (int) value;
    return 0;
  }
  if (strcmp(name, "step_blocks") == 0 || strcmp(name, "step_blocks_multi") == 0) {
    if (value < 8 || value > 1048576)
      return fail(ctx, "step_blocks must be in 8 ... 1048576");
    (strcmp(name, "step_blocks") == 0 ? ctx->step_blocks : ctx->step_blocks_multi) = (int) value;
    return 0;
  }
  if (strcmp(name, "fuse_sort") == 0) {    // 0: module_sort re-orders every array in its own pass
    ctx->fuse_sort = value != 0;
    return 0;
  }
  if (strcmp(name, "lazy_meteo") == 0) {   // 0: run module_meteo inside every time step that schedules it
    if (value == 0 && flush_meteo(ctx))
      return 1;
    ctx->lazy_meteo = value != 0;
    return 0;
  }
  if (strcmp(name, "pin_host_atm") == 0) {   // page-lock the arrays handed to mphip_update_atm / mphip_get_atm
    ctx->pin_host_atm = value != 0;
    return 0;
  }
  if (strcmp(name, "multi_step") == 0) {   // mphip_run_timesteps: most steps per launch, 0 = one launch per step
    if (value < 0 || value > 4096)
      return fail(ctx, "multi_step must be in 0 ... 4096");
    ctx->multi_step = (int) value;
    return 0;
  }
  if (strcmp(name, "fuse_sort_quantities") == 0) {   // 0: module_sort moves the quantity arrays in a pass of its own
    ctx->fuse_quantities = value != 0;
    return 0;
  }
  if (strcmp(name, "perm_records") == 0) {
    ctx->perm_records = value != 0;
    return 0;
  }
  if (strcmp(name, "big_grid") == 0) {
    ctx->big_grid = value != 0;
    return 0;
  }
  if (strcmp(name, "xcd_map") == 0) {
    ctx->xcd_map = value != 0;
    return 0;
  }
  if (strcmp(name, "generic_kernel") == 0) {   // tuning aid: never pick a specialised instantiation
    ctx->force_generic = value != 0;
    return 0;
  }
  if (strcmp(name, "sort_bits") == 0) {
    if (!(value == 0 || (value >= 8 && value <= kRadixMaxBits)))
      return fail(ctx, "sort_bits must be 0 (automatic), 8, 9 or 10");
    ctx->sort_bits = (int) value;
    return 0;
  }
  if (strcmp(name, "mix_exchange_levels") == 0) {   // 0: the exchange of a mixing step covers the whole grid
    ctx->mix_exchange_levels = value != 0;
    return 0;
  }
  if (strcmp(name, "lds_tile") == 0) {   // cells of the LDS wind tile for runs of pure trajectory steps (24 bytes each; 0: off)
    if (value != 0 && (value < 64 || value > 2400))
      return fail(ctx, "lds_tile must be 0 or 64 ... 2400 cells");
    ctx->lds_tile = (int) value;
    return 0;
  }
  if (strcmp(name, "ahead_priority") == 0) {   // (before the first sort ahead: the stream is created then)
    ctx->ahead_priority = value != 0;
    return 0;
  }
  if (strcmp(name, "emit_keys") == 0) {   // 0: the keys of the sort ahead always come from a kernel of their own
    ctx->emit_keys = value != 0;
    return 0;
  }
  if (strcmp(name, "sort_repair") == 0) {   // 0: the module_sort that runs ahead always sorts from scratch
    ctx->sort_repair = value != 0;
    return 0;
  }
  if (strcmp(name, "compact_depo") == 0) {   // 0: deposition-only launches run the tail of the fused kernel
    ctx->compact_depo = value != 0;
    return 0;
  }
  if (strcmp(name, "ahead_box") == 0) {   // tuning aid
    ctx->ahead_box = value != 0;
    return 0;
  }
  if (strcmp(name, "sort_ahead") == 0) {   // 0: module_sort runs when mphip_run_timestep reaches it
    if (ahead_drop(ctx))
      return 1;
    ctx->sort_ahead = value != 0;
    return 0;
  }
  if (strcmp(name, "sum_path") == 0) {   // tests: force one of the two ordered-sum algorithms (0: choose by crowding)
    if (!(value == 0 || value == 1 || value == 2))
      return fail(ctx, "sum_path must be 0, 1 or 2");
    ctx->sum_path = (int) value;
    return 0;
  }
  if (strcmp(name, "deterministic_sums") == 0) {
    // 1 (default): cell sums of module_mixing / the gridded output add in the reference's serial order;
    // 0: floating-point atomics (order of arrival)
    if (!(value == 0 || value == 1))
      return fail(ctx, "deterministic_sums must be 0 or 1");
    ctx->deterministic_sums = (int) value;
    return 0;
  }
  if (strcmp(name, "grid_records") == 0) {   // tuning / tests: 0 = gather the gridded sums' values from the arrays
    ctx->grid_records = value != 0;
    return 0;
  }
  if (strcmp(name, "chain_blocks") == 0) {
    ctx->chain_blocks = (int) value;
    return 0;
  }
  if (strcmp(name, "locality_zorder") == 0) {
    ctx->locality_zorder = value != 0;
    ctx->steps_since_resort = 1 << 30;
    return 0;
  }
  if (strcmp(name, "locality_tile") == 0) {
    if (value < 0 || value > 64)
      return fail(ctx, "locality_tile must be in 0 ... 64");
    ctx->locality_tile = (int) value;
    ctx->steps_since_resort = 1 << 30;
    return 0;
  }
  return fail(ctx, std::string("unknown option ") + name);
}

int mphip_synchronize(mphip_ctx *ctx) {
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
}

int mphip_profile_begin(mphip_ctx *ctx) {
  if (!ctx)
    return 1;
  ctx->prof = true;
  ctx->ev_used = 0;
  return 0;
}

int mphip_profile_end(mphip_ctx *ctx, long long *launches, double *kernel_ms) {
  if (!ctx)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double total = 0;
  for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[k], ctx->ev[k + 1]));
    total += ms;
  }
  if (launches)
    *launches = (long long) (ctx->ev_used / 2);
  if (kernel_ms)
    *kernel_ms = total;
  ctx->prof = false;
  ctx->ev_used = 0;
  return 0;
}

int mphip_test_sincosf(mphip_ctx *ctx, uint32_t bits_first, uint32_t count, float *cos_out, float *sin_out) {
  if (!ctx || !cos_out || !sin_out)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  float *dc = nullptr, *ds = nullptr;
  HIPCHK(hipMalloc((void **) &dc, (size_t) count * sizeof(float)));
  HIPCHK(hipMalloc((void **) &ds, (size_t) count * sizeof(float)));
  hipLaunchKernelGGL(test_sincosf_kernel, dim3(grid_for(count)), dim3(256), 0, ctx->stream, bits_first, count, dc, ds);
  HIPCHK(hipMemcpyAsync(cos_out, dc, (size_t) count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(sin_out, ds, (size_t) count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipFree(dc));
  HIPCHK(hipFree(ds));
  return 0;
}

int mphip_test_libm(mphip_ctx *ctx, int op, const double *x, const double *y, long long n, double *out) {
  if (!ctx || !x || !out || n < 0 || ((op & 15) == 2 && !y) || (op & ~31) || (op & 15) > 5)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t bytes = (size_t) std::max<long long>(n, 1) * sizeof(double);
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc((void **) &dx, bytes));
  HIPCHK(hipMalloc((void **) &dout, bytes));
  HIPCHK(hipMemcpyAsync(dx, x, (size_t) n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if ((op & 15) == 2) {
    HIPCHK(hipMalloc((void **) &dy, bytes));
    HIPCHK(hipMemcpyAsync(dy, y, (size_t) n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(test_libm_kernel, dim3(grid_for(std::max<long long>(n, 1))), dim3(256), 0, ctx->stream, op, dx, dy, n, dout);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dout, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipFree(dx));
  HIPCHK(hipFree(dout));
  if (dy)
    HIPCHK(hipFree(dy));
  return 0;
}

int mphip_test_piece(mphip_ctx *ctx, int piece, int reps, double *checksum) {
  if (!ctx || !ctx->have_ctl || ctx->np == 0)
    return fail(ctx, "mphip_test_piece needs control parameters and particles");
  HIPCHK(hipSetDevice(ctx->device));
  if (ensure_packed(ctx))
    return 1;
  if (!ctx->have_clim)
    return fail(ctx, "climatological tropopause data were not uploaded");
  StepParams S;
  memset(&S.emit, 0, sizeof(S.emit));
  S.depo_busy = nullptr;
  S.ctl = ctx->ctl;
  S.met = dev_met(ctx);
  S.atm = dev_atm(ctx);
  S.clim = ctx->d_clim;
  S.tracers = ctx->d_tracers;
  S.t = 0;
  S.mask = 0;
  long long per_block = (ctx->np + ctx->step_blocks - 1) / ctx->step_blocks;
  per_block = std::max<long long>(256, (per_block + 255) / 256 * 256);
  int nb = (int) ((ctx->np + per_block - 1) / per_block);
  nb = (nb + 7) & ~7;
  S.nblocks_logical = nb;
  S.per_block = per_block;
  S.xcd_map = ctx->xcd_map;
  S.ctr_turb = 1000;
  S.ctr_meso = 5000;
  S.ctr_conv = 9000;
  S.ctr_pbl = 0;
  double *d = nullptr;
  HIPCHK(hipMalloc((void **) &d, (size_t) ctx->np * sizeof(double)));
  const size_t lds = axes_lds_bytes(ctx) + sizeof(DevClim) + (size_t) kLibmLogExpDoubles * sizeof(double);
  switch (piece) {
#define PIECE_CASE(K)                                                                                  \
  case K:                                                                                              \
    hipLaunchKernelGGL(piece_kernel<K>, dim3(nb), dim3(256), lds, ctx->stream, S, reps, d);            \
    break;
#ifdef MPHIP_QUICK
    PIECE_CASE(0)
#else
    PIECE_CASE(0) PIECE_CASE(1) PIECE_CASE(2) PIECE_CASE(3) PIECE_CASE(4) PIECE_CASE(5) PIECE_CASE(6) PIECE_CASE(7)
    PIECE_CASE(8) PIECE_CASE(9) PIECE_CASE(10) PIECE_CASE(11) PIECE_CASE(12) PIECE_CASE(13) PIECE_CASE(14)
    PIECE_CASE(15) PIECE_CASE(16) PIECE_CASE(17) PIECE_CASE(18) PIECE_CASE(19) PIECE_CASE(20) PIECE_CASE(21)
    PIECE_CASE(22) PIECE_CASE(23) PIECE_CASE(24) PIECE_CASE(25) PIECE_CASE(26)
#endif
#undef PIECE_CASE
  default:
    (void) hipFree(d);
    return fail(ctx, "unknown piece");
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (checksum) {
    std::vector<double> h((size_t) std::min<long long>(ctx->np, 1024));
    HIPCHK(hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    double sum = 0;
    for (double v : h)
      sum += v;
    *checksum = sum;
  }
  HIPCHK(hipFree(d));
  return 0;
}

int mphip_test_rng(mphip_ctx *ctx, uint64_t ctr, long long n, int method, double *out) {
  if (!ctx || !out || n < 0)
    return 1;
  HIPCHK(hipSetDevice(ctx->device));
  double *d = nullptr;
  HIPCHK(hipMalloc((void **) &d, (size_t) std::max<long long>(n, 1) * sizeof(double)));
  hipLaunchKernelGGL(test_rng_kernel, dim3(grid_for(std::max<long long>(n, 1))), dim3(256), 0, ctx->stream, ctr, n,
                     method, d);
  HIPCHK(hipMemcpyAsync(out, d, (size_t) n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipFree(d));
  return 0;
}

}   // extern "C"
