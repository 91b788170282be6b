"""Named parity cases: control settings + synthetic inputs."""
import numpy as np

from mptrac_amd.clim import load_clim_tropo
from mptrac_amd.ctl import ctl_from_quantities
from mptrac_amd.synth import FIELDS_METEO_ONLY, synthetic_met, synthetic_particles

BASE = dict(advect=4, dt_mod=180.0, t_stop=3600.0, dt_met=3600.0, rng_type=1)

CASES = {
    # RK4 advection only (BASELINE config 0)
    "advect": dict(BASE),
    "advect_midpoint": dict(BASE, advect=2),
    "advect_euler": dict(BASE, advect=1),
    # + turbulent diffusion (config 1)
    "turb": dict(BASE, diffusion=1, turb_mesox=0.0, turb_mesoz=0.0, turb_dz_trop=0.1, turb_dz_pbl=0.5),
    # Hanna / FLEXPART closure inside the boundary layer (TURB_PBL_SCHEME 1)
    "pbl": dict(BASE, diffusion=1, turb_pbl_scheme=1, turb_dz_trop=0.1, turb_mesox=0.0, turb_mesoz=0.0),
    # (the closure keeps uvwp[2] in m/s, module_diff_meso in hPa/s: with both vertical parts on the reference
    # itself drives pressures negative, so the combined case keeps the mesoscale part horizontal)
    "pbl_meso": dict(BASE, diffusion=1, turb_pbl_scheme=1, turb_mesoz=0.0, conv_cape=0.0),
    # + mesoscale diffusion
    "diff": dict(BASE, diffusion=1, turb_dz_trop=0.1),
    # + convection + sedimentation (config 2)
    "conv_sedi": dict(BASE, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0, conv_mix_pbl=1,
                      conv_pbl_trans=0.1, turb_pbl_trans=0.2),
    # CAPE threshold that splits the particle set
    "conv_thresh": dict(BASE, conv_cape=150.0, conv_cin=5.0),
    # + decay, mixing, wet and dry deposition, sort (config 4)
    "full": dict(BASE, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0, sort_dt=360.0,
                 mixing_trop=1e-3, mixing_strat=1e-6, mixing_dt=360.0, mixing_nx=36, mixing_ny=18, mixing_nz=20,
                 tdec_trop=259200.0, tdec_strat=259200.0, dry_depo_vdep=0.15,
                 wet_depo_ic_a=1e-4, wet_depo_ic_b=0.8, wet_depo_bc_a=5e-5, wet_depo_bc_b=0.6),
    # model-level advection: zeta (with module_advect_init) and eta coordinates
    "advect_zeta": dict(BASE, advect_vert_coord=1),
    "advect_zeta_midpoint": dict(BASE, advect=2, advect_vert_coord=1),
    "advect_eta": dict(BASE, advect_vert_coord=3),
    "zeta_full": dict(BASE, advect_vert_coord=1, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0),
    # ADVECT_VERT_COORD 2: pressure advection with u, v, omega interpolated from the model levels (mptrac.c:3649-3659)
    "advect_mlp": dict(BASE, advect_vert_coord=2),
    "advect_mlp_midpoint": dict(BASE, advect=2, advect_vert_coord=2),
    "mlp_full": dict(BASE, advect_vert_coord=2, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0),
    # Henry-law wet deposition with SO2 pH correction
    "wet_henry": dict(BASE, wet_depo_ic_h=(1.3e-2, 2900.0), wet_depo_bc_h=(1.3e-2, 2900.0),
                      wet_depo_so2_ph=4.5, wet_depo_ic_ret_ratio=0.5, wet_depo_bc_ret_ratio=0.3),
    # module_meteo every step (MET_DT_OUT default 0.1 < DT_MOD) on top of the stochastic modules
    "meteo": dict(BASE, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0),
    # module_meteo only when fmod(t, MET_DT_OUT) == 0, interleaved with module_sort
    "meteo_gated": dict(BASE, met_dt_out=1800.0, sort_dt=360.0, diffusion=1, turb_dz_trop=0.1),
    # module_isosurf: isobaric, isopycnic, isentropic and balloon-pressure trajectories
    "isosurf_p": dict(BASE, isosurf=1, diffusion=1, turb_dz_trop=0.1),
    "isosurf_rho": dict(BASE, isosurf=2),
    "isosurf_theta": dict(BASE, isosurf=3, diffusion=1, turb_dz_trop=0.1, sort_dt=900.0),
    "isosurf_balloon": dict(BASE, isosurf=4),
    # module_bound_cond: surface layer by pressure depth / PBL / zeta, mass + vmr with trend + age of air,
    # together with decay and mixing (which also mixes aoa)
    "bound": dict(BASE, bound_lat0=-60.0, bound_lat1=60.0, bound_p0=1100.0, bound_p1=300.0, bound_dps=400.0,
                  bound_mass=2.5, bound_mass_trend=1e-4, bound_vmr=3e-9, bound_vmr_trend=1e-13,
                  tdec_trop=259200.0, tdec_strat=259200.0, diffusion=1, turb_dz_trop=0.1,
                  mixing_trop=1e-3, mixing_strat=1e-6, mixing_dt=360.0, mixing_nx=36, mixing_ny=18, mixing_nz=20),
    "bound_pbl_zeta": dict(BASE, bound_lat0=-90.0, bound_lat1=90.0, bound_p0=1100.0, bound_p1=100.0, bound_pbl=1,
                           bound_zetas=330.0, bound_dzs=3.0, bound_mass=1.0, conv_cape=0.0),
}

# pressure time series a balloon would report (ISOSURF 4): 2 h, one value per 10 min
BALLOON = ([-600.0 + 600.0 * k for k in range(13)], [80.0 - 2.5 * k + 0.3 * (k % 3) for k in range(13)])

CASE_QUANTITIES = {
    "meteo": ("m", "rp", "rhop", "t", "u", "zg", "pv", "ps", "pt", "theta", "rh", "zeta_d", "sst", "lapse", "vh", "o3"),
    "meteo_gated": ("m", "t", "w", "h2o", "tdew", "plfc", "cc", "rho"),
    "bound": ("m", "vmr", "aoa", "loss_rate", "mloss_decay"),
    "bound_pbl_zeta": ("m", "aoa"),
}

QUANTITIES = ("m", "rp", "rhop", "vmr", "loss_rate", "mloss_decay", "mloss_wet", "mloss_dry")
QUANTITIES_ML = QUANTITIES + ("zeta", "eta")
PRESSURE_LEVEL_FIELDS = ("u", "v", "w", "t", "lwc", "rwc", "iwc", "swc", "h2o", "ps", "pbl", "cape", "cin", "pel",
                         "pct", "pcb", "cl", "ess", "nss", "shf")


def make_case(name, n=10000, grid="C1", seed=12345, quantities=None, lon0=-180.0, fields=None):
    ctl = dict(CASES[name])
    ml = ctl.get("advect_vert_coord", 0) in (1, 3)
    if quantities is None:
        quantities = CASE_QUANTITIES.get(name, QUANTITIES_ML if ml else QUANTITIES)
    ml = ml or ctl.get("advect_vert_coord", 0) == 2      # model-level fields are generated
    if fields is None and not ml:
        fields = PRESSURE_LEVEL_FIELDS          # model-level fields only where they are used
        if name.startswith("meteo"):
            fields = fields + FIELDS_METEO_ONLY
    ctl.update(ctl_from_quantities(quantities))
    if name.startswith(("advect", "isosurf", "bound")) or name in ("turb", "diff", "conv_thresh", "pbl", "meteo_gated"):
        # no sedimentation in these
        ctl["qnt_rp"] = ctl["qnt_rhop"] = -1
    met0 = synthetic_met(grid, 0.0, 1.0, fields=fields, lon0=lon0)
    met1 = synthetic_met(grid, 3600.0, 1.25, fields=fields, lon0=lon0)
    atm = synthetic_particles(n, seed=seed, quantities=quantities)
    if ctl.get("turb_pbl_scheme", 0):      # half of the particles inside the boundary layer
        atm["p"][::2] = 1013.25 * np.exp(-(0.02 + 0.9 * (atm["lon"][::2] + 180.0) / 360.0) / 7.0)
    if name.startswith("isosurf"):         # some particles are released later: module_isosurf also acts on dt = 0
        atm["time"][::7] = 540.0
    for name_q in ("zeta", "eta"):
        if name_q in quantities:     # a vertical coordinate inside the range of the synthetic zetal field
            atm["q"][list(quantities).index(name_q)] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
    return ctl, load_clim_tropo(), met0, met1, atm


def prepare(engine):
    """Inputs a driver hands over besides ctl / met / atm (same call on the oracle and the product)."""
    if engine.ctl.isosurf == 4:
        engine.set_balloon(*BALLOON)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    nan = np.isnan(b)
    if nan.any():      # undefined values (e.g. sst over land) must be undefined on both sides
        if not np.array_equal(np.isnan(a), nan):
            return float("inf")
        a, b = np.where(nan, 0.0, a), np.where(nan, 0.0, b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))


def rel_err_strict(a, b, floor=1e-300):
    """True relative error |a - b| / |b| (quantities far below 1, e.g. mixing ratios)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nan = np.isnan(b)
    if not np.array_equal(np.isnan(a), nan):
        return float("inf")
    a, b = np.where(nan, 0.0, a), np.where(nan, 0.0, b)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def q_rows_err(ctl, a, b, floor=1e-3):
    """Largest error over the quantity rows, each row on ITS OWN scale:
    |a - b| / max(|b|, floor * max|row|).  rel_err() floors the denominator at
    1, which for rows far below 1 (vmr ~ 3e-9, loss_rate ~ 4e-6, mloss_*) is
    an absolute bar and checks nothing.  The mloss_* rows accumulate
    m * (1 - aux) -- a difference of numbers near 1 -- so their natural scale is
    the mass row's.  Returns (error, row index of the worst row)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    worst, where = 0.0, -1
    qm = getattr(ctl, "qnt_m", -1)
    mloss = {getattr(ctl, k, -1) for k in ("qnt_mloss_wet", "qnt_mloss_dry", "qnt_mloss_decay")}
    for iq in range(b.shape[0]):
        nan = np.isnan(b[iq])
        if not np.array_equal(np.isnan(a[iq]), nan):
            return float("inf"), iq
        x, y = np.where(nan, 0.0, a[iq]), np.where(nan, 0.0, b[iq])
        top = float(np.max(np.abs(y))) if y.size else 0.0
        if iq in mloss and qm >= 0 and b[qm].size:
            top = max(top, float(np.nanmax(np.abs(b[qm]))))
        if top == 0.0:
            err = 0.0 if np.array_equal(x, y) else float("inf")
        else:
            err = float(np.max(np.abs(x - y) / np.maximum(np.abs(y), floor * top))) if y.size else 0.0
        if err > worst:
            worst, where = err, iq
    return worst, where


def step_times(ctl):
    """The driver's time loop (src/trac.c:131-137)."""
    t = ctl.t_start
    out = []
    while ctl.direction * (t - ctl.t_stop) < ctl.dt_mod:
        if ctl.direction * (t - ctl.t_stop) > 0:
            t = ctl.t_stop
        out.append(t)
        t += ctl.direction * ctl.dt_mod
    return out
