"""Hot-path subset of the reference's ``ctl_t`` (src/mptrac.h:2494-3553).

Field names, meaning and defaults follow ``mptrac_read_ctl``
(src/mptrac.c:6723-7741; defaults cited per group below).  The field order is
the layout of ``mphip_ctl_t`` in include/mptrac_hip.h (checked at load time
through ``mphip_sizeof_ctl``).
"""
import ctypes as C

# (name, ctype, default)
CTL_FIELDS = [
    # time control, mptrac.c:7015-7024
    ("direction", C.c_int, 1),
    ("met_coord_type", C.c_int, 0),            # mptrac.c:6974
    ("t_start", C.c_double, 0.0),              # set by module_timesteps_init
    ("t_stop", C.c_double, 1e100),
    ("dt_mod", C.c_double, 180.0),
    ("dt_met", C.c_double, 3600.0),
    ("met_utm_ref_lat", C.c_double, 0.0),      # mptrac.c:6981
    # quantities, mptrac.c:6737-6971 (-1 = not present)
    ("nq", C.c_int, 0),
    ("qnt_m", C.c_int, -1),
    ("qnt_vmr", C.c_int, -1),
    ("qnt_rp", C.c_int, -1),
    ("qnt_rhop", C.c_int, -1),
    ("qnt_ens", C.c_int, -1),
    ("qnt_loss_rate", C.c_int, -1),
    ("qnt_mloss_decay", C.c_int, -1),
    ("qnt_mloss_wet", C.c_int, -1),
    ("qnt_mloss_dry", C.c_int, -1),
    ("qnt_zeta", C.c_int, -1),
    ("qnt_eta", C.c_int, -1),
    ("nens", C.c_int, 0),                      # mptrac.c:7610
    # modules, mptrac.c:7196-7263
    ("advect", C.c_int, 2),
    ("advect_vert_coord", C.c_int, 0),
    ("rng_type", C.c_int, 1),
    ("diffusion", C.c_int, 0),
    ("turb_pbl_scheme", C.c_int, 0),
    ("conv_mix_pbl", C.c_int, 0),
    ("turb_dx_pbl", C.c_double, 50.0),
    ("turb_dx_trop", C.c_double, 50.0),
    ("turb_dx_strat", C.c_double, 0.0),
    ("turb_dz_pbl", C.c_double, 0.0),
    ("turb_dz_trop", C.c_double, 0.0),
    ("turb_dz_strat", C.c_double, 0.1),
    ("turb_mesox", C.c_double, 0.16),
    ("turb_mesoz", C.c_double, 0.16),
    ("turb_pbl_trans", C.c_double, 0.0),
    ("conv_pbl_trans", C.c_double, 0.0),
    ("conv_cape", C.c_double, -999.0),
    ("conv_cin", C.c_double, -999.0),
    ("conv_dt", C.c_double, -999.0),
    ("sort_dt", C.c_double, -999.0),           # mptrac.c:7200
    ("tdec_trop", C.c_double, 0.0),            # mptrac.c:7541-7543
    ("tdec_strat", C.c_double, 0.0),
    # mixing, mptrac.c:7479-7507
    ("mixing_dt", C.c_double, 3600.0),
    ("mixing_trop", C.c_double, -999.0),
    ("mixing_strat", C.c_double, -999.0),
    ("mixing_z0", C.c_double, -5.0),
    ("mixing_z1", C.c_double, 85.0),
    ("mixing_lon0", C.c_double, -180.0),
    ("mixing_lon1", C.c_double, 180.0),
    ("mixing_lat0", C.c_double, -90.0),
    ("mixing_lat1", C.c_double, 90.0),
    ("mixing_nx", C.c_int, 360),
    ("mixing_ny", C.c_int, 180),
    ("mixing_nz", C.c_int, 90),
    ("pad0", C.c_int, 0),
    # wet / dry deposition, mptrac.c:7425-7456
    ("wet_depo_pre", C.c_double * 2, (0.5, 0.36)),
    ("wet_depo_ic_a", C.c_double, 0.0),
    ("wet_depo_ic_b", C.c_double, 0.0),
    ("wet_depo_bc_a", C.c_double, 0.0),
    ("wet_depo_bc_b", C.c_double, 0.0),
    ("wet_depo_ic_h", C.c_double * 2, (0.0, 0.0)),
    ("wet_depo_bc_h", C.c_double * 2, (0.0, 0.0)),
    ("wet_depo_so2_ph", C.c_double, 0.0),
    ("wet_depo_ic_ret_ratio", C.c_double, 1.0),
    ("wet_depo_bc_ret_ratio", C.c_double, 1.0),
    ("dry_depo_vdep", C.c_double, 0.0),
    ("dry_depo_dp", C.c_double, 30.0),
    # gridded output, mptrac.c:7631-7648
    ("grid_z0", C.c_double, -5.0),
    ("grid_z1", C.c_double, 85.0),
    ("grid_lon0", C.c_double, -180.0),
    ("grid_lon1", C.c_double, 180.0),
    ("grid_lat0", C.c_double, -90.0),
    ("grid_lat1", C.c_double, 90.0),
    ("grid_nx", C.c_int, 360),
    ("grid_ny", C.c_int, 180),
    ("grid_nz", C.c_int, 1),
    ("pad1", C.c_int, 0),
    # module_meteo, mptrac.c:7197 and 7921-7924; qnt_met[k] = ctl->qnt_<METEO_QUANTITIES[k]>
    ("met_dt_out", C.c_double, 0.1),
    ("qnt_met", C.c_int * 60, (-1,) * 60),
    # module_isosurf (mptrac.c:7208) and module_bound_cond (mptrac.c:7266-7289)
    ("isosurf", C.c_int, 0),
    ("bound_pbl", C.c_int, 0),
    ("qnt_aoa", C.c_int, -1),
    ("pad3", C.c_int, 0),
    ("bound_mass", C.c_double, -999.0),
    ("bound_mass_trend", C.c_double, 0.0),
    ("bound_vmr", C.c_double, -999.0),
    ("bound_vmr_trend", C.c_double, 0.0),
    ("bound_lat0", C.c_double, -999.0),
    ("bound_lat1", C.c_double, -999.0),
    ("bound_p0", C.c_double, -999.0),
    ("bound_p1", C.c_double, -999.0),
    ("bound_dps", C.c_double, -999.0),
    ("bound_dzs", C.c_double, -999.0),
    ("bound_zetas", C.c_double, -999.0),
    # module_meteo's OH climatology (clim_oh, mptrac.c:89-120): diurnal scaling, reference longitude of a Cartesian grid
    ("oh_chem_beta", C.c_double, 0.0),
    ("met_utm_ref_lon", C.c_double, 0.0),
    # trace gases of module_bound_cond / module_mixing: qnt_Cccl4, qnt_Cccl3f, qnt_Cccl2f2, qnt_Cn2o, qnt_Csf6 (TRACERS)
    ("qnt_tracer", C.c_int * 5, (-1,) * 5),
    ("pad4", C.c_int, 0),
]

# quantities module_meteo fills, in the order of its SET_ATM list (mptrac.c:5091-5157) = MPHIP_MQ_*
METEO_QUANTITIES = (
    "ps", "ts", "zs", "us", "vs", "ess", "nss", "shf", "lsm", "sst", "pbl", "pt", "tt", "zt", "h2ot", "zg", "p",
    "t", "rho", "u", "v", "w", "h2o", "o3", "lwc", "rwc", "iwc", "swc", "cc", "pct", "pcb", "cl", "plcl", "plfc",
    "pel", "cape", "cin", "o3c", "vh", "vz", "psat", "psice", "pw", "sh", "rh", "rhice", "theta", "zeta_d",
    "tvirt", "lapse", "pv", "tdew", "tice",
    # from the zonal-mean climatologies of clim_t (ZONAL_MEANS below)
    "hno3", "oh", "h2o2", "ho2", "o1d", "tnat", "tsts")
assert len(METEO_QUANTITIES) == 60

# zonal-mean climatologies module_meteo samples (clim_zm_t members of clim_t, mptrac.h:3805-3817) = MPHIP_ZM_*
ZONAL_MEANS = ("hno3", "oh", "h2o2", "ho2", "o1d")

# trace gases with a surface time series (clim_ts_t members ccl4, ccl3f, ccl2f2, n2o, sf6 of clim_t,
# mptrac.h:3820-3832) = MPHIP_TR_*: quantity names and names of the tables
TRACERS = ("Cccl4", "Cccl3f", "Cccl2f2", "Cn2o", "Csf6")
TRACER_SERIES = ("ccl4", "ccl3f", "ccl2f2", "n2o", "sf6")


def make_ctl_struct(name):
    """ctypes mirror of the C struct; one class per consumer so that the
    library bindings do not share a type object."""
    return type(name, (C.Structure,), {"_fields_": [(n, t) for n, t, _ in CTL_FIELDS]})


def fill_ctl(struct, **kw):
    """Set defaults (as mptrac_read_ctl would) and then the given keys."""
    known = {n for n, _, _ in CTL_FIELDS}
    for k in kw:
        if k not in known:
            raise KeyError(f"unknown control parameter {k!r}")
    for n, t, d in CTL_FIELDS:
        v = kw.get(n, d)
        if isinstance(d, tuple):
            arr = getattr(struct, n)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(struct, n, v)
    return struct


def ctl_from_quantities(names):
    """Resolve quantity names to qnt_* slots like mptrac.c:6737-6971."""
    out = {"nq": len(names)}
    table = {"m": "qnt_m", "vmr": "qnt_vmr", "rp": "qnt_rp", "rhop": "qnt_rhop", "ens": "qnt_ens",
             "loss_rate": "qnt_loss_rate", "mloss_decay": "qnt_mloss_decay",
             "mloss_wet": "qnt_mloss_wet", "mloss_dry": "qnt_mloss_dry", "zeta": "qnt_zeta", "eta": "qnt_eta",
             "aoa": "qnt_aoa"}
    met = [-1] * len(METEO_QUANTITIES)
    tracer = [-1] * len(TRACERS)
    for i, n in enumerate(names):
        if n in table:
            out[table[n]] = i
        elif n in METEO_QUANTITIES:
            met[METEO_QUANTITIES.index(n)] = i
        elif n in TRACERS:
            tracer[TRACERS.index(n)] = i
    if any(v >= 0 for v in met):
        out["qnt_met"] = tuple(met)
    if any(v >= 0 for v in tracer):
        out["qnt_tracer"] = tuple(tracer)
    return out
