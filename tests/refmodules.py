"""Independent numpy statements of the modules no reference-held golden file reaches (every test of the
reference uses ADVECT 2 and none runs convection, mixing or deposition) -- a second opinion on the oracle, written
from the text of the reference (src/mptrac.c, src/mptrac.h; line ranges at each function), NOT from oracle/:

  intpol_met_space_3d / _2d, intpol_met_time_3d / _2d, locate_irr / locate_reg, intpol_check_lon_lat
  module_advect with ADVECT 4 / 2 / 1 on pressure levels       mptrac.c:3597-3678
  module_diff_turb (horizontal and vertical branch)            mptrac.c:4588-4734
  module_convection                                            mptrac.c:4102-4171
  module_sedi + sedi()                                         mptrac.c:5869-5882, 12506-12535
  module_mixing + module_mixing_help                           mptrac.c:5169-5347
  module_wet_depo, module_dry_depo                             mptrac.c:6155-6290, 4738-4797

Vectorised over the particles (one numpy array per scalar of the C loops), same operation order, IEEE doubles;
the float arrays of met_t are widened where C widens them.  The random numbers of the stochastic modules are an input
(cache->rs as module_rng left it -- the generator is pinned separately by the reference's coord_test golden and the
known-answer values).  tests/test_oracle_second_opinion.py compares every module with the oracle to 1e-13.
"""
import numpy as np

# mptrac.h:255-345
G0, H0, KB, MA, P0, RE, RI = 9.80665, 7.0, 1.3806504e-23, 28.9644, 1013.25, 6367.421, 8.3144598
RA = 1e3 * RI / MA
M_AIR_MOLECULE = 4.8096e-26
T_REF, WD_T_LIQUID, WD_T_ICE, WD_T_LIQUID_BC = 298.15, 273.15, 238.15, 270.0
SO2_K1_REF, SO2_K1_TEMP, SO2_K2_REF, SO2_K2_TEMP = 1.23e-2, 2.01e3, 6e-8, 1.12e3


def Z(p):                      # mptrac.h:2243
    return H0 * np.log(P0 / p)


def LIN(x0, y0, x1, y1, x):    # mptrac.h:1351
    return y0 + (y1 - y0) / (x1 - x0) * (x - x0)


def FMOD(x, y):                # mptrac.h:1121: x - (int) (x / y) * y, truncation towards zero
    return x - np.trunc(x / y) * y


def DX2DEG(dx, lat):           # mptrac.h:904
    with np.errstate(all="ignore"):
        v = dx * 180.0 / (np.pi * RE * np.cos(lat * (np.pi / 180.0)))
    return np.where((lat < -89.999) | (lat > 89.999), 0.0, v)


def DY2DEG(dy):                # mptrac.h:922
    return dy * 180.0 / (np.pi * RE)


def DZ2DP(dz, p):              # mptrac.h:941
    return -dz * p / H0


def locate_irr(xx, x):
    """mptrac.c:3495-3521, the bisection as written (xx ascending or descending), for an array of x"""
    xx = np.asarray(xx, dtype=np.float64)
    n = len(xx)
    x = np.atleast_1d(np.asarray(x, dtype=np.float64))
    ilo = np.zeros(x.shape, dtype=np.int64)
    ihi = np.full(x.shape, n - 1, dtype=np.int64)
    asc = xx[(n - 1) >> 1] < xx[((n - 1) >> 1) + 1]
    while True:
        active = ihi > ilo + 1
        if not active.any():
            return ilo
        i = (ihi + ilo) >> 1
        up = (xx[i] > x) if asc else (xx[i] <= x)
        ihi = np.where(active & up, i, ihi)
        ilo = np.where(active & ~up, i, ilo)


def locate_reg(xx, x):
    """mptrac.c:3559-3574"""
    n = len(xx)
    with np.errstate(invalid="ignore"):
        i = np.trunc((x - xx[0]) / (xx[1] - xx[0]))
    return np.clip(np.nan_to_num(i, nan=0.0), 0, n - 2).astype(np.int64)


class Weights:
    """ci / cw of INTPOL_INIT: what an interpolation call with init = 1 leaves behind"""

    def __init__(self, met, p, lon, lat, three_d=True):
        # intpol_check_lon_lat, mptrac.c:2755-2778 (latitude / longitude grids only here)
        lon2 = FMOD(lon, 360.0)
        lon2 = np.where(lon2 < met.lon[0], lon2 + 360.0, np.where(lon2 > met.lon[-1], lon2 - 360.0, lon2))
        lo, hi = (met.lat[0], met.lat[-1]) if met.lat[0] < met.lat[-1] else (met.lat[-1], met.lat[0])
        lat2 = np.minimum(np.maximum(lat, lo), hi)
        self.ix = locate_reg(met.lon, lon2)
        self.iy = locate_irr(met.lat, lat2)
        self.wx = (met.lon[self.ix + 1] - lon2) / (met.lon[self.ix + 1] - met.lon[self.ix])
        self.wy = (met.lat[self.iy + 1] - lat2) / (met.lat[self.iy + 1] - met.lat[self.iy])
        if three_d:
            self.ip = locate_irr(met.p, p)
            self.wp = (met.p[self.ip + 1] - p) / (met.p[self.ip + 1] - met.p[self.ip])


def space_3d(arr, w):
    """intpol_met_space_3d, mptrac.c:3024-3044: the corner difference is a float difference"""
    def col(dx, dy):
        a0 = arr[w.ix + dx, w.iy + dy, w.ip]
        a1 = arr[w.ix + dx, w.iy + dy, w.ip + 1]
        return w.wp * (a0 - a1).astype(np.float64) + a1.astype(np.float64)       # float - float, then widened
    aux00, aux01, aux10, aux11 = col(0, 0), col(0, 1), col(1, 0), col(1, 1)
    aux0 = w.wy * (aux00 - aux01) + aux01
    aux1 = w.wy * (aux10 - aux11) + aux11
    return w.wx * (aux0 - aux1) + aux1


def space_2d(arr, w):
    """intpol_met_space_2d, mptrac.c:3086-3113 (nearest corner if a corner is not finite)"""
    a00 = arr[w.ix, w.iy].astype(np.float64)
    a01 = arr[w.ix, w.iy + 1].astype(np.float64)
    a10 = arr[w.ix + 1, w.iy].astype(np.float64)
    a11 = arr[w.ix + 1, w.iy + 1].astype(np.float64)
    fin = np.isfinite(a00) & np.isfinite(a01) & np.isfinite(a10) & np.isfinite(a11)
    with np.errstate(invalid="ignore"):
        aux0 = w.wy * (a00 - a01) + a01
        aux1 = w.wy * (a10 - a11) + a11
        smooth = w.wx * (aux0 - aux1) + aux1
    nearest = np.where(w.wy < 0.5, np.where(w.wx < 0.5, a11, a01), np.where(w.wx < 0.5, a10, a00))
    return np.where(fin, smooth, nearest)


class Ref:
    def __init__(self, ctl, clim, met0, met1):
        """ctl: the ctypes control structure of the oracle binding (plain parameter values, no code);
        clim: (time[12], lat[73], tropo[12][73], ...); met0 / met1: mptrac_amd.synth.Met"""
        self.c, self.m0, self.m1 = ctl, met0, met1
        self.tropo_time, self.tropo_lat, self.tropo = (np.asarray(a, dtype=np.float64) for a in clim[:3])

    # -- interpolation in time -------------------------------------------------------------------------------
    def wt(self, ts):
        return (self.m1.time - ts) / (self.m1.time - self.m0.time)

    def time_3d(self, name, ts, p, lon, lat, w=None):
        """intpol_met_time_3d, mptrac.c:3117-3143: both snapshots with the indices and weights of met0"""
        w = w or Weights(self.m0, p, lon, lat)
        v0, v1 = space_3d(self.m0.f3[name], w), space_3d(self.m1.f3[name], w)
        return self.wt(ts) * (v0 - v1) + v1

    def time_2d(self, name, ts, lon, lat, w=None):
        """intpol_met_time_2d, mptrac.c:3147-3179"""
        w = w or Weights(self.m0, None, lon, lat, three_d=False)
        v0, v1 = space_2d(self.m0.f2[name], w), space_2d(self.m1.f2[name], w)
        wt = self.wt(ts)
        with np.errstate(invalid="ignore"):
            both = wt * (v0 - v1) + v1
        return np.where(np.isfinite(v0) & np.isfinite(v1), both, np.where(wt < 0.5, v1, v0))

    # -- climatological tropopause and the weights -------------------------------------------------------------
    def clim_tropo(self, t, lat):
        """mptrac.c:213-237"""
        sec = FMOD(t, 365.25 * 86400.0)
        sec = np.where(sec < 0, sec + 365.25 * 86400.0, sec)      # (one wrap suffices for |t| < a year: asserted)
        assert np.all(sec >= 0)
        isec = locate_irr(self.tropo_time, sec)
        ilat = locate_reg(self.tropo_lat, lat)
        la0, la1 = self.tropo_lat[ilat], self.tropo_lat[ilat + 1]
        p0 = LIN(la0, self.tropo[isec, ilat], la1, self.tropo[isec, ilat + 1], lat)
        p1 = LIN(la0, self.tropo[isec + 1, ilat], la1, self.tropo[isec + 1, ilat + 1], lat)
        return LIN(self.tropo_time[isec], p0, self.tropo_time[isec + 1], p1, sec)

    def tropo_weight(self, t, lat, p):
        """mptrac.c:12748-12770"""
        pt = self.clim_tropo(t, lat)
        p1, p0 = pt * 0.866877899, pt / 0.866877899
        return np.where(p > p0, 1.0, np.where(p < p1, 0.0, LIN(p0, 1.0, p1, 0.0, p)))

    def pbl_weight(self, p, pbl, ps):
        """mptrac.c:8358-8376"""
        p1 = pbl - self.c.turb_pbl_trans * (ps - pbl)
        p0 = pbl
        with np.errstate(all="ignore"):
            mid = LIN(p0, 1.0, p1, 0.0, p)
        return np.where(p > p0, 1.0, np.where(p < p1, 0.0, mid))

    # -- module_advect, pressure levels (mptrac.c:3609-3678) ------------------------------------------------------
    def advect(self, time, lon, lat, p, dt):
        n_nodes = self.c.advect
        u, v, w = [None] * 4, [None] * 4, [None] * 4
        um = vm = wm = 0.0
        x1 = lat
        for i in range(n_nodes):
            if i == 0:
                dts = 0.0 * dt
                x0, x1, x2 = lon, lat, p
            else:
                dts = (1.0 if i == 3 else 0.5) * dt
                x0 = lon + DX2DEG(dts * u[i - 1] / 1000.0, lat)       # every node starts from the OLD position
                x1 = lat + DY2DEG(dts * v[i - 1] / 1000.0)
                x2 = p + dts * w[i - 1]
            tm = time + dts
            wts = Weights(self.m0, x2, x0, x1)
            u[i] = self.time_3d("u", tm, x2, x0, x1, wts)
            v[i] = self.time_3d("v", tm, x2, x0, x1, wts)
            w[i] = self.time_3d("w", tm, x2, x0, x1, wts)
            k = 1.0
            if n_nodes == 2:
                k = 0.0 if i == 0 else 1.0
            elif n_nodes == 4:
                k = 1.0 / 6.0 if i in (0, 3) else 2.0 / 6.0
            um, vm, wm = um + k * u[i], vm + k * v[i], wm + k * w[i]
        # the longitude step uses the old latitude -- except ADVECT 2, which takes the latitude of its last node
        new_lon = lon + DX2DEG(dt * um / 1000.0, x1 if n_nodes == 2 else lat)
        return time + dt, new_lon, lat + DY2DEG(dt * vm / 1000.0), p + dt * wm

    # -- module_diff_turb (mptrac.c:4588-4734), TURB_PBL_SCHEME 0 ------------------------------------------------
    def diff_turb(self, time, lon, lat, p, dt, rs):
        c = self.c
        w2 = Weights(self.m0, None, lon, lat, three_d=False)
        pbl = self.time_2d("pbl", time, lon, lat, w2)
        ps = self.time_2d("ps", time, lon, lat, w2)
        ptop = self.m0.p[-1]

        def kz_at(pp, la):
            wpbl = self.pbl_weight(pp, pbl, ps)
            wtrop = self.tropo_weight(time, la, pp) * (1.0 - wpbl)
            wstrat = 1.0 - wpbl - wtrop
            return wpbl, wtrop, wstrat, wpbl * c.turb_dz_pbl + wtrop * c.turb_dz_trop + wstrat * c.turb_dz_strat

        wpbl, wtrop, wstrat, Kz = kz_at(p, lat)
        Kx = wpbl * c.turb_dx_pbl + wtrop * c.turb_dx_trop + wstrat * c.turb_dx_strat
        dt_abs = np.abs(dt)
        sigma_h = np.sqrt(2.0 * Kx * dt_abs)
        horiz = Kx > 0
        new_lon = np.where(horiz, lon + DX2DEG(rs[0::3] * sigma_h / 1000.0, lat), lon)
        new_lat = np.where(horiz, lat + DY2DEG(rs[1::3] * sigma_h / 1000.0), lat)
        # vertical part: the weights of the two probes see the latitude the horizontal part has just written
        sigma_z = np.sqrt(2.0 * Kz * dt_abs) * 1e-3
        eps_km = 0.01
        p_up = p + DZ2DP(eps_km, p)
        p_dn = p + DZ2DP(-eps_km, p)
        Kz_up = kz_at(np.maximum(ptop, np.minimum(ps, p_up)), new_lat)[3]
        Kz_dn = kz_at(np.maximum(ptop, np.minimum(ps, p_dn)), new_lat)[3]
        dKz_dz = (Kz_up - Kz_dn) / (2.0 * eps_km * 1e3)
        w_drift = dKz_dz + Kz * (-1.0 / (1e3 * H0))
        dz_tot = rs[2::3] * sigma_z + w_drift * dt_abs * 1e-3
        ptrial = p + DZ2DP(dz_tot, p)
        for _ in range(10):
            over, under = ptrial > ps, ptrial < ptop
            with np.errstate(all="ignore"):
                ptrial = np.where(over, ps * ps / ptrial, np.where(under, ptop * ptop / ptrial, ptrial))
        new_p = np.where(Kz > 0, np.maximum(ptop, np.minimum(ps, ptrial)), p)
        return new_lon, new_lat, new_p

    # -- module_convection (mptrac.c:4102-4171) -----------------------------------------------------------------
    def convection(self, time, lon, lat, p, rs):
        c = self.c
        w2 = Weights(self.m0, None, lon, lat, three_d=False)
        ps = self.time_2d("ps", time, lon, lat, w2)
        pbot, ptop = ps, ps
        if c.conv_mix_pbl:
            pbl = self.time_2d("pbl", time, lon, lat, w2)
            ptop = pbl - c.conv_pbl_trans * (ps - pbl)
        if c.conv_cape >= 0:
            cape = self.time_2d("cape", time, lon, lat, w2)
            cin = self.time_2d("cin", time, lon, lat, w2)
            pel = self.time_2d("pel", time, lon, lat, w2)
            with np.errstate(invalid="ignore"):
                deep = np.isfinite(cape) & (cape >= c.conv_cape)
                if c.conv_cin > 0:
                    deep &= np.isfinite(cin) & (cin >= c.conv_cin)
                ptop = np.where(deep, np.minimum(ptop, pel), ptop)
        act = (ptop != pbot) & (p >= ptop)
        tbot = self.time_3d("t", time, pbot, lon, lat)
        ttop = self.time_3d("t", time, ptop, lon, lat)
        rhobot, rhotop = pbot / tbot, ptop / ttop
        rho = rhobot + (rhotop - rhobot) * rs
        with np.errstate(all="ignore"):
            mixed = LIN(rhobot, pbot, rhotop, ptop, rho)
        return np.where(act, mixed, p)

    # -- sedi (mptrac.c:12506-12535) and module_sedi (mptrac.c:5869-5882) --------------------------------------------
    @staticmethod
    def sedi(p, T, rp, rhop):
        r = rp * 1e-6
        rho = 100.0 * p / (RA * T)
        eta = 1.8325e-5 * (416.16 / (T + 120.0)) * np.power(T / 296.16, 1.5)
        v = np.sqrt(8.0 * KB * T / (np.pi * M_AIR_MOLECULE))
        lam = 2.0 * eta / (rho * v)
        K = lam / r
        G = 1.0 + K * (1.249 + 0.42 * np.exp(-0.87 / K))
        return 2.0 * (r * r) * (rhop - rho) * G0 / (9.0 * eta) * G

    def sedimentation(self, time, lon, lat, p, dt, rp, rhop):
        t = self.time_3d("t", time, p, lon, lat)
        v_s = self.sedi(p, t, rp, rhop)
        return p + DZ2DP(v_s * dt / 1000.0, p)

    # -- module_mixing (mptrac.c:5169-5347), NENS 0 -----------------------------------------------------------------
    def mixing(self, t, time, lon, lat, p, q_rows):
        """q_rows: the arrays of the mixed quantities, in the order the reference visits them; returns new arrays"""
        c = self.c
        dz = (c.mixing_z1 - c.mixing_z0) / c.mixing_nz
        dlon = (c.mixing_lon1 - c.mixing_lon0) / c.mixing_nx
        dlat = (c.mixing_lat1 - c.mixing_lat0) / c.mixing_ny
        t0, t1 = t - 0.5 * c.dt_mod, t + 0.5 * c.dt_mod
        zpart = Z(p)
        out = ((time < t0) | (time > t1) | (lon < c.mixing_lon0) | (lon >= c.mixing_lon1) | (lat < c.mixing_lat0)
               | (lat >= c.mixing_lat1) | (zpart < c.mixing_z0) | (zpart >= c.mixing_z1))
        ixs = np.trunc((lon - c.mixing_lon0) / dlon).astype(np.int64)
        iys = np.trunc((lat - c.mixing_lat0) / dlat).astype(np.int64)
        izs = np.trunc((zpart - c.mixing_z0) / dz).astype(np.int64)
        inside = ~out & ~((ixs >= c.mixing_nx) | (iys >= c.mixing_ny) | (izs >= c.mixing_nz))
        idx = (ixs * c.mixing_ny + iys) * c.mixing_nz + izs          # ARRAY_3D
        ngrid = c.mixing_nx * c.mixing_ny * c.mixing_nz
        mixparam = np.ones_like(p)
        if c.mixing_trop < 1 or c.mixing_strat < 1:
            w = self.tropo_weight(time, lat, p)
            mixparam = w * c.mixing_trop + (1.0 - w) * c.mixing_strat
        res = []
        sel = np.nonzero(inside)[0]
        for q in q_rows:
            cmean = np.zeros(ngrid)
            count = np.zeros(ngrid, dtype=np.int64)
            for ip in sel:                      # the serial accumulation order of the reference's loop
                cmean[idx[ip]] += q[ip]
                count[idx[ip]] += 1
            with np.errstate(invalid="ignore"):
                cmean = np.where(count > 0, cmean / np.maximum(count, 1), cmean)
            new = q.copy()
            new[sel] = q[sel] + (cmean[idx[sel]] - q[sel]) * mixparam[sel]
            res.append(new)
        return res

    # -- module_wet_depo (mptrac.c:6155-6290): returns (acts, aux, lambda) ------------------------------------------
    def wet_depo(self, time, lon, lat, p, dt):
        c = self.c
        w2 = Weights(self.m0, None, lon, lat, three_d=False)
        pct = self.time_2d("pct", time, lon, lat, w2)
        with np.errstate(invalid="ignore"):
            act = np.isfinite(pct) & ~(p <= pct)
        pcb = self.time_2d("pcb", time, lon, lat, w2)
        cl = self.time_2d("cl", time, lon, lat, w2)
        with np.errstate(all="ignore"):
            Is = np.power(1.0 / c.wet_depo_pre[0] * cl, 1.0 / c.wet_depo_pre[1])
            act &= ~(Is < 0.01)
        w3 = Weights(self.m0, p, lon, lat)
        lwc, rwc, iwc, swc = (self.time_3d(k, time, p, lon, lat, w3) for k in ("lwc", "rwc", "iwc", "swc"))
        inside = (lwc > 0) | (rwc > 0) | (iwc > 0) | (swc > 0)
        t = self.time_3d("t", time, p, lon, lat, w3)
        with np.errstate(all="ignore"):
            dzc = 1e3 * (Z(pct) - Z(pcb))
            # in-cloud
            eta_ic = np.where(t > WD_T_LIQUID, 1.0, np.where(t <= WD_T_ICE, c.wet_depo_ic_ret_ratio,
                                                             LIN(WD_T_LIQUID, 1.0, WD_T_ICE, c.wet_depo_ic_ret_ratio, t)))
            lam_ic = np.zeros_like(p)
            if c.wet_depo_ic_a > 0:
                lam_ic = c.wet_depo_ic_a * np.power(Is, c.wet_depo_ic_b) * eta_ic
            elif c.wet_depo_ic_h[0] > 0:
                h = c.wet_depo_ic_h[0] * np.exp(c.wet_depo_ic_h[1] * (1.0 / t - 1.0 / T_REF))
                if c.wet_depo_so2_ph > 0:
                    H_ion = np.power(10.0, -c.wet_depo_so2_ph)
                    K_1 = SO2_K1_REF * np.exp(SO2_K1_TEMP * (1.0 / t - 1.0 / T_REF))
                    K_2 = SO2_K2_REF * np.exp(SO2_K2_TEMP * (1.0 / t - 1.0 / T_REF))
                    h = h * (1.0 + K_1 / H_ion + K_1 * K_2 / (H_ion * H_ion))
                lam_ic = h * RI * t * Is / 3.6e6 / dzc * eta_ic
            # below cloud
            eta_bc = np.where(t > WD_T_LIQUID_BC, 1.0, c.wet_depo_bc_ret_ratio)
            lam_bc = np.zeros_like(p)
            if c.wet_depo_bc_a > 0:
                lam_bc = c.wet_depo_bc_a * np.power(Is, c.wet_depo_bc_b) * eta_bc
            elif c.wet_depo_bc_h[0] > 0:
                h = c.wet_depo_bc_h[0] * np.exp(c.wet_depo_bc_h[1] * (1.0 / t - 1.0 / T_REF))
                lam_bc = h * RI * t * Is / 3.6e6 / dzc * eta_bc
            lam = np.where(inside, lam_ic, lam_bc)
            aux = np.exp(-dt * lam)
        return act, aux, lam

    # -- module_dry_depo (mptrac.c:4738-4797): returns (acts, aux, rate) ---------------------------------------------
    def dry_depo(self, time, lon, lat, p, dt, rp=None, rhop=None):
        c = self.c
        ps = self.time_2d("ps", time, lon, lat)
        act = ~(p < ps - c.dry_depo_dp)
        dz = 1000.0 * (Z(ps - c.dry_depo_dp) - Z(ps))
        if c.qnt_rp > 0 and c.qnt_rhop > 0:                 # "> 0" as the reference writes it
            t = self.time_3d("t", time, p, lon, lat)
            v_dep = self.sedi(p, t, rp, rhop)
        else:
            v_dep = np.full_like(p, c.dry_depo_vdep)
        aux = np.exp(-dt * v_dep / dz)
        return act, aux, v_dep / dz

    def apply_loss(self, q, act, aux, rate, which):
        """the bookkeeping both deposition modules share (mptrac.c:6276-6287, 4784-4794); q: dict name -> array"""
        q = {k: v.copy() for k, v in q.items()}
        if "m" in q:
            if which in q:
                q[which] = np.where(act, q[which] + q["m"] * (1 - aux), q[which])
            q["m"] = np.where(act, q["m"] * aux, q["m"])
            if "loss_rate" in q:
                q["loss_rate"] = np.where(act, q["loss_rate"] + rate, q["loss_rate"])
        if "vmr" in q:
            q["vmr"] = np.where(act, q["vmr"] * aux, q["vmr"])
        return q
