"""ORACLE (test infrastructure) -- D-grid shallow-water step (Fortran d_sw) restated in numpy.

Follows fv3core/pace/fv3core/stencils/fxadv.py:10-661 (FiniteVolumeFluxPrep) and
fv3core/pace/fv3core/stencils/d_sw.py:33-1237 (DGridShallowWaterLagrangianDynamics), using
oracle/ppm_transport.py (fvtp2d, delnflux, xtp_u/ytp_v) and oracle/damping.py.
Parity status: see oracle/ppm_transport.py header.
"""
import numpy as np

from . import damping as damping_mod
from . import ppm_transport as tr
from ._np import kcol, put, sh

DCON_THRESHOLD = 1e-5  # d_sw.py:30


def _contra(v1, v2, cosa, rsin2):
    """d2a2c_vect.py:225-281 contravariant."""
    return (v1 - v2 * cosa) * rsin2


def fxadv(g, uc, vc, crx, cry, xfx, yfx, uc_contra, vc_contra, dt):
    """FiniteVolumeFluxPrep.__call__ (fxadv.py:565-661); uc_contra/vc_contra keep values the
    reference leaves untouched (fxadv.py:34-43 'utmp')."""
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    full_o, full_d = (0, 0), (n + 6, n + 6)
    I, J = g.I, g.J
    cosa_u, cosa_v, rsin_u, rsin_v = g.m2("cosa_u"), g.m2("cosa_v"), g.m2("rsin_u"), g.m2("rsin_v")
    sg1, sg2, sg3, sg4 = g.m2("sin_sg1"), g.m2("sin_sg2"), g.m2("sin_sg3"), g.m2("sin_sg4")
    with np.errstate(all="ignore"):
        # main_uc_vc_contra :10-48
        vbar = 0.25 * (sh(vc, -1, 0) + vc + sh(vc, -1, 1) + sh(vc, 0, 1))
        m = (I >= is_ - 1) & (I <= ie + 2) & ~(((J >= js - 1) & (J <= js)) | ((J >= je) & (J <= je + 1)))
        put(uc_contra, _contra(uc, vbar, cosa_u, rsin_u), full_o, full_d, mask=m)
        ubar = 0.25 * (sh(uc, 0, -1) + sh(uc, 1, -1) + uc + sh(uc, 1, 0))
        m = (J >= js - 1) & (J <= je + 2)
        put(vc_contra, _contra(vc, ubar, cosa_v, rsin_v), full_o, full_d, mask=m)
        # uc_contra_y_edge :51-77
        m = (I == is_) | (I == ie + 1)
        put(uc_contra, np.where(uc > 0, uc / sh(sg3, -1, 0), uc / sg1), full_o, full_d, mask=m)
        # vc_contra_y_edge :80-125
        ucb = 0.25 * (sh(uc_contra, 0, -1) + sh(uc_contra, 1, -1) + uc_contra + sh(uc_contra, 1, 0))
        strip = ((I >= is_ - 1) & (I <= is_)) | ((I >= ie) & (I <= ie + 1))
        m = strip & (J >= js) & (J <= je + 1) & ~(((J >= js) & (J <= js + 1)) | ((J >= je) & (J <= je + 1)))
        put(vc_contra, _contra(vc, ucb, cosa_v, 1.0), full_o, full_d, mask=m)
        # vc_contra_x_edge :128-145
        m = (J == js) | (J == je + 1)
        put(vc_contra, np.where(vc > 0, vc / sh(sg4, 0, -1), vc / sg2), full_o, full_d, mask=m)
        # uc_contra_x_edge :148-180
        vcb = 0.25 * (sh(vc_contra, -1, 0) + vc_contra + sh(vc_contra, -1, 1) + sh(vc_contra, 0, 1))
        rows = ((J >= js - 1) & (J <= js)) | ((J >= je) & (J <= je + 1))
        m = rows & (I >= is_) & (I <= ie + 1) & ~(((I >= is_) & (I <= is_ + 1)) | ((I >= ie) & (I <= ie + 1)))
        put(uc_contra, _contra(uc, vcb, cosa_u, 1.0), full_o, full_d, mask=m)
        # uc_contra_corners :183-300 on origin_full(add=(1,1,0)), domain_full(add=(-1,-1,0))
        co, cd = (1, 1), (n + 5, n + 5)
        ucc = uc_contra.copy()
        new = uc_contra.copy()
        vt = vc_contra

        def set_u(mask, val):
            nonlocal new
            new = np.where(mask, val, new)

        damp = 1.0 / (1.0 - 0.0625 * cosa_u * sh(cosa_v, -1, 0))
        set_u((I == is_ + 1) & ((J == js - 1) | (J == je)),
              (uc - 0.25 * cosa_u * (sh(vt, -1, 1) + sh(vt, 0, 1) + vt + sh(vc, -1, 0)
                                     - 0.25 * sh(cosa_v, -1, 0) * (sh(ucc, -1, 0) + sh(ucc, -1, -1) + sh(ucc, 0, -1)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * cosa_u * sh(cosa_v, -1, 1))
        set_u((I == is_ + 1) & ((J == js) | (J == je + 1)),
              (uc - 0.25 * cosa_u * (sh(vt, -1, 0) + vt + sh(vt, 0, 1) + sh(vc, -1, 1)
                                     - 0.25 * sh(cosa_v, -1, 1) * (sh(ucc, -1, 0) + sh(ucc, -1, 1) + sh(ucc, 0, 1)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * cosa_u * cosa_v)
        set_u((I == ie) & ((J == js - 1) | (J == je)),
              (uc - 0.25 * cosa_u * (sh(vt, 0, 1) + sh(vt, -1, 1) + sh(vt, -1, 0) + vc
                                     - 0.25 * cosa_v * (sh(ucc, 1, 0) + sh(ucc, 1, -1) + sh(ucc, 0, -1)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * cosa_u * sh(cosa_v, 0, 1))
        set_u((I == ie) & ((J == js) | (J == je + 1)),
              (uc - 0.25 * cosa_u * (vt + sh(vt, -1, 0) + sh(vt, -1, 1) + sh(vc, 0, 1)
                                     - 0.25 * sh(cosa_v, 0, 1) * (sh(ucc, 1, 0) + sh(ucc, 1, 1) + sh(ucc, 0, 1)))) * damp)
        put(uc_contra, new, co, cd)
        # vc_contra_corners :303-404
        ut = uc_contra
        vcc = vc_contra.copy()
        new = vc_contra.copy()

        def set_v(mask, val):
            nonlocal new
            new = np.where(mask, val, new)

        damp = 1.0 / (1.0 - 0.0625 * sh(cosa_u, 0, -1) * cosa_v)
        set_v(((I == is_ - 1) | (I == ie)) & (J == js + 1),
              (vc - 0.25 * cosa_v * (sh(ut, 1, -1) + sh(ut, 1, 0) + ut + sh(uc, 0, -1)
                                     - 0.25 * sh(cosa_u, 0, -1) * (sh(vcc, 0, -1) + sh(vcc, -1, -1) + sh(vcc, -1, 0)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * sh(cosa_u, 1, -1) * cosa_v)
        set_v(((I == is_) | (I == ie + 1)) & (J == js + 1),
              (vc - 0.25 * cosa_v * (sh(ut, 0, -1) + ut + sh(ut, 1, 0) + sh(uc, 1, -1)
                                     - 0.25 * sh(cosa_u, 1, -1) * (sh(vcc, 0, -1) + sh(vcc, 1, -1) + sh(vcc, 1, 0)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * sh(cosa_u, 1, 0) * cosa_v)
        set_v(((I == ie + 1) | (I == is_)) & (J == je),
              (vc - 0.25 * cosa_v * (ut + sh(ut, 0, -1) + sh(ut, 1, -1) + sh(uc, 1, 0)
                                     - 0.25 * sh(cosa_u, 1, 0) * (sh(vcc, 0, 1) + sh(vcc, 1, 1) + sh(vcc, 1, 0)))) * damp)
        damp = 1.0 / (1.0 - 0.0625 * cosa_u * cosa_v)
        set_v(((I == ie) | (I == is_ - 1)) & (J == je),
              (vc - 0.25 * cosa_v * (sh(ut, 1, 0) + sh(ut, 1, -1) + sh(ut, 0, -1) + uc
                                     - 0.25 * cosa_u * (sh(vcc, 0, 1) + sh(vcc, -1, 1) + sh(vcc, -1, 0)))) * damp)
        put(vc_contra, new, co, cd)
        # fxadv_fluxes_stencil :436-486
        m = (I >= is_) & (I <= ie + 1)
        pos = uc_contra > 0
        put(crx, np.where(pos, dt * uc_contra * sh(g.m2("rdxa"), -1, 0), dt * uc_contra * g.m2("rdxa")), full_o, full_d, mask=m)
        put(xfx, np.where(pos, g.m2("dy") * dt * uc_contra * sh(sg3, -1, 0), g.m2("dy") * dt * uc_contra * sg1), full_o, full_d, mask=m)
        m = (J >= js) & (J <= je + 1)
        pos = vc_contra > 0
        put(cry, np.where(pos, dt * vc_contra * sh(g.m2("rdya"), 0, -1), dt * vc_contra * g.m2("rdya")), full_o, full_d, mask=m)
        put(yfx, np.where(pos, g.m2("dx") * dt * vc_contra * sh(sg4, 0, -1), g.m2("dx") * dt * vc_contra * sg2), full_o, full_d, mask=m)


def expand_col(values4, nk):
    """Per-level array from the reference's (level0, level1, level2, level>=3) externals."""
    out = np.full(nk, float(values4[3]))
    out[: min(3, nk)] = np.asarray(values4[:3], dtype=float)[: min(3, nk)]
    return out


class DSWState:
    """Persistent temporaries of DGridShallowWaterLagrangianDynamics (d_sw.py:765-784)."""

    def __init__(self, shape):
        self.uc_contra = np.zeros(shape)
        self.vc_contra = np.zeros(shape)


def d_sw(g, col, cfg, st, delpc, delp, pt, u, v, w, uc, vc, ua, va, divgd, mfx, mfy, cx, cy, crx, cry, xfx, yfx,
         q_con, zh, heat_source, diss_est, dt):
    """DGridShallowWaterLagrangianDynamics.__call__ (d_sw.py:935-1237).

    col: column namelist dict of K arrays (d_sw.get_column_namelist, d_sw.py:633-683);
    cfg: dict with hord_dp, hord_tm, hord_vt, hord_mt, dddmp, d4_bg, nord, d_con, do_skeb.
    """
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    nk = g.nk
    nkt = u.shape[2]
    I, J = g.I, g.J
    rarea = g.m2("rarea")
    co, cd = (is_, js), (n, n)
    cd1 = (n + 1, n + 1)
    shape = u.shape
    z = lambda: np.zeros(shape)  # noqa: E731
    fx, fy, gx, gy, fx2, fy2, dw, wk, heat_s = z(), z(), z(), z(), z(), z(), z(), z(), z()
    ut, vt, ke, vort_a, vort_b, abs_vort, vort_x_delta, vort_y_delta, damped = z(), z(), z(), z(), z(), z(), z(), z(), z()
    damp_w3, damp_vt3 = kcol(col["damp_w"], nkt), kcol(col["damp_vt"], nkt)
    ke_bg3, dcon3 = kcol(col["ke_bg"], nkt), kcol(col["d_con"], nkt)
    with np.errstate(all="ignore"):
        fxadv(g, uc, vc, crx, cry, xfx, yfx, st.uc_contra, st.vc_contra, dt)
        tr.fvtp2d(g, delp, crx, cry, xfx, yfx, fx, fy, cfg["hord_dp"], nord_k=col["nord_v"], damp_c_k=col["damp_vt"])
        # flux_capacitor :33-60, compute domain + halo 3
        put(cx, cx + crx, (0, 0), (n + 6, n + 6), k1=nk)
        put(cy, cy + cry, (0, 0), (n + 6, n + 6), k1=nk)
        put(mfx, mfx + fx, (0, 0), (n + 6, n + 6), k1=nk)
        put(mfy, mfy + fy, (0, 0), (n + 6, n + 6), k1=nk)
        damp_w = tr.calc_damp(col["damp_w"][:nk], g.da_min_c, col["nord_w"][:nk])
        tr.delnflux_nosg(g, w, fx2, fy2, damp_w, wk, col["nord_w"])
        # heat_diss :63-103
        on = damp_w3 > 1e-5
        dwn = (fx2 - sh(fx2, 1, 0) + fy2 - sh(fy2, 0, 1)) * rarea
        put(dw, dwn, co, cd, mask=on, k1=nk)
        dd8 = ke_bg3 * abs(dt)
        hs = np.where(on, dd8 - dw * (w + 0.5 * dw), 0.0)
        put(heat_s, hs, co, cd, k1=nk)
        put(diss_est, hs, co, cd, k1=nk)
        tr.fvtp2d(g, w, crx, cry, xfx, yfx, gx, gy, cfg["hord_vt"], x_mass_flux=fx, y_mass_flux=fy)
        # apply_fluxes :122-145
        put(w, w * delp + (gx - sh(gx, 1, 0) + gy - sh(gy, 0, 1)) * rarea, co, cd, k1=nk)
        tr.fvtp2d(g, q_con, crx, cry, xfx, yfx, gx, gy, cfg["hord_dp"], x_mass_flux=fx, y_mass_flux=fy, mass=delp,
                  nord_k=col["nord_t"], damp_c_k=col["damp_t"])
        put(q_con, q_con * delp + (gx - sh(gx, 1, 0) + gy - sh(gy, 0, 1)) * rarea, co, cd, k1=nk)
        tr.fvtp2d(g, pt, crx, cry, xfx, yfx, gx, gy, cfg["hord_tm"], x_mass_flux=fx, y_mass_flux=fy, mass=delp,
                  nord_k=col["nord_v"], damp_c_k=col["damp_vt"])
        # apply_pt_delp_fluxes :148-201 (region local compute)
        ptn = pt * delp + (gx - sh(gx, 1, 0) + gy - sh(gy, 0, 1)) * rarea
        delpn = delp + (fx - sh(fx, 1, 0) + fy - sh(fy, 0, 1)) * rarea
        put(pt, ptn / delpn, co, cd, k1=nk)
        put(delp, delpn, co, cd, k1=nk)
        # adjust_w_and_qcon :331-350
        wn = w / delp
        wn = np.where(on, wn + dw, wn)
        put(w, wn, co, cd, k1=nk)
        put(q_con, q_con / delp, co, cd, k1=nk)
        # compute_kinetic_energy :204-256
        uct, vct = st.uc_contra, st.vc_contra
        cosa, rsina = g.m2("cosa"), g.m2("rsina")
        ub_cov = 0.5 * (sh(uc, 0, -1) + uc)
        vb_cov = 0.5 * (sh(vc, -1, 0) + vc)
        ub = (ub_cov - vb_cov * cosa) * rsina
        vb = (vb_cov - ub_cov * cosa) * rsina
        jedge = (J == js) | (J == je + 1)
        iedge = (I == is_) | (I == ie + 1)
        ub = np.where(jedge, 0.25 * (-sh(uct, 0, -2) + 3.0 * (sh(uct, 0, -1) + uct) - sh(uct, 0, 1)), ub)
        ub = np.where(iedge, 0.5 * (sh(uct, 0, -1) + uct), ub)
        vb = np.where(iedge, 0.25 * (-sh(vct, -2, 0) + 3.0 * (sh(vct, -1, 0) + vct) - sh(vct, 1, 0)), vb)
        vb = np.where(jedge, 0.5 * (sh(vct, -1, 0) + vct), vb)
        adv_v = tr.advect_wind_1d(v, vb, g.rdy, g.dy, g.dya, dt, g, 1, cfg["hord_mt"])
        adv_u = tr.advect_wind_1d(u, ub, g.rdx, g.dx, g.dxa, dt, g, 0, cfg["hord_mt"])
        ken = 0.5 * dt * (ub * adv_u + vb * adv_v)
        dt6 = dt / 6.0

        def corner_ke(io1, jo1, io2, vsign):
            # d_sw.py:259-281
            return dt6 * (
                (uct + sh(uct, 0, -1)) * ((io1 + 1) * u - (io1 * sh(u, -1, 0)))
                + (vct + sh(vct, -1, 0)) * ((jo1 + 1) * v - (jo1 * sh(v, 0, -1)))
                + (((jo1 + 1) * uct - (jo1 * sh(uct, 0, -1))) + vsign * ((io1 + 1) * vct - (io1 * sh(vct, -1, 0))))
                * ((io2 + 1) * u - (io2 * sh(u, -1, 0)))
            )

        ken = np.where((I == is_) & (J == js), corner_ke(0, 0, -1, 1), ken)
        ken = np.where((I == ie + 1) & (J == js), corner_ke(-1, 0, 0, -1), ken)
        ken = np.where((I == ie + 1) & (J == je + 1), corner_ke(-1, -1, 0, 1), ken)
        ken = np.where((I == is_) & (J == je + 1), corner_ke(0, -1, -1, -1), ken)
        put(ke, ken, co, cd1, k1=nk)
        # compute_vorticity :301-328 (compute + halo 3)
        dx, dy = g.m2("dx"), g.m2("dy")
        va_n = (u - sh(u, 0, 1) * sh(dx, 0, 1) / dx) * (rarea * dx) + (sh(v, 1, 0) * sh(dy, 1, 0) / dy - v) * (rarea * dy)
        put(vort_a, va_n, (0, 0), (n + 6, n + 6), k1=nk)
        damping_mod.divergence_damping(
            g, u, v, va, vort_b, ua, divgd, vc, uc, delpc, ke, vort_a, dt, nord_k=col["nord"], d2_bg_k=col["d2_divg"],
            dddmp=cfg["dddmp"], d4_bg=cfg["d4_bg"], nord=cfg["nord"],
        )
        # rel_vorticity_to_abs :389-402
        put(abs_vort, vort_a + g.m2("fC_agrid"), (0, 0), (n + 6, n + 6), k1=nk)
        tr.fvtp2d(g, abs_vort, crx, cry, xfx, yfx, fx, fy, cfg["hord_vt"])
        # u_and_v_from_ke :439-477
        put(u, u * dx + ke - sh(ke, 1, 0) + fy, co, (n, n + 1), k1=nk)
        put(v, v * dy + ke - sh(ke, 0, 1) - fx, co, (n + 1, n), k1=nk)
        damp_vt = tr.calc_damp(col["damp_vt"][:nk], g.da_min_c, col["nord_v"][:nk])
        tr.delnflux_nosg(g, vort_a, ut, vt, damp_vt, damped, col["nord_v"])
        # vort_differencing :353-380 -- the reference tests dcon[0] of the *current level* (K-field offset 0)
        don = dcon3 > DCON_THRESHOLD
        put(vort_x_delta, vort_b - sh(vort_b, 1, 0), co, (n, n + 1), mask=don, k1=nk)
        put(vort_y_delta, vort_b - sh(vort_b, 0, 1), co, (n + 1, n), mask=don, k1=nk)
        # heat_source_from_vorticity_damping :493-577 on compute+1
        rdx, rdy = g.m2("rdx"), g.m2("rdy")
        ubt = (vort_x_delta + vt) * rdx
        fy_ = u * rdx
        gy_ = fy_ * ubt
        vbt = (vort_y_delta - ut) * rdy
        fx_ = v * rdy
        gx_ = fx_ * vbt
        do_skeb = cfg.get("do_skeb", False)
        cond = (dcon3 > DCON_THRESHOLD) | do_skeb
        u2 = fy_ + sh(fy_, 0, 1)
        du2 = ubt + sh(ubt, 0, 1)
        v2 = fx_ + sh(fx_, 1, 0)
        dv2 = vbt + sh(vbt, 1, 0)
        dampterm = g.m2("rsin2") * 0.25 * (
            (ubt * ubt + sh(ubt, 0, 1) * sh(ubt, 0, 1) + vbt * vbt + sh(vbt, 1, 0) * sh(vbt, 1, 0))
            + 2.0 * (gy_ + sh(gy_, 0, 1) + gx_ + sh(gx_, 1, 0))
            - g.m2("cosa_s") * (u2 * dv2 + v2 * du2 + du2 * dv2)
        )
        hs_new = delp * (heat_s - dcon3 * dampterm)
        put(heat_s, hs_new, co, cd1, mask=cond, k1=nk)
        if cfg["d_con"] > DCON_THRESHOLD or do_skeb:
            put(heat_source, heat_source + heat_s, co, cd, k1=nk)
            if do_skeb:
                put(diss_est, diss_est - dampterm, co, cd, k1=nk)
        # update_u_and_v :582-608
        von = damp_vt3 > 1e-5
        put(u, u + vt, co, (n, n + 1), mask=von, k1=nk)
        put(v, v - ut, co, (n + 1, n), mask=von, k1=nk)
