"""ORACLE (test infrastructure) -- C-grid half step (Fortran c_sw) and the D->A->C wind
interpolation (d2a2c_vect), restated in numpy.

Follows fv3core/pace/fv3core/stencils/d2a2c_vect.py:14-655, c_sw.py:19-766 and the multiplier corner
fills stencils/pace/stencils/corners.py:129-305.  Parity status: see oracle/ppm_transport.py header.
"""
import numpy as np

from ._np import put, sh

# d2a2c_vect.py:14-17, a2b_ord4.py:31-33
C1 = -2.0 / 14.0
C2 = 11.0 / 14.0
C3 = 5.0 / 14.0
A1 = 9.0 / 16.0
A2 = -1.0 / 16.0


def _contra(v1, v2, cosa, rsin2):
    return (v1 - v2 * cosa) * rsin2


def fill_corners_cells_mult(q, qc, g, direction, ncells, sw=1.0, se=1.0, nw=1.0, ne=1.0, ks=slice(None)):
    """corners.py:129-305 fill_corners_{2,3}cells_mult_{x,y}: q's corner row/column from qc, in place.
    Closed form: a = 1..ncells cells beyond the edge."""
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    todo = []
    for a in range(1, ncells + 1):
        if direction == "x":
            todo += [((is_ - a, js - 1), sw, (is_ - 1, js + a - 1)), ((ie + a, js - 1), se, (ie + 1, js + a - 1)),
                     ((is_ - a, je + 1), nw, (is_ - 1, je + 1 - a)), ((ie + a, je + 1), ne, (ie + 1, je + 1 - a))]
        else:
            todo += [((is_ - 1, js - a), sw, (is_ + a - 1, js - 1)), ((ie + 1, js - a), se, (ie + 1 - a, js - 1)),
                     ((is_ - 1, je + a), nw, (is_ + a - 1, je + 1)), ((ie + 1, je + a), ne, (ie + 1 - a, je + 1))]
    vals = [m * qc[s[0], s[1], ks] for (_, m, s) in todo]
    for (d, _, _), v in zip(todo, vals):
        q[d[0], d[1], ks] = v


class D2A2CState:
    def __init__(self, shape):
        self.utmp = np.zeros(shape)
        self.vtmp = np.zeros(shape)


def d2a2c_vect(g, st, uc, vc, u, v, ua, va, utc, vtc):
    """DGrid2AGrid2CGridVectors.__call__ (d2a2c_vect.py:529-655), dord4=True, npt=4."""
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    nk = g.nk
    I, J = g.I, g.J
    utmp, vtmp = st.utmp, st.vtmp
    npt = 4 if (4 <= n - 1) else 0
    avg_off = 3 if npt else -1
    with np.errstate(all="ignore"):
        put(utmp, 1e30, (0, 0), (n + 6, n + 6), k1=nk)
        put(vtmp, 1e30, (0, 0), (n + 6, n + 6), k1=nk)
        lo, hi = npt + 2, (ie + 1) - npt  # OFFSET = 2
        if hi >= lo:
            put(utmp, A2 * (sh(u, 0, -1) + sh(u, 0, 2)) + A1 * (u + sh(u, 0, 1)), (lo, lo), (hi - lo + 1, hi - lo + 1), k1=nk)
            put(vtmp, A2 * (sh(v, -1, 0) + sh(v, 2, 0)) + A1 * (v + sh(v, 1, 0)), (lo, lo), (hi - lo + 1, hi - lo + 1), k1=nk)
        m = (J < js + avg_off) | (J >= je - avg_off + 1) | (I < is_ + avg_off) | (I >= ie - avg_off + 1)
        put(utmp, 0.5 * (u + sh(u, 0, 1)), (0, 0), (n + 6, n + 6), mask=m, k1=nk)
        put(vtmp, 0.5 * (v + sh(v, 1, 0)), (0, 0), (n + 6, n + 6), mask=m, k1=nk)
        cs, r2 = g.m2("cosa_s"), g.m2("rsin2")
        put(ua, _contra(utmp, vtmp, cs, r2), (is_ - 2, js - 2), (n + 4, n + 4), k1=nk)
        put(va, _contra(vtmp, utmp, cs, r2), (is_ - 2, js - 2), (n + 4, n + 4), k1=nk)
        ks = slice(0, nk)
        fill_corners_cells_mult(utmp, vtmp, g, "x", 3, sw=-1, se=1, ne=-1, nw=1, ks=ks)
        fill_corners_cells_mult(ua, va, g, "x", 2, sw=-1, se=1, ne=-1, nw=1, ks=ks)
        cu, ru = g.m2("cosa_u"), g.m2("rsin_u")
        # ut_main
        ucn = A2 * (sh(utmp, -2, 0) + sh(utmp, 1, 0)) + A1 * (sh(utmp, -1, 0) + utmp)
        put(uc, ucn, (is_ + 2, js - 1), (n - 3, n + 2), k1=nk)
        put(utc, _contra(uc, v, cu, ru), (is_ + 2, js - 1), (n - 3, n + 2), k1=nk)
        # east_west_edges
        rows = (J >= js - 1) & (J <= je + 1)
        eo, ed = (is_ - 3, js - 3), (n + 6, n + 6)
        dxa = g.m2("dxa")
        sg1, sg3 = g.m2("sin_sg1"), g.m2("sin_sg3")
        cub = C1 * sh(utmp, -2, 0) + C2 * sh(utmp, -1, 0) + C3 * utmp
        rev = C1 * sh(utmp, 1, 0) + C2 * utmp + C3 * sh(utmp, -1, 0)
        t1 = sh(dxa, -2, 0) + sh(dxa, -1, 0)
        t2 = dxa + sh(dxa, 1, 0)
        n1 = (t1 + sh(dxa, -1, 0)) * sh(ua, -1, 0) - sh(dxa, -1, 0) * sh(ua, -2, 0)
        n2 = (t1 + dxa) * ua - dxa * sh(ua, 1, 0)
        edge = 0.5 * (n1 / t1 + n2 / t2)
        for i0 in (is_, ie + 1):
            put(uc, cub, eo, ed, mask=rows & (I == i0 - 1), k1=nk)
            put(utc, edge, eo, ed, mask=rows & (I == i0), k1=nk)
            put(uc, np.where(utc > 0, utc * sh(sg3, -1, 0), utc * sg1), eo, ed, mask=rows & (I == i0), k1=nk)
            put(uc, rev, eo, ed, mask=rows & (I == i0 + 1), k1=nk)
            put(utc, _contra(uc, v, cu, ru), eo, ed, mask=rows & ((I == i0 - 1) | (I == i0 + 1)), k1=nk)
        fill_corners_cells_mult(vtmp, utmp, g, "y", 3, sw=-1, se=1, ne=-1, nw=1, ks=ks)
        fill_corners_cells_mult(va, ua, g, "y", 2, sw=-1, se=1, ne=-1, nw=1, ks=ks)
        # north_south_edges
        cv, rv = g.m2("cosa_v"), g.m2("rsin_v")
        sg2, sg4 = g.m2("sin_sg2"), g.m2("sin_sg4")
        dya = g.m2("dya")
        cols = (I >= is_ - 1) & (I <= ie + 1)
        lag = A2 * (sh(vtmp, 0, -2) + sh(vtmp, 0, 1)) + A1 * (sh(vtmp, 0, -1) + vtmp)
        put(vc, lag, eo, ed, mask=cols & (J >= js - 1) & (J <= je + 2), k1=nk)
        put(vtc, _contra(vc, u, cv, rv), eo, ed, mask=cols & (J >= js - 1) & (J <= je + 2), k1=nk)
        cub = C1 * sh(vtmp, 0, -2) + C2 * sh(vtmp, 0, -1) + C3 * vtmp
        rev = C1 * sh(vtmp, 0, 1) + C2 * vtmp + C3 * sh(vtmp, 0, -1)
        t1 = sh(dya, 0, -2) + sh(dya, 0, -1)
        t2 = dya + sh(dya, 0, 1)
        n1 = (t1 + sh(dya, 0, -1)) * sh(va, 0, -1) - sh(dya, 0, -1) * sh(va, 0, -2)
        n2 = (t1 + dya) * va - dya * sh(va, 0, 1)
        edge = 0.5 * (n1 / t1 + n2 / t2)
        for j0 in (js, je + 1):
            put(vc, cub, eo, ed, mask=cols & (J == j0 - 1), k1=nk)
            put(vtc, _contra(vc, u, cv, rv), eo, ed, mask=cols & (J == j0 - 1), k1=nk)
            put(vtc, edge, eo, ed, mask=cols & (J == j0), k1=nk)
            put(vc, np.where(vtc > 0, vtc * sh(sg4, 0, -1), vtc * sg2), eo, ed, mask=cols & (J == j0), k1=nk)
            put(vc, rev, eo, ed, mask=cols & (J == j0 + 1), k1=nk)
            put(vtc, _contra(vc, u, cv, rv), eo, ed, mask=cols & (J == j0 + 1), k1=nk)
        # vt_main
        put(vc, lag, (is_ - 1, js + 2), (n + 2, n - 3), k1=nk)
        put(vtc, _contra(vc, u, cv, rv), (is_ - 1, js + 2), (n + 2, n - 3), k1=nk)


class CSWState:
    def __init__(self, shape):
        self.d2a2c = D2A2CState(shape)
        self.delpc = np.zeros(shape)
        self.ptc = np.zeros(shape)
        self.ke = np.zeros(shape)
        self.vort = np.zeros(shape)
        self.fx = np.zeros(shape)
        self.fx1 = np.zeros(shape)
        self.fx2 = np.zeros(shape)


def c_sw(g, st, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2, nord=3):
    """CGridShallowWaterDynamics.__call__ (c_sw.py:599-766).  Results delpc/ptc are in st."""
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    nk = g.nk
    I, J = g.I, g.J
    ks = slice(0, nk)
    m2 = g.m2
    with np.errstate(all="ignore"):
        put(st.delpc, 0.0, (0, 0), (n + 6, n + 6), k1=nk)
        put(st.ptc, 0.0, (0, 0), (n + 6, n + 6), k1=nk)
        d2a2c_vect(g, st.d2a2c, uc, vc, u, v, ua, va, ut, vt)
        if nord > 0:
            # divergence_corner :31-156 (compute + 1)
            sg1, sg2, sg3, sg4 = m2("sin_sg1"), m2("sin_sg2"), m2("sin_sg3"), m2("sin_sg4")
            cg1, cg2, cg3, cg4 = m2("cos_sg1"), m2("cos_sg2"), m2("cos_sg3"), m2("cos_sg4")
            dxc, dyc, rarea_c = m2("dxc"), m2("dyc"), m2("rarea_c")
            uf = (u - 0.25 * (sh(va, 0, -1) + va) * (sh(cg4, 0, -1) + cg2)) * dyc * 0.5 * (sh(sg4, 0, -1) + sg2)
            vf = (v - 0.25 * (sh(ua, -1, 0) + ua) * (sh(cg3, -1, 0) + cg1)) * dxc * 0.5 * (sh(sg3, -1, 0) + sg1)
            d = (sh(vf, 0, -1) - vf + sh(uf, -1, 0) - uf) * rarea_c
            iedge = (I == is_) | (I == ie + 1)
            jedge = (J == js) | (J == je + 1)
            vf0 = v * dxc * 0.5 * (sh(sg3, -1, 0) + sg1)
            vf1 = sh(v, 0, -1) * sh(dxc, 0, -1) * 0.5 * (sh(sg3, -1, -1) + sh(sg1, 0, -1))
            uf1 = ((sh(u, -1, 0) - 0.25 * (sh(va, -1, -1) + sh(va, -1, 0)) * (sh(cg4, -1, -1) + sh(cg2, -1, 0)))
                   * sh(dyc, -1, 0) * 0.5 * (sh(sg4, -1, -1) + sh(sg2, -1, 0)))
            d = np.where(iedge, (vf1 - vf0 + uf1 - uf) * rarea_c, d)
            uf0 = u * dyc * 0.5 * (sh(sg4, 0, -1) + sg2)
            uf1b = sh(u, -1, 0) * sh(dyc, -1, 0) * 0.5 * (sh(sg4, -1, -1) + sh(sg2, -1, 0))
            vf1b = ((sh(v, 0, -1) - 0.25 * (sh(ua, -1, -1) + sh(ua, 0, -1)) * (sh(cg3, -1, -1) + sh(cg1, 0, -1)))
                    * sh(dxc, 0, -1) * 0.5 * (sh(sg3, -1, -1) + sh(sg1, 0, -1)))
            d = np.where(jedge, (vf1b - vf + uf1b - uf0) * rarea_c, d)
            d = np.where(iedge & (J == js), (-vf0 + uf1b - uf0) * rarea_c, d)
            d = np.where(iedge & (J == je + 1), (vf1 + uf1b - uf0) * rarea_c, d)
            put(divgd, d, (is_, js), (n + 1, n + 1), k1=nk)
        # geoadjust_ut / vt :159-203 (halo 1)
        dy, dx = m2("dy"), m2("dx")
        sg1, sg2, sg3, sg4 = m2("sin_sg1"), m2("sin_sg2"), m2("sin_sg3"), m2("sin_sg4")
        put(ut, np.where(ut > 0, dt2 * ut * dy * sh(sg3, -1, 0), dt2 * ut * dy * sg1), (is_ - 1, js - 1), (n + 3, n + 2), k1=nk)
        put(vt, np.where(vt > 0, dt2 * vt * dx * sh(sg4, 0, -1), dt2 * vt * dx * sg2), (is_ - 1, js - 1), (n + 2, n + 3), k1=nk)
        fc = fill_corners_cells_mult
        for f in (delp, pt, w):
            fc(f, f, g, "x", 2, ks=ks)
        # compute_nonhydrostatic_fluxes_x :231-259 (halo 1)
        up = ut > 0.0
        fx1 = ut * np.where(up, sh(delp, -1, 0), delp)
        fx = fx1 * np.where(up, sh(pt, -1, 0), pt)
        fx2 = fx1 * np.where(up, sh(w, -1, 0), w)
        o, dm = (is_ - 1, js - 1), (n + 3, n + 2)
        put(st.fx1, fx1, o, dm, k1=nk)
        put(st.fx, fx, o, dm, k1=nk)
        put(st.fx2, fx2, o, dm, k1=nk)
        for f in (delp, pt, w):
            fc(f, f, g, "y", 2, ks=ks)
        # transportdelp_update_vorticity_and_kineticenergy :262-364 (halo 1)
        rarea = m2("rarea")
        upy = vt > 0.0
        fy1 = vt * np.where(upy, sh(delp, 0, -1), delp)
        fy = fy1 * np.where(upy, sh(pt, 0, -1), pt)
        fy2 = fy1 * np.where(upy, sh(w, 0, -1), w)
        delpc = delp + (st.fx1 - sh(st.fx1, 1, 0) + fy1 - sh(fy1, 0, 1)) * rarea
        ptc = (pt * delp + (st.fx - sh(st.fx, 1, 0) + fy - sh(fy, 0, 1)) * rarea) / delpc
        wc = (w * delp + (st.fx2 - sh(st.fx2, 1, 0) + fy2 - sh(fy2, 0, 1)) * rarea) / delpc
        o, dm = (is_ - 1, js - 1), (n + 2, n + 2)
        put(st.delpc, delpc, o, dm, k1=nk)
        put(st.ptc, ptc, o, dm, k1=nk)
        put(omga, wc, o, dm, k1=nk)
        cg1, cg2, cg3, cg4 = m2("cos_sg1"), m2("cos_sg2"), m2("cos_sg3"), m2("cos_sg4")
        ke = np.where(ua > 0.0, uc, sh(uc, 1, 0))
        vort = np.where(va > 0.0, vc, sh(vc, 0, 1))
        vort = np.where(((J == js - 1) | (J == je)) & (va <= 0.0), vort * sg4 + sh(u, 0, 1) * cg4, vort)
        vort = np.where(((J == js) | (J == je + 1)) & (va > 0.0), vort * sg2 + u * cg2, vort)
        ke = np.where(((I == ie) | (I == is_ - 1)) & (ua <= 0.0), ke * sg3 + sh(v, 1, 0) * cg3, ke)
        ke = np.where(((I == ie + 1) | (I == is_)) & (ua > 0.0), ke * sg1 + v * cg1, ke)
        ke = 0.5 * dt2 * (ua * ke + va * vort)
        put(st.ke, ke, o, dm, k1=nk)
        put(st.vort, vort, o, dm, k1=nk)
        # circulation_cgrid :367-397 + absolute_vorticity :400-408 (compute + 1)
        dxc, dyc = m2("dxc"), m2("dyc")
        fxc = dxc * uc
        fyc = dyc * vc
        fx1c = sh(dxc, 0, -1) * sh(uc, 0, -1)
        fy1c = sh(dyc, -1, 0) * sh(vc, -1, 0)
        vc_ = fx1c - fxc - fy1c + fyc
        vc_ = np.where((I == is_) & ((J == js) | (J == je + 1)), fx1c - fxc + fyc, vc_)
        vc_ = np.where((I == ie + 1) & ((J == js) | (J == je + 1)), fx1c - fxc - fy1c, vc_)
        put(st.vort, vc_, (is_, js), (n + 1, n + 1), k1=nk)
        put(st.vort, m2("fC") + m2("rarea_c") * st.vort, (is_, js), (n + 1, n + 1), k1=nk)
        # update_y_velocity :445-480 (X, Y_INTERFACE)
        vort, ke = st.vort, st.ke
        tmp = dt2 * (u - vc * m2("cosa_v")) / m2("sina_v")
        tmp = np.where((J == js) | (J == je + 1), dt2 * u, tmp)
        flux = np.where(tmp > 0.0, vort, sh(vort, 1, 0))
        put(vc, vc - tmp * flux + m2("rdyc") * (sh(ke, 0, -1) - ke), (is_, js), (n, n + 1), k1=nk)
        # update_x_velocity :411-442 (X_INTERFACE, Y)
        tmp = dt2 * (v - uc * m2("cosa_u")) / m2("sina_u")
        tmp = np.where((I == is_) | (I == ie + 1), dt2 * v, tmp)
        flux = np.where(tmp > 0.0, vort, sh(vort, 0, 1))
        put(uc, uc + tmp * flux + m2("rdxc") * (sh(ke, -1, 0) - ke), (is_, js), (n + 1, n), k1=nk)
