"""ORACLE (test infrastructure) -- A-grid -> B-grid 4th-order interpolation and the D-grid
divergence damping, restated in numpy.

Follows fv3core/pace/fv3core/stencils/a2b_ord4.py:22-761 and
fv3core/pace/fv3core/stencils/divergence_damping.py:23-632 (+ corner fills from
stencils/pace/stencils/corners.py via oracle/corner_ops.py).
Parity status: see oracle/ppm_transport.py header.
"""
import numpy as np

from . import corner_ops
from ._np import kcol, put, sh

# a2b_ord4.py:22-33
C1 = 2.0 / 3.0
C2 = -1.0 / 6.0
B1 = 7.0 / 12.0
B2 = -1.0 / 12.0
A1 = 9.0 / 16.0
A2 = -1.0 / 16.0


def _gcd(p1a, p1b, p2a, p2b):
    """a2b_ord4.py:36-40 great_circle_dist."""
    tb = np.sin((p1b - p2b) / 2.0) ** 2.0
    ta = np.sin((p1a - p2a) / 2.0) ** 2.0
    return np.arcsin(np.sqrt(tb + np.cos(p1b) * np.cos(p2b) * ta)) * 2.0


def _extrap(g, i, j, o1, o2, qin):
    """a2b_ord4.py:43-56 extrap_corner at corner point (i, j); o1/o2 = (di, dj) of the two A-grid cells."""
    p0a, p0b = g.lon[i, j], g.lat[i, j]
    p1a, p1b = g.lon_agrid[i + o1[0], j + o1[1]], g.lat_agrid[i + o1[0], j + o1[1]]
    p2a, p2b = g.lon_agrid[i + o2[0], j + o2[1]], g.lat_agrid[i + o2[0], j + o2[1]]
    qa = qin[i + o1[0], j + o1[1], :]
    qb = qin[i + o2[0], j + o2[1], :]
    x1 = _gcd(p1a, p1b, p0a, p0b)
    x2 = _gcd(p2a, p2b, p0a, p0b)
    return qa + x1 / (x2 - x1) * (qa - qb)


_DIAG = {
    "ur": ((0, 0), (1, 1)),  # cell to the upper-right of the corner point, and the next one out
    "ul": ((-1, 0), (-2, 1)),
    "lr": ((0, -1), (1, -2)),
    "ll": ((-1, -1), (-2, -2)),
}
# which three diagonals each corner stencil uses (a2b_ord4.py:59-273); note the reference places
# "_nw_corner" at (iec+1, jsc) and "_se_corner" at (isc, jec+1) (a2b_ord4.py:570-583) -- kept.
_CORNER_SETS = {"sw": ("ur", "ul", "lr"), "nw": ("ul", "ll", "ur"), "ne": ("ll", "lr", "ul"), "se": ("lr", "ll", "ur")}


def _jvec(a, g):
    a = np.asarray(a)
    return a[0, :] if a.ndim == 2 else a


def a2b_ord4(g, qin, qout, k0=0, k1=None, replace=False):
    """AGrid2BGridFourthOrder.__call__ (a2b_ord4.py:668-761) on levels k0:k1."""
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    if k1 is None:
        k1 = g.nk
    ks = slice(k0, k1)
    edges = np.zeros_like(qin)
    dxa, dya = g.m2("dxa"), g.m2("dya")
    with np.errstate(all="ignore"):
        for name, (i, j) in (("sw", (is_, js)), ("nw", (ie + 1, js)), ("ne", (ie + 1, je + 1)), ("se", (is_, je + 1))):
            tot = 0.0
            for dname in _CORNER_SETS[name]:
                o1, o2 = _DIAG[dname]
                tot = tot + _extrap(g, i, j, o1, o2, qin)
            val = tot * (1.0 / 3.0)
            qout[i, j, ks] = val[ks]
            edges[i, j, ks] = val[ks]
        # qout_x_edge :286-304 on i = is_ and ie+1, j in [js+1, je]
        q2 = (sh(qin, -1, 0) * dxa + qin * sh(dxa, -1, 0)) / (sh(dxa, -1, 0) + dxa)
        for i, ew in ((is_, _jvec(g.edge_w, g)), (ie + 1, _jvec(g.edge_e, g))):
            e3 = ew.reshape(1, -1, 1)
            val = e3 * sh(q2, 0, -1) + (1.0 - e3) * q2
            put(qout, val, (i, js + 1), (1, n - 1), k0=k0, k1=k1)
            put(edges, val, (i, js + 1), (1, n - 1), k0=k0, k1=k1)
        # qout_y_edge :307-325 on j = js and je+1, i in [is+1, ie]
        q1 = (sh(qin, 0, -1) * dya + qin * sh(dya, 0, -1)) / (sh(dya, 0, -1) + dya)
        for j, es in ((js, np.asarray(g.edge_s)), (je + 1, np.asarray(g.edge_n))):
            e3 = es.reshape(-1, 1, 1)
            val = e3 * sh(q1, -1, 0) + (1.0 - e3) * q1
            put(qout, val, (is_ + 1, j), (n - 1, 1), k0=k0, k1=k1)
            put(edges, val, (is_ + 1, j), (n - 1, 1), k0=k0, k1=k1)
        # ppm_volume_mean_x :429-450
        def s(a, d):
            return sh(a, d, 0)

        qx = B2 * (s(qin, -2) + s(qin, 1)) + B1 * (s(qin, -1) + qin)
        g_in = s(dxa, 1) / dxa
        g_ou = s(dxa, -2) / s(dxa, -1)
        west = 0.5 * (((2.0 + g_in) * qin - s(qin, 1)) / (1.0 + g_in) + ((2.0 + g_ou) * s(qin, -1) - s(qin, -2)) / (1.0 + g_ou))
        g_in = dxa / s(dxa, -1)
        g_ou = s(dxa, -3) / s(dxa, -2)
        left = 0.5 * (((2.0 + g_in) * s(qin, -1) - qin) / (1.0 + g_in) + ((2.0 + g_ou) * s(qin, -2) - s(qin, -3)) / (1.0 + g_ou))
        right = B2 * (s(qin, -1) + s(qin, 2)) + B1 * (qin + s(qin, 1))
        west2 = (3.0 * (g_in * s(qin, -1) + qin) - (g_in * left + right)) / (2.0 + 2.0 * g_in)
        g_in = s(dxa, -2) / s(dxa, -1)
        g_ou = s(dxa, 1) / dxa
        east = 0.5 * (((2.0 + g_in) * s(qin, -1) - s(qin, -2)) / (1.0 + g_in) + ((2.0 + g_ou) * qin - s(qin, 1)) / (1.0 + g_ou))
        g_in = s(dxa, -1) / dxa
        g_ou = s(dxa, 2) / s(dxa, 1)
        right = 0.5 * (((2.0 + g_in) * qin - s(qin, -1)) / (1.0 + g_in) + ((2.0 + g_ou) * s(qin, 1) - s(qin, 2)) / (1.0 + g_ou))
        left = B2 * (s(qin, -3) + qin) + B1 * (s(qin, -2) + s(qin, -1))
        east2 = (3.0 * (s(qin, -1) + g_in * qin) - (g_in * right + left)) / (2.0 + 2.0 * g_in)
        qx = np.where(g.I == is_, west, qx)
        qx = np.where(g.I == is_ + 1, west2, qx)
        qx = np.where(g.I == ie + 1, east, qx)
        qx = np.where(g.I == ie, east2, qx)

        # ppm_volume_mean_y :453-473
        def t(a, d):
            return sh(a, 0, d)

        qy = B2 * (t(qin, -2) + t(qin, 1)) + B1 * (t(qin, -1) + qin)
        g_in = t(dya, 1) / dya
        g_ou = t(dya, -2) / t(dya, -1)
        south = 0.5 * (((2.0 + g_in) * qin - t(qin, 1)) / (1.0 + g_in) + ((2.0 + g_ou) * t(qin, -1) - t(qin, -2)) / (1.0 + g_ou))
        g_in = dya / t(dya, -1)
        g_ou = t(dya, -3) / t(dya, -2)
        lower = 0.5 * (((2.0 + g_in) * t(qin, -1) - qin) / (1.0 + g_in) + ((2.0 + g_ou) * t(qin, -2) - t(qin, -3)) / (1.0 + g_ou))
        upper = B2 * (t(qin, -1) + t(qin, 2)) + B1 * (qin + t(qin, 1))
        south2 = (3.0 * (g_in * t(qin, -1) + qin) - (g_in * lower + upper)) / (2.0 + 2.0 * g_in)
        g_in = t(dya, -2) / t(dya, -1)
        g_ou = t(dya, 1) / dya
        north = 0.5 * (((2.0 + g_in) * t(qin, -1) - t(qin, -2)) / (1.0 + g_in) + ((2.0 + g_ou) * qin - t(qin, 1)) / (1.0 + g_ou))
        g_in = t(dya, -1) / dya
        g_ou = t(dya, 2) / t(dya, 1)
        lower = B2 * (t(qin, -3) + qin) + B1 * (t(qin, -2) + t(qin, -1))
        upper = 0.5 * (((2.0 + g_in) * qin - t(qin, -1)) / (1.0 + g_in) + ((2.0 + g_ou) * t(qin, 1) - t(qin, 2)) / (1.0 + g_ou))
        north2 = (3.0 * (t(qin, -1) + g_in * qin) - (g_in * upper + lower)) / (2.0 + 2.0 * g_in)
        qy = np.where(g.J == js, south, qy)
        qy = np.where(g.J == js + 1, south2, qy)
        qy = np.where(g.J == je + 1, north, qy)
        qy = np.where(g.J == je, north2, qy)
        # the reference commits qx on i in [is, ie+1], j in [js-2, je+2] and qy transposed; values
        # outside are never read by a2b_interpolation's window, so the full-array values serve.
        # a2b_interpolation :476-506 on i in [is+1, ie], j in [js+1, je]
        qxx = A2 * (t(qx, -2) + t(qx, 1)) + A1 * (t(qx, -1) + qx)
        qyy = A2 * (s(qy, -2) + s(qy, 1)) + A1 * (s(qy, -1) + qy)
        up = A2 * (t(qx, -1) + t(qx, 2)) + A1 * (qx + t(qx, 1))
        qxx = np.where(g.J == js + 1, C1 * (t(qx, -1) + qx) + C2 * (t(edges, -1) + up), qxx)
        lo = A2 * (t(qx, -3) + qx) + A1 * (t(qx, -2) + t(qx, -1))
        qxx = np.where(g.J == je, C1 * (t(qx, -1) + qx) + C2 * (t(edges, 1) + lo), qxx)
        rt = A2 * (s(qy, -1) + s(qy, 2)) + A1 * (qy + s(qy, 1))
        qyy = np.where(g.I == is_ + 1, C1 * (s(qy, -1) + qy) + C2 * (s(edges, -1) + rt), qyy)
        lf = A2 * (s(qy, -3) + qy) + A1 * (s(qy, -2) + s(qy, -1))
        qyy = np.where(g.I == ie, C1 * (s(qy, -1) + qy) + C2 * (s(edges, 1) + lf), qyy)
        put(qout, 0.5 * (qxx + qyy), (is_ + 1, js + 1), (n - 1, n - 1), k0=k0, k1=k1)
    if replace:
        put(qin, qout, (is_, js), (n + 1, n + 1), k0=k0, k1=k1)


def divergence_damping(g, u, v, va, vort_b, ua, divg_d, vc, uc, delpc, ke, wk, dt, *, nord_k, d2_bg_k,
                       dddmp, d4_bg, nord, stretched_grid=False, grid_type=0):
    """DivergenceDamping.__call__ (divergence_damping.py:482-632).

    vort_b = damped_rel_vort_bgrid (out), wk = rel_vort_agrid (in).  nord_k / d2_bg_k are the
    column-namelist K arrays; the reference splits the column at the first level with nord > 0
    (divergence_damping.py:307-331).
    """
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    nk = g.nk
    nord_k = np.asarray(nord_k, dtype=float)[:nk]
    kstart, nonzero_nord = 0, int(nord)
    for k in range(nk):
        if nord_k[k] > 0:
            kstart, nonzero_nord = k, int(nord_k[k])
            break
    do_zero_order = kstart > 0
    nkt = u.shape[2]
    d2bg3 = kcol(d2_bg_k, nkt)
    cp = (is_, js)  # compute origin
    cd = (n + 1, n + 1)
    da_min_c = g.da_min_c
    with np.errstate(all="ignore"):
        if do_zero_order:
            # compute_u_contra_dyc :30-63, dims [X, Y_INTERFACE] halos (1, 0)
            vc_from_va = 0.5 * (sh(va, 0, -1) + va)
            u_contra = (u - vc_from_va * g.m2("cosa_v")) * g.m2("sina_v")
            edge = np.where(vc > 0, u * sh(g.m2("sin_sg4"), 0, -1), u * g.m2("sin_sg2"))
            u_contra = np.where((g.J == js) | (g.J == je + 1), edge, u_contra)
            u_contra_dyc = np.zeros_like(u)
            put(u_contra_dyc, u_contra * g.m2("dyc"), (is_ - 1, js), (n + 2, n + 1), k1=kstart)
            # compute_v_contra_dxc :66-98, dims [X_INTERFACE, Y] halos (0, 1)
            uc_from_ua = 0.5 * (sh(ua, -1, 0) + ua)
            v_contra = (v - uc_from_ua * g.m2("cosa_u")) * g.m2("sina_u")
            edge = np.where(uc > 0, v * sh(g.m2("sin_sg3"), -1, 0), v * g.m2("sin_sg1"))
            v_contra = np.where((g.I == is_) | (g.I == ie + 1), edge, v_contra)
            v_contra_dxc = np.zeros_like(u)
            put(v_contra_dxc, v_contra * g.m2("dxc"), (is_, js - 1), (n + 1, n + 2), k1=kstart)
            # delpc_computation :101-135  (note the argument order at the call site :561-566:
            # u_contra_dxc <- self.u_contra_dyc, v_contra_dyc <- self.v_contra_dxc)
            a, b = u_contra_dyc, v_contra_dxc
            d = sh(b, 0, -1) - b + sh(a, -1, 0) - a
            d = np.where(((g.I == is_) | (g.I == ie + 1)) & (g.J == js), d - sh(b, 0, -1), d)
            d = np.where(((g.I == is_) | (g.I == ie + 1)) & (g.J == je + 1), d + b, d)
            put(delpc, g.m2("rarea_c") * d, cp, cd, k1=kstart)
            # damping :138-158
            delpcdt = delpc * dt
            damp = da_min_c * np.maximum(d2bg3, np.minimum(0.2, dddmp * np.abs(delpcdt)))
            vort = damp * delpc
            put(vort_b, vort, cp, cd, k1=kstart)
            put(ke, ke + vort, cp, cd, k1=kstart)
        hk = np.arange(kstart, nk)
        put(delpc, divg_d, cp, cd, k0=kstart, k1=nk)
        for it in range(nonzero_nord):
            nt = nonzero_nord - (it + 1)
            fillc = (it + 1 != nonzero_nord) and grid_type < 3
            if fillc:
                corner_ops.fill_corners_bgrid(divg_d, g, "x", ks=hk)
            # vc_from_divg :188-197
            put(vc, (sh(divg_d, 1, 0) - divg_d) * g.m2("divg_u"), (is_ - nt - 1, js - nt), (n + 2 * nt + 2, n + 2 * nt + 1),
                k0=kstart, k1=nk)
            if fillc:
                corner_ops.fill_corners_bgrid(divg_d, g, "y", ks=hk)
            put(uc, (sh(divg_d, 0, 1) - divg_d) * g.m2("divg_v"), (is_ - nt, js - nt - 1), (n + 2 * nt + 1, n + 2 * nt + 2),
                k0=kstart, k1=nk)
            if fillc:
                corner_ops.fill_corners_dgrid(vc, uc, g, -1.0, ks=hk)
            # redo_divg_d :212-240
            d = sh(uc, 0, -1) - uc + sh(vc, -1, 0) - vc
            d = np.where(((g.I == is_) | (g.I == ie + 1)) & (g.J == js), d - sh(uc, 0, -1), d)
            d = np.where(((g.I == is_) | (g.I == ie + 1)) & (g.J == je + 1), d + uc, d)
            if not stretched_grid:
                d = d * g.m2("rarea_c")
            put(divg_d, d, (is_ - nt, js - nt), (n + 2 * nt + 1, n + 2 * nt + 1), k0=kstart, k1=nk)
        if dddmp < 1e-5:
            put(vort_b, 0.0, (0, 0), (n + 7, n + 7), k0=kstart, k1=nk)
        else:
            a2b_ord4(g, wk, vort_b, k0=kstart, k1=nk)
            # smagorinsky_diffusion_approx :243-251
            val = abs(dt) * (delpc ** 2.0 + vort_b ** 2.0) ** 0.5
            put(vort_b, val, cp, cd, k0=kstart, k1=nk)
        if stretched_grid:
            dd8 = g.da_min * d4_bg ** (nonzero_nord + 1)
        else:
            dd8 = (da_min_c * d4_bg) ** (nonzero_nord + 1)
        # damping_nord_highorder_stencil :161-185
        damp = da_min_c * np.maximum(d2bg3, np.minimum(0.2, dddmp * np.abs(vort_b)))
        vort = damp * delpc + dd8 * divg_d
        put(vort_b, vort, cp, cd, k0=kstart, k1=nk)
        put(ke, ke + vort, cp, cd, k0=kstart, k1=nk)
