"""ORACLE (test infrastructure) -- piecewise-parabolic 1-D flux operators, the Putman-Lin 2-D
finite-volume transport operator and the del-n damping fluxes, restated in numpy.

Follows (all under fv3core/pace/fv3core/stencils/):
  ppm.py:6-35 (constants, standard constraint), xppm.py:19-181,290-355 (mord<8 path) and
  :76-102,185-266 (ord 8), yppm.py (mechanical transpose of xppm.py), fvtp2d.py:34-346,
  delnflux.py:21-328,945-1261, xtp_u.py:9-91, ytp_v.py:9-91.
Parity status: cross-checked against the reference's own stencil source executed by
tools/gtinterp (tools/crosscheck_oracle.py) and pinned by tests/golden fixtures generated the
same way; NOT pinned by Fortran savepoint data (unavailable offline) -- see DESIGN.md.
"""
import numpy as np

from . import corner_ops
from ._np import kcol, put, sh

# ppm.py:6-19
C1 = -2.0 / 14.0
C2 = 11.0 / 14.0
C3 = 5.0 / 14.0
P1 = 7.0 / 12.0
P2 = -1.0 / 12.0
S11 = 11.0 / 14.0
S14 = 4.0 / 7.0
S15 = 3.0 / 14.0


def _sd(a, d, axis):
    """value at offset d along the PPM axis (0 -> i, 1 -> j)."""
    return sh(a, d, 0) if axis == 0 else sh(a, 0, d)


def _edge_masks(g, axis):
    """masks for 'index == tile start + n' / 'tile end + n' along the PPM axis."""
    idx = g.I if axis == 0 else g.J
    s, e = (g.is_, g.ie) if axis == 0 else (g.js, g.je)
    return idx, s, e


def compute_al(q, dxa2, g, axis):
    """xppm.py:148-181 / yppm.py same lines: interface values for mord < 8."""
    idx, s, e = _edge_masks(g, axis)
    qm1, qm2, qp1 = _sd(q, -1, axis), _sd(q, -2, axis), _sd(q, 1, axis)
    al = P1 * (qm1 + q) + P2 * (qm2 + qp1)
    m = (idx == s - 1) | (idx == e)
    al = np.where(m, C1 * qm2 + C2 * qm1 + C3 * q, al)
    dm1, dm2, dp1 = _sd(dxa2, -1, axis), _sd(dxa2, -2, axis), _sd(dxa2, 1, axis)
    edge = 0.5 * (
        ((2.0 * dm1 + dm2) * qm1 - dm1 * qm2) / (dm2 + dm1)
        + ((2.0 * dxa2 + dp1) * q - dxa2 * qp1) / (dxa2 + dp1)
    )
    m = (idx == s) | (idx == e + 1)
    al = np.where(m, edge, al)
    m = (idx == s + 1) | (idx == e + 2)
    al = np.where(m, C3 * qm1 + C2 * q + C1 * qp1, al)
    return al


def _fx1(c, br, b0, bl, axis):
    """xppm.py:32-44."""
    return np.where(
        c > 0.0,
        (1.0 - c) * (_sd(br, -1, axis) - c * _sd(b0, -1, axis)),
        (1.0 + c) * (bl + c * b0),
    )


def _advection_mask(bl, b0, br, mord, axis):
    """xppm.py:48-61."""
    if mord == 5:
        smt5 = bl * br < 0
    else:
        smt5 = (3.0 * np.abs(b0)) < np.abs(bl - br)
    smt5 = np.where(np.isnan(b0), False, smt5)
    prev = _sd(smt5.astype(float), -1, axis)
    prev = np.where(np.isnan(prev), 0.0, prev) > 0.5
    return np.where(prev | smt5, 1.0, 0.0)


def _sign(a, b):
    """basic_operations.py:33-40."""
    return np.where(b > 0, np.abs(a), -np.abs(a))


def _pert_ppm_standard_constraint(a0, al, ar):
    """ppm.py:22-35."""
    da1 = al - ar
    da2 = da1 ** 2
    a6da = 3.0 * (al + ar) * da1
    neg = al * ar < 0.0
    ar_n = np.where(a6da < -da2, -2.0 * al, ar)
    al_n = np.where((~(a6da < -da2)) & (a6da > da2), -2.0 * ar, al)
    return np.where(neg, al_n, 0.0), np.where(neg, ar_n, 0.0)


def _blbr_ord8(q, dxa2, g, axis, minmax):
    """xppm.py:76-102,105-145,185-287: monotone (ord 8) edge perturbations bl, br."""
    idx, s, e = _edge_masks(g, axis)
    qm1, qm2, qp1, qp2 = _sd(q, -1, axis), _sd(q, -2, axis), _sd(q, 1, axis), _sd(q, 2, axis)
    # dm_iord8plus :76-81
    xt = 0.25 * (qp1 - qm1)
    dqr = np.maximum(np.maximum(q, qm1), qp1) - q
    dql = q - np.minimum(np.minimum(q, qm1), qp1)
    dm = _sign(np.minimum(np.minimum(np.abs(xt), dqr), dql), xt)
    # al_iord8plus :84-86
    al = 0.5 * (qm1 + q) + 1.0 / 3.0 * (_sd(dm, -1, axis) - dm)
    # blbr_iord8 :89-94
    xt2 = 2.0 * dm
    bl = -1.0 * _sign(np.minimum(np.abs(xt2), np.abs(al - q)), xt2)
    br = _sign(np.minimum(np.abs(xt2), np.abs(_sd(al, 1, axis) - q)), xt2)
    # bl_br_edges :185-266
    dm1, dm2, dp1, dp2 = _sd(dxa2, -1, axis), _sd(dxa2, -2, axis), _sd(dxa2, 1, axis), _sd(dxa2, 2, axis)
    e0 = 0.5 * (
        ((2.0 * dxa2 + dm1) * q - dxa2 * qm1) / (dm1 + dxa2)
        + ((2.0 * dp1 + dp2) * qp1 - dp1 * qp2) / (dp1 + dp2)
    )
    e1 = 0.5 * (
        ((2.0 * dm1 + dm2) * qm1 - dm1 * qm2) / (dm2 + dm1)
        + ((2.0 * dxa2 + dp1) * q - dxa2 * qp1) / (dxa2 + dp1)
    )
    if minmax:
        e0 = np.minimum(np.maximum(e0, np.minimum(np.minimum(np.minimum(qm1, q), qp1), qp2)),
                        np.maximum(np.maximum(np.maximum(qm1, q), qp1), qp2))
        e1 = np.minimum(np.maximum(e1, np.minimum(np.minimum(np.minimum(qm2, qm1), q), qp1)),
                        np.maximum(np.maximum(np.maximum(qm2, qm1), q), qp1))
    al_ip1 = _sd(al, 1, axis)
    # dm of the left / right neighbour, written out as in the reference
    xl = 0.25 * (q - qm2)
    dqr_l = np.maximum(np.maximum(qm1, qm2), q) - qm1
    dql_l = qm1 - np.minimum(np.minimum(qm1, qm2), q)
    dm_left = _sign(np.minimum(np.minimum(np.abs(xl), dqr_l), dql_l), xl)
    xr = 0.25 * (qp2 - q)
    dqr_r = np.maximum(np.maximum(qp1, q), qp2) - qp1
    dql_r = qp1 - np.minimum(np.minimum(qp1, q), qp2)
    dm_right = _sign(np.minimum(np.minimum(np.abs(xr), dqr_r), dql_r), xr)
    xt_bl = np.full(q.shape, np.nan)
    xt_br = np.full(q.shape, np.nan)
    m = idx == s - 1
    xt_bl = np.where(m, S14 * dm_left + S11 * (qm1 - q) + q, xt_bl)
    xt_br = np.where(m, e0, xt_br)
    m = idx == s
    xt_bl = np.where(m, e1, xt_bl)
    xt_br = np.where(m, S15 * q + S11 * qp1 - S14 * dm_right, xt_br)
    m = idx == s + 1
    xt_bl = np.where(m, S15 * qm1 + S11 * q - S14 * dm, xt_bl)
    xt_br = np.where(m, al_ip1, xt_br)
    m = idx == e - 1
    xt_bl = np.where(m, al, xt_bl)
    xt_br = np.where(m, S15 * qp1 + S11 * q + S14 * dm, xt_br)
    m = idx == e
    xt_bl = np.where(m, S15 * q + S11 * qm1 + S14 * dm_left, xt_bl)
    xt_br = np.where(m, e0, xt_br)
    m = idx == e + 1
    xt_bl = np.where(m, e1, xt_bl)
    xt_br = np.where(m, S11 * (qp1 - q) - S14 * dm_right + q, xt_br)
    medge = ((idx >= s - 1) & (idx <= s + 1)) | ((idx >= e - 1) & (idx <= e + 1))
    bl = np.where(medge, xt_bl - q, bl)
    br = np.where(medge, xt_br - q, br)
    return bl, br, medge


def ppm_flux(q, c, dxa, g, axis, ord_, out, origin, domain):
    """XPiecewiseParabolic / YPiecewiseParabolic.__call__ (xppm.py:290-355, yppm.py:290-355).

    q, c, out: (ni, nj, nk) arrays; dxa: 2-D metric (dxa for axis 0, dya for axis 1).
    Writes the interface-mean advected value into out on [origin, origin+domain).
    """
    mord = abs(ord_)
    dxa2 = dxa[:, :, None]
    with np.errstate(all="ignore"):
        if mord < 8:
            al = compute_al(q, dxa2, g, axis)
            bl = al - q
            br = _sd(al, 1, axis) - q
            b0 = bl + br
            mask = _advection_mask(bl, b0, br, mord, axis)
            fx1 = _fx1(c, br, b0, bl, axis)
        else:
            assert ord_ == 8
            bl, br, medge = _blbr_ord8(q, dxa2, g, axis, minmax=True)
            cbl, cbr = _pert_ppm_standard_constraint(q, bl, br)
            bl = np.where(medge, cbl, bl)
            br = np.where(medge, cbr, br)
            b0 = bl + br
            fx1 = _fx1(c, br, b0, bl, axis)
            mask = 1.0
        flux = np.where(c > 0.0, _sd(q, -1, axis) + fx1 * mask, q + fx1 * mask)
    put(out, flux, origin, domain)


def advect_wind_1d(u, ub_contra, rdx, dx, dxa, dt, g, axis, ord_):
    """xtp_u.advect_u_along_x / ytp_v.advect_v_along_y (xtp_u.py:9-91, ytp_v.py:9-91).

    Returns the full-array value (caller commits).  Note the reference passes the D-grid spacing
    ``dx`` (not dxa) to compute_al for ord < 8 (xtp_u.py:22).
    """
    idx, s, e = _edge_masks(g, axis)
    oidx = g.J if axis == 0 else g.I
    os_, oe = (g.js, g.je) if axis == 0 else (g.is_, g.ie)
    dx2, rdx2 = dx[:, :, None], rdx[:, :, None]
    with np.errstate(all="ignore"):
        if abs(ord_) < 8:
            al = compute_al(u, dx2, g, axis)
            bl = al - u
            br = _sd(al, 1, axis) - u
        else:
            bl, br, _ = _blbr_ord8(u, dxa[:, :, None], g, axis, minmax=False)
            m = (idx == s + 1) | (idx == e - 1)
            cbl, cbr = _pert_ppm_standard_constraint(u, bl, br)
            bl = np.where(m, cbl, bl)
            br = np.where(m, cbr, br)
        # zero corners :41-49
        zc = (((idx >= s - 1) & (idx <= s)) | ((idx >= e) & (idx <= e + 1))) & ((oidx == os_) | (oidx == oe + 1))
        bl = np.where(zc, 0.0, bl)
        br = np.where(zc, 0.0, br)
        b0 = bl + br
        cfl = np.where(ub_contra > 0, ub_contra * dt * _sd(rdx2, -1, axis), ub_contra * dt * rdx2)
        fx0 = _fx1(cfl, br, b0, bl, axis)
        if abs(ord_) < 8:
            mask = _advection_mask(bl, b0, br, abs(ord_), axis)
        else:
            mask = 1.0
        return np.where(ub_contra > 0.0, _sd(u, -1, axis) + fx0 * mask, u + fx0 * mask)


# --------------------------------------------------------------------------- del-n damping
def calc_damp(damp_c, da_min, nord):
    """delnflux.py:21-38."""
    return (np.asarray(damp_c, dtype=float) * da_min) ** (np.asarray(nord, dtype=float) + 1)


def delnflux_nosg(g, q, fx2, fy2, damp_k, d2, nord_k, mass_given=False, nk=None):
    """DelnFluxNoSG.__call__ (delnflux.py:1050-1261).

    nord_k, damp_k: per-level arrays (len >= nk).  The reference selects the order through the
    externals nord0..nord3 for levels 0,1,2,>=3 (delnflux.py:41-82); callers expand those to a
    per-level array.  q is read only; fx2, fy2, d2 are written (d2 holds the last iterate).
    """
    if nk is None:
        nk = g.nk
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    nord_k = np.asarray(nord_k, dtype=float)[:nk]
    nmax = int(nord_k.max())
    n = g.n
    del6_v, del6_u, rarea = g.m2("del6_v"), g.m2("del6_u"), g.m2("rarea")
    nkt = q.shape[2]
    hi = np.zeros((1, 1, nkt), dtype=bool)
    hi[0, 0, :nk] = nord_k > 0
    lo = np.zeros((1, 1, nkt), dtype=bool)
    lo[0, 0, :nk] = nord_k == 0
    damp3 = kcol(damp_k, nkt)
    # d2_damp_interval :208-256 / copy_stencil_interval :259-307
    o = (is_ - 1 - nmax, js - 1 - nmax)
    d = (n + 2 + 2 * nmax, n + 2 + 2 * nmax)
    src = q if mass_given else damp3 * q
    inner = g.reg(is_ - 1, ie + 1, js - 1, je + 1)
    put(d2, src, o, d, mask=hi | (lo & inner), k1=nk)

    def corners(direction):
        # copy_corners_{x,y}_nord :331-942: only levels with nord > 0
        ks = np.nonzero(nord_k > 0)[0]
        if ks.size:
            corner_ops.copy_corners(d2, g, direction, ks=ks)

    corners("x")
    fo = (is_ - nmax, js - nmax)
    with np.errstate(all="ignore"):
        # fx_calc_stencil_nord :41-82
        val = del6_v * (sh(d2, -1, 0) - d2)
        put(fx2, val, fo, (n + 1 + 2 * nmax, n + 2 * nmax), mask=hi | (lo & g.reg(is_, ie + 1, js, je)), k1=nk)
        corners("y")
        val = del6_u * (sh(d2, 0, -1) - d2)
        put(fy2, val, fo, (n + 2 * nmax, n + 1 + 2 * nmax), mask=hi | (lo & g.reg(is_, ie, js, je + 1)), k1=nk)
        for it in range(nmax):
            nt = nmax - 1 - it
            # d2_highorder_stencil :183-205
            val = (fx2 - sh(fx2, 1, 0) + fy2 - sh(fy2, 0, 1)) * rarea
            put(d2, val, (is_ - nt - 1, js - nt - 1), (n + 2 + 2 * nt, n + 2 + 2 * nt), mask=hi, k1=nk)
            corners("x")
            val = -del6_v * (sh(d2, -1, 0) - d2)
            put(fx2, val, (is_ - nt, js - nt), (n + 1 + 2 * nt, n + 2 * nt), mask=hi, k1=nk)
            corners("y")
            val = -del6_u * (sh(d2, 0, -1) - d2)
            put(fy2, val, (is_ - nt, js - nt), (n + 2 * nt, n + 1 + 2 * nt), mask=hi, k1=nk)


def delnflux(g, q, fx, fy, nord_k, damp_c_k, da_min, mass=None, d2=None, nk=None):
    """DelnFlux.__call__ (delnflux.py:945-1047): compute and add the damping fluxes to fx, fy."""
    if nk is None:
        nk = g.nk
    damp_c_k = np.asarray(damp_c_k, dtype=float)[:nk]
    if (damp_c_k <= 1e-4).all():
        return
    nord_k = np.asarray(nord_k, dtype=float)[:nk]
    damp = calc_damp(damp_c_k, da_min, nord_k)
    fx2 = np.zeros_like(q)
    fy2 = np.zeros_like(q)
    if d2 is None:
        d2 = np.zeros_like(q)
    delnflux_nosg(g, q, fx2, fy2, damp, d2, nord_k, mass_given=mass is not None, nk=nk)
    o, d = (g.is_, g.js), (g.n + 1, g.n + 1)
    if mass is None:
        put(fx, fx + fx2, o, d, k1=nk)
        put(fy, fy + fy2, o, d, k1=nk)
    else:
        damp3 = kcol(damp, q.shape[2])
        with np.errstate(all="ignore"):
            put(fx, fx + 0.5 * damp3 * (sh(mass, -1, 0) + mass) * fx2, o, d, k1=nk)
            put(fy, fy + 0.5 * damp3 * (sh(mass, 0, -1) + mass) * fy2, o, d, k1=nk)


# --------------------------------------------------------------------------- fvtp2d
def fvtp2d(g, q, crx, cry, xfx, yfx, q_x_flux, q_y_flux, hord, x_mass_flux=None, y_mass_flux=None,
           mass=None, nord_k=None, damp_c_k=None, nk=None):
    """FiniteVolumeTransport.__call__ (fvtp2d.py:262-346).  q's corner halos are overwritten in
    place exactly as in the reference (copy_corners y then x)."""
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    area = g.m2("area")
    ord_outer = hord
    ord_inner = 8 if hord == 10 else hord
    nkt = q.shape[2]
    x_unit = xfx if x_mass_flux is None else x_mass_flux
    y_unit = yfx if y_mass_flux is None else y_mass_flux
    q_y_adv_mean = np.zeros_like(q)
    q_adv_y = np.zeros_like(q)
    q_adv_y_x_mean = np.zeros_like(q)
    q_x_adv_mean = np.zeros_like(q)
    q_adv_x = np.zeros_like(q)
    q_adv_x_y_mean = np.zeros_like(q)
    with np.errstate(all="ignore"):
        corner_ops.copy_corners(q, g, "y")
        ppm_flux(q, cry, g.dya, g, 1, ord_inner, q_y_adv_mean, (is_ - 3, js), (n + 7, n + 1))
        # q_i_stencil :34-56, origin_full(add=(0,3,0)), domain_full(add=(0,-3,1))
        fyy = yfx * q_y_adv_mean
        q_i = (q * area + fyy - sh(fyy, 0, 1)) / (area + yfx - sh(yfx, 0, 1))
        put(q_adv_y, q_i, (0, 3), (n + 6, n + 3))
        ppm_flux(q_adv_y, crx, g.dxa, g, 0, ord_outer, q_adv_y_x_mean, (is_, js), (n + 1, n + 1))
        corner_ops.copy_corners(q, g, "x")
        ppm_flux(q, crx, g.dxa, g, 0, ord_inner, q_x_adv_mean, (is_, js - 3), (n + 1, n + 7))
        # q_j_stencil :59-77
        fx1 = xfx * q_x_adv_mean
        q_j = (q * area + fx1 - sh(fx1, 1, 0)) / (area + xfx - sh(xfx, 1, 0))
        put(q_adv_x, q_j, (3, 0), (n + 3, n + 6))
        ppm_flux(q_adv_x, cry, g.dya, g, 1, ord_outer, q_adv_x_y_mean, (is_, js), (n + 1, n + 1))
        # final_fluxes :80-119 (regions [:, :-1] and [:-1, :])
        put(q_x_flux, 0.5 * (q_adv_y_x_mean + q_x_adv_mean) * x_unit, (is_, js), (n + 1, n))
        put(q_y_flux, 0.5 * (q_adv_x_y_mean + q_y_adv_mean) * y_unit, (is_, js), (n, n + 1))
    if nord_k is not None and damp_c_k is not None:
        delnflux(g, q, q_x_flux, q_y_flux, nord_k, damp_c_k, g.da_min, mass=mass, nk=nk)
