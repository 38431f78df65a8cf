"""ORACLE (test infrastructure) -- the remaining pieces of the acoustic loop in numpy.

Follows (fv3core/pace/fv3core/stencils/): dyn_core.py:51-171 (zero_data, gz_from_surface..., interface pressure,
compute_geopotential, p_grad_c_stencil), nh_p_grad.py:11-255, pe_halo.py:6-34, pk3_halo.py:11-69,
ray_fast.py:24-206, del2cubed.py:16-194, temperature_adjust.py:8-43.
Parity status: see oracle/ppm_transport.py header.
"""
import numpy as np

from . import constants as c
from . import corner_ops
from . import damping
from ._np import kcol, put, sh


def p_grad_c(g, uc, vc, delpc, pkc, gz, dt2):
    """dyn_core.p_grad_c_stencil (dyn_core.py:120-171), hydrostatic=False; domain compute + 1."""
    n, nk = g.n, g.nk
    rdxc, rdyc = g.m2("rdxc"), g.m2("rdyc")
    with np.errstate(all="ignore"):
        wk = delpc
        un = uc + dt2 * rdxc / (sh(wk, -1, 0) + wk) * (
            (sh(gz, -1, 0, 1) - gz) * (sh(pkc, 0, 0, 1) - sh(pkc, -1, 0))
            + (sh(gz, -1, 0) - sh(gz, 0, 0, 1)) * (sh(pkc, -1, 0, 1) - pkc)
        )
        vn = vc + dt2 * rdyc / (sh(wk, 0, -1) + wk) * (
            (sh(gz, 0, -1, 1) - gz) * (sh(pkc, 0, 0, 1) - sh(pkc, 0, -1))
            + (sh(gz, 0, -1) - sh(gz, 0, 0, 1)) * (sh(pkc, 0, -1, 1) - pkc)
        )
        put(uc, un, (g.is_, g.js), (n + 1, n + 1), k1=nk)
        put(vc, vn, (g.is_, g.js), (n + 1, n + 1), k1=nk)


def nh_p_grad(g, u, v, pp, gz, pk3, delp, dt, ptop, akap):
    """NonHydrostaticPressureGradient.__call__ (nh_p_grad.py:187-255)."""
    n, nk = g.n, g.nk
    K = nk + 1
    is_, js = g.is_, g.js
    top_value = ptop ** akap
    wk1 = np.zeros(u.shape)
    wk = np.zeros(u.shape)
    damping.a2b_ord4(g, pp, wk1, k0=1, k1=K, replace=True)
    damping.a2b_ord4(g, pk3, wk1, k0=1, k1=K, replace=True)
    damping.a2b_ord4(g, gz, wk1, k0=0, k1=K, replace=True)
    damping.a2b_ord4(g, delp, wk1, k0=0, k1=nk, replace=False)
    W = (slice(is_, is_ + n + 1), slice(js, js + n + 1))
    with np.errstate(all="ignore"):
        pp[W + (0,)] = 0.0
        pk3[W + (0,)] = top_value
        put(wk, sh(pk3, 0, 0, 1) - pk3, (is_, js), (n + 1, n + 1), k1=nk)
        rdx, rdy = g.m2("rdx"), g.m2("rdy")
        du = dt / (wk + sh(wk, 1, 0)) * (
            (sh(gz, 0, 0, 1) - sh(gz, 1, 0)) * (sh(pk3, 1, 0, 1) - pk3) + (gz - sh(gz, 1, 0, 1)) * (sh(pk3, 0, 0, 1) - sh(pk3, 1, 0))
        )
        un = (u + du + dt / (wk1 + sh(wk1, 1, 0)) * (
            (sh(gz, 0, 0, 1) - sh(gz, 1, 0)) * (sh(pp, 1, 0, 1) - pp) + (gz - sh(gz, 1, 0, 1)) * (sh(pp, 0, 0, 1) - sh(pp, 1, 0))
        )) * rdx
        put(u, un, (is_, js), (n, n + 1), k1=nk)
        dv = dt / (wk + sh(wk, 0, 1)) * (
            (sh(gz, 0, 0, 1) - sh(gz, 0, 1)) * (sh(pk3, 0, 1, 1) - pk3) + (gz - sh(gz, 0, 1, 1)) * (sh(pk3, 0, 0, 1) - sh(pk3, 0, 1))
        )
        vn = (v + dv + dt / (wk1 + sh(wk1, 0, 1)) * (
            (sh(gz, 0, 0, 1) - sh(gz, 0, 1)) * (sh(pp, 0, 1, 1) - pp) + (gz - sh(gz, 0, 1, 1)) * (sh(pp, 0, 0, 1) - sh(pp, 0, 1))
        )) * rdy
        put(v, vn, (is_, js), (n + 1, n), k1=nk)


def _ring_mask(g, width):
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    I, J = g.I[:, :, 0], g.J[:, :, 0]
    outer = (I >= is_ - width) & (I <= ie + width) & (J >= js - width) & (J <= je + width)
    inner = (I >= is_) & (I <= ie) & (J >= js) & (J <= je)
    return outer & ~inner


def edge_pe(g, pe, delp, ptop):
    """pe_halo.edge_pe (pe_halo.py:6-34): 1-wide ring around the compute domain."""
    m = _ring_mask(g, 1)
    K = g.nk + 1
    pe[:, :, 0][m] = ptop
    for k in range(1, K):
        pe[:, :, k][m] = (pe[:, :, k - 1] + delp[:, :, k - 1])[m]


def pk3_halo(g, pk3, delp, ptop, akap):
    """PK3Halo.__call__ (pk3_halo.py:11-69): 2-wide ring."""
    m = _ring_mask(g, 2)
    K = g.nk + 1
    pe = np.zeros(pk3.shape[:2])
    pe[m] = ptop
    with np.errstate(all="ignore"):
        for k in range(1, K):
            pe[m] = (pe + delp[:, :, k - 1])[m]
            pk3[:, :, k][m] = (pe ** akap)[m]


def ray_fast(g, u, v, w, dp, pfull, dt, ptop, rf_cutoff, tau, hydrostatic=False):
    """RayleighDamping.__call__ (ray_fast.py:48-206), literal statement order."""
    SDAY = 86400.0
    nk, n = g.nk, g.n
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    nudge = rf_cutoff + min(100.0, 10.0 * ptop)
    dp = np.asarray(dp, dtype=float)[:nk]
    pfull = np.asarray(pfull, dtype=float)[:nk]
    W = (slice(is_, ie + 2), slice(js, je + 2))
    shape2 = (n + 1, n + 1)
    with np.errstate(all="ignore"):
        rf = np.full(nk, np.nan)
        act = pfull < rf_cutoff
        rffvals = dt / (tau * SDAY) * np.sin(0.5 * c.PI * np.log(rf_cutoff / pfull) / np.log(rf_cutoff / ptop)) ** 2
        rf[act] = (1.0 / (1.0 + rffvals))[act]
        nz_ = pfull < nudge
        p_ref = np.full((shape2 + (nk,)), np.nan)
        if nz_[0]:
            p_ref[:, :, 0] = dp[0]
        for k in range(1, nk):
            p_ref[:, :, k] = p_ref[:, :, k - 1]
            if nz_[k]:
                p_ref[:, :, k] += dp[k]
        for k in range(nk - 2, -1, -1):
            if nz_[k]:
                p_ref[:, :, k] = p_ref[:, :, k + 1]

        def damp(wind, imax, jmax):
            wv = wind[W][:, :, :nk]
            reg = np.zeros(shape2, dtype=bool)
            reg[: imax, : jmax] = True
            dmdir = np.full(shape2 + (nk,), np.nan)
            for k in range(nk):
                if k > 0:
                    dmdir[:, :, k][reg] = dmdir[:, :, k - 1][reg]
                if act[k]:
                    layer = (1.0 - rf[k]) * dp[k] * wv[:, :, k]
                    if k == 0:
                        dmdir[:, :, k][reg] = layer[reg]
                    else:
                        dmdir[:, :, k][reg] = (dmdir[:, :, k] + layer)[reg]
                    wv[:, :, k][reg] = (wv[:, :, k] * rf[k])[reg]
                elif k == 0:
                    p_ref[:, :, 0][reg] = 0
            for k in range(nk - 2, -1, -1):
                if act[k]:
                    dmdir[:, :, k] = dmdir[:, :, k + 1]
            for k in range(nk):
                if nz_[k]:
                    wv[:, :, k][reg] = (wv[:, :, k] + dmdir[:, :, k] / p_ref[:, :, k])[reg]

        damp(u, n, n + 1)
        damp(v, n + 1, n)
        if not hydrostatic:
            wv = w[W][:, :, :nk]
            for k in range(nk):
                if act[k]:
                    wv[:n, :n, k] = wv[:n, :n, k] * rf[k]


def del2_cubed(g, qdel, cd, nmax, nk=None):
    """HyperdiffusionDamping.__call__ (del2cubed.py:168-194)."""
    if nk is None:
        nk = g.nk
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    ntimes = int(min(3, nmax))
    third = 1.0 / 3.0
    I, J = g.I, g.J
    q = np.zeros(qdel.shape)
    fx, fy = np.zeros(qdel.shape), np.zeros(qdel.shape)
    rarea = g.m2("rarea")
    with np.errstate(all="ignore"):
        for it in range(ntimes):
            nt = ntimes - (it + 1)
            # corner_fill :31-68
            a = qdel
            qn = a.copy()
            for (ci, cj, o1, o2) in (
                (is_, js, (-1, 0), (0, -1)), (is_ - 1, js, (1, 0), (1, -1)), (is_, js - 1, (0, 1), (-1, 1)),
                (ie, js, (1, 0), (0, -1)), (ie + 1, js, (-1, 0), (-1, -1)), (ie, js - 1, (0, 1), (1, 1)),
                (ie, je, (1, 0), (0, 1)), (ie + 1, je, (-1, 0), (-1, 1)), (ie, je + 1, (0, -1), (1, -1)),
                (is_, je, (-1, 0), (0, 1)), (is_ - 1, je, (1, 0), (1, 1)), (is_, je + 1, (0, -1), (-1, -1)),
            ):
                m = (I == ci) & (J == cj)
                # operand order as written in the reference for each case
                order = {
                    (is_, js): (a, sh(a, -1, 0), sh(a, 0, -1)), (is_ - 1, js): (sh(a, 1, 0), a, sh(a, 1, -1)),
                    (is_, js - 1): (sh(a, 0, 1), sh(a, -1, 1), a), (ie, js): (a, sh(a, 1, 0), sh(a, 0, -1)),
                    (ie + 1, js): (sh(a, -1, 0), a, sh(a, -1, -1)), (ie, js - 1): (sh(a, 0, 1), sh(a, 1, 1), a),
                    (ie, je): (a, sh(a, 1, 0), sh(a, 0, 1)), (ie + 1, je): (sh(a, -1, 0), a, sh(a, -1, 1)),
                    (ie, je + 1): (sh(a, 0, -1), sh(a, 1, -1), a), (is_, je): (a, sh(a, -1, 0), sh(a, 0, 1)),
                    (is_ - 1, je): (sh(a, 1, 0), a, sh(a, 1, 1)), (is_, je + 1): (sh(a, 0, -1), sh(a, -1, -1), a),
                }[(ci, cj)]
                qn = np.where(m, (order[0] + order[1] + order[2]) * third, qn)
            put(q, qn, (0, 0), (n + 6, n + 6), k1=nk)
            ks = slice(0, nk)
            if nt > 0:
                corner_ops.copy_corners(q, g, "x", ks=ks)
            put(fx, g.m2("del6_v") * (sh(q, -1, 0) - q), (is_ - nt, js - nt), (n + 1 + 2 * nt, n + 2 * nt), k1=nk)
            if nt > 0:
                corner_ops.copy_corners(q, g, "y", ks=ks)
            put(fy, g.m2("del6_u") * (sh(q, 0, -1) - q), (is_ - nt, js - nt), (n + 2 * nt, n + 1 + 2 * nt), k1=nk)
            put(qdel, q, (0, 0), (n + 6, n + 6), k1=nk)
            put(qdel, qdel + cd * rarea * (fx - sh(fx, 1, 0) + fy - sh(fy, 0, 1)), (is_ - nt, js - nt), (n + 2 * nt, n + 2 * nt), k1=nk)


def apply_diffusive_heating(g, delp, delz, cappa, heat_source, pt, delt_time_factor, nk):
    """temperature_adjust.apply_diffusive_heating (temperature_adjust.py:8-43), compute domain, nk levels."""
    W = (slice(g.is_, g.ie + 1), slice(g.js, g.je + 1), slice(0, nk))
    with np.errstate(all="ignore"):
        pkz = (c.RDG * delp[W] / delz[W] * pt[W]) ** (cappa[W] / (1.0 - cappa[W]))
        dtmp = heat_source[W] / (c.CV_AIR * delp[W])
        fac = np.full(nk, 1.0)
        fac[0] = 0.1
        if nk > 1:
            fac[1] = 0.5
        lim = delt_time_factor * fac[None, None, :]
        mag = np.minimum(lim, np.abs(dtmp))
        deltmin = np.where(dtmp > 0, np.abs(mag), -np.abs(mag))
        pt[W] = pt[W] + deltmin / pkz
