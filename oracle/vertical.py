"""ORACLE (test infrastructure) -- column (vertical) solvers of the acoustic step in numpy.

Follows fv3core/pace/fv3core/stencils/sim1_solver.py:20-219 (semi-implicit solver),
riem_solver3.py:26-321 (D-grid nonhydrostatic solver) and riem_solver_c.py:21-250 (C-grid).
k-sequential computations are python loops over levels with 2-D numpy slabs.
Parity status: see oracle/ppm_transport.py header.
"""
import math

import numpy as np

from . import constants as c


def sim1_solve(w, dm, gm, dz, pt, pm, pe, pem, ws, cp3, dt, p_fac, win, km):
    """sim1_solver.py:20-141 + Sim1Solver.__call__ :165-219 on the horizontal window
    win=(i0, i1, j0, j1) (half-open) and km layers (arrays hold km+1 levels).
    w, dz inout; pe out (nonhydrostatic perturbation pressure on interfaces)."""
    i0, i1, j0, j1 = win
    t1g = 2.0 * dt * dt
    rdt = 1.0 / dt
    W = (slice(i0, i1), slice(j0, j1))

    def v(a):
        return a[W]

    w_, dm_, gm_, dz_, pt_, pm_, pe_, pem_, cp3_ = (v(a) for a in (w, dm, gm, dz, pt, pm, pe, pem, cp3))
    ws_ = ws[W]
    ni, nj = w_.shape[0], w_.shape[1]
    K = km + 1
    with np.errstate(all="ignore"):
        pe_[:, :, :km] = np.exp(gm_[:, :, :km] * np.log(-dm_[:, :, :km] / dz_[:, :, :km] * c.RDGAS * pt_[:, :, :km])) - pm_[:, :, :km]
        w1 = w_[:, :, :km].copy()
        g_rat = np.zeros((ni, nj, K))
        bb = np.zeros((ni, nj, K))
        dd = np.zeros((ni, nj, K))
        g_rat[:, :, : km - 1] = dm_[:, :, : km - 1] / dm_[:, :, 1:km]
        bb[:, :, : km - 1] = 2.0 * (1.0 + g_rat[:, :, : km - 1])
        dd[:, :, : km - 1] = 3.0 * (pe_[:, :, : km - 1] + g_rat[:, :, : km - 1] * pe_[:, :, 1:km])
        bb[:, :, km - 1] = 2.0
        dd[:, :, km - 1] = 3.0 * pe_[:, :, km - 1]
        bet = np.zeros((ni, nj, K))
        bet[:, :, :km] = bb[:, :, 0:1]
        pp = np.zeros((ni, nj, K))
        gam = np.zeros((ni, nj, K))
        aa = np.zeros((ni, nj, K))
        pp[:, :, 1] = dd[:, :, 0] / bet[:, :, 1]
        for k in range(1, km):
            gam[:, :, k] = g_rat[:, :, k - 1] / bet[:, :, k - 1]
            bet[:, :, k] = bb[:, :, k] - gam[:, :, k]
        for k in range(2, K):
            pp[:, :, k] = (dd[:, :, k - 1] - pp[:, :, k - 1]) / bet[:, :, k - 1]
        for k in range(km - 1, 0, -1):
            pp[:, :, k] = pp[:, :, k] - gam[:, :, k] * pp[:, :, k + 1]
            aa[:, :, k] = (
                t1g * 0.5 * (gm_[:, :, k - 1] + gm_[:, :, k]) / (dz_[:, :, k - 1] + dz_[:, :, k]) * (pem_[:, :, k] + pp[:, :, k])
            )
        bet[:, :, 0] = dm_[:, :, 0] - aa[:, :, 1]
        for k in range(1, K):
            bet[:, :, k] = bet[:, :, k - 1]
        w_[:, :, 0] = (dm_[:, :, 0] * w1[:, :, 0] + dt * pp[:, :, 1]) / bet[:, :, 0]
        for k in range(1, km - 1):
            gam[:, :, k] = aa[:, :, k] / bet[:, :, k - 1]
            bet[:, :, k] = dm_[:, :, k] - (aa[:, :, k] + aa[:, :, k + 1] + aa[:, :, k] * gam[:, :, k])
            w_[:, :, k] = (
                dm_[:, :, k] * w1[:, :, k] + dt * (pp[:, :, k + 1] - pp[:, :, k]) - aa[:, :, k] * w_[:, :, k - 1]
            ) / bet[:, :, k]
        k = km - 1
        p1 = np.zeros((ni, nj, K))
        p1[:, :, k] = t1g * gm_[:, :, k] / dz_[:, :, k] * (pem_[:, :, k + 1] + pp[:, :, k + 1])
        gam[:, :, k] = aa[:, :, k] / bet[:, :, k - 1]
        bet[:, :, k] = dm_[:, :, k] - (aa[:, :, k] + p1[:, :, k] + aa[:, :, k] * gam[:, :, k])
        w_[:, :, k] = (
            dm_[:, :, k] * w1[:, :, k] + dt * (pp[:, :, k + 1] - pp[:, :, k]) - p1[:, :, k] * ws_ - aa[:, :, k] * w_[:, :, k - 1]
        ) / bet[:, :, k]
        for k in range(km - 2, -1, -1):
            w_[:, :, k] = w_[:, :, k] - gam[:, :, k + 1] * w_[:, :, k + 1]
        pe_[:, :, 0] = 0.0
        for k in range(1, K):
            pe_[:, :, k] = pe_[:, :, k - 1] + dm_[:, :, k - 1] * (w_[:, :, k - 1] - w1[:, :, k - 1]) * rdt
        k = km - 1
        p1[:, :, k] = (pe_[:, :, k] + 2.0 * pe_[:, :, k + 1]) * 1.0 / 3.0
        for k in range(km - 2, -1, -1):
            p1[:, :, k] = (pe_[:, :, k] + bb[:, :, k] * pe_[:, :, k + 1] + g_rat[:, :, k] * pe_[:, :, k + 2]) * 1.0 / 3.0 - g_rat[
                :, :, k
            ] * p1[:, :, k + 1]
        s = slice(0, km)
        # NB the reference compares p_fac * delta_mass (not p_fac * pm) -- sim1_solver.py:134
        maxp = np.where(p_fac * dm_[:, :, s] > p1[:, :, s] + pm_[:, :, s], p_fac * pm_[:, :, s], p1[:, :, s] + pm_[:, :, s])
        dz_[:, :, s] = -dm_[:, :, s] * c.RDGAS * pt_[:, :, s] * np.exp((cp3_[:, :, s] - 1.0) * np.log(maxp))


def riem_solver3(g, last_call, dt, cappa, ptop, zs, ws, delz, q_con, delp, pt, zh, pe, ppe, pk3, pk, peln, w, p_fac,
                 beta=0.0, use_logp=False):
    """NonhydrostaticVerticalSolver.__call__ (riem_solver3.py:208-321) on the compute domain."""
    km = g.nk
    K = km + 1
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    win = (is_, ie + 1, js, je + 1)
    W = (slice(is_, ie + 1), slice(js, je + 1))
    shape = delp.shape
    peln1 = math.log(ptop)
    ptk = math.exp(c.KAPPA * peln1)
    dm = np.zeros(shape)
    pe_init = np.zeros(shape)
    p_int = np.zeros(shape)
    logp = np.zeros(shape)
    gamma = np.zeros(shape)
    p_gas = np.zeros(shape)
    with np.errstate(all="ignore"):
        # precompute :26-90
        dm[W] = delp[W]
        pe_init[W] = pe[W]
        pg = np.zeros(shape)
        logpg = np.zeros(shape)
        p_int[W + (0,)] = ptop
        logp[W + (0,)] = peln1
        pk3[W + (0,)] = ptk
        pg[W + (0,)] = ptop
        logpg[W + (0,)] = peln1
        for k in range(1, K):
            p_int[W + (k,)] = p_int[W + (k - 1,)] + dm[W + (k - 1,)]
            logp[W + (k,)] = np.log(p_int[W + (k,)])
            pg[W + (k,)] = pg[W + (k - 1,)] + dm[W + (k - 1,)] * (1.0 - q_con[W + (k - 1,)])
            logpg[W + (k,)] = np.log(pg[W + (k,)])
            pk3[W + (k,)] = np.exp(c.KAPPA * logp[W + (k,)])
        gamma[W] = 1.0 / (1.0 - cappa[W])
        dm[W] = dm[W] * c.RGRAV
        s, s1 = slice(0, km), slice(1, K)
        p_gas[W + (s,)] = (pg[W + (s1,)] - pg[W + (s,)]) / (logpg[W + (s1,)] - logpg[W + (s,)])
        delz[W + (s,)] = zh[W + (s1,)] - zh[W + (s,)]
        sim1_solve(w, dm, gamma, delz, pt, p_gas, pe, p_int, ws, cappa, dt, p_fac, win, km)
        # finalize :93-145
        if use_logp:
            pk3[W] = logp[W]
        ppe[W] = (pe[W] + p_int[W]) if beta < -0.1 else pe[W]
        if last_call:
            peln[W] = logp[W]
            pk[W] = pk3[W]
            pe[W] = p_int[W]
        else:
            pe[W] = pe_init[W]
        zh[W + (km,)] = zs[W]
        for k in range(km - 1, -1, -1):
            zh[W + (k,)] = zh[W + (k + 1,)] - delz[W + (k,)]


def riem_solver_c(g, dt2, cappa, ptop, hs, ws, ptc, q_con, delpc, gz, pef, w3, p_fac):
    """NonhydrostaticVerticalSolverCGrid.__call__ (riem_solver_c.py:160-250) on compute +- 1."""
    km = g.nk
    K = km + 1
    is_, ie, js, je = g.is_, g.ie, g.js, g.je
    win = (is_ - 1, ie + 2, js - 1, je + 2)
    W = (slice(is_ - 1, ie + 2), slice(js - 1, je + 2))
    shape = delpc.shape
    dm, w, pem, pe, gm, dz, pm = (np.zeros(shape) for _ in range(7))
    with np.errstate(all="ignore"):
        # precompute :21-88
        dm[W] = delpc[W]
        w[W] = w3[W]
        peg = np.zeros(shape)
        pem[W + (0,)] = ptop
        peg[W + (0,)] = ptop
        for k in range(1, K):
            pem[W + (k,)] = pem[W + (k - 1,)] + dm[W + (k - 1,)]
            peg[W + (k,)] = peg[W + (k - 1,)] + dm[W + (k - 1,)] * (1.0 - q_con[W + (k - 1,)])
        s, s1 = slice(0, km), slice(1, K)
        dz[W + (s,)] = gz[W + (s1,)] - gz[W + (s,)]
        gm[W] = 1.0 / (1.0 - cappa[W])
        dm[W] = dm[W] / c.GRAV
        pm[W + (s,)] = (peg[W + (s1,)] - peg[W + (s,)]) / np.log(peg[W + (s1,)] / peg[W + (s,)])
        sim1_solve(w, dm, gm, dz, ptc, pm, pe, pem, ws, cappa, dt2, p_fac, win, km)
        # finalize :91-123
        pef[W + (0,)] = ptop
        pef[W + (s1,)] = pe[W + (s1,)] + pem[W + (s1,)]
        gz[W + (km,)] = hs[W]
        for k in range(km - 1, -1, -1):
            gz[W + (k,)] = gz[W + (k + 1,)] - dz[W + (k,)] * c.GRAV


def update_dz_c(g, dp_ref, zs, ut, vt, gz, ws, dt, gz_x=None, gz_y=None):
    """UpdateGeopotentialHeightOnCGrid.__call__ (updatedzc.py:172-207)."""
    from . import cgrid_sw
    from ._np import put, sh

    km = g.nk
    K = km + 1
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    gz_x = gz.copy()
    gz_y = gz.copy()
    ks = slice(0, K)
    cgrid_sw.fill_corners_cells_mult(gz_x, gz_x, g, "x", 2, ks=ks)
    cgrid_sw.fill_corners_cells_mult(gz_y, gz_y, g, "y", 2, ks=ks)
    area = g.m2("area")
    dp = np.zeros(K + 1)
    dp[:km] = np.asarray(dp_ref)[:km]
    with np.errstate(all="ignore"):
        xfx = np.zeros(gz.shape)
        yfx = np.zeros(gz.shape)
        for src, dst in ((ut, xfx), (vt, yfx)):
            # p_weighted_average_top / domain / bottom :15-31
            ratio = dp[0] / (dp[0] + dp[1])
            dst[:, :, 0] = src[:, :, 0] + (src[:, :, 0] - src[:, :, 1]) * ratio
            for k in range(1, km):
                int_ratio = 1.0 / (dp[k - 1] + dp[k])
                dst[:, :, k] = (dp[k] * src[:, :, k - 1] + dp[k - 1] * src[:, :, k]) * int_ratio
            ratio = dp[km - 1] / (dp[km - 2] + dp[km - 1])
            dst[:, :, km] = src[:, :, km - 1] + (src[:, :, km - 1] - src[:, :, km - 2]) * ratio
        fx = xfx * np.where(xfx > 0.0, sh(gz_x, -1, 0), gz_x)
        fy = yfx * np.where(yfx > 0.0, sh(gz_y, 0, -1), gz_y)
        new = (gz * area + fx - sh(fx, 1, 0) + fy - sh(fy, 0, 1)) / (area + xfx - sh(xfx, 1, 0) + yfx - sh(yfx, 0, 1))
        put(gz, new, (is_ - 1, js - 1), (n + 2, n + 2), k1=K)
        W = (slice(is_ - 1, ie + 2), slice(js - 1, je + 2))
        rdt = 1.0 / dt
        ws[W] = (zs[W] - gz[W + (km,)]) * rdt
        for k in range(km - 1, -1, -1):
            lim = gz[W + (k + 1,)] + c.DZ_MIN
            gz[W + (k,)] = np.where(gz[W + (k,)] > lim, gz[W + (k,)], lim)


def cubic_spline_constants(dp0):
    """updatedzd.cubic_spline_interpolation_constants (updatedzd.py:129-154); dp0 = dp_ref[:nz]."""
    dp0 = np.asarray(dp0, dtype=float)
    nz = dp0.shape[0]
    gk, beta, gamma = np.zeros(nz), np.zeros(nz), np.zeros(nz)
    gk[0] = dp0[1] / dp0[0]
    beta[0] = gk[0] * (gk[0] + 0.5)
    gamma[0] = (1.0 + gk[0] * (gk[0] + 1.5)) / beta[0]
    gk[1:] = dp0[:-1] / dp0[1:]
    for i in range(1, nz):
        beta[i] = 2.0 + 2.0 * gk[i] - gamma[i - 1]
        gamma[i] = gk[i] / beta[i]
    return gk, beta, gamma


def spline_to_interfaces(q_center, q_int, gk, beta, gamma, km, n_full):
    """cubic_spline_interpolation_from_layer_center_to_interfaces (updatedzd.py:157-196) on the full
    domain (origin_full, domain_full(add=(0,0,1)))."""
    W = (slice(0, n_full), slice(0, n_full))
    qc, qi = q_center[W], q_int[W]
    with np.errstate(all="ignore"):
        xt1 = 2.0 * gk[0] * (gk[0] + 1.0)
        qi[:, :, 0] = (xt1 * qc[:, :, 0] + qc[:, :, 1]) / beta[0]
        for k in range(1, km):
            qi[:, :, k] = (3.0 * (qc[:, :, k - 1] + gk[k] * qc[:, :, k]) - qi[:, :, k - 1]) / beta[k]
        a_bot = 1.0 + gk[km - 1] * (gk[km - 1] + 1.5)
        xt1 = 2.0 * gk[km - 1] * (gk[km - 1] + 1.0)
        xt2 = gk[km - 1] * (gk[km - 1] + 0.5) - a_bot * gamma[km - 1]
        qi[:, :, km] = (xt1 * qc[:, :, km - 1] + qc[:, :, km - 2] - a_bot * qi[:, :, km - 1]) / xt2
        for k in range(km - 1, -1, -1):
            qi[:, :, k] = qi[:, :, k] - gamma[k] * qi[:, :, k + 1]


def update_dz_d(g, col, dp_ref, zs, zh, crx, cry, xfx, yfx, wsd, dt, hord_tm=6):
    """UpdateHeightOnDGrid.__call__ (updatedzd.py:281-356)."""
    from . import ppm_transport as tr
    from ._np import put, sh

    km = g.nk
    K = km + 1
    is_, ie, js, je, n = g.is_, g.ie, g.js, g.je, g.n
    gk, beta, gamma = cubic_spline_constants(np.asarray(dp_ref)[:km])
    shape = zh.shape
    crx_i, cry_i, xfx_i, yfx_i = (np.zeros(shape) for _ in range(4))
    for src, dst in ((crx, crx_i), (xfx, xfx_i), (cry, cry_i), (yfx, yfx_i)):
        spline_to_interfaces(src, dst, gk, beta, gamma, km, n + 6)
    fx, fy = np.zeros(shape), np.zeros(shape)
    tr.fvtp2d(g, zh, crx_i, cry_i, xfx_i, yfx_i, fx, fy, hord_tm)
    fx2, fy2, wk = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    # damp_vt K-field holds nz+1 entries, the last is 0 (quantity_factory.zeros), as is nord_v's
    nord_k = np.zeros(K)
    nord_v = np.asarray(col["nord_v"], dtype=float)
    nord_k[:] = nord_v[3]
    nord_k[:3] = nord_v[:3]
    damp_k = np.zeros(K)
    damp_k[:km] = np.asarray(col["damp_vt"], dtype=float)[:km]
    tr.delnflux_nosg(g, zh, fx2, fy2, damp_k, wk, nord_k, nk=K)
    area = g.m2("area")
    with np.errstate(all="ignore"):
        area_after = (area + xfx_i - sh(xfx_i, 1, 0)) + (area + yfx_i - sh(yfx_i, 0, 1)) - area
        adv = (zh * area + fx - sh(fx, 1, 0) + fy - sh(fy, 0, 1)) / area_after
        new = adv + (fx2 - sh(fx2, 1, 0) + fy2 - sh(fy2, 0, 1)) / area
        put(zh, new, (is_, js), (n, n), k1=K)
        W = (slice(is_, ie + 1), slice(js, je + 1))
        wsd[W] = (zs[W] - zh[W + (km,)]) / dt
        for k in range(km - 1, -1, -1):
            other = zh[W + (k + 1,)] + c.DZ_MIN
            zh[W + (k,)] = np.where(zh[W + (k,)] > other, zh[W + (k,)], other)
