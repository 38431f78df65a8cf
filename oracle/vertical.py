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
