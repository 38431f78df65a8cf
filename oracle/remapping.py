"""CPU oracle (TEST INFRASTRUCTURE, not product) for the vertical remapping operators of the reference:

* ``remap_profile``  -- RemapProfile.__call__, fv3core/pace/fv3core/stencils/remap_profile.py:566-681 (Fortran cs_profile),
  for |kord| in {9, 10} (the reference asserts kord <= 10; kord < 9 is not restated) and every ``iv``;
* ``map_single``     -- MapSingle.__call__, fv3core/pace/fv3core/stencils/map_single.py:96-200 (Fortran map_single /
  map1_ppm / map_scalar).

Plain numpy, vectorised over the columns, sequential in k where the reference is.  Every expression keeps the
reference's operand order.  Pinned bit for bit against a run of the reference itself (tools/make_golden_remap.py ->
tests/golden/remap_c12.npz, tests/test_oracle_golden.py).

Array convention: fields are (ni, nj, km + 1); ``km`` layers, level ``km`` is the extra (interface) level that the
reference's storages carry too.  ``pe1`` / ``pe2`` hold the km + 1 interface pressures.
"""
import numpy as np


def _w(c, a, b):
    return np.where(c, a, b)


def _min3(a, p, l):
    # a if (a < p) and (a < l) else p if p < l else l   (remap_profile.py:417-423)
    return _w((a < p) & (a < l), a, _w(p < l, p, l))


def _max3(a, p, l):
    return _w((a > p) & (a > l), a, _w(p > l, p, l))


def _posdef_constraint_iv0(a1, a2, a3, a4):
    """remap_profile.py:52-87"""
    with np.errstate(all="ignore"):
        c0 = a1 <= 0.0
        c1 = (np.abs(a3 - a2) < -a4) & ((a1 + 0.25 * (a3 - a2) ** 2 / a4 + a4 * (1.0 / 12.0)) < 0.0)
    c2 = (a1 < a3) & (a1 < a2)
    c3 = a3 > a2
    inner = ~c0 & c1
    # branch A (c0, or inner & c2): a2 = a3 = a1, a4 = 0
    A = c0 | (inner & c2)
    B = inner & ~c2 & c3    # a4 = 3 (a2 - a1); a3 = a2 - a4
    C = inner & ~c2 & ~c3   # a4 = 3 (a3 - a1); a2 = a3 - a4
    a4B = 3.0 * (a2 - a1)
    a3B = a2 - a4B
    a4C = 3.0 * (a3 - a1)
    a2C = a3 - a4C
    n2 = _w(A, a1, _w(C, a2C, a2))
    n3 = _w(A, a1, _w(B, a3B, a3))
    n4 = _w(A, 0.0, _w(B, a4B, _w(C, a4C, a4)))
    return a1, n2, n3, n4


def _posdef_constraint_iv1(a1, a2, a3, a4):
    """remap_profile.py:90-117"""
    da1 = a3 - a2
    da2 = da1 * da1
    a6da = a4 * da1
    A = ((a1 - a2) * (a1 - a3)) >= 0.0
    B = ~A & (a6da < -1.0 * da2)
    C = ~A & ~B & (a6da > da2)
    a4B = 3.0 * (a2 - a1)
    a3B = a2 - a4B
    a4C = 3.0 * (a3 - a1)
    a2C = a3 - a4C
    n2 = _w(A, a1, _w(C, a2C, a2))
    n3 = _w(A, a1, _w(B, a3B, a3))
    n4 = _w(A, 0.0, _w(B, a4B, _w(C, a4C, a4)))
    return a1, n2, n3, n4


def _remap_constraint(a1, a2, a3, a4, extm):
    """remap_profile.py:120-151"""
    da1 = a3 - a2
    da2 = da1 * da1
    a6da = a4 * da1
    A = extm
    B = ~A & (a6da < -da2)
    C = ~A & ~B & (a6da > da2)
    a4B = 3.0 * (a2 - a1)
    a3B = a2 - a4B
    a4C = 3.0 * (a3 - a1)
    a2C = a3 - a4C
    n2 = _w(A, a1, _w(C, a2C, a2))
    n3 = _w(A, a1, _w(B, a3B, a3))
    n4 = _w(A, 0.0, _w(B, a4B, _w(C, a4C, a4)))
    return a1, n2, n3, n4


def remap_profile(qs, a1, a2, a3, a4, delp, km, kord, iv, qmin=0.0, gam=None):
    """In place on a2, a3, a4 (and a1 where the reference writes it -- it never changes its value).  ``qs``: (ni, nj)
    bottom boundary value (used for iv == -2 only).  ``gam``: the RemapProfile object's persistent work field (level 0
    keeps whatever an earlier call left when iv == -2; zeros for a fresh object)."""
    kord = abs(kord)
    assert kord in (9, 10), "oracle restates kord 9 and 10"
    shp = a1.shape
    if gam is None:
        gam = np.zeros(shp)
    q = np.zeros(shp)
    # ---- set_initial_vals (remap_profile.py:154-264); k-domain = km + 1 levels
    with np.errstate(all="ignore"):
        if iv == -2:
            q[:, :, 0] = 1.5 * a1[:, :, 0]
            gam[:, :, 1] = 0.5
            gr = delp[:, :, 0] / delp[:, :, 1]
            bet = 2.0 + gr + gr - gam[:, :, 1]
            q[:, :, 1] = (3.0 * (a1[:, :, 0] + a1[:, :, 1]) - q[:, :, 0]) / bet
            grid_ratio = np.zeros(shp)
            for k in range(2, km):
                old_gr = delp[:, :, k - 2] / delp[:, :, k - 1]
                old_bet = 2.0 + old_gr + old_gr - gam[:, :, k - 1]
                gam[:, :, k] = old_gr / old_bet
                grid_ratio[:, :, k] = delp[:, :, k - 1] / delp[:, :, k]
            for k in range(2, km - 1):
                g = grid_ratio[:, :, k]
                bet = 2.0 + g + g - gam[:, :, k]
                q[:, :, k] = (3.0 * (a1[:, :, k - 1] + a1[:, :, k]) - q[:, :, k - 1]) / bet
            k = km - 1
            g = grid_ratio[:, :, k]
            q[:, :, k] = (3.0 * (a1[:, :, k - 1] + a1[:, :, k]) - g * qs - q[:, :, k - 1]) / (2.0 + g + g - gam[:, :, k])
            q[:, :, km] = qs
            for k in range(km - 2, -1, -1):
                q[:, :, k] = q[:, :, k] - gam[:, :, k + 1] * q[:, :, k + 1]
        else:
            gr = delp[:, :, 1] / delp[:, :, 0]
            bet = gr * (gr + 0.5)
            q[:, :, 0] = ((gr + gr) * (gr + 1.0) * a1[:, :, 0] + a1[:, :, 1]) / bet
            gam[:, :, 0] = (1.0 + gr * (gr + 1.5)) / bet
            for k in range(1, km):
                d4 = delp[:, :, k - 1] / delp[:, :, k]
                bet = 2.0 + d4 + d4 - gam[:, :, k - 1]
                q[:, :, k] = (3.0 * (a1[:, :, k - 1] + d4 * a1[:, :, k]) - q[:, :, k - 1]) / bet
                gam[:, :, k] = d4 / bet
            d4 = delp[:, :, km - 2] / delp[:, :, km - 1]
            a_bot = 1.0 + d4 * (d4 + 1.5)
            q[:, :, km] = (2.0 * d4 * (d4 + 1.0) * a1[:, :, km - 1] + a1[:, :, km - 2] - a_bot * q[:, :, km - 1]) / (
                d4 * (d4 + 0.5) - a_bot * gam[:, :, km - 1])
            for k in range(km - 1, -1, -1):
                q[:, :, k] = q[:, :, k] - gam[:, :, k] * q[:, :, k + 1]

    # ---- apply_constraints (remap_profile.py:267-337); k-domain = km levels
    tmp = np.zeros(shp)
    tmp2 = np.zeros(shp)
    a10 = a1[:, :, 0:km - 1]
    a11 = a1[:, :, 1:km]
    tmp[:, :, 1:km] = _w(a10 > a11, a10, a11)
    tmp2[:, :, 1:km] = _w(a10 < a11, a10, a11)
    gam[:, :, 1:km] = a11 - a10

    def clamp_hi(k):
        q[:, :, k] = _w(q[:, :, k] >= tmp[:, :, k], tmp[:, :, k], q[:, :, k])

    def clamp_lo(k):
        q[:, :, k] = _w(q[:, :, k] <= tmp2[:, :, k], tmp2[:, :, k], q[:, :, k])

    clamp_hi(1)
    clamp_lo(1)
    for k in range(2, km - 1):
        qk = q[:, :, k]
        g = gam[:, :, k - 1] * gam[:, :, k + 1]
        A = g > 0
        B = ~A & (gam[:, :, k - 1] > 0)
        C = ~A & ~B
        hi = _w(qk >= tmp[:, :, k], tmp[:, :, k], qk)
        both = _w(hi <= tmp2[:, :, k], tmp2[:, :, k], hi)
        lo = _w(qk <= tmp2[:, :, k], tmp2[:, :, k], qk)
        qc = hi
        if iv == 0:
            qc = _w(qc < 0.0, 0.0, qc)
        q[:, :, k] = _w(A, both, _w(B, lo, qc))
    clamp_hi(km - 1)
    clamp_lo(km - 1)
    a2[:, :, 0:km] = q[:, :, 0:km]
    a3[:, :, 0:km] = q[:, :, 1:km + 1]
    extm = np.zeros(shp, dtype=bool)
    extm[:, :, 0] = (a2[:, :, 0] - a1[:, :, 0]) * (a3[:, :, 0] - a1[:, :, 0]) > 0.0
    extm[:, :, 1:km - 1] = gam[:, :, 1:km - 1] * gam[:, :, 2:km] < 0.0
    extm[:, :, km - 1] = (a2[:, :, km - 1] - a1[:, :, km - 1]) * (a3[:, :, km - 1] - a1[:, :, km - 1]) > 0.0
    ext5 = np.zeros(shp, dtype=bool)
    ext6 = np.zeros(shp, dtype=bool)
    if kord > 9:
        s = slice(0, km)
        x0 = 2.0 * a1[:, :, s] - (a2[:, :, s] + a3[:, :, s])
        x1 = np.abs(a2[:, :, s] - a3[:, :, s])
        a4[:, :, s] = 3.0 * x0
        ext5[:, :, s] = np.abs(x0) > x1
        ext6[:, :, s] = np.abs(a4[:, :, s]) > x1

    # ---- set_interpolation_coefficients (remap_profile.py:340-563); k-domain = km levels
    def a4_std(k):
        a4[:, :, k] = 3.0 * (2.0 * a1[:, :, k] - (a2[:, :, k] + a3[:, :, k]))

    if iv == 0:
        a2[:, :, 0] = _w(a2[:, :, 0] < 0.0, 0.0, a2[:, :, 0])
        a4_std(0); a4_std(1)
    if iv == -1:
        a2[:, :, 0] = _w(a2[:, :, 0] * a1[:, :, 0] <= 0.0, 0.0, a2[:, :, 0])
        a4_std(0); a4_std(1)
    if iv == 2:
        a2[:, :, 0] = a1[:, :, 0]
        a3[:, :, 0] = a1[:, :, 0]
        a4[:, :, 0] = 0.0
        a4_std(1)
    if iv < -1 or iv == 1 or iv > 2:
        a4_std(0); a4_std(1)
    if iv != 2:
        _, a2[:, :, 0], a3[:, :, 0], a4[:, :, 0] = _posdef_constraint_iv1(a1[:, :, 0], a2[:, :, 0], a3[:, :, 0], a4[:, :, 0])
    _, a2[:, :, 1], a3[:, :, 1], a4[:, :, 1] = _remap_constraint(a1[:, :, 1], a2[:, :, 1], a3[:, :, 1], a4[:, :, 1], extm[:, :, 1])

    s = slice(2, km - 2)
    if km - 2 > 2:
        A1, A2, A3, A4 = a1[:, :, s], a2[:, :, s].copy(), a3[:, :, s].copy(), a4[:, :, s].copy()
        g0, gm, gp, gpp = gam[:, :, s], gam[:, :, 1:km - 3], gam[:, :, 3:km - 1], gam[:, :, 4:km]
        pmp_1 = A1 - 2.0 * gp
        lac_1 = pmp_1 + 1.5 * gpp
        pmp_2 = A1 + 2.0 * g0
        lac_2 = pmp_2 - 1.5 * gm
        if kord == 9:
            em, e0, ep = extm[:, :, 1:km - 3], extm[:, :, s], extm[:, :, 3:km - 1]
            cond = (e0 & em) | (e0 & ep) | (e0 & ((qmin > 0.0) & (A1 < qmin)))
            a4e = 6.0 * A1 - 3.0 * (A2 + A3)
            big = np.abs(a4e) > np.abs(A2 - A3)
            tmin = _min3(A1, pmp_1, lac_1)
            tmax0 = _w(A2 > tmin, A2, tmin)
            tmax = _max3(A1, pmp_1, lac_1)
            n2 = _w(tmax0 < tmax, tmax0, tmax)
            tmin = _min3(A1, pmp_2, lac_2)
            tmax0 = _w(A3 > tmin, A3, tmin)
            tmax = _max3(A1, pmp_2, lac_2)
            n3 = _w(tmax0 < tmax, tmax0, tmax)
            n4 = 6.0 * A1 - 3.0 * (n2 + n3)
            R2 = _w(cond, A1, _w(big, n2, A2))
            R3 = _w(cond, A1, _w(big, n3, A3))
            R4 = _w(cond, 0.0, _w(big, n4, a4e))
            A2, A3, A4 = R2, R3, R4
        else:  # kord == 10
            tmin2 = _min3(A1, pmp_1, lac_1)
            tmax2 = _max3(A1, pmp_1, lac_1)
            t2 = _w(A2 > tmin2, A2, tmin2)
            tmin3 = _w(A1 < pmp_2, A1, pmp_2)
            tmin3 = _w(lac_2 < tmin3, lac_2, tmin3)
            tmax3 = _w(A1 > pmp_2, A1, pmp_2)
            tmax3 = _w(lac_2 > tmax3, lac_2, tmax3)
            t3 = _w(A3 > tmin3, A3, tmin3)
            lim2 = _w(t2 < tmax2, t2, tmax2)
            lim3 = _w(t3 < tmax3, t3, tmax3)
            e5, e5m, e5p = ext5[:, :, s], ext5[:, :, 1:km - 3], ext5[:, :, 3:km - 1]
            e6, e6m, e6p = ext6[:, :, s], ext6[:, :, 1:km - 3], ext6[:, :, 3:km - 1]
            n5 = e5m | e5p
            n6 = e6m | e6p
            flat = e5 & n5
            limited = (e5 & ~n5 & n6) | (~e5 & e6 & n5)
            A2 = _w(flat, A1, _w(limited, lim2, A2))
            A3 = _w(flat, A1, _w(limited, lim3, A3))
            A4 = 3.0 * (2.0 * A1 - (A2 + A3))
        if iv == 0:
            _, A2, A3, A4 = _posdef_constraint_iv0(A1, A2, A3, A4)
        a2[:, :, s], a3[:, :, s], a4[:, :, s] = A2, A3, A4

    kb = km - 1
    if iv == 0:
        a3[:, :, kb] = _w(a3[:, :, kb] < 0.0, 0.0, a3[:, :, kb])
    if iv == -1:
        a3[:, :, kb] = _w(a3[:, :, kb] * a1[:, :, kb] <= 0.0, 0.0, a3[:, :, kb])
    a4_std(km - 2); a4_std(km - 1)
    k = km - 2
    _, a2[:, :, k], a3[:, :, k], a4[:, :, k] = _remap_constraint(a1[:, :, k], a2[:, :, k], a3[:, :, k], a4[:, :, k], extm[:, :, k])
    k = km - 1
    _, a2[:, :, k], a3[:, :, k], a4[:, :, k] = _posdef_constraint_iv1(a1[:, :, k], a2[:, :, k], a3[:, :, k], a4[:, :, k])
    return gam


def lagrangian_contributions(q, pe1, pe2, a1, a2, a3, a4, dp1, km):
    """map_single.py:21-93: layer means of the piecewise-parabolic profile between the target interfaces pe2.  The
    reference's per-column relative index ``lev`` is kept as the absolute source layer L."""
    ni, nj = q.shape[0], q.shape[1]
    L = np.zeros((ni, nj), dtype=np.int64)
    I, J = np.meshgrid(np.arange(ni), np.arange(nj), indexing="ij")

    def at(f, idx):
        return f[I, J, np.minimum(idx, f.shape[2] - 1)]

    with np.errstate(all="ignore"):
        for k in range(km):
            p2a, p2b = pe2[:, :, k], pe2[:, :, k + 1]
            pl = (p2a - at(pe1, L)) / at(dp1, L)
            inside = p2b <= at(pe1, L + 1)
            pr = (p2b - at(pe1, L)) / at(dp1, L)
            q_in = (at(a2, L) + 0.5 * (at(a4, L) + at(a3, L) - at(a2, L)) * (pr + pl)
                    - at(a4, L) * 1.0 / 3.0 * (pr * (pr + pl) + pl * pl))
            qsum = (at(pe1, L + 1) - p2a) * (at(a2, L) + 0.5 * (at(a4, L) + at(a3, L) - at(a2, L)) * (1.0 + pl)
                                            - at(a4, L) * 1.0 / 3.0 * (1.0 + pl * (1.0 + pl)))
            out = ~inside
            L = np.where(out, L + 1, L)
            while True:
                go = out & (L + 1 <= km) & (at(pe1, L + 1) < p2b)
                if not go.any():
                    break
                qsum = np.where(go, qsum + at(dp1, L) * at(a1, L), qsum)
                L = np.where(go, L + 1, L)
            dp = p2b - at(pe1, L)
            esl = dp / at(dp1, L)
            qsum = qsum + dp * (at(a2, L) + 0.5 * esl * (at(a3, L) - at(a2, L) + at(a4, L) * (1.0 - (2.0 / 3.0) * esl)))
            q[:, :, k] = np.where(inside, q_in, qsum / (p2b - p2a))


def map_single(q1, pe1, pe2, km, kord, iv, qs=None, qmin=0.0, gam=None):
    """MapSingle.__call__ (map_single.py:154-200): q1 (ni, nj, >= km) is remapped in place from the layers bounded by pe1
    to those bounded by pe2."""
    shp = q1.shape
    a1 = q1.copy()
    a2, a3, a4 = np.zeros(shp), np.zeros(shp), np.zeros(shp)
    dp1 = np.zeros(shp)
    dp1[:, :, 0:km] = pe1[:, :, 1:km + 1] - pe1[:, :, 0:km]
    if qs is None:
        qs = np.zeros(shp[:2])
    remap_profile(qs, a1, a2, a3, a4, dp1, km, kord, iv, qmin, gam)
    lagrangian_contributions(q1, pe1, pe2, a1, a2, a3, a4, dp1, km)
    return q1


def fillz(q, dp, km):
    """fix_tracer (fv3core/pace/fv3core/stencils/fillz.py:15-117, Fortran fillz): negative tracer masses borrow from the
    layers above / below, then the column is rescaled so that its mass (below the top layer) is unchanged.  In place on
    q (ni, nj, >= km); dp: layer thicknesses."""
    ni, nj = q.shape[:2]
    zfix = np.zeros((ni, nj), dtype=np.int64)
    lower = np.zeros((ni, nj, km))
    upper = np.zeros((ni, nj, km))
    dm = np.zeros((ni, nj, km))
    dm_pos = np.zeros((ni, nj, km))
    with np.errstate(all="ignore"):
        # fix_top (BACKWARD: level 1, then level 0)
        c = q[:, :, 0] < 0.0
        q[:, :, 1] = _w(c, q[:, :, 1] + q[:, :, 0] * dp[:, :, 0] / dp[:, :, 1], q[:, :, 1])
        q[:, :, 0] = _w(q[:, :, 0] < 0, 0.0, q[:, :, 0])
        dm[:, :, 0] = q[:, :, 0] * dp[:, :, 0]
        # fix_interior
        for k in range(1, km - 1):
            qk, dk = q[:, :, k], dp[:, :, k]
            lf = lower[:, :, k - 1]
            qk = _w(lf != 0.0, qk - (lf / dk), qk)
            neg = qk < 0.0
            zfix = zfix + neg
            up = q[:, :, k - 1] * dp[:, :, k - 1]
            need = -(qk * dk)
            dq = _w(up < need, up, need)
            b1 = neg & (q[:, :, k - 1] > 0.0)
            qk = _w(b1, qk + dq / dk, qk)
            upper[:, :, k] = _w(b1, dq, upper[:, :, k])
            lo = q[:, :, k + 1] * dp[:, :, k + 1]
            need = -(qk * dk)
            dq = _w(lo < need, lo, need)
            b2 = neg & (qk < 0.0) & (q[:, :, k + 1] > 0.0)
            qk = _w(b2, qk + dq / dk, qk)
            lower[:, :, k] = _w(b2, dq, lower[:, :, k])
            q[:, :, k] = qk
        s = slice(0, km - 1)
        uf = upper[:, :, 1:km]
        q[:, :, s] = _w(uf != 0.0, q[:, :, s] - uf / dp[:, :, s], q[:, :, s])
        dm[:, :, s] = q[:, :, s] * dp[:, :, s]
        dm_pos[:, :, s] = _w(dm[:, :, s] > 0.0, dm[:, :, s], 0.0)
        # fix_bottom
        k = km - 1
        qk, dk = q[:, :, k], dp[:, :, k]
        lf = lower[:, :, k - 1]
        qk = _w(lf != 0.0, qk - (lf / dk), qk)
        qup = q[:, :, k - 1] * dp[:, :, k - 1]
        qly = -qk * dk
        dup = _w(qup < qly, qup, qly)
        b = (qk < 0.0) & (q[:, :, k - 1] > 0.0)
        zfix = zfix + b
        qk = _w(b, qk + (dup / dk), qk)
        upper[:, :, k] = _w(b, dup, upper[:, :, k])
        q[:, :, k] = qk
        dm[:, :, k] = qk * dk
        dm_pos[:, :, k] = _w(dm[:, :, k] > 0.0, dm[:, :, k], 0.0)
        k = km - 2
        uf = upper[:, :, km - 1]
        q[:, :, k] = _w(uf != 0.0, q[:, :, k] - (uf / dp[:, :, k]), q[:, :, k])
        dm[:, :, k] = _w(uf != 0.0, q[:, :, k] * dp[:, :, k], dm[:, :, k])
        dm_pos[:, :, k] = _w(uf != 0.0, _w(dm[:, :, k] > 0.0, dm[:, :, k], 0.0), dm_pos[:, :, k])
        sum0 = np.zeros((ni, nj))
        sum1 = np.zeros((ni, nj))
        for k in range(1, km):
            sum0 = sum0 + dm[:, :, k]
            sum1 = sum1 + dm_pos[:, :, k]
        fac = _w(sum0 > 0.0, sum0 / sum1, 0.0)
        act = (zfix > 0) & (fac > 0.0)
        for k in range(1, km):
            v = fac * dm[:, :, k] / dp[:, :, k]
            q[:, :, k] = _w(act, _w(v > 0.0, v, 0.0), q[:, :, k])
    return zfix


# =====================================================================================================================
# LagrangianToEulerian (fv3core/pace/fv3core/stencils/remapping.py:286-695), do_sat_adj = False, non-hydrostatic,
# kord_tm < 0 (the only modes the reference implements besides the saturation adjustment)
# =====================================================================================================================
def _moist_cvm_gz(t):
    """moist_cv_nwat6_fn + moist_cvm (moist_cv.py:22-46); t: dict of the six water species."""
    from . import constants as c

    ql = t["qliquid"] + t["qrain"]
    qs = t["qice"] + t["qsnow"] + t["qgraupel"]
    gz = ql + qs
    cvm = (1.0 - (t["qvapor"] + gz)) * c.CV_AIR + t["qvapor"] * c.CV_VAP + ql * c.C_LIQ + qs * c.C_ICE
    return cvm, gz


TRACER_ORDER = ["qvapor", "qliquid", "qrain", "qice", "qsnow", "qgraupel", "qo3mr", "qsgs_tke", "qcld"]


def lagrangian_to_eulerian(f, tracers, ak, bk, ptop, akap, zvir, last_step, n, km, o=3, kord_tm=-9, kord_tr=9, kord_wz=9,
                           kord_mt=9, fill=True, nq=8, work=None):
    """In place on the dict ``f`` of arrays (ni, nj, km + 1): pt, delp, delz, peln, u, v, w, cappa, q_con, pkz, pk, pe and
    the 2-D ps, wsd; ``tracers``: dict name -> array.  ``o``: index of the first compute cell in both horizontal
    directions, ``n``: cells per direction.  ``work``: the operator's persistent pe2 (level values of the extra row
    je+1 survive from call to call in the reference; zeros for a fresh object)."""
    from . import constants as c

    C = (slice(o, o + n), slice(o, o + n))          # compute columns
    CJ = (slice(o, o + n), slice(o, o + n + 1))     # + the extra row (domain_jextra)
    CI = (slice(o, o + n + 1), slice(o, o + n))     # + the extra column
    pt, delp, delz, peln, pe = f["pt"], f["delp"], f["delz"], f["peln"], f["pe"]
    shp = pe.shape
    pe1, pe2 = np.zeros(shp), (np.zeros(shp) if work is None else work)
    # init_pe (remapping.py:42-56), domain_jextra
    pe2[CJ + (0,)] = ptop
    pe2[CJ + (km,)] = pe[CJ + (km,)]
    pe1[CJ] = pe[CJ]
    # moist_cv_pt_pressure (remapping.py:85-171)
    K = slice(0, km)
    t = {k: v[C + (K,)] for k, v in tracers.items()}
    cvm, gz = _moist_cvm_gz(t)
    f["q_con"][C + (K,)] = gz
    cappa = c.RDGAS / (c.RDGAS + cvm / (1.0 + zvir * t["qvapor"]))
    f["cappa"][C + (K,)] = cappa
    p = pt[C + (K,)]
    pt[C + (K,)] = p * np.exp(cappa / (1.0 - cappa) * np.log(c.RDG * delp[C + (K,)] / delz[C + (K,)] * p))
    delz[C + (K,)] = -delz[C + (K,)] / delp[C + (K,)]
    ps = pe[C + (km,)].copy()
    f["ps"][C] = ps
    pn2 = np.zeros(shp)
    pn2[C + (0,)] = peln[C + (0,)]
    for k in range(1, km):
        pe2[C + (k,)] = ak[k] + bk[k] * ps
    pn2[C + (km,)] = peln[C + (km,)]
    dp2 = np.zeros(shp)
    dp2[C + (K,)] = pe2[C + (slice(1, km + 1),)] - pe2[C + (K,)]
    delp[C + (K,)] = dp2[C + (K,)]
    # pn2_pk_delp (remapping.py:174-193)
    pn2[C + (K,)] = np.log(pe2[C + (K,)])
    f["pk"][C + (K,)] = np.exp(akap * pn2[C + (K,)])

    def remap(field, a, b, kord, iv, win, qs=None, qmin=0.0):
        q = field[win].copy()
        map_single(q, a[win], b[win], km, kord, iv, qs=qs, qmin=qmin)
        field[win + (K,)] = q[:, :, :km]

    # the remaps (remapping.py:587-592)
    remap(pt, peln, pn2, abs(kord_tm), 1, C, qmin=184.0)
    for t_index, name in enumerate(TRACER_ORDER[:nq]):
        remap(tracers[name], pe1, pe2, 9 if t_index == 5 else abs(kord_tr), 0, C)
    if fill:
        for name in TRACER_ORDER[:nq]:
            q = tracers[name][C].copy()
            fillz(q, dp2[C], km)
            tracers[name][C + (K,)] = q[:, :, :km]
    remap(f["w"], pe1, pe2, kord_wz, -2, C, qs=f["wsd"][C])
    remap(delz, pe1, pe2, kord_wz, 1, C)
    # undo_delz_adjust_and_copy_peln (remapping.py:59-80)
    delz[C + (K,)] = -delz[C + (K,)] * delp[C + (K,)]
    pe0 = np.zeros(shp)
    pe0[C] = peln[C]
    peln[C] = pn2[C]
    # moist_pkz (moist_cv.py:130-172)
    t = {k: v[C + (K,)] for k, v in tracers.items()}
    cvm, gz = _moist_cvm_gz(t)
    f["q_con"][C + (K,)] = gz
    cappa = c.RDGAS / (c.RDGAS + cvm / (1.0 + zvir * t["qvapor"]))
    f["cappa"][C + (K,)] = cappa
    f["pkz"][C + (K,)] = np.exp(cappa * np.log(c.RDG * delp[C + (K,)] / delz[C + (K,)] * pt[C + (K,)]))
    # pressures_mapu + u (remapping.py:196-227, 624)
    pe3 = np.zeros(shp)
    js = slice(o - 1, o + n)  # the row to the south of every row of CJ
    bot = pe[:, :, km]
    pe0[CJ + (0,)] = pe[CJ + (0,)]
    for k in range(1, km + 1):
        pe0[CJ + (k,)] = 0.5 * (pe[C[0], js, k] + pe1[CJ + (k,)])
    for k in range(km + 1):
        bkh = 0.5 * bk[k]
        pe3[CJ + (k,)] = ak[k] + bkh * (bot[C[0], js] + bot[CJ])
    remap(f["u"], pe0, pe3, kord_mt, -1, CJ)
    # pressures_mapv + v (remapping.py:230-254, 627)
    iw = slice(o - 1, o + n)
    pe3[CI + (0,)] = ak[0]
    pe0[CI + (0,)] = pe[CI + (0,)]
    for k in range(1, km + 1):
        bkh = 0.5 * bk[k]
        pe0[CI + (k,)] = 0.5 * (pe[iw, C[1], k] + pe[CI + (k,)])
        pe3[CI + (k,)] = ak[k] + bkh * (bot[iw, C[1]] + bot[CI])
    remap(f["v"], pe0, pe3, kord_mt, -1, CI)
    # update_ua (remapping.py:257-273) + copy_from_below (:276-283): pe = the Eulerian interfaces
    for k in range(1, km):
        pe[C + (k,)] = pe2[C + (k,)]
    # last step / not (remapping.py:675-695)
    if last_step:
        KK = slice(0, km + 1)
        tt = {k: v[C + (KK,)] for k, v in tracers.items()}
        gz = tt["qliquid"] + tt["qrain"] + tt["qice"] + tt["qsnow"] + tt["qgraupel"]
        with np.errstate(all="ignore"):
            pt[C + (KK,)] = (pt[C + (KK,)] + 0.0 * f["pkz"][C + (KK,)]) / ((1.0 + zvir * tt["qvapor"]) * (1.0 - gz))
    else:
        pt[C + (K,)] = pt[C + (K,)] / f["pkz"][C + (K,)]
    return pe2
