"""CPU oracle (TEST INFRASTRUCTURE, not product) for the operators DynamicalCore runs around the acoustic loop, the tracer
advection and the remapping (fv3core/pace/fv3core/stencils/fv_dynamics.py:424-624):

* ``neg_adj3``   -- AdjustNegativeTracerMixingRatio, fv3core/pace/fv3core/stencils/neg_adj3.py:11-420
* ``c2l_ord2`` / ``c2l_ord4`` -- CubedToLatLon, stencils/pace/stencils/c2l_ord.py:15-112
* ``fv_setup``, ``pt_to_potential_density_pt``, ``omega_from_w`` -- moist_cv.py:175-234, fv_dynamics.py:41-67

numpy / plain Python, the reference's operand order.  neg_adj3 is pinned bit for bit against a run of the reference
(tools/make_golden_dycore.py -> tests/golden/negadj_c12.npz); the others through the whole-step fixtures.
"""
import numpy as np

from . import constants as c


def _fix_negative_ice(qv, qi, qs, qg, qr, ql, pt, lcpk, icpk):
    """neg_adj3.py:13-54"""
    qsum = qi + qs
    if qsum > 0.0:
        if qi < 0.0:
            qi = 0.0
            qs = qsum
        elif qs < 0.0:
            qs = 0.0
            qi = qsum
    else:
        qi = 0.0
        qs = 0.0
        qg = qg + qsum
    if qg < 0.0:
        dq = qs if qs < -qg else -qg
        qs = qs - dq
        qg = qg + dq
        if qg < 0.0:
            dq = qi if qi < -qg else -qg
            qi = qi - dq
            qg = qg + dq
    if qg < 0.0 and qr > 0.0:
        dq = qr if qr < -qg else -qg
        qg = qg + dq
        ql = ql - dq
        pt = pt + dq * icpk
    if qg < 0.0 and ql > 0.0:
        dq = ql if ql < -qg else -qg
        qg = qg + dq
        ql = ql - dq
        pt = pt + dq * icpk
    if qg < 0.0 and qv > 0.0:
        dq = 0.999 * qv if 0.999 * qv < -qg else -qg
        qg = qg + dq
        qv = qv - dq
        pt = pt + dq * (icpk + lcpk)
    return qv, qi, qs, qg, qr, ql, pt


def _fix_negative_liq(qv, qi, qs, qg, qr, ql, pt, lcpk, icpk):
    """neg_adj3.py:57-98"""
    qsum = ql + qr
    pos_qg = 0.0 if 0.0 > qg else qg
    if qsum > 0.0:
        if qr < 0.0:
            qr = 0.0
            ql = qsum
        elif ql < 0.0:
            ql = 0.0
            qr = qsum
    else:
        ql = 0.0
        qr_tmp = qsum
        dq = pos_qg if pos_qg < -qr_tmp else -qr_tmp
        qr_tmp = qr_tmp + dq
        qg = qg - dq
        pt = pt - dq * icpk
        if qr < 0.0:
            dq = qi + qs if (qi + qs) < -qr_tmp else -qr_tmp
            qr_tmp = qr_tmp + dq
            dq1 = dq if dq < qs else qs
            qs = qs - dq1
            qi = qi + dq1 - dq
            pt = pt - dq * icpk
        qr = qr_tmp
        if qr < 0.0 and qv > 0.0:
            dq = 0.999 * qv if 0.999 * qv < -qr else -qr
            qv = qv - dq
            qr = qr + dq
            pt = pt + dq * lcpk
    return qv, qi, qs, qg, qr, ql, pt


def _fix_neg_water_point(pt, qv, ql, qr, qs, qi, qg, lv00, d0_vap):
    """fix_neg_water, neg_adj3.py:101-140"""
    q_liq = 0.0 if 0.0 > ql + qr else ql + qr
    q_sol = 0.0 if 0.0 > qi + qs else qi + qs
    cpm = (1.0 - (qv + q_liq + q_sol)) * c.CV_AIR + qv * c.CV_VAP + q_liq * c.C_LIQ + q_sol * c.C_ICE
    lcpk = (lv00 + d0_vap * pt) / cpm
    icpk = (c.LI0 + c.DC_ICE * pt) / cpm
    qv, qi, qs, qg, qr, ql, pt = _fix_negative_ice(qv, qi, qs, qg, qr, ql, pt, lcpk, icpk)
    qv, qi, qs, qg, qr, ql, pt = _fix_negative_liq(qv, qi, qs, qg, qr, ql, pt, lcpk, icpk)
    return pt, qv, ql, qr, qs, qi, qg


def _w(cnd, a, b):
    return np.where(cnd, a, b)


def fillq(q, dp, km):
    """neg_adj3.py:143-170"""
    s1 = np.zeros(q.shape[:2])
    s2 = np.zeros(q.shape[:2])
    for k in range(km):
        s1 = _w(q[:, :, k] > 0, s1 + q[:, :, k] * dp[:, :, k], s1)
    for k in range(km - 1, -1, -1):
        qk, dk = q[:, :, k], dp[:, :, k]
        m = (qk < 0.0) & (s1 >= 0)
        dq = _w(s1 < -qk * dk, s1, -qk * dk)
        s1 = _w(m, s1 - dq, s1)
        s2 = _w(m, s2 + dq, s2)
        q[:, :, k] = _w(m, qk + dq / dk, qk)
    for k in range(km - 1, -1, -1):
        qk, dk = q[:, :, k], dp[:, :, k]
        m = (qk > 0.0) & (s1 >= 1e-12) & (s2 > 0)
        dq = _w(s2 < qk * dk, s2, qk * dk)
        s2 = _w(m, s2 - dq, s2)
        q[:, :, k] = _w(m, qk - dq / dk, qk)


def fix_water_vapor_down(q, dp, km):
    """neg_adj3.py:174-248"""
    shp = q.shape
    upper = np.zeros(shp)
    lower = np.zeros(shp)
    q[:, :, 1] = _w(q[:, :, 0] < 0, q[:, :, 1] + q[:, :, 0] * dp[:, :, 0] / dp[:, :, 1], q[:, :, 1])
    q[:, :, 0] = _w(q[:, :, 0] < 0.0, 0.0, q[:, :, 0])
    for k in range(1, km - 1):
        qk, dk = q[:, :, k], dp[:, :, k]
        dq = q[:, :, k - 1] * dp[:, :, k - 1]
        lf = lower[:, :, k - 1]
        qk = _w(lf != 0, qk + lf / dk, qk)
        m = (qk < 0) & (q[:, :, k - 1] > 0)
        dq2 = _w(dq < -qk * dk, dq, -qk * dk)
        upper[:, :, k] = _w(m, dq2, upper[:, :, k])
        qk = _w(m, qk + dq2 / dk, qk)
        m2 = qk < 0
        lower[:, :, k] = _w(m2, qk * dk, lower[:, :, k])
        qk = _w(m2, 0.0, qk)
        q[:, :, k] = qk
    s = slice(0, km - 2)
    uf = upper[:, :, 1:km - 1]
    q[:, :, s] = _w(uf != 0, q[:, :, s] - uf / dp[:, :, s], q[:, :, s])
    kb = km - 1
    q[:, :, kb] = _w(lower[:, :, kb - 1] > 0, q[:, :, kb] + lower[:, :, kb] / dp[:, :, kb], q[:, :, kb])
    upper[:, :, kb] = q[:, :, kb]
    dpb = dp[:, :, kb]
    for k in range(km - 2, -1, -1):
        qk, dk = q[:, :, k], dp[:, :, k]
        dq = qk * dk
        un = upper[:, :, k + 1]
        m = (un < 0) & (qk > 0)
        dq = _w(m & (dq >= -un * dpb), -un * dpb, dq)
        q[:, :, k] = _w(m, qk - dq / dk, qk)
        upper[:, :, k] = _w(m, un + dq / dpb, un)
    q[:, :, kb] = upper[:, :, 0]


def fix_neg_cloud(dp, q, km):
    """neg_adj3.py:251-281"""
    for k in range(1, km - 1):
        q[:, :, k] = _w(q[:, :, k - 1] < 0.0, q[:, :, k] + q[:, :, k - 1] * dp[:, :, k - 1] / dp[:, :, k], q[:, :, k])
    s = slice(1, km - 1)
    q[:, :, s] = _w(q[:, :, s] < 0.0, 0.0, q[:, :, s])
    k = km - 2
    qk, dk, qn, dn = q[:, :, k], dp[:, :, k], q[:, :, k + 1], dp[:, :, k + 1]
    m = (qn < 0.0) & (qk > 0)
    dq = _w(-qk * dk < qn * dn, -qk * dk, qn * dn)
    q[:, :, k] = _w(m, qk - dq / dk, qk)
    k = km - 1
    qk, dk, qm, dm = q[:, :, k], dp[:, :, k], q[:, :, k - 1], dp[:, :, k - 1]
    m = (qk < 0) & (qm > 0.0)
    dq = _w(-qk * dk < qm * dm, -qk * dk, qm * dm)
    qn = qk + dq / dk
    qn = _w(0.0 > qn, 0.0, qn)
    q[:, :, k] = _w(m, qn, qk)


def neg_adj3(qvapor, qliquid, qrain, qsnow, qice, qgraupel, qcld, pt, delp, km):
    """AdjustNegativeTracerMixingRatio.__call__ (neg_adj3.py:377-420), non-hydrostatic; in place on (ni, nj, >= km)."""
    d0_vap = c.CV_VAP - c.C_LIQ
    lv00 = c.HLV - d0_vap * c.TICE
    ni, nj = pt.shape[:2]
    for i in range(ni):
        for j in range(nj):
            for k in range(km):
                r = _fix_neg_water_point(float(pt[i, j, k]), float(qvapor[i, j, k]), float(qliquid[i, j, k]), float(qrain[i, j, k]),
                                         float(qsnow[i, j, k]), float(qice[i, j, k]), float(qgraupel[i, j, k]), lv00, d0_vap)
                (pt[i, j, k], qvapor[i, j, k], qliquid[i, j, k], qrain[i, j, k], qsnow[i, j, k], qice[i, j, k],
                 qgraupel[i, j, k]) = r
    with np.errstate(all="ignore"):
        fillq(qgraupel, delp, km)
        fillq(qrain, delp, km)
        fix_water_vapor_down(qvapor, delp, km)
        fix_neg_cloud(delp, qcld, km)


C1, C2 = 1.125, -0.125


def c2l_ord2(u, v, dx, dy, a11, a12, a21, a22, n, km, o=3):
    """c2l_ord.py:15-51 on the compute domain + 1 halo cell (compute_halos = (1, 1)).  Returns ua, va (full arrays)."""
    ua, va = np.zeros(u.shape), np.zeros(u.shape)
    w = (slice(o - 1, o + n + 1), slice(o - 1, o + n + 1))
    jn = (w[0], slice(o, o + n + 2))
    ie = (slice(o, o + n + 2), w[1])
    K = slice(0, km)
    wu = u * dx[:, :, None]
    wv = v * dy[:, :, None]
    u1 = 2.0 * (wu[w + (K,)] + wu[jn + (K,)]) / (dx[w] + dx[jn])[:, :, None]
    v1 = 2.0 * (wv[w + (K,)] + wv[ie + (K,)]) / (dy[w] + dy[ie])[:, :, None]
    ua[w + (K,)] = a11[w][:, :, None] * u1 + a12[w][:, :, None] * v1
    va[w + (K,)] = a21[w][:, :, None] * u1 + a22[w][:, :, None] * v1
    return ua, va


def c2l_ord4(u, v, dx, dy, a11, a12, a21, a22, n, km, o=3):
    """ord4_transform, c2l_ord.py:54-112 (u, v with their halos updated), compute domain."""
    ua, va = np.zeros(u.shape), np.zeros(u.shape)
    I, J = slice(o, o + n), slice(o, o + n)
    K = slice(0, km)

    def sh(a, di, dj):
        return a[o + di:o + n + di, o + dj:o + n + dj]

    utmp = C2 * (sh(u, 0, -1)[:, :, K] + sh(u, 0, 2)[:, :, K]) + C1 * (sh(u, 0, 0)[:, :, K] + sh(u, 0, 1)[:, :, K])
    vtmp = C2 * (sh(v, -1, 0)[:, :, K] + sh(v, 2, 0)[:, :, K]) + C1 * (sh(v, 0, 0)[:, :, K] + sh(v, 1, 0)[:, :, K])
    dxe, dye = dx[:, :, None], dy[:, :, None]
    v_e = 2.0 * ((sh(v, 0, 0)[:, :, K] * sh(dye, 0, 0)) + (sh(v, 1, 0)[:, :, K] * sh(dye, 1, 0))) / (sh(dye, 0, 0) + sh(dye, 1, 0))
    u_e = 2.0 * (sh(u, 0, 0)[:, :, K] * sh(dxe, 0, 0) + sh(u, 0, 1)[:, :, K] * sh(dxe, 0, 1)) / (sh(dxe, 0, 0) + sh(dxe, 0, 1))
    edge = np.zeros((n, n, 1), dtype=bool)
    edge[0, :], edge[-1, :], edge[:, 0], edge[:, -1] = True, True, True, True
    utmp = np.where(edge, u_e, utmp)
    vtmp = np.where(edge, v_e, vtmp)
    ua[I, J, K] = a11[I, J][:, :, None] * utmp + a12[I, J][:, :, None] * vtmp
    va[I, J, K] = a21[I, J][:, :, None] * utmp + a22[I, J][:, :, None] * vtmp
    return ua, va


def fv_setup(t, pt, delp, delz):
    """moist_cv.fv_setup (moist_cv.py:175-234, moist_phys): returns q_con, cvm, pkz, cappa, dp1 for arrays of one shape;
    t: dict of the six water species."""
    ql = t["qliquid"] + t["qrain"]
    qs = t["qice"] + t["qsnow"] + t["qgraupel"]
    gz = ql + qs
    cvm = (1.0 - (t["qvapor"] + gz)) * c.CV_AIR + t["qvapor"] * c.CV_VAP + ql * c.C_LIQ + qs * c.C_ICE
    dp1 = c.ZVIR * t["qvapor"]
    cappa = c.RDGAS / (c.RDGAS + cvm / (1.0 + dp1))
    pkz = np.exp(cappa * np.log(c.RDG * delp * pt * (1.0 + dp1) * (1.0 - gz) / delz))
    return gz, cvm, pkz, cappa, dp1
