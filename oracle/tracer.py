"""ORACLE (test infrastructure) -- horizontal tracer advection (Fortran tracer_2d_1l) for all six tiles, in numpy.

Follows fv3core/pace/fv3core/stencils/tracer_2d_1l.py:19-392 statement by statement (cmax is hard-coded to 2.0 there, so
n_split = 3); the transport is oracle/ppm_transport.fvtp2d with the monotone ord-8 PPM, halo updates are oracle/halo.py.
Pinned by tests/golden/tracer_c12_tile*.npz (a run of the reference itself, tools/make_golden_tracer.py).
"""
import math

import numpy as np

from . import halo
from . import ppm_transport as tr
from ._np import put, sh


def flux_compute(g, cx, cy, xfx, yfx, nk):
    """flux_compute (tracer_2d_1l.py:19-77)."""
    n = g.n
    dxa, dya, dx, dy = g.m2("dxa"), g.m2("dya"), g.m2("dx"), g.m2("dy")
    sg1, sg2, sg3, sg4 = g.m2("sin_sg1"), g.m2("sin_sg2"), g.m2("sin_sg3"), g.m2("sin_sg4")
    with np.errstate(all="ignore"):
        x = np.where(cx > 0, cx * sh(dxa, -1, 0) * dy * sh(sg3, -1, 0), cx * dxa * dy * sg1)
        y = np.where(cy > 0, cy * sh(dya, 0, -1) * dx * sh(sg4, 0, -1), cy * dya * dx * sg2)
    put(xfx, x, (g.is_, g.js - 3), (n + 1, n + 6), k1=nk)
    put(yfx, y, (g.is_ - 3, g.js), (n + 6, n + 1), k1=nk)


def tracer_advection(grids, tracers, dp1, mfx, mfy, cx, cy, n, nk, hord=8):
    """TracerAdvection.__call__ (tracer_2d_1l.py:262-392).  tracers: dict name -> list of 6 arrays; the others: lists of 6
    arrays.  Everything is updated in place like the reference does."""
    T = range(6)
    shape = dp1[0].shape
    xfx = [np.zeros(shape) for _ in T]
    yfx = [np.zeros(shape) for _ in T]
    for t in T:
        flux_compute(grids[t], cx[t], cy[t], xfx[t], yfx[t], nk)
    n_split = math.floor(1.0 + 2.0)
    frac = 1.0 / n_split
    for t in T:
        for a in (cx[t], xfx[t], mfx[t], cy[t], yfx[t], mfy[t]):
            a[:, :, :nk] = a[:, :, :nk] * frac
    for q in tracers.values():
        halo.halo_update(q, n, nk=nk)
    dp2 = [np.zeros(shape) for _ in T]
    for it in range(n_split):
        last = it == n_split - 1
        for t in T:
            g = grids[t]
            rarea = g.m2("rarea")
            W = (slice(g.is_, g.ie + 1), slice(g.js, g.je + 1), slice(0, nk))
            with np.errstate(all="ignore"):
                d2 = dp1[t] + (mfx[t] - sh(mfx[t], 1, 0) + mfy[t] - sh(mfy[t], 0, 1)) * rarea
            dp2[t][W] = d2[W]
            for q in tracers.values():
                fx, fy = np.zeros(shape), np.zeros(shape)
                tr.fvtp2d(g, q[t], cx[t], cy[t], xfx[t], yfx[t], fx, fy, hord, x_mass_flux=mfx[t], y_mass_flux=mfy[t], nk=nk)
                with np.errstate(all="ignore"):
                    qn = (q[t] * dp1[t] + (fx - sh(fx, 1, 0) + fy - sh(fy, 0, 1)) * rarea) / dp2[t]
                q[t][W] = qn[W]
        if not last:
            for q in tracers.values():
                halo.halo_update(q, n, nk=nk)
            for t in T:
                g = grids[t]
                W = (slice(g.is_, g.ie + 1), slice(g.js, g.je + 1), slice(0, nk))
                tmp = dp1[t][W].copy()
                dp1[t][W] = dp2[t][W]
                dp2[t][W] = tmp
