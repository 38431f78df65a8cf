"""ORACLE (test infrastructure) -- the acoustic loop of AcousticDynamics.__call__ for all six tiles of a (1, 1) layout,
in numpy, built from the other oracle modules.  Follows fv3core/pace/fv3core/stencils/dyn_core.py:670-970 statement by
statement (non-hydrostatic, rf_fast, nord > 0, d_con > 0); halo exchanges are oracle/halo.py.
Pinned by tests/golden/acoustic_c12_tile*.npz (a run of the reference itself, tools/make_golden_acoustic.py).
"""
import numpy as np

from . import acoustic_parts as ap
from . import cgrid_sw, dgrid_sw, halo, vertical
from . import constants as c

HUGE_R = 1.0e40


class TileState:
    """Per-tile persistent temporaries (dyn_core.py:192-218 and the operator classes' own)."""

    def __init__(self, shape):
        z = lambda: np.zeros(shape)  # noqa: E731
        self.ut, self.vt, self.gz, self.zh, self.pkc, self.pk3 = z(), z(), z(), z(), z(), z()
        self.heat_source, self.divgd, self.crx, self.cry, self.xfx, self.yfx = z(), z(), z(), z(), z(), z()
        self.ws3, self.wsd = np.zeros(shape[:2]), np.zeros(shape[:2])
        self.pk3[:] = HUGE_R
        self.csw = cgrid_sw.CSWState(shape)
        self.dsw = dgrid_sw.DSWState(shape)


def acoustic_dynamics(grids, col, cfg, states, cappas, timestep, n_split, n, nz, first_timestep=True, end_step=True):
    """grids: 6 oracle Grids; states: 6 dicts name -> ndarray (mutated in place); cappas: 6 arrays.
    cfg: dict with the d_sw keys of helpers.DSW_CFG plus p_fac, rf_cutoff, tau, delt_max, hord_tm."""
    T = range(6)
    dt = timestep / n_split
    dt2 = 0.5 * dt
    akap = c.KAPPA
    shape = states[0]["u"].shape
    tmp = [TileState(shape) for _ in T]
    zs = [states[t]["phis"] / c.GRAV for t in T]
    ptop = grids[0].ptop
    f = lambda name: [states[t][name] for t in T]  # noqa: E731
    halo.halo_update(f("q_con"), n, nk=nz)
    halo.halo_update(cappas, n, nk=nz)
    halo.halo_update(f("delp"), n, nk=nz)
    halo.halo_update(f("pt"), n, nk=nz)
    halo.vector_halo_update(f("u"), f("v"), n, grid="d", nk=nz)
    for t in T:
        for k in ("mfxd", "mfyd", "cxd", "cyd"):
            states[t][k][:-1, :-1, :nz] = 0.0
        if first_timestep:
            tmp[t].heat_source[3:-4, 3:-4, :nz] = 0.0
            states[t]["diss_estd"][3:-4, 3:-4, :nz] = 0.0
    for it in range(n_split):
        remap_step = it == n_split - 1
        halo.halo_update(f("w"), n, nk=nz)
        if it == 0:
            for t in T:
                W = (slice(3, 3 + n), slice(3, 3 + n))
                tmp[t].gz[W + (nz,)] = zs[t][W]
                for k in range(nz - 1, -1, -1):
                    tmp[t].gz[W + (k,)] = tmp[t].gz[W + (k + 1,)] - states[t]["delz"][W + (k,)]
            halo.halo_update([tmp[t].gz for t in T], n)
        for t in T:
            s, tt, g = states[t], tmp[t], grids[t]
            cgrid_sw.c_sw(g, tt.csw, s["delp"], s["pt"], s["u"], s["v"], s["w"], s["uc"], s["vc"], s["ua"], s["va"], tt.ut, tt.vt,
                          tt.divgd, s["omga"], dt2, nord=cfg["nord"])
        halo.halo_update([tmp[t].divgd for t in T], n, xi=1, yi=1, nk=nz)
        for t in T:
            s, tt, g = states[t], tmp[t], grids[t]
            if it == 0:
                tt.zh[:-1, :-1, :] = tt.gz[:-1, :-1, :]
            else:
                tt.gz[:-1, :-1, :] = tt.zh[:-1, :-1, :]
            vertical.update_dz_c(g, g.dp_ref, zs[t], tt.ut, tt.vt, tt.gz, tt.ws3, dt2)
            vertical.riem_solver_c(g, dt2, cappas[t], ptop, s["phis"], tt.ws3, tt.csw.ptc, s["q_con"], tt.csw.delpc, tt.gz, tt.pkc,
                                   s["omga"], p_fac=cfg["p_fac"])
            ap.p_grad_c(g, s["uc"], s["vc"], tt.csw.delpc, tt.pkc, tt.gz, dt2)
        halo.vector_halo_update(f("uc"), f("vc"), n, grid="c", nk=nz)
        for t in T:
            s, tt, g = states[t], tmp[t], grids[t]
            dgrid_sw.d_sw(g, col, cfg, tt.dsw, tt.vt, s["delp"], s["pt"], s["u"], s["v"], s["w"], s["uc"], s["vc"], s["ua"], s["va"],
                          tt.divgd, s["mfxd"], s["mfyd"], s["cxd"], s["cyd"], tt.crx, tt.cry, tt.xfx, tt.yfx, s["q_con"], tt.zh,
                          tt.heat_source, s["diss_estd"], dt)
        for name in ("delp", "pt", "q_con"):
            halo.halo_update(f(name), n, nk=nz)
        for t in T:
            s, tt, g = states[t], tmp[t], grids[t]
            vertical.update_dz_d(g, col, g.dp_ref, zs[t], tt.zh, tt.crx, tt.cry, tt.xfx, tt.yfx, tt.wsd, dt, hord_tm=cfg["hord_tm"])
            vertical.riem_solver3(g, remap_step, dt, cappas[t], ptop, zs[t], tt.wsd, s["delz"], s["q_con"], s["delp"], s["pt"], tt.zh,
                                  s["pe"], tt.pkc, tt.pk3, s["pk"], s["peln"], s["w"], p_fac=cfg["p_fac"])
        halo.halo_update([tmp[t].zh for t in T], n)
        halo.halo_update([tmp[t].pkc for t in T], n, n_pts=2)
        for t in T:
            s, tt, g = states[t], tmp[t], grids[t]
            if remap_step:
                ap.edge_pe(g, s["pe"], s["delp"], ptop)
            ap.pk3_halo(g, tt.pk3, s["delp"], ptop, akap)
            tt.gz[1:-2, 1:-2, :] = tt.zh[1:-2, 1:-2, :] * c.GRAV
            ap.nh_p_grad(g, s["u"], s["v"], tt.pkc, tt.gz, tt.pk3, s["delp"], dt, ptop, akap)
            ap.ray_fast(g, s["u"], s["v"], s["w"], g.dp_ref, g.p, dt, ptop, rf_cutoff=cfg["rf_cutoff"], tau=cfg["tau"])
        if it != n_split - 1:
            halo.vector_halo_update(f("u"), f("v"), n, grid="d", nk=nz)
        else:
            halo.synchronize_vector_interfaces(f("u"), f("v"), n, nk=nz)
    halo.halo_update([tmp[t].heat_source for t in T], n, nk=nz)
    for t in T:
        s, tt, g = states[t], tmp[t], grids[t]
        ap.del2_cubed(g, tt.heat_source, c.CNST_0P20 * g.da_min, 3)
        ap.apply_diffusive_heating(g, s["delp"], s["delz"], cappas[t], tt.heat_source, s["pt"], abs(dt * cfg["delt_max"]), nz)
    return tmp
