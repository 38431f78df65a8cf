"""The per-stencil device implementations behind FrozenStencil for the gtscript definitions the reference's Translate tests
launch as stencils of their own (pace_amd/dsl/device_stencils.py -> pace_stencil -> csrc/k_stencils.hip): built through
StencilFactory.from_origin_domain with the windows those tests use and compared, exactly, with numpy restatements of the
definitions (the corner fills: oracle/corner_ops.py, which the whole-operator tests already hold to the reference's runs).

translate_d_sw.py:104 (ubke), :151 (vbke), :184 (flux_capacitor), :226 (heat_diss), :251 (apply_fluxes);
translate_xtp_u.py:13-23 / translate_ytp_v.py (xtp_u_stencil_defn, ytp_v_stencil_defn);
translate_corners.py:29-36 (fill_corners_2cells_{x,y}_stencil), :120 (fill_corners_dgrid_defn); corners.py:17-59 (CopyCorners).
CPU: the kernel sources under emulation; `-m gpu`: the HIP library.
"""
import types

import numpy as np
import pytest

from helpers import Env, build_emu, oracle_grid

N, NZ = 12, 4


def _defn(module, name, args):
    """A stand-in for the reference's definition function: FrozenStencil only looks at its identity and argument names."""
    f = eval(f"lambda {', '.join(args)}: None")
    f.__name__ = name
    f.__module__ = module
    return f


def _env(lib, device):
    from pace_amd import synthetic

    m = synthetic.tile_metrics(N, NZ)
    return Env(lib, device, m, N, NZ), m


def _rand(rng):
    return rng.standard_normal((N + 7, N + 7, NZ + 1))


def run_all(lib, device):
    env, m = _env(lib, device)
    sf, gi = env.stencil_factory, env.grid_indexing
    rng = np.random.default_rng(11)
    g = oracle_grid(m, N, NZ)
    W = (slice(3, 3 + N), slice(3, 3 + N), slice(0, NZ))

    # flux_capacitor: full domain
    a = {k: _rand(rng) for k in "cx cy xflux yflux crx_adv cry_adv fx fy".split()}
    q = {k: env.q3(v) for k, v in a.items()}
    st = sf.from_origin_domain(_defn("pace.fv3core.stencils.d_sw", "flux_capacitor", list(a)), origin=gi.origin_full(), domain=gi.domain_full())
    st(**q)
    F = (slice(0, N + 6), slice(0, N + 6), slice(0, NZ))
    for acc, inc in (("cx", "crx_adv"), ("cy", "cry_adv"), ("xflux", "fx"), ("yflux", "fy")):
        assert np.array_equal(q[acc].numpy()[F], (a[acc] + a[inc])[F]), acc
        assert np.array_equal(q[acc].numpy()[N + 6], a[acc][N + 6]), "written outside the window"

    # heat_diss: compute domain
    a = {k: _rand(rng) for k in "fx2 fy2 w heat_source diss_est dw".split()}
    damp_w, ke_bg = np.array([0.0, 1e-6, 0.02, 0.03, 0.0]), np.array([0.1, 0.2, 0.3, 0.4, 0.0])
    q = {k: env.q3(v) for k, v in a.items()}
    st = sf.from_origin_domain(_defn("pace.fv3core.stencils.d_sw", "heat_diss", ["fx2", "fy2", "w", "rarea", "heat_source", "diss_est", "dw", "damp_w", "ke_bg", "dt"]),
                               origin=gi.origin_compute(), domain=gi.domain_compute())
    dt = -7.5
    st(q["fx2"], q["fy2"], q["w"], env.grid_data.rarea, q["heat_source"], q["diss_est"], q["dw"], env.kq(damp_w), env.kq(ke_bg), dt)
    rarea = m["rarea"][:, :, None]
    dwr = (a["fx2"] - np.roll(a["fx2"], -1, 0) + a["fy2"] - np.roll(a["fy2"], -1, 1)) * rarea
    on = (damp_w > 1e-5)[None, None, :]
    hs = np.where(on, ke_bg[None, None, :] * abs(dt) - dwr * (a["w"] + 0.5 * dwr), 0.0)
    assert np.array_equal(q["heat_source"].numpy()[W], hs[W]) and np.array_equal(q["diss_est"].numpy()[W], hs[W])
    assert np.array_equal(q["dw"].numpy()[W], np.where(on, dwr, a["dw"])[W])
    assert np.array_equal(q["heat_source"].numpy()[0], a["heat_source"][0])

    # apply_fluxes: compute domain
    a = {k: _rand(rng) for k in "q delp gx gy".split()}
    q = {k: env.q3(v) for k, v in a.items()}
    st = sf.from_origin_domain(_defn("pace.fv3core.stencils.d_sw", "apply_fluxes", ["q", "delp", "gx", "gy", "rarea"]),
                               origin=gi.origin_compute(), domain=gi.domain_compute())
    st(q["q"], q["delp"], q["gx"], q["gy"], env.grid_data.rarea)
    ref = a["q"] * a["delp"] + (a["gx"] - np.roll(a["gx"], -1, 0) + a["gy"] - np.roll(a["gy"], -1, 1)) * rarea
    assert np.array_equal(q["q"].numpy()[W], ref[W])

    # ubke / vbke: compute + 1 (the B-grid points)
    a = {k: _rand(rng) for k in "uc vc ut vt".split()}
    q = {k: env.q3(v) for k, v in a.items()}
    ub, vb = env.q3(), env.q3()
    o, d = gi.origin_compute(), gi.domain_compute(add=(1, 1, 0))
    su = sf.from_origin_domain(_defn("translate_d_sw", "ubke", ["uc", "vc", "cosa", "rsina", "ut", "ub", "dt4", "dt5"]), origin=o, domain=d)
    sv = sf.from_origin_domain(_defn("translate_d_sw", "vbke", ["vc", "uc", "cosa", "rsina", "vt", "vb", "dt4", "dt5"]), origin=o, domain=d)
    dt5 = 0.5 * 3.7
    su(q["uc"], q["vc"], env.grid_data.cosa, env.grid_data.rsina, q["ut"], ub, 0.25 * 3.7, dt5)
    sv(q["vc"], q["uc"], env.grid_data.cosa, env.grid_data.rsina, q["vt"], vb, 0.25 * 3.7, dt5)
    sh = lambda x, di, dj: np.roll(np.roll(x, -di, 0), -dj, 1)  # noqa: E731  value at (i + di, j + dj)
    cosa, rsina = m["cosa"][:, :, None], m["rsina"][:, :, None]
    ub_cov = 0.5 * (sh(a["uc"], 0, -1) + a["uc"])
    vb_cov = 0.5 * (sh(a["vc"], -1, 0) + a["vc"])
    I, J = np.meshgrid(np.arange(N + 7), np.arange(N + 7), indexing="ij")
    jedge = ((J == 3) | (J == 3 + N))[:, :, None]
    iedge = ((I == 3) | (I == 3 + N))[:, :, None]
    rub = (ub_cov - vb_cov * cosa) * rsina
    rub = np.where(jedge, 0.25 * (-sh(a["ut"], 0, -2) + 3.0 * (sh(a["ut"], 0, -1) + a["ut"]) - sh(a["ut"], 0, 1)), rub)
    rub = np.where(iedge, 0.5 * (sh(a["ut"], 0, -1) + a["ut"]), rub) * (2.0 * dt5)
    rvb = (vb_cov - ub_cov * cosa) * rsina
    rvb = np.where(iedge, 0.25 * (-sh(a["vt"], -2, 0) + 3.0 * (sh(a["vt"], -1, 0) + a["vt"]) - sh(a["vt"], 1, 0)), rvb)
    rvb = np.where(jedge, 0.5 * (sh(a["vt"], -1, 0) + a["vt"]), rvb) * (2.0 * dt5)
    B = (slice(3, 4 + N), slice(3, 4 + N), slice(0, NZ))
    assert np.array_equal(ub.numpy()[B], rub[B]) and np.array_equal(vb.numpy()[B], rvb[B])


    # xtp_u / ytp_v: compute + 1 (translate_xtp_u.py:41-42), iord 5 and 6, against the oracle's advect_wind_1d with dt = 1
    from oracle import ppm_transport as tr

    for iord in (5, 6):
        a = {k: _rand(rng) for k in "c u".split()}
        a["c"] *= 0.3  # a Courant number
        for name, axis, fargs, margs in (("xtp_u_stencil_defn", 0, ["ub_contra_times_dt", "u", "updated_u"], ("dx", "dxa", "rdx")),
                                         ("ytp_v_stencil_defn", 1, ["vb_contra_times_dt", "v", "updated_v"], ("dy", "dya", "rdy"))):
            q = {k: env.q3(v) for k, v in a.items()}
            out = env.q3()
            st = sf.from_origin_domain(_defn("translate_" + name[:5], name, fargs + list(margs)), origin=gi.origin_compute(),
                                       domain=gi.domain_compute(add=(1, 1, 0)), externals={"iord": iord, "mord": iord, "xt_minmax": False})
            st(q["c"], q["u"], out, *[getattr(env.grid_data, k) for k in margs])
            ref = tr.advect_wind_1d(a["u"], a["c"], m[margs[2]], m[margs[0]], m[margs[1]], 1.0, g, axis, iord)
            B = (slice(3, 4 + N), slice(3, 4 + N), slice(0, NZ))
            assert np.array_equal(out.numpy()[B], ref[B]), (name, iord)
            assert not out.numpy()[2].any() and not out.numpy()[:, 2].any(), "written outside the window"

    # moist_pt_last_step: compute domain (translate_last_step.py:16)
    names = "qvapor qliquid qrain qsnow qice qgraupel gz pt pkz".split()
    a = {k: np.abs(_rand(rng)) * (0.01 if k.startswith("q") else 1.0) + (200.0 if k == "pt" else 0.0) for k in names}
    q = {k: env.q3(v) for k, v in a.items()}
    st = sf.from_origin_domain(_defn("pace.fv3core.stencils.moist_cv", "moist_pt_last_step", names + ["dtmp", "r_vir"]),
                               origin=gi.origin_compute(), domain=gi.domain_compute())
    dtmp, r_vir = 0.37, 0.6078
    st(*[q[k] for k in names], dtmp, r_vir)
    cond = a["qliquid"] + a["qrain"] + a["qice"] + a["qsnow"] + a["qgraupel"]
    ptn = (a["pt"] + dtmp * a["pkz"]) / ((1.0 + r_vir * a["qvapor"]) * (1.0 - cond))
    assert np.array_equal(q["gz"].numpy()[W], cond[W]) and np.array_equal(q["pt"].numpy()[W], ptn[W])
    assert np.array_equal(q["pt"].numpy()[0], a["pt"][0]), "written outside the window"

    # moist_pkz / moist_pt: one row of the compute domain (translate_moistcvpluspkz_2d.py:19-24, translate_moistcvpluspt_2d.py:50-55)
    from oracle import constants as oc

    water = "qvapor qliquid qrain qsnow qice qgraupel".split()
    a = {k: np.abs(_rand(rng)) * 0.004 for k in water}
    a.update(pt=np.abs(_rand(rng)) * 3.0 + 250.0, delp=np.abs(_rand(rng)) * 50.0 + 500.0, delz=-(np.abs(_rand(rng)) * 40.0 + 300.0))
    r_vir = 0.6078
    ql, qs = a["qliquid"] + a["qrain"], a["qice"] + a["qsnow"] + a["qgraupel"]
    gz = ql + qs
    cvm = (1.0 - (a["qvapor"] + gz)) * oc.CV_AIR + a["qvapor"] * oc.CV_VAP + ql * oc.C_LIQ + qs * oc.C_ICE
    cappa = oc.RDGAS / (oc.RDGAS + cvm / (1.0 + r_vir * a["qvapor"]))
    row = (slice(3, 3 + N), slice(5, 6), slice(0, NZ))
    o_row, d_row = (3, 5, 0), (N, 1, NZ)
    names = water + "q_con gz cvm pkz pt cappa delp delz".split()
    q = {k: env.q3(a.get(k, np.zeros_like(gz))) for k in names}
    st = sf.from_origin_domain(_defn("pace.fv3core.stencils.moist_cv", "moist_pkz", names + ["r_vir"]), origin=o_row, domain=d_row)
    st(*[q[k] for k in names], r_vir)
    pkz = np.exp(cappa * np.log(oc.RDG * a["delp"] / a["delz"] * a["pt"]))
    for k, ref in (("q_con", gz), ("gz", gz), ("cvm", cvm), ("cappa", cappa)):
        assert np.array_equal(q[k].numpy()[row], ref[row]), k
    assert np.abs(q["pkz"].numpy()[row] / pkz[row] - 1.0).max() < 1e-14  # (the device's exp / log)
    assert not q["pkz"].numpy()[:, 4].any() and not q["pkz"].numpy()[:, 6].any(), "written outside the window"
    names = water + "q_con pt cappa delp delz".split()
    q = {k: env.q3(a.get(k, np.zeros_like(gz))) for k in names}
    st = sf.from_origin_domain(_defn("translate_moistcvpluspt_2d", "moist_pt", names + ["r_vir"]), origin=o_row, domain=d_row)
    st(*[q[k] for k in names], r_vir)
    ptn = a["pt"] * np.exp(cappa / (1.0 - cappa) * np.log(oc.RDG * a["delp"] / a["delz"] * a["pt"]))
    assert np.array_equal(q["q_con"].numpy()[row], gz[row]) and np.array_equal(q["cappa"].numpy()[row], cappa[row])
    assert np.abs(q["pt"].numpy()[row] / ptn[row] - 1.0).max() < 1e-14
    assert np.array_equal(q["pt"].numpy()[:, 4], a["pt"][:, 4]), "written outside the window"

    # corner fills: full domain (translate_corners.py)
    from oracle import corner_ops as co

    of, df = gi.origin_full(), gi.domain_full()
    ks = slice(0, NZ)
    for name, direction, oracle_fn in (("copy_corners_x_stencil_defn", "x", co.copy_corners), ("copy_corners_y_stencil_defn", "y", co.copy_corners),
                                       ("fill_corners_bgrid_x_defn", "x", co.fill_corners_bgrid), ("fill_corners_bgrid_y_defn", "y", co.fill_corners_bgrid)):
        a0 = _rand(rng)
        qq = env.q3(a0)
        # (FillCornersBGrid builds its stencils on the interface dims: one more point each way, corners.py:545-588)
        dom = gi.domain_full(add=(1, 1, 0)) if "bgrid" in name else df
        st = sf.from_origin_domain(_defn("pace.stencils.corners", name, ["q_in", "q_out"]), origin=of, domain=dom)
        st(qq, qq)
        ref = a0.copy()
        oracle_fn(ref, g, direction, ks)
        assert np.array_equal(qq.numpy()[:, :, ks], ref[:, :, ks]), name
    for mysign in (1.0, -1.0):
        x0, y0 = _rand(rng), _rand(rng)
        qx, qy = env.q3(x0), env.q3(y0)
        st = sf.from_origin_domain(_defn("pace.stencils.corners", "fill_corners_dgrid_defn", ["x_in", "x_out", "y_in", "y_out", "mysign"]),
                                   origin=of, domain=gi.domain_full(add=(1, 1, 0)))
        st(qx, qx, qy, qy, mysign)
        rx, ry = x0.copy(), y0.copy()
        co.fill_corners_dgrid(rx, ry, g, mysign, ks)
        assert np.array_equal(qx.numpy()[:, :, ks], rx[:, :, ks]) and np.array_equal(qy.numpy()[:, :, ks], ry[:, :, ks])
    # fill_corners_2cells_{x,y}_stencil (corners.py:130-177, 224-262): eight cells each
    is_, ie, js, je = 3, 2 + N, 3, 2 + N
    for name, table in (("fill_corners_2cells_x_stencil",
                         [((is_ - 1, js - 1), (0, 1)), ((is_ - 2, js - 1), (1, 2)), ((ie + 1, js - 1), (0, 1)), ((ie + 2, js - 1), (-1, 2)),
                          ((is_ - 1, je + 1), (0, -1)), ((is_ - 2, je + 1), (1, -2)), ((ie + 1, je + 1), (0, -1)), ((ie + 2, je + 1), (-1, -2))]),
                        ("fill_corners_2cells_y_stencil",
                         [((is_ - 1, js - 1), (1, 0)), ((is_ - 1, js - 2), (2, 1)), ((ie + 1, js - 1), (-1, 0)), ((ie + 1, js - 2), (-2, 1)),
                          ((is_ - 1, je + 1), (1, 0)), ((is_ - 1, je + 2), (2, -1)), ((ie + 1, je + 1), (-1, 0)), ((ie + 1, je + 2), (-2, -1))])):
        a0 = _rand(rng)
        qq = env.q3(a0)
        st = sf.from_origin_domain(_defn("pace.stencils.corners", name, ["q_out", "q_in"]), origin=of, domain=df)
        st(qq, qq)
        ref = a0.copy()
        for (di, dj), (oi, oj) in table:
            ref[di, dj, ks] = a0[di + oi, dj + oj, ks]
        assert np.array_equal(qq.numpy()[:, :, ks], ref[:, :, ks]), name


def test_unregistered_definition_is_refused():
    from pace_amd import _lib

    env, _ = _env(_lib.Library(build_emu()), "cpu")
    with pytest.raises(NotImplementedError):
        env.stencil_factory.from_origin_domain(_defn("pace.fv3core.stencils.d_sw", "no_such_stencil", ["q"]),
                                               origin=env.grid_indexing.origin_compute(), domain=env.grid_indexing.domain_compute())


def test_compare_to_numpy_is_refused_not_ignored():
    """dsl/pace/dsl/stencil.py:166-234: the reference's compare-to-numpy mode has no counterpart at run time here; asking for it
    raises at construction instead of being dropped silently."""
    import copy

    from pace_amd import _lib

    env, _ = _env(_lib.Library(build_emu()), "cpu")
    cfg = copy.copy(env.stencil_factory.config)
    cfg.compare_to_numpy = True
    from pace_amd.dsl.stencil import FrozenStencil

    with pytest.raises(NotImplementedError, match="compare_to_numpy"):
        FrozenStencil(_defn("pace.fv3core.stencils.d_sw", "flux_capacitor", ["cx"]), (0, 0, 0), (1, 1, 1), cfg, factory=env.stencil_factory)


def test_device_stencils_emulated():
    from pace_amd import _lib

    run_all(_lib.Library(build_emu()), "cpu")


@pytest.mark.gpu
def test_device_stencils_gpu():
    from pace_amd import _lib

    run_all(_lib.load(), "cuda")
