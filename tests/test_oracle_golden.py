"""The oracle (oracle/*.py) against fixtures produced by running the reference's own source
(tools/make_golden.py).  CPU only.  Tolerances are the reference's translate-test tolerances
 or tighter."""
import numpy as np
import pytest

from helpers import DSW_ARGS, DSW_CFG, RIEM_ARGS, column_for_levels, compare, dsw_window, expand_riem_fixture, golden, oracle_grid, window


@pytest.mark.parametrize("name,tile", [("d_sw_c12_tile0_call1.npz", 0), ("d_sw_c12_tile1_call3.npz", 1)])
def test_d_sw_oracle_matches_reference(name, tile):
    from oracle import dgrid_sw

    fix = golden(name)
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    g = oracle_grid(golden(f"grid_c12_tile{tile}.npz"), 12, nk)
    col = column_for_levels(k_sel)
    st = dgrid_sw.DSWState(fix["in_u"].shape)
    st.uc_contra[...] = fix["in_uc_contra"]
    st.vc_contra[...] = fix["in_vc_contra"]
    a = {k: fix["in_" + k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, DSW_CFG, st, *[a[k] for k in DSW_ARGS], float(fix["dt"]))
    for k in DSW_ARGS:
        if k == "zh":
            continue  # scratch after d_sw (d_sw.py:1032-1033) / untouched input
        err = compare(fix["out_" + k][dsw_window(k, 12, nk)], a[k][dsw_window(k, 12, nk)])
        assert err < 3.2e-10, (k, err)  # translate_d_sw.py:19


@pytest.mark.parametrize("name", ["riem_solver3_c12_tile0_call2.npz", "riem_solver3_c12_tile0_call3.npz"])
def test_riem_solver3_oracle_matches_reference(name):
    from oracle import vertical

    fix = golden(name)
    g = oracle_grid(golden("grid_c12_tile0.npz"), 12, 79)
    a = expand_riem_fixture(fix)
    vertical.riem_solver3(g, bool(fix["last_call"]), float(fix["dt"]), a["cappa"], float(fix["ptop"]), a["zs"], a["ws"], a["delz"],
                          a["q_con"], a["delp"], a["pt"], a["zh"], a["p"], a["ppe"], a["pk3"], a["pk"], a["log_p_interface"], a["w"],
                          p_fac=0.05)
    for k in ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w"):
        nk = 79 if k in ("delz", "w") else 80
        got = a[k][3:15, 3:7, :nk]
        err = compare(fix["out_" + k][:, :, :nk], got)
        assert err < 1e-12, (k, err)  # reference tolerance is 5e-6 (overrides/standard.yaml:49-61)


@pytest.mark.parametrize("name,nord,damp,mass", [("fvtp2d_c12_tile0_call6.npz", "nord_v", "damp_vt", False),
                                                ("fvtp2d_c12_tile0_call8.npz", "nord_t", "damp_t", True)])
def test_fvtp2d_oracle_matches_reference(name, nord, damp, mass):
    from oracle import ppm_transport

    fix = golden(name)
    k_sel = golden("d_sw_c12_tile0_call1.npz")["k_sel"]
    nk = len(k_sel)
    g = oracle_grid(golden("grid_c12_tile0.npz"), 12, nk)
    col = column_for_levels(k_sel)
    q = fix["in_q"].copy()
    fx, fy = np.zeros_like(q), np.zeros_like(q)
    kw = dict(x_mass_flux=fix["in_x_mass_flux"], y_mass_flux=fix["in_y_mass_flux"], mass=fix["in_mass"]) if mass else {}
    ppm_transport.fvtp2d(g, q, fix["in_crx"], fix["in_cry"], fix["in_x_area_flux"], fix["in_y_area_flux"], fx, fy, 6,
                         nord_k=col[nord], damp_c_k=col[damp], **kw)
    assert compare(fix["out_q_x_flux"][window(12, 1, 0, nk)], fx[window(12, 1, 0, nk)]) < 1e-14
    assert compare(fix["out_q_y_flux"][window(12, 0, 1, nk)], fy[window(12, 0, 1, nk)]) < 1e-14


def test_ppm_known_answers():
    """Hand-derivable properties of the PPM operator (xppm.py:148-181): a constant field is transported
    unchanged, and the interior interface formula is exact for cubics' cell means only up to its
    4th-order truncation -- for a linear field the interface value is the exact midpoint."""
    from oracle import ppm_transport
    from oracle._np import Grid

    n = 12
    rng = np.random.default_rng(0)
    metrics = dict(golden("grid_c12_tile0.npz"))
    g = Grid(n, 4, metrics)
    shape = (n + 7, n + 7, 5)
    c = rng.random(shape) - 0.5
    out = np.zeros(shape)
    ppm_transport.ppm_flux(np.full(shape, 3.25), c, g.dxa, g, 0, 6, out, (3, 3), (n + 1, n))
    assert np.all(out[3 : 3 + n + 1, 3 : 3 + n, :4] == 3.25)
    lin = np.arange(n + 7, dtype=float).reshape(-1, 1, 1) * np.ones(shape)
    al = ppm_transport.compute_al(lin, np.ones((n + 7, n + 7, 1)), g, 0)
    np.testing.assert_allclose(al[5 : n, 3 : 3 + n, 0], lin[5:n, 3 : 3 + n, 0] - 0.5, rtol=0, atol=1e-13)


def test_tridiagonal_solver_against_scipy():
    """sim1_solver's first system (sim1_solver.py:60-88) is the tridiagonal system
    [1+.. g_rat] pp = dd; check our restatement against scipy.linalg.solve_banded on one column."""
    import scipy.linalg

    from oracle import vertical

    fix = golden("riem_solver3_c12_tile0_call2.npz")
    g = oracle_grid(golden("grid_c12_tile0.npz"), 12, 79)
    a = expand_riem_fixture(fix)
    before_w = a["w"].copy()
    vertical.riem_solver3(g, False, float(fix["dt"]), a["cappa"], float(fix["ptop"]), a["zs"], a["ws"], a["delz"], a["q_con"],
                          a["delp"], a["pt"], a["zh"], a["p"], a["ppe"], a["pk3"], a["pk"], a["log_p_interface"], a["w"], p_fac=0.05)
    # independent check of the w system: rebuild it from the solver's own outputs is circular, so check
    # instead the discrete identity pe[k] - pe[k-1] = dm[k-1] (w - w_old)[k-1] / dt  (sim1_solver.py:118-126)
    dm = fix["in_delp"][:, :, :79] / 9.80665
    dpe = np.diff(a["ppe"][3:15, 3:7, :80], axis=2)
    rhs = dm * (a["w"][3:15, 3:7, :79] - before_w[3:15, 3:7, :79]) / float(fix["dt"])
    np.testing.assert_allclose(dpe, rhs, rtol=1e-9, atol=1e-9)
    # and the banded-solve sanity on a synthetic diagonally dominant system of the same shape
    n = 79
    lo, di, up = np.full(n, 1.0), np.full(n, 4.0), np.full(n, 1.0)
    rhs1 = np.arange(n, dtype=float)
    ab = np.zeros((3, n))
    ab[0, 1:], ab[1], ab[2, :-1] = up[:-1], di, lo[1:]
    x = scipy.linalg.solve_banded((1, 1), ab, rhs1)
    bet, gam, y = di[0], np.zeros(n), np.zeros(n)
    y[0] = rhs1[0] / bet
    for k in range(1, n):
        gam[k] = up[k - 1] / bet
        bet = di[k] - lo[k] * gam[k]
        y[k] = (rhs1[k] - lo[k] * y[k - 1]) / bet
    for k in range(n - 2, -1, -1):
        y[k] -= gam[k + 1] * y[k + 1]
    np.testing.assert_allclose(x, y, rtol=1e-12)


def test_oracle_acoustic_dynamics_against_reference_run():
    """The whole oracle (c_sw, d2a2c_vect, updatedzc, riem_solver_c, p_grad_c, d_sw, updatedzd, riem_solver3, pe/pk3 halo,
    nh_p_grad, ray_fast, del2cubed, heating + every halo-exchange flavour) composed into the acoustic loop of
    dyn_core.py:670-970 for all six tiles, against the reference run's output (tests/golden/acoustic_c12_tile*.npz)."""
    from helpers import ACOUSTIC_OUT, DSW_CFG, acoustic_errors, acoustic_fixture, golden, oracle_grid

    from oracle import dyn_core

    n, nz = 12, 79
    fixes = [acoustic_fixture(t) for t in range(6)]
    grids = [oracle_grid({k[5:]: v for k, v in fx.items() if k.startswith("grid_")}, n, nz) for fx in fixes]
    states = [{k[3:]: v.copy() for k, v in fx.items() if k.startswith("in_") and k != "in_cappa"} for fx in fixes]
    cappas = [fx["in_cappa"].copy() for fx in fixes]
    col = {k: v for k, v in golden("column_namelist_c12.npz").items()}
    cfg = dict(DSW_CFG, p_fac=0.05, rf_cutoff=3000.0, tau=10.0, delt_max=0.002, hord_tm=6)
    tmp = dyn_core.acoustic_dynamics(grids, col, cfg, states, cappas, float(fixes[0]["timestep"]), int(fixes[0]["n_split"]), n, nz)
    for t in range(6):
        out = dict(states[t])
        out["heat_source"] = tmp[t].heat_source
        for k, e in acoustic_errors(fixes[t], out).items():
            assert e == 0.0, (t, k, e)  # bit-for-bit


def _tracer_case():
    from helpers import golden, oracle_grid

    n = 12
    fixes = [golden(f"tracer_c12_tile{t}.npz") for t in range(6)]
    nk = len(fixes[0]["k_sel"])
    grids = [oracle_grid({k[5:]: v for k, v in golden(f"acoustic_c12_tile{t}.npz").items() if k.startswith("grid_")}, n, nk)
             for t in range(6)]
    return n, nk, fixes, grids


def test_oracle_tracer_advection_against_reference_run():
    """oracle/tracer.py (flux_compute, sub-cycling, ord-8 monotone PPM transport, tracer halo updates) for all six tiles
    against the reference's TracerAdvection run (tools/make_golden_tracer.py): bit for bit."""
    from oracle import tracer

    n, nk, fixes, grids = _tracer_case()
    f = lambda k: [fx["in_" + k].copy() for fx in fixes]  # noqa: E731
    tracers = {"qvapor": f("qvapor"), "q2": f("q2")}
    dp1, mfx, mfy, cx, cy = f("dp1"), f("mfxd"), f("mfyd"), f("cxd"), f("cyd")
    tracer.tracer_advection(grids, tracers, dp1, mfx, mfy, cx, cy, n, nk)
    W = (slice(3, 3 + n), slice(3, 3 + n), slice(0, nk))
    for t in range(6):
        for name, q in tracers.items():
            assert np.array_equal(q[t][W], fixes[t]["out_" + name][W]), (t, name)
        assert np.array_equal(mfx[t][3 : 4 + n, 3 : 3 + n, :nk], fixes[t]["out_mfxd"][3 : 4 + n, 3 : 3 + n, :nk])
        assert np.array_equal(cy[t][3 : 3 + n, 3 : 4 + n, :nk], fixes[t]["out_cyd"][3 : 3 + n, 3 : 4 + n, :nk])


@pytest.mark.parametrize("name", sorted(__import__("helpers").REMAP_CASES))
def test_map_single_oracle_against_reference_run(name):
    """oracle/remapping.py reproduces the reference's MapSingle (RemapProfile + lagrangian_contributions) bit for bit: pt
    in log-pressure with qmin, a tracer, w with its bottom value, delz, u on its staggered window; kord 9 and 10; the
    weakly deformed coordinate of a real acoustic call and a strongly deformed one (several source layers per target)."""
    from helpers import REMAP_CASES, REMAP_KM

    from oracle import remapping

    d = golden("remap_c12.npz")
    kord, iv, src, dst, qs, qmin = REMAP_CASES[name]
    q = d[name + "_in"].copy()
    remapping.map_single(q, d[src], d[dst], REMAP_KM, kord, iv, qs=None if qs is None else d[qs], qmin=qmin)
    ref = d[name + "_out"]
    assert np.array_equal(q[:, :, :REMAP_KM], ref[:, :, :REMAP_KM])
    assert np.abs(ref[:, :, :REMAP_KM] - d[name + "_in"][:, :, :REMAP_KM]).max() > 0  # the fixture does remap something


@pytest.mark.parametrize("t", [0, 1, 2])
def test_fillz_oracle_against_reference_run(t):
    """oracle fillz == the reference's FillNegativeTracerValues on tracers with sprinkled / dense negative masses."""
    from helpers import REMAP_KM

    from oracle import remapping

    d = golden("remap_c12.npz")
    q = d[f"fillz{t}_in"].copy()
    remapping.fillz(q, d["fillz_dp"], REMAP_KM)
    assert np.array_equal(q[:, :, :REMAP_KM], d[f"fillz{t}_out"][:, :, :REMAP_KM])
    assert (d[f"fillz{t}_in"][:, :, :REMAP_KM] != q[:, :, :REMAP_KM]).mean() > 0.3


def _l2e_inputs():
    d = golden("l2e_c12.npz")
    f = {k[3:]: d[k].copy() for k in d if k.startswith("in_") and not k.startswith("in_tr_")}
    tr = {k[6:]: d[k].copy() for k in d if k.startswith("in_tr_")}
    return d, f, tr


@pytest.mark.parametrize("last_step", [False, True])
def test_lagrangian_to_eulerian_oracle_against_reference_run(last_step):
    """oracle.remapping.lagrangian_to_eulerian == the reference's LagrangianToEulerian (do_sat_adj = False) on tile 0 after
    an acoustic call, every output, bit for bit (fixture window: compute domain + 1 halo cell, origin 1)."""
    from oracle import constants as c
    from oracle import remapping

    d, f, tr = _l2e_inputs()
    km, n = 79, 12
    remapping.lagrangian_to_eulerian(f, tr, d["ak"], d["bk"], float(d["ptop"]), c.KAPPA, c.ZVIR, last_step, n, km, o=1, nq=8)
    cw = (slice(1, 13), slice(1, 13))
    if last_step:
        assert np.array_equal(f["pt"][cw][:, :, :km], d["out_last_pt"][cw][:, :, :km])
        return
    for key in d:
        if not key.startswith("out_") or key == "out_last_pt":
            continue
        name = key[4:]
        got = tr[name[3:]] if name.startswith("tr_") else f[name]
        win = {"u": (slice(1, 13), slice(1, 14)), "v": (slice(1, 14), slice(1, 13))}.get(name, cw)
        if d[key].ndim == 2:
            assert np.array_equal(got[cw], d[key][cw]), name
        else:
            kk = km + 1 if name in ("pe", "peln", "pk") else km
            assert np.array_equal(got[win][:, :, :kk], d[key][win][:, :, :kk]), name


def test_neg_adj3_oracle_against_reference_run():
    from oracle import dycore_parts

    d = golden("negadj_c12.npz")
    names = ["qvapor", "qliquid", "qrain", "qsnow", "qice", "qgraupel", "qcld"]
    f = {k: d["in_" + k].copy() for k in names + ["pt", "delp"]}
    dycore_parts.neg_adj3(*[f[k] for k in names], f["pt"], f["delp"], 79)
    for k in names + ["pt"]:
        assert np.array_equal(f[k][:, :, :79], d["out_" + k][:, :, :79]), k
    assert np.abs(d["out_pt"] - d["in_pt"]).max() > 1.0


def test_d_sw_order5_oracle_matches_reference():
    """The oracle's d_sw with every advection order 5 against the reference's run of that namelist: bit for bit."""
    from oracle import dgrid_sw

    fix = golden("d_sw_h5_c12_tile0_call1.npz")
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    cfg = dict(DSW_CFG, hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5)
    col = {k[4:]: np.ascontiguousarray(v[np.asarray(k_sel)]) for k, v in fix.items() if k.startswith("col_")}
    g = oracle_grid(golden("grid_c12_tile0.npz"), 12, nk)
    st = dgrid_sw.DSWState(fix["in_u"].shape)
    st.uc_contra[:] = fix["in_uc_contra"]
    st.vc_contra[:] = fix["in_vc_contra"]
    a = {k: fix["in_" + k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, cfg, st, *[a[k] for k in DSW_ARGS], float(fix["dt"]))
    for k in DSW_ARGS:
        if k != "zh":
            assert compare(fix["out_" + k][dsw_window(k, 12, nk)], a[k][dsw_window(k, 12, nk)]) == 0.0, k


@pytest.mark.parametrize("variant", ["nord2", "dcon0", "skeb", "dddmp0"])
def test_d_sw_namelist_variants_oracle_matches_reference(variant):
    """The oracle's d_sw against runs of the reference with one namelist option changed (two damping passes instead of three;
    no dissipative heating; the dissipation estimate kept; no Smagorinsky term): bit for bit."""
    from helpers import dsw_variant_fixture

    from oracle import dgrid_sw

    fix, cfg, col, nk = dsw_variant_fixture(variant)
    g = oracle_grid(golden("grid_c12_tile0.npz"), 12, nk)
    st = dgrid_sw.DSWState(fix["in_u"].shape)
    st.uc_contra[:] = fix["in_uc_contra"]
    st.vc_contra[:] = fix["in_vc_contra"]
    a = {k: fix["in_" + k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, cfg, st, *[a[k] for k in DSW_ARGS], float(fix["dt"]))
    for k in DSW_ARGS:
        if k != "zh":
            assert compare(fix["out_" + k][dsw_window(k, 12, nk)], a[k][dsw_window(k, 12, nk)]) == 0.0, k


def test_oracle_acoustic_dynamics_variant_against_reference_run():
    """The whole oracle loop with several namelist options changed at once (nord = 2 -- in c_sw's corner divergence, the
    divergence damping, every del-n damping --, d_con = 0, all advection orders 5) against the reference's run of that namelist
    on the same inputs (tools/make_golden_acoustic.py v2): bit for bit."""
    from helpers import ACOUSTIC_VARIANTS, DSW_CFG, acoustic_errors, acoustic_variant_fixture, golden, oracle_grid

    from oracle import dyn_core

    n, nz = 12, 79
    fixes = [acoustic_variant_fixture(t, "v2") for t in range(6)]
    var = golden("acoustic_c12_v2.npz")
    grids = [oracle_grid({k[5:]: v for k, v in fx.items() if k.startswith("grid_")}, n, nz) for fx in fixes]
    states = [{k[3:]: v.copy() for k, v in fx.items() if k.startswith("in_") and k != "in_cappa"} for fx in fixes]
    cappas = [fx["in_cappa"].copy() for fx in fixes]
    col = {k[9:]: v for k, v in var.items() if k.startswith("namelist_")}
    opts = ACOUSTIC_VARIANTS["v2"]
    cfg = dict(DSW_CFG, p_fac=0.05, rf_cutoff=3000.0, tau=10.0, delt_max=0.002, **opts)
    tmp = dyn_core.acoustic_dynamics(grids, col, cfg, states, cappas, float(fixes[0]["timestep"]), int(fixes[0]["n_split"]), n, nz)
    for t in range(6):
        out = dict(states[t])
        out["heat_source"] = tmp[t].heat_source
        for k, e in acoustic_errors(fixes[t], out).items():
            assert e == 0.0, (t, k, e)


def test_lagrangian_to_eulerian_order_10_oracle_against_reference_run():
    """oracle.remapping.lagrangian_to_eulerian with every remapping order 10 == the reference's run with that namelist
    (tools/make_golden_l2e.py kord10; negatives in four condensates), every output, bit for bit."""
    from helpers import l2e_k10_fixture

    from oracle import constants as c
    from oracle import remapping

    d = l2e_k10_fixture()
    f = {k[3:]: d[k].copy() for k in d if k.startswith("in_") and not k.startswith("in_tr_")}
    tr = {k[6:]: d[k].copy() for k in d if k.startswith("in_tr_")}
    km, n = 79, 12
    remapping.lagrangian_to_eulerian(f, tr, d["ak"], d["bk"], float(d["ptop"]), c.KAPPA, c.ZVIR, False, n, km, o=1, nq=8,
                                     kord_tm=-10, kord_tr=10, kord_wz=10, kord_mt=10)
    cw = (slice(1, 13), slice(1, 13))
    for key in d:
        if not key.startswith("out_") or key == "out_last_pt":
            continue
        name = key[4:]
        got = tr[name[3:]] if name.startswith("tr_") else f[name]
        win = {"u": (slice(1, 13), slice(1, 14)), "v": (slice(1, 14), slice(1, 13))}.get(name, cw)
        if d[key].ndim == 2:
            assert np.array_equal(got[cw], d[key][cw]), name
        else:
            kk = km + 1 if name in ("pe", "peln", "pk") else km
            assert np.array_equal(got[win][:, :, :kk], d[key][win][:, :, :kk]), name


def test_loop_conditioning_of_diss_estd():
    """Why helpers.ACOUSTIC_TOL holds diss_estd to 3e-6 and not to the reference's DynCore bound of 2e-6 (translate_dyncore.py:120):
    the ORACLE ITSELF, on the reference run's inputs, moves by more than 1e-6 in diss_estd when nothing but the ORDER of two
    additions per level changes -- the column sums of the interface pressures (riem_solver3.py:63-81) accumulated in extended
    precision and rounded once per level, which leaves every pressure within an ulp of the sequential double sum.  Any column
    solver that does not walk the 79 levels one after the other (the device's scans over sixteen lanes) re-associates exactly
    these sums.  Everything else stays far inside 2e-6 (w 1.7e-7, omga 5.6e-8, vc 8e-8)."""
    import inspect

    from helpers import DSW_CFG, acoustic_errors, acoustic_fixture, golden, oracle_grid

    import oracle.vertical as V
    from oracle import dyn_core

    src = inspect.getsource(V.riem_solver3)
    old_g = "            pg[W + (k,)] = pg[W + (k - 1,)] + dm[W + (k - 1,)] * (1.0 - q_con[W + (k - 1,)])\n"
    old_p = "            p_int[W + (k,)] = p_int[W + (k - 1,)] + dm[W + (k - 1,)]\n"
    loop = "        for k in range(1, K):\n"
    assert old_g in src and old_p in src and loop in src
    src = src.replace(old_g, "            _g = _g + (dm[W + (k - 1,)] * (1.0 - q_con[W + (k - 1,)])).astype(np.longdouble)\n"
                             "            pg[W + (k,)] = _g.astype(np.float64)\n")
    src = src.replace(old_p, "            _p = _p + dm[W + (k - 1,)].astype(np.longdouble)\n            p_int[W + (k,)] = _p.astype(np.float64)\n")
    src = src.replace(loop, "        _g = np.full(pg[W + (0,)].shape, ptop, dtype=np.longdouble)\n        _p = _g.copy()\n" + loop, 1)
    original = V.riem_solver3
    try:
        exec(src, V.__dict__)  # (oracle.dyn_core calls vertical.riem_solver3 through the module)
        n, nz = 12, 79
        fixes = [acoustic_fixture(t) for t in range(6)]
        grids = [oracle_grid({k[5:]: v for k, v in fx.items() if k.startswith("grid_")}, n, nz) for fx in fixes]
        states = [{k[3:]: v.copy() for k, v in fx.items() if k.startswith("in_") and k != "in_cappa"} for fx in fixes]
        cappas = [fx["in_cappa"].copy() for fx in fixes]
        col = {k: v for k, v in golden("column_namelist_c12.npz").items()}
        cfg = dict(DSW_CFG, p_fac=0.05, rf_cutoff=3000.0, tau=10.0, delt_max=0.002, hord_tm=6)
        tmp = dyn_core.acoustic_dynamics(grids, col, cfg, states, cappas, float(fixes[0]["timestep"]), int(fixes[0]["n_split"]), n, nz)
    finally:
        V.riem_solver3 = original
    worst = {}
    for t in range(6):
        out = dict(states[t])
        out["heat_source"] = tmp[t].heat_source
        for k, e in acoustic_errors(fixes[t], out).items():
            worst[k] = max(worst.get(k, 0.0), e)
    assert worst["diss_estd"] > 1e-6, worst  # the oracle against itself: at the reference's bound already
    assert all(e < 2e-6 for k, e in worst.items() if k != "diss_estd") and worst["delp"] < 1e-15, worst
