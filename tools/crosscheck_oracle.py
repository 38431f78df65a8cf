"""Cross-check the hand-written oracle (oracle/*.py) against the reference's own stencil source
executed by tools/gtinterp, on random inputs.  Dev-container only (reads /root/reference).

    python tools/crosscheck_oracle.py [group ...]
"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

import refenv  # noqa: E402

from oracle._np import Grid  # noqa: E402

N, NZ = 12, 79
_envs = None


def env():
    global _envs
    if _envs is None:
        _envs = refenv.build_all(N, NZ, with_state=False)
    return _envs[0]


def oracle_grid(e, nk=NZ):
    m = {}
    for name in dir(e.grid_data):
        if name.startswith("_"):
            continue
        try:
            v = getattr(e.grid_data, name)
        except Exception:  # noqa: BLE001
            continue
        if hasattr(v, "data") and hasattr(v, "dims"):
            m[name] = np.array(v.data)
        elif isinstance(v, (float, int)):
            m[name] = v
    for name in ["del6_u", "del6_v", "divg_u", "divg_v"]:
        m[name] = np.array(getattr(e.damping, name).data)
    m["da_min"] = e.damping.da_min
    m["da_min_c"] = e.damping.da_min_c
    return Grid(N, nk, m)


def rand(rng, shape=(N + 7, N + 7, NZ + 1)):
    return rng.random(shape)


def report(name, a, b, window=None, tol=1e-13):
    if window is not None:
        a, b = a[window], b[window]
    both_nan = np.isnan(a) & np.isnan(b)
    diff = np.abs(a - b)
    diff[both_nan] = 0
    bad = np.isnan(diff)
    denom = np.abs(a) + np.abs(b)
    rel = np.where(denom > 0, 2 * diff / np.where(denom > 0, denom, 1), 0)
    rel[bad] = np.inf
    mx = rel.max() if rel.size else 0.0
    status = "ok " if mx < tol else "FAIL"
    print(f"  [{status}] {name}: max rel {mx:.3e} (nan-mismatch {int(bad.sum())})")
    return mx < tol


def check_corners():
    from pace.stencils import corners as rc

    from oracle import corner_ops as oc

    e = env()
    g = oracle_grid(e)
    sf = e.stencil_factory
    rng = np.random.default_rng(0)
    ok = True
    for d in "xy":
        q = rand(rng)
        ref = q.copy()
        rc.CopyCorners(d, sf)(ref)
        mine = q.copy()
        oc.copy_corners(mine, g, d)
        ok &= report(f"copy_corners_{d}", ref[:, :, :NZ], mine[:, :, :NZ])
        q = rand(rng)
        ref = q.copy()
        rc.FillCornersBGrid(d, stencil_factory=sf)(ref)
        mine = q.copy()
        oc.fill_corners_bgrid(mine, g, d)
        ok &= report(f"fill_corners_bgrid_{d}", ref[:, :, :NZ], mine[:, :, :NZ])
    from pace.util import X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM

    st = sf.from_dims_halo(func=rc.fill_corners_dgrid_defn, compute_dims=[X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM],
                           compute_halos=(3, 3))
    x, y = rand(rng), rand(rng)
    rx, ry = x.copy(), y.copy()
    st(rx, rx, ry, ry, -1.0)
    mx, my = x.copy(), y.copy()
    oc.fill_corners_dgrid(mx, my, g, -1.0)
    ok &= report("fill_corners_dgrid x", rx[:, :, :NZ], mx[:, :, :NZ])
    ok &= report("fill_corners_dgrid y", ry[:, :, :NZ], my[:, :, :NZ])
    return ok


def _W(g, di=0, dj=0, nk=NZ):
    return (slice(g.is_, g.ie + 1 + di), slice(g.js, g.je + 1 + dj), slice(0, nk))


def _quant(e, arr):
    q = e.qf.zeros(["x", "y", "z"], "")
    q.data[:] = arr
    return q


def check_transport():
    from pace.fv3core.stencils import delnflux as rdel
    from pace.fv3core.stencils import fvtp2d as rf
    from pace.fv3core.stencils import xppm as rx
    from pace.fv3core.stencils import yppm as ry

    from oracle import ppm_transport as op

    e = env()
    g = oracle_grid(e)
    sf, gi = e.stencil_factory, e.grid_indexing
    rng = np.random.default_rng(1)
    ok = True
    for ord_ in (5, 6, 8):
        for axis, cls, dxa in ((0, rx.XPiecewiseParabolic, e.grid_data.dxa), (1, ry.YPiecewiseParabolic, e.grid_data.dya)):
            q, c = rand(rng), rand(rng) - 0.5
            origin = gi.origin_compute(add=(0, -3, 0)) if axis == 0 else gi.origin_compute(add=(-3, 0, 0))
            domain = gi.domain_compute(add=(1, 7, 1)) if axis == 0 else gi.domain_compute(add=(7, 1, 1))
            ref = np.zeros_like(q)
            cls(sf, dxa, 0, ord_, origin, domain)(q, c, ref)
            mine = np.zeros_like(q)
            op.ppm_flux(q, c, g.dxa if axis == 0 else g.dya, g, axis, ord_, mine, origin[:2], domain[:2])
            w = (slice(3, 3 + N + 1), slice(0, N + 6), slice(0, NZ)) if axis == 0 else (slice(0, N + 6), slice(3, 3 + N + 1), slice(0, NZ))
            ok &= report(f"ppm axis={axis} ord={ord_}", ref, mine, w)
    # column parameters as in d_sw.get_column_namelist for the baroclinic config
    nord = np.full(NZ + 1, 2.0); nord[:2] = 0
    damp = np.full(NZ + 1, 0.06); damp[0] = 0.1; damp[1] = 0.05
    def kq(a):
        q = e.qf.zeros(["z"], "")
        q.data[:] = a
        return q
    for hord, use_mass, with_damp in ((6, False, False), (6, False, True), (6, True, True), (8, True, False), (5, False, True)):
        q = rand(rng) + 0.5
        crx, cry = rand(rng) - 0.5, rand(rng) - 0.5
        xfx = crx * g.area[:, :, None] * 0.5
        yfx = cry * g.area[:, :, None] * 0.5
        mfx, mfy, mass = rand(rng), rand(rng), rand(rng) + 1
        kw = dict(nord=kq(nord), damp_c=kq(damp)) if with_damp else {}
        obj = rf.FiniteVolumeTransport(sf, e.qf, e.grid_data, e.damping, 0, hord, **kw)
        rq, rfx, rfy = q.copy(), np.zeros_like(q), np.zeros_like(q)
        obj(rq, crx, cry, xfx, yfx, rfx, rfy, **(dict(x_mass_flux=mfx, y_mass_flux=mfy, mass=mass) if use_mass else {}))
        mq, mfx_o, mfy_o = q.copy(), np.zeros_like(q), np.zeros_like(q)
        op.fvtp2d(g, mq, crx, cry, xfx, yfx, mfx_o, mfy_o, hord, **(dict(x_mass_flux=mfx, y_mass_flux=mfy, mass=mass) if use_mass else {}),
                  **(dict(nord_k=nord, damp_c_k=damp) if with_damp else {}))
        tag = f"fvtp2d hord={hord} mass={use_mass} damp={with_damp}"
        ok &= report(tag + " fx", rfx, mfx_o, _W(g, 1, 0))
        ok &= report(tag + " fy", rfy, mfy_o, _W(g, 0, 1))
        ok &= report(tag + " q(corners)", rq[:, :, :NZ], mq[:, :, :NZ])
    for nord_col in ([0, 0, 2, 2], [0, 0, 0, 2], [2, 2, 2, 2]):
        nk_arr = np.full(NZ + 1, float(nord_col[3])); nk_arr[:3] = nord_col[:3]
        obj = rdel.DelnFluxNoSG(sf, e.damping, e.grid_data.rarea, kq(nk_arr))
        q = rand(rng)
        dk = rand(rng, (NZ + 1,))
        rfx, rfy, rd2 = np.zeros_like(q), np.zeros_like(q), np.zeros_like(q)
        obj(q, rfx, rfy, dk, rd2)
        mfx_o, mfy_o, md2 = np.zeros_like(q), np.zeros_like(q), np.zeros_like(q)
        op.delnflux_nosg(g, q, mfx_o, mfy_o, dk, md2, nk_arr)
        ok &= report(f"delnflux_nosg nord={nord_col} fx", rfx, mfx_o, _W(g, 1, 0))
        ok &= report(f"delnflux_nosg nord={nord_col} fy", rfy, mfy_o, _W(g, 0, 1))
        ok &= report(f"delnflux_nosg nord={nord_col} d2", rd2, md2, _W(g, 0, 0))
    return ok


def column_namelist():
    """Values of d_sw.get_column_namelist for the baroclinic config (d_sw.py:633-683)."""
    col = {}
    z = NZ + 1
    col["nord"] = np.full(z, 3.0); col["nord"][:3] = 0
    col["nord_v"] = np.full(z, 2.0); col["nord_v"][:2] = 0
    col["nord_w"] = np.full(z, 2.0); col["nord_w"][:3] = 0
    col["nord_t"] = np.full(z, 2.0)
    col["damp_vt"] = np.full(z, 0.06); col["damp_vt"][:2] = [0.1, 0.05]
    col["damp_w"] = np.full(z, 0.06); col["damp_w"][:3] = [0.2, 0.1, 0.02]
    col["damp_t"] = np.full(z, 0.06)
    col["d2_divg"] = np.zeros(z); col["d2_divg"][:3] = [0.2, 0.1, 0.02]
    col["d_con"] = np.full(z, 1.0); col["d_con"][:3] = 0
    col["ke_bg"] = np.zeros(z)
    return col


def check_damping():
    from pace.fv3core.stencils import a2b_ord4 as ra
    from pace.fv3core.stencils import divergence_damping as rd

    from oracle import damping as od

    e = env()
    g = oracle_grid(e)
    sf = e.stencil_factory
    rng = np.random.default_rng(2)
    ok = True
    obj = ra.AGrid2BGridFourthOrder(sf, e.qf, e.grid_data, 0, replace=False)
    qin = rand(rng)
    rq, ro = qin.copy(), np.zeros_like(qin)
    obj(rq, ro)
    mq, mo = qin.copy(), np.zeros_like(qin)
    od.a2b_ord4(g, mq, mo)
    ok &= report("a2b_ord4 qout", ro, mo, _W(g, 1, 1))
    obj = ra.AGrid2BGridFourthOrder(sf.restrict_vertical(k_start=1), e.qf, e.grid_data, 0, replace=True)
    rq, ro = qin.copy(), np.zeros_like(qin)
    obj(rq, ro)
    mq, mo = qin.copy(), np.zeros_like(qin)
    od.a2b_ord4(g, mq, mo, k0=1, replace=True)
    ok &= report("a2b_ord4(k>=1, replace) qout", ro, mo, _W(g, 1, 1))
    ok &= report("a2b_ord4(k>=1, replace) qin", rq[:, :, :NZ], mq[:, :, :NZ])
    col = column_namelist()

    def kq(a):
        q = e.qf.zeros(["z"], "")
        q.data[:] = a
        return q

    obj = rd.DivergenceDamping(sf, e.qf, e.grid_data, e.damping, False, False, 0.5, 0.15, 3, 0, kq(col["nord"]), kq(col["d2_divg"]))
    names = ["u", "v", "va", "vort_b", "ua", "divg_d", "vc", "uc", "delpc", "ke", "wk"]
    base = {nm: rand(rng) - 0.5 for nm in names}
    base["divg_d"] *= 1e-6
    base["wk"] *= 1e-5
    ref = {k: v.copy() for k, v in base.items()}
    obj(*[ref[nm] for nm in names], 10.0)
    mine = {k: v.copy() for k, v in base.items()}
    od.divergence_damping(g, *[mine[nm] for nm in names], 10.0, nord_k=col["nord"], d2_bg_k=col["d2_divg"], dddmp=0.5, d4_bg=0.15, nord=3)
    for nm in ("vort_b", "ke", "delpc"):
        ok &= report(f"divergence_damping {nm}", ref[nm], mine[nm], _W(g, 1, 1))
    for nm in ("divg_d", "uc", "vc"):
        ok &= report(f"divergence_damping {nm} (scratch, full)", ref[nm][:, :, :NZ], mine[nm][:, :, :NZ])
    return ok


_capture = None


def captured():
    """Reference AcousticDynamics run (2 calls x n_split=2) with per-component in/out records.
    Cached in /tmp/ref_capture.pkl (produced by tools/make_golden.py --cache)."""
    global _capture
    if _capture is None:
        import pickle

        path = "/tmp/ref_capture.pkl"
        if not os.path.exists(path):
            import make_golden

            make_golden.build_capture(path)
        _capture = pickle.load(open(path, "rb"))
    return _capture


def grid_from_capture(cap, rank=0, nk=NZ):
    gm = dict(cap[f"grid{rank}"])
    return Grid(N, nk, gm)


DSW_CFG = dict(hord_dp=6, hord_tm=6, hord_vt=6, hord_mt=6, dddmp=0.5, d4_bg=0.15, nord=3, d_con=1.0, do_skeb=False)
DSW_ARGS = ("delpc delp pt u v w uc vc ua va divgd mfx mfy cx cy crx cry xfx yfx q_con zh heat_source diss_est").split()


def check_dsw():
    from oracle import dgrid_sw as od

    cap = captured()
    ok = True
    col = column_namelist()
    for rank in (0, 1):
        g = grid_from_capture(cap, rank)
        recs = cap["records"][(f"rank{rank}", "FiniteVolumeFluxPrep")]
        st = od.DSWState(recs[0]["in"]["uc"].shape)
        names = ["uc", "vc", "crx", "cry", "x_area_flux", "y_area_flux", "uc_contra", "vc_contra"]
        for n_, r in enumerate(recs[:2]):
            a = {k: r["in"][k].copy() for k in names}
            od.fxadv(g, *[a[k] for k in names], r["in"]["dt"])
            for k in names[2:]:
                ok &= report(f"rank{rank} fxadv[{n_}] {k}", r["out"][k][:, :, :NZ], a[k][:, :, :NZ])
        recs = cap["records"][(f"rank{rank}", "DGridShallowWaterLagrangianDynamics")]
        st = od.DSWState(recs[0]["in"]["u"].shape)
        for n_, r in enumerate(recs):
            a = {k: r["in"][k].copy() for k in DSW_ARGS}
            od.d_sw(g, col, DSW_CFG, st, *[a[k] for k in DSW_ARGS], r["in"]["dt"])
            for k in DSW_ARGS:
                if k in ("delpc", "divgd", "uc", "vc"):
                    continue  # scratch after d_sw (d_sw.py:1032-1033); checked inside check_damping
                # windows of translate_d_sw.py:36-65 (x-interface / y-interface / centre variables)
                di = 1 if k in ("mfx", "cx", "crx", "xfx", "v") else 0
                dj = 1 if k in ("mfy", "cy", "cry", "yfx", "u") else 0
                ok &= report(f"rank{rank} d_sw[{n_}] {k}", r["out"][k], a[k], _W(g, di, dj), tol=1e-11)
    return ok


def check_riem3():
    from oracle import vertical as ov

    cap = captured()
    ok = True
    names = "cappa zs ws delz q_con delp pt zh p ppe pk3 pk log_p_interface w".split()
    for rank in (0, 1):
        g = grid_from_capture(cap, rank)
        for n_, r in enumerate(cap["records"][(f"rank{rank}", "NonhydrostaticVerticalSolver")]):
            a = {k: r["in"][k].copy() for k in names}
            ov.riem_solver3(g, r["in"]["last_call"], r["in"]["dt"], a["cappa"], r["in"]["ptop"], a["zs"], a["ws"], a["delz"],
                            a["q_con"], a["delp"], a["pt"], a["zh"], a["p"], a["ppe"], a["pk3"], a["pk"], a["log_p_interface"],
                            a["w"], p_fac=0.05)
            for k in ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w"):
                nk = NZ if k in ("delz", "w") else NZ + 1
                ok &= report(f"rank{rank} riem_solver3[{n_}] last={r['in']['last_call']} {k}", r["out"][k], a[k], _W(g, 0, 0, nk), tol=1e-12)
    return ok


CSW_ARGS = "delp pt u v w uc vc ua va ut vt divgd omga".split()


def check_csw():
    from oracle import cgrid_sw as oc

    cap = captured()
    ok = True
    for rank in (0, 1):
        g = grid_from_capture(cap, rank)
        recs = cap["records"][(f"rank{rank}", "DGrid2AGrid2CGridVectors")]
        st = oc.D2A2CState(recs[0]["in"]["u"].shape)
        names = ["uc", "vc", "u", "v", "ua", "va", "utc", "vtc"]
        for n_, r in enumerate(recs[:2]):
            a = {k: r["in"][k].copy() for k in names}
            oc.d2a2c_vect(g, st, *[a[k] for k in names])
            for k in ("uc", "vc", "ua", "va", "utc", "vtc"):
                ok &= report(f"rank{rank} d2a2c[{n_}] {k}", r["out"][k][:, :, :NZ], a[k][:, :, :NZ], tol=1e-12)
        recs = cap["records"][(f"rank{rank}", "CGridShallowWaterDynamics")]
        st = oc.CSWState(recs[0]["in"]["u"].shape)
        for n_, r in enumerate(recs):
            a = {k: r["in"][k].copy() for k in CSW_ARGS}
            oc.c_sw(g, st, *[a[k] for k in CSW_ARGS], r["in"]["dt2"])
            for k in CSW_ARGS:
                ok &= report(f"rank{rank} c_sw[{n_}] {k}", r["out"][k][:, :, :NZ], a[k][:, :, :NZ], tol=1e-11)
    return ok


def _rec(cap, rank, name):
    return cap["records"][(f"rank{rank}", name)]


def check_parts():
    from oracle import acoustic_parts as oa
    from oracle import vertical as ov

    cap = captured()
    ok = True
    col = column_namelist()
    for rank in (0, 1):
        g = grid_from_capture(cap, rank)
        gm = cap[f"grid{rank}"]
        full = (slice(None), slice(None), slice(0, NZ + 1))
        for n_, r in enumerate(_rec(cap, rank, "NonhydrostaticVerticalSolverCGrid")):
            i_ = r["in"]
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in i_.items()}
            ov.riem_solver_c(g, a["dt2"], a["cappa"], a["ptop"], a["hs"], a["ws"], a["ptc"], a["q_con"], a["delpc"], a["gz"], a["pef"],
                             a["w3"], p_fac=0.05)
            w = (slice(2, N + 5), slice(2, N + 5), slice(0, NZ + 1))
            for k in ("gz", "pef"):
                ok &= report(f"rank{rank} riem_solver_c[{n_}] {k}", r["out"][k], a[k], w, tol=1e-12)
        for n_, r in enumerate(_rec(cap, rank, "UpdateGeopotentialHeightOnCGrid")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            ov.update_dz_c(g, gm["dp_ref"], a["zs"], a["ut"], a["vt"], a["gz"], a["ws"], a["dt"])
            w = (slice(2, N + 5), slice(2, N + 5), slice(0, NZ + 1))
            ok &= report(f"rank{rank} updatedzc[{n_}] gz", r["out"]["gz"], a["gz"], w, tol=1e-13)
            ok &= report(f"rank{rank} updatedzc[{n_}] ws", r["out"]["ws"], a["ws"], w[:2], tol=1e-13)
        for n_, r in enumerate(_rec(cap, rank, "UpdateHeightOnDGrid")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            ov.update_dz_d(g, col, gm["dp_ref"], a["surface_height"], a["height"], a["courant_number_x"], a["courant_number_y"],
                           a["x_area_flux"], a["y_area_flux"], a["ws"], a["dt"])
            ok &= report(f"rank{rank} updatedzd[{n_}] zh", r["out"]["height"], a["height"], _W(g, 0, 0, NZ + 1), tol=1e-13)
            ok &= report(f"rank{rank} updatedzd[{n_}] ws", r["out"]["ws"], a["ws"], _W(g)[:2], tol=1e-13)
        for n_, r in enumerate(_rec(cap, rank, "NonHydrostaticPressureGradient")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            oa.nh_p_grad(g, a["u"], a["v"], a["pp"], a["gz"], a["pk3"], a["delp"], a["dt"], a["ptop"], a["akap"])
            ok &= report(f"rank{rank} nh_p_grad[{n_}] u", r["out"]["u"], a["u"], _W(g, 0, 1), tol=1e-12)
            ok &= report(f"rank{rank} nh_p_grad[{n_}] v", r["out"]["v"], a["v"], _W(g, 1, 0), tol=1e-12)
            for k in ("pp", "gz", "pk3"):
                ok &= report(f"rank{rank} nh_p_grad[{n_}] {k}", r["out"][k], a[k], _W(g, 1, 1, NZ + 1), tol=1e-12)
        for n_, r in enumerate(_rec(cap, rank, "PK3Halo")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            oa.pk3_halo(g, a["pk3"], a["delp"], a["ptop"], a["akap"])
            ok &= report(f"rank{rank} pk3_halo[{n_}]", r["out"]["pk3"][full], a["pk3"][full], tol=1e-13)
        for n_, r in enumerate(_rec(cap, rank, "RayleighDamping")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            oa.ray_fast(g, a["u"], a["v"], a["w"], a["dp"], a["pfull"], a["dt"], a["ptop"], rf_cutoff=3000.0, tau=10.0)
            for k in ("u", "v", "w"):
                ok &= report(f"rank{rank} ray_fast[{n_}] {k}", r["out"][k][:, :, :NZ], a[k][:, :, :NZ], tol=1e-13)
        for n_, r in enumerate(_rec(cap, rank, "HyperdiffusionDamping")):
            a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r["in"].items()}
            oa.del2_cubed(g, a["qdel"], a["cd"], 3)
            ok &= report(f"rank{rank} del2cubed[{n_}]", r["out"]["qdel"][:, :, :NZ], a["qdel"][:, :, :NZ], tol=1e-13)
    return ok


def check_halo():
    """Reference CubedSphereCommunicator (6 ranks on threads) vs oracle/halo.py on random fields."""
    import pace.util
    from threadcomm import run_ranks

    from oracle import halo as oh

    n, nz = 12, 4
    rng = np.random.default_rng(5)
    dims = {"c": ["x", "y", "z"], "xi": ["x_interface", "y", "z"], "yi": ["x", "y_interface", "z"],
            "b": ["x_interface", "y_interface", "z"], "zi": ["x", "y", "z_interface"]}
    base = {k: [rng.random((n + 7, n + 7, nz + 1)) for _ in range(6)] for k in dims}

    def rank(comm):
        part = pace.util.CubedSpherePartitioner(pace.util.TilePartitioner((1, 1)))
        cube = pace.util.CubedSphereCommunicator(comm, part)
        sizer = pace.util.SubtileGridSizer.from_tile_params(nx_tile=n, ny_tile=n, nz=nz, n_halo=3, extra_dim_lengths={},
                                                            layout=(1, 1), tile_partitioner=part.tile, tile_rank=0)
        qf = pace.util.QuantityFactory.from_backend(sizer, "numpy")
        r = comm.Get_rank()
        out = {}

        def q(key, dd):
            x = qf.zeros(dd, "")
            x.data[:] = base[key][r]
            return x

        s = q("c", dims["c"]); cube.halo_update(s, n_points=3); out["c"] = s.data.copy()
        s = q("b", dims["b"]); cube.halo_update(s, n_points=3); out["b"] = s.data.copy()
        s = q("zi", dims["zi"]); cube.halo_update(s, n_points=2); out["zi"] = s.data.copy()
        u, v = q("yi", dims["yi"]), q("xi", dims["xi"])
        cube.vector_halo_update(u, v, n_points=3); out["du"], out["dv"] = u.data.copy(), v.data.copy()
        uc, vc = q("xi", dims["xi"]), q("yi", dims["yi"])
        cube.vector_halo_update(uc, vc, n_points=3); out["cu"], out["cv"] = uc.data.copy(), vc.data.copy()
        u, v = q("yi", dims["yi"]), q("xi", dims["xi"])
        cube.synchronize_vector_interfaces(u, v); out["su"], out["sv"] = u.data.copy(), v.data.copy()
        return out

    ref = run_ranks(6, rank)
    ok = True
    f = [a.copy() for a in base["c"]]; oh.halo_update(f, n, nk=nz)
    g_ = [a.copy() for a in base["b"]]; oh.halo_update(g_, n, xi=1, yi=1, nk=nz)
    z = [a.copy() for a in base["zi"]]; oh.halo_update(z, n, n_pts=2)
    du, dv = [a.copy() for a in base["yi"]], [a.copy() for a in base["xi"]]; oh.vector_halo_update(du, dv, n, grid="d", nk=nz)
    cu, cv = [a.copy() for a in base["xi"]], [a.copy() for a in base["yi"]]; oh.vector_halo_update(cu, cv, n, grid="c", nk=nz)
    su, sv = [a.copy() for a in base["yi"]], [a.copy() for a in base["xi"]]; oh.synchronize_vector_interfaces(su, sv, n, nk=nz)
    for t in range(6):
        for name, mine in (("c", f), ("b", g_), ("zi", z), ("du", du), ("dv", dv), ("cu", cu), ("cv", cv), ("su", su), ("sv", sv)):
            ok &= report(f"tile{t} halo {name}", ref[t][name], mine[t])
    return ok


GROUPS = {"halo": check_halo, "parts": check_parts, "csw": check_csw, "riem3": check_riem3, "corners": check_corners, "transport": check_transport, "damping": check_damping, "dsw": check_dsw}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GROUPS)
    allok = True
    for n in names:
        print(n)
        allok &= bool(GROUPS[n]())
    sys.exit(0 if allok else 1)
