"""`-m gpu`: the HIP library on a real MI355X, through the C ABI and the host classes.

* against the reference-generated fixtures (C12),
* against the oracle on synthetic inputs at C48 x 79 (all levels, tile seams of the LDS kernels),
* at the BASELINE size (C192 x 79) through size-independent properties.
"""
import numpy as np
import pytest

from helpers import (acoustic_errors, check_tracer_outputs, run_in_child, DSW_ARGS, DSW_CFG, RIEM_ARGS, Env, column_for_levels, compare, dsw_window, expand_riem_fixture, golden,
                     oracle_grid, run_d_sw, run_riem3, window)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pace_amd import _lib

    return _lib.load()  # raises if libpace_hip.so is missing: no fallback


def full_column(nz):
    col = golden("column_namelist_c12.npz")
    return {k: np.ascontiguousarray(v[:nz]) for k, v in col.items()}


@pytest.mark.parametrize("name,tile", [("d_sw_c12_tile0_call1.npz", 0), ("d_sw_c12_tile1_call3.npz", 1)])
def test_d_sw_matches_reference_fixture(lib, name, tile):
    fix = golden(name)
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    env = Env(lib, "cuda", golden(f"grid_c12_tile{tile}.npz"), 12, nk)
    out, _ = run_d_sw(env, column_for_levels(k_sel), {k: fix["in_" + k] for k in DSW_ARGS}, float(fix["dt"]),
                      ut0=fix["in_uc_contra"], vt0=fix["in_vc_contra"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        err = compare(fix["out_" + k][dsw_window(k, 12, nk)], out[k][dsw_window(k, 12, nk)])
        assert err < 3.2e-10, (k, err)  # translate_d_sw.py:19


@pytest.mark.parametrize("name", ["riem_solver3_c12_tile0_call2.npz", "riem_solver3_c12_tile0_call3.npz"])
def test_riem_solver3_matches_reference_fixture(lib, name):
    fix = golden(name)
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, 79)
    out = run_riem3(env, expand_riem_fixture(fix), bool(fix["last_call"]), float(fix["dt"]), float(fix["ptop"]))
    for k in ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w"):
        nk = 79 if k in ("delz", "w") else 80
        err = compare(fix["out_" + k][:, :, :nk], out[k][3:15, 3:7, :nk], near_zero=1e-12)
        assert err < 5e-6, (k, err)  # overrides/standard.yaml:49-61


def test_fvtp2d_matches_reference_fixture(lib):
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport

    fix = golden("fvtp2d_c12_tile0_call8.npz")
    k_sel = golden("d_sw_c12_tile0_call1.npz")["k_sel"]
    nk = len(k_sel)
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, nk)
    col = column_for_levels(k_sel)
    op = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 6, nord=env.kq(col["nord_t"]),
                               damp_c=env.kq(col["damp_t"]))
    f = {k[3:]: env.q3(v) for k, v in fix.items() if k.startswith("in_")}
    fx, fy = env.q3(), env.q3()
    op(f["q"], f["crx"], f["cry"], f["x_area_flux"], f["y_area_flux"], fx, fy, x_mass_flux=f["x_mass_flux"],
       y_mass_flux=f["y_mass_flux"], mass=f["mass"])
    assert compare(fix["out_q_x_flux"][window(12, 1, 0, nk)], fx.numpy()[window(12, 1, 0, nk)]) < 1e-13
    assert compare(fix["out_q_y_flux"][window(12, 0, 1, nk)], fy.numpy()[window(12, 0, 1, nk)]) < 1e-13


@pytest.mark.parametrize("n", [48, 96])
def test_d_sw_and_riem3_match_oracle(lib, n):
    """Synthetic C48 / C96 x 79: exercises every level class (k = 0, 1, 2, >= 3), the tile seams of the LDS kernels and
    (C96) workgroups that touch no tile edge, which run the straight-line interior PPM / delnflux paths."""
    from oracle import dgrid_sw, vertical
    from pace_amd import synthetic

    nz = 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = full_column(nz)
    env = Env(lib, "cuda", metrics, n, nz)
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    g = oracle_grid(metrics, n, nz)
    st = dgrid_sw.DSWState(s["u"].shape)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, DSW_CFG, st, *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        scale = float(np.abs(a[k][dsw_window(k, n, nz)]).max())
        err = compare(a[k][dsw_window(k, n, nz)], out[k][dsw_window(k, n, nz)], near_zero=1e-12 * max(scale, 1e-300))
        assert err < 3.2e-10, (k, err)
    # riem_solver3 on the d_sw-updated state
    inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": a["q_con"], "delp": a["delp"],
           "pt": a["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
           "log_p_interface": s["peln"], "w": a["w"]}
    got = run_riem3(env, inp, False, s["dt"], metrics["ptop"])
    b = {k: v.copy() for k, v in inp.items()}
    vertical.riem_solver3(g, False, s["dt"], b["cappa"], metrics["ptop"], b["zs"], b["ws"], b["delz"], b["q_con"], b["delp"], b["pt"],
                          b["zh"], b["p"], b["ppe"], b["pk3"], b["pk"], b["log_p_interface"], b["w"], p_fac=0.05)
    for k in ("delz", "zh", "ppe", "pk3", "w"):
        nk = nz if k in ("delz", "w") else nz + 1
        scale = float(np.abs(b[k][window(n, 0, 0, nk)]).max())
        # The reference's metric is relative (5e-6, overrides/standard.yaml:49-61); the perturbation pressure and w of this
        # synthetic state cross zero, where a relative metric is meaningless: values below 1e-5 of the field's magnitude are
        # exempt (measured absolute errors are < 1e-11 of the magnitude for both, tools/riem_check.py)
        err = compare(b[k][window(n, 0, 0, nk)], got[k][window(n, 0, 0, nk)], near_zero=(1e-5 if k in ("ppe", "w") else 1e-9) * scale)
        assert err < 5e-6, (k, err)
        assert float(np.abs(b[k][window(n, 0, 0, nk)] - got[k][window(n, 0, 0, nk)]).max()) < 1e-10 * scale, k


def test_c192_properties(lib):
    """BASELINE size (C192 x 79): properties that need no oracle.
    (1) a constant scalar is transported as q * unit flux (fvtp2d consistency);
    (2) d_sw is deterministic (bitwise) and leaves no NaN in its outputs;
    (3) delp is updated in flux form: sum(delp*area) changes only by the boundary fluxes."""
    import torch

    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport
    from pace_amd.fv3core.stencils.fxadv import FiniteVolumeFluxPrep

    n, nz = 192, 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(lib, "cuda", metrics, n, nz)
    col = full_column(nz)
    # (1)
    prep = FiniteVolumeFluxPrep(env.stencil_factory, env.grid_data)
    f = {k: env.q3(s[k]) for k in ("uc", "vc")}
    crx, cry, xfx, yfx, ut, vt = (env.q3() for _ in range(6))
    prep(f["uc"], f["vc"], crx, cry, xfx, yfx, ut, vt, s["dt"])
    tp = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 6)
    q = env.q3(np.full(s["u"].shape, 2.5))
    fx, fy = env.q3(), env.q3()
    tp(q, crx, cry, xfx, yfx, fx, fy)
    torch.cuda.synchronize()
    w = window(n, 1, 0, nz)
    np.testing.assert_allclose(fx.numpy()[w], 2.5 * xfx.numpy()[w], rtol=1e-13, atol=0)
    # (2) + (3)
    out1, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    out2, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    for k in ("delp", "pt", "u", "v", "w", "q_con", "mfx", "mfy", "heat_source"):
        a, b = out1[k][dsw_window(k, n, nz)], out2[k][dsw_window(k, n, nz)]
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, b), k
    area = metrics["area"][3 : 3 + n, 3 : 3 + n, None]
    c = window(n, 0, 0, nz)
    dm = ((out1["delp"][c] - s["delp"][c]) * area).sum(axis=(0, 1))
    mfx, mfy = out1["mfx"], out1["mfy"]
    boundary = (mfx[3, 3 : 3 + n, :nz].sum(0) - mfx[3 + n, 3 : 3 + n, :nz].sum(0) + mfy[3 : 3 + n, 3, :nz].sum(0)
                - mfy[3 : 3 + n, 3 + n, :nz].sum(0))
    total = (s["delp"][c] * area).sum(axis=(0, 1))
    np.testing.assert_allclose(dm / total, boundary / total, rtol=0, atol=1e-12)


def test_riem_solver3_c192_sampled_columns_match_oracle(lib):
    """riem_solver3 at the BASELINE size (C192 x 79).  The columns of the vertical solver are independent, so the oracle is run
    on 12 x 12 blocks of them (the four corners of the tile, an edge and the centre: 864 columns, embedded in C12-sized arrays)
    and compared with the same columns of the C192 device run at the operator's tolerance; the device run is also bitwise
    reproducible and leaves nothing non-finite."""
    from oracle import vertical
    from pace_amd import synthetic

    n, nz = 192, 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(lib, "cuda", metrics, n, nz)
    inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": s["q_con"], "delp": s["delp"],
           "pt": s["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
           "log_p_interface": s["peln"], "w": s["w"]}
    got = run_riem3(env, inp, True, s["dt"], metrics["ptop"])
    again = run_riem3(env, inp, True, s["dt"], metrics["ptop"])
    outs = ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w")
    for k in outs:
        nk = nz if k in ("delz", "w") else nz + 1
        a = got[k][window(n, 0, 0, nk)]
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, again[k][window(n, 0, 0, nk)]), k
    m12 = synthetic.tile_metrics(12, nz)
    g12 = oracle_grid(m12, 12, nz)
    for (bi, bj) in ((0, 0), (180, 0), (0, 180), (180, 180), (90, 0), (90, 90)):
        sub = {}
        for k, v in inp.items():
            blk = v[3 + bi:3 + bi + 12, 3 + bj:3 + bj + 12]
            full = np.zeros((19, 19) + v.shape[2:])
            full[3:15, 3:15] = blk
            sub[k] = full
        vertical.riem_solver3(g12, True, s["dt"], sub["cappa"], metrics["ptop"], sub["zs"], sub["ws"], sub["delz"], sub["q_con"],
                              sub["delp"], sub["pt"], sub["zh"], sub["p"], sub["ppe"], sub["pk3"], sub["pk"], sub["log_p_interface"],
                              sub["w"], p_fac=0.05)
        for k in outs:
            nk = nz if k in ("delz", "w") else nz + 1
            ref = sub[k][3:15, 3:15, :nk]
            dev = got[k][3 + bi:3 + bi + 12, 3 + bj:3 + bj + 12, :nk]
            scale = float(np.abs(ref).max())
            err = compare(ref, dev, near_zero=(1e-5 if k in ("ppe", "w") else 1e-9) * scale)
            assert err < 5e-6, (bi, bj, k, err)  # overrides/standard.yaml:49-61
            assert float(np.abs(ref - dev).max()) < 1e-10 * scale, (bi, bj, k)


def test_halo_updates_six_tiles_on_one_gpu_equal_the_reference_run(lib):
    """The HIP pack / unpack kernels and the exchange of six tiles resident on one device against what the reference's own
    pace.util (run natively, tools/make_golden_halo.py) left in the same arrays: exactly."""
    from pace_amd.util import run_tiles
    from test_halo import check_native, native_fixture, native_tile_program

    n, nz, base, exp = native_fixture()
    results = run_tiles(6, lambda comm: native_tile_program(comm, lib, base, n, nz, device="cuda:0"))
    check_native(results, exp)


def test_acoustic_dynamics_six_tiles_matches_reference_run(lib, tmp_path):
    """One whole AcousticDynamics call (n_split = 2, every operator of the loop and all halo-update groups) for the six C12
    tiles resident on one device, against the reference run's output.  Tolerance: see
    test_emu_kernels.test_acoustic_dynamics_six_tiles_emulated."""
    import json
    import os

    fixes, outs = run_in_child("acoustic", tmp_path)
    errs = [acoustic_errors(fixes[t], outs[t]) for t in range(6)]
    worst = {k: max(e[k] for e in errs) for k in errs[0]}
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(worst, open(os.path.join(out_dir, "acoustic_c12_gpu_errors.json"), "w"), indent=1)
    # Device exp/log are 1-2 ulp off numpy's; the two tridiagonal solves per substep amplify that on near-zero w (polar
    # tiles).  5e-6 is the reference's own Riem_Solver3 bound on every backend (overrides/standard.yaml:49-61); fields
    # the vertical solver does not feed stay at 1e-7.
    from helpers import ACOUSTIC_TOL, ACOUSTIC_TOL_DEFAULT

    for k, e in worst.items():
        assert e < ACOUSTIC_TOL.get(k, ACOUSTIC_TOL_DEFAULT), (k, e)


def test_acoustic_dynamics_variant_six_tiles_matches_reference_run(lib, tmp_path):
    """One whole AcousticDynamics call with nord = 2, d_con = 0 and all advection orders 5 against the reference's run of that
    namelist (tools/make_golden_acoustic.py v2), six tiles on one device."""
    from helpers import ACOUSTIC_TOL, ACOUSTIC_TOL_DEFAULT

    fixes, outs = run_in_child("acoustic_v2", tmp_path)
    errs = [acoustic_errors(fixes[t], outs[t]) for t in range(6)]
    worst = {k: max(e[k] for e in errs) for k in errs[0]}
    for k, e in worst.items():
        assert e < ACOUSTIC_TOL.get(k, ACOUSTIC_TOL_DEFAULT), (k, e)


def test_fused_transport_update_matches_oracle_c96(lib):
    """pace_fvtp2d_update (PPM transport + DelnFlux + apply_fluxes in one kernel) against the three oracle steps,
    C96 x 79 (edge, corner and interior workgroups), bit-exact."""
    import ctypes as C

    import torch

    from oracle import ppm_transport as tr
    from pace_amd import synthetic
    from pace_amd.util.grid import geom_struct

    n, nz = 96, 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = full_column(nz)
    env = Env(lib, "cuda", metrics, n, nz)
    g = oracle_grid(metrics, n, nz)
    # Courant numbers / area fluxes from the oracle's fxadv on the synthetic C-grid winds; mass fluxes: transport of delp
    from oracle import dgrid_sw

    for k in ("crx", "cry", "xfx", "yfx"):
        s[k] = np.zeros_like(s["pt"])
    dgrid_sw.fxadv(g, s["uc"], s["vc"], s["crx"], s["cry"], s["xfx"], s["yfx"], np.zeros_like(s["pt"]), np.zeros_like(s["pt"]), s["dt"])
    fx, fy = np.zeros_like(s["pt"]), np.zeros_like(s["pt"])
    tr.fvtp2d(g, s["delp"].copy(), s["crx"], s["cry"], s["xfx"], s["yfx"], fx, fy, 6)
    gx, gy = np.zeros_like(s["pt"]), np.zeros_like(s["pt"])
    tr.fvtp2d(g, s["pt"].copy(), s["crx"], s["cry"], s["xfx"], s["yfx"], gx, gy, 6, x_mass_flux=fx, y_mass_flux=fy)
    nord = np.asarray(col["nord_t"], dtype=float)
    damp_c = np.asarray(col["damp_t"], dtype=float)
    tr.delnflux(g, s["pt"], gx, gy, nord, damp_c, metrics["da_min"], mass=s["delp"])
    rarea = metrics["rarea"][:, :, None]
    expect = np.zeros_like(s["pt"])
    W = window(n, 0, 0, nz)
    inc = (gx[:-1, :-1] - gx[1:, :-1] + gy[:-1, :-1] - gy[:-1, 1:]) * rarea[:-1, :-1]
    expect[:-1, :-1] = s["pt"][:-1, :-1] * s["delp"][:-1, :-1] + inc
    f = {k: env.q3(v) for k, v in (("pt", s["pt"]), ("crx", s["crx"]), ("cry", s["cry"]), ("xfx", s["xfx"]), ("yfx", s["yfx"]),
                                   ("fx", fx), ("fy", fy), ("delp", s["delp"]))}
    out = env.q3()
    kdev = torch.as_tensor(np.concatenate([(damp_c[:nz] * metrics["da_min"]) ** (nord[:nz] + 1), nord[:nz]]), device="cuda")
    geom = geom_struct(env.qf)
    lib.call("pace_fvtp2d_update", C.byref(geom), C.byref(env.grid_data.c_struct()), f["pt"].ptr, f["crx"].ptr, f["cry"].ptr,
             f["xfx"].ptr, f["yfx"].ptr, f["fx"].ptr, f["fy"].ptr, f["delp"].ptr, kdev.data_ptr(), kdev.data_ptr() + 8 * nz,
             int(nord.max()), out.ptr, 6, nz, None)
    torch.cuda.synchronize()
    assert np.array_equal(expect[W], out.numpy()[W])


def test_tracer_advection_six_tiles_matches_reference_run(lib, tmp_path):
    """TracerAdvection (k_fvtp2d<8, -1, 0> + the tracer_2d_1l stencils + tracer halo updates), six tiles on one device,
    against the reference's own run: bit for bit."""
    fixes, outs = run_in_child("tracer", tmp_path)
    check_tracer_outputs(fixes, outs)


def test_ord8_transport_matches_oracle_c96(lib):
    """Monotone PPM at C96 x 7 levels: edge, corner and interior workgroups of the ord-8 kernel against the oracle."""
    import torch

    from oracle import dgrid_sw
    from oracle import ppm_transport as tr
    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport

    n, nz = 96, 7
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    g = oracle_grid(metrics, n, nz)
    for k in ("crx", "cry", "xfx", "yfx"):
        s[k] = np.zeros_like(s["pt"])
    dgrid_sw.fxadv(g, s["uc"], s["vc"], s["crx"], s["cry"], s["xfx"], s["yfx"], np.zeros_like(s["pt"]), np.zeros_like(s["pt"]), s["dt"])
    env = Env(lib, "cuda", metrics, n, nz)
    op = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 8)
    f = {k: env.q3(s[k]) for k in ("pt", "crx", "cry", "xfx", "yfx")}
    fx, fy = env.q3(), env.q3()
    op(f["pt"], f["crx"], f["cry"], f["xfx"], f["yfx"], fx, fy)
    torch.cuda.synchronize()
    ofx, ofy = np.zeros_like(s["pt"]), np.zeros_like(s["pt"])
    tr.fvtp2d(g, s["pt"].copy(), s["crx"], s["cry"], s["xfx"], s["yfx"], ofx, ofy, 8)
    assert np.array_equal(ofx[window(n, 1, 0, nz)], fx.numpy()[window(n, 1, 0, nz)])
    assert np.array_equal(ofy[window(n, 0, 1, nz)], fy.numpy()[window(n, 0, 1, nz)])


@pytest.mark.parametrize("name", sorted(__import__("helpers").REMAP_CASES))
def test_map_single_matches_reference_run(lib, name):
    """MapSingle on the GPU against the run of the reference's MapSingle (tests/golden/remap_c12.npz): bit for bit -- the
    kernels contain only IEEE add / multiply / divide and are built without FMA contraction."""
    from helpers import REMAP_KM, run_map_single

    d = golden("remap_c12.npz")
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, REMAP_KM)
    out = run_map_single(env, name, d)
    assert np.array_equal(out, d[name + "_out"][:, :, :REMAP_KM])


def _remap_columns(n, km, seed, deform):
    """Synthetic columns: hybrid-like target interfaces, a source coordinate deformed by up to `deform` layers, a smooth field
    plus noise (so the monotonicity constraints engage)."""
    rng = np.random.default_rng(seed)
    ni = n + 7
    sig = np.linspace(0.0, 1.0, km + 1) ** 1.6
    ps = 1.0e5 * (1.0 + 0.02 * rng.random((ni, ni)))
    ptop = 300.0
    pe2 = ptop + (ps - ptop)[:, :, None] * sig[None, None, :]
    amp = deform / km * rng.random((ni, ni))
    s1 = sig[None, None, :] + amp[:, :, None] * np.sin(2.0 * np.pi * sig)[None, None, :]
    s1[:, :, 0], s1[:, :, km] = 0.0, 1.0
    pe1 = ptop + (ps - ptop)[:, :, None] * s1
    q = np.zeros((ni, ni, km + 1))
    q[:, :, :km] = 250.0 + 40.0 * np.cos(3.0 * np.pi * sig[:km])[None, None, :] + 3.0 * rng.standard_normal((ni, ni, km))
    return q, pe1, pe2


@pytest.mark.parametrize("kord,iv", [(9, 1), (9, 0), (10, 1), (9, -1), (9, -2)])
def test_map_single_matches_oracle_c48(lib, kord, iv):
    """All levels, 48 x 48 columns, strongly deformed coordinate (up to 3 layers): bit-exact against the oracle."""
    import torch

    from oracle import remapping
    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.map_single import MapSingle

    n, km = 48, 79
    env = Env(lib, "cuda", synthetic.tile_metrics(n, km), n, km)
    q, pe1, pe2 = _remap_columns(n, km, seed=3 + kord + iv, deform=3.0)
    qs = 0.1 * q[:, :, km - 1]
    fq, f1, f2, fs = env.q3(q), env.q3(pe1), env.q3(pe2), env.q2(qs)
    MapSingle(env.stencil_factory, env.qf, kord, iv, ["x", "y", "z"])(fq, f1, f2, qs=fs if iv == -2 else None, qmin=200.0 if iv == 1 else 0.0)
    torch.cuda.synchronize()
    w = (slice(3, 3 + n), slice(3, 3 + n))
    ref = q[w].copy()
    remapping.map_single(ref, pe1[w], pe2[w], km, kord, iv, qs=qs[w] if iv == -2 else None, qmin=200.0 if iv == 1 else 0.0)
    assert np.array_equal(fq.numpy()[w][:, :, :km], ref[:, :, :km])


def test_map_single_c192_properties(lib):
    """BASELINE size: (a) the remap conserves the column integral sum(q dp) (iv = 1, kord 9) to rounding; (b) remapping
    onto the same coordinate returns the field to rounding; (c) a constant field stays constant to rounding."""
    import torch

    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.map_single import MapSingle

    n, km = 192, 79
    env = Env(lib, "cuda", synthetic.tile_metrics(n, km), n, km)
    q, pe1, pe2 = _remap_columns(n, km, seed=17, deform=2.5)
    w = (slice(3, 3 + n), slice(3, 3 + n))
    op = MapSingle(env.stencil_factory, env.qf, 9, 1, ["x", "y", "z"])
    fq, f1, f2 = env.q3(q), env.q3(pe1), env.q3(pe2)
    op(fq, f1, f2)
    torch.cuda.synchronize()
    out = fq.numpy()[w][:, :, :km]
    before = (q[w][:, :, :km] * np.diff(pe1[w], axis=2)).sum(axis=2)
    after = (out * np.diff(pe2[w], axis=2)).sum(axis=2)
    assert np.max(np.abs(after - before) / np.abs(before)) < 1e-13
    fq = env.q3(q)
    op(fq, f1, f1)
    torch.cuda.synchronize()
    assert np.max(np.abs(fq.numpy()[w][:, :, :km] - q[w][:, :, :km])) < 1e-11
    c = np.zeros_like(q)
    c[:, :, :km] = 7.25
    fq = env.q3(c)
    op(fq, f1, f2)
    torch.cuda.synchronize()
    assert np.max(np.abs(fq.numpy()[w][:, :, :km] - 7.25)) < 1e-12


@pytest.mark.parametrize("kord", [9, 10])
def test_mapn_tracer_and_fillz_match_oracle(lib, kord):
    """MapNTracer (batched remap of seven tracers + fillz) on the GPU against the oracle: bit for bit."""
    from helpers import REMAP_KM, run_mapn_tracer

    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, REMAP_KM)
    got, exp = run_mapn_tracer(env, golden("remap_c12.npz"), kord)
    for t, (g, e) in enumerate(zip(got, exp)):
        assert np.array_equal(g, e), t


def test_lagrangian_to_eulerian_order_10_matches_reference_run(lib):
    """LagrangianToEulerian with every remapping order 10 on the inputs of the reference's own run with that namelist: tracers,
    winds, w, delz exact or at rounding level; pt / pkz (log-pressure coordinate, discontinuous limiter) within 1e-6."""
    from helpers import check_l2e, l2e_k10_fixture, run_l2e

    d = l2e_k10_fixture()
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, 79)
    check_l2e(run_l2e(env, d, False, kord=10), d, False, 1e-12, loose={"pt": 1e-6, "pkz": 1e-6})


@pytest.mark.parametrize("last_step", [False, True])
def test_lagrangian_to_eulerian_matches_reference_run(lib, last_step):
    """LagrangianToEulerian on the GPU against the run of the reference (tests/golden/l2e_c12.npz).  The reference's own
    bound for this operator is 2e-8; device exp / log differ from numpy's in the last place, and a remap
    fed with them can take another limiter branch, so: 1e-11 on every variable, and the mass fields exactly."""
    from helpers import check_l2e, run_l2e

    d = golden("l2e_c12.npz")
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, 79)
    worst = check_l2e(run_l2e(env, d, last_step), d, last_step, 1e-11)
    if not last_step:
        for name in ("delp", "pe", "ps", "tr_qvapor", "w", "u", "v"):
            assert worst[name] == 0.0, (name, worst[name])


def test_neg_adj3_matches_reference_run(lib):
    from pace_amd.fv3core.stencils.neg_adj3 import AdjustNegativeTracerMixingRatio
    import torch

    d = golden("negadj_c12.npz")
    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, 79)

    def embed(a):
        full = np.full((19, 19, 80), np.nan)
        full[3:15, 3:15, :] = a
        return env.q3(full)

    names = ["qvapor", "qliquid", "qrain", "qsnow", "qice", "qgraupel", "qcld"]
    f = {k: embed(d["in_" + k]) for k in names + ["pt", "delp"]}
    AdjustNegativeTracerMixingRatio(env.stencil_factory, env.qf, False, False)(*[f[k] for k in names], f["pt"], f["delp"])
    torch.cuda.synchronize()
    for k in names + ["pt"]:
        assert np.array_equal(f[k].numpy()[3:15, 3:15, :79], d["out_" + k][:, :, :79]), k


def test_dynamical_core_step_six_tiles_matches_reference_run(lib, tmp_path):
    """One whole DynamicalCore.step_dynamics for the six C12 tiles resident on one device against the run of the
    reference's DynamicalCore (see the emulated twin of this test for what it covers and helpers.check_dycore for the
    tolerances)."""
    import json
    import os

    from helpers import check_dycore

    fixes, outs = run_in_child("dycore", tmp_path)
    worst = check_dycore(fixes, outs)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(worst, open(os.path.join(out_dir, "dycore_c12_gpu_errors.json"), "w"), indent=1)


@pytest.mark.parametrize("n,last_step", [(48, False), (96, True)])
def test_lagrangian_to_eulerian_matches_oracle(lib, n, last_step):
    """LagrangianToEulerian at C48 / C96 x 79 (BASELINE's smaller configurations) against the oracle on a synthetic state
    whose Lagrangian surfaces are displaced by up to two layers: masses, pressures and remapped winds / tracers exactly or
    to rounding, log-pressure-driven fields (pt, pkz) to 1e-11."""
    import torch

    from helpers import l2e_synthetic_case

    from oracle import constants as oc
    from oracle import remapping
    from pace_amd import synthetic
    from pace_amd.fv3core import RemappingConfig
    from pace_amd.fv3core.stencils.remapping import LagrangianToEulerian
    from pace_amd.util import constants as c

    km = 79
    f, tr, ak, bk, ptop = l2e_synthetic_case(n, km)
    env = Env(lib, "cuda", synthetic.tile_metrics(n, km), n, km)
    qf = {k: (env.q3(v) if v.ndim == 3 else env.q2(v)) for k, v in f.items()}
    qt = {k: env.q3(v) for k, v in tr.items()}
    op = LagrangianToEulerian(env.stencil_factory, env.qf, RemappingConfig(), None, 8, None, qt)
    op(qt, qf["pt"], qf["delp"], qf["delz"], qf["peln"], qf["u"], qf["v"], qf["w"], qf["cappa"], qf["q_con"], qf["qcld"],
       qf["pkz"], qf["pk"], qf["pe"], qf["phis"], qf["ps"], qf["wsd"], env.kq(ak), env.kq(bk), None, ptop, c.KAPPA, c.ZVIR, last_step,
       0.0, 100.0)
    torch.cuda.synchronize()
    remapping.lagrangian_to_eulerian(f, tr, ak, bk, ptop, oc.KAPPA, oc.ZVIR, last_step, n, km, o=3, nq=8)
    cw = (slice(3, 3 + n), slice(3, 3 + n))
    for name in ("delp", "pe", "ps"):
        got, ref = qf[name].numpy()[cw], f[name][cw]
        assert np.array_equal(got[..., :km] if got.ndim == 3 else got, ref[..., :km] if ref.ndim == 3 else ref), name
    for name in ("pt", "delz", "peln", "w", "q_con", "pkz", "pk", "cappa", "u", "v"):
        win = {"u": (slice(3, 3 + n), slice(3, 4 + n)), "v": (slice(3, 4 + n), slice(3, 3 + n))}.get(name, cw)
        e = compare(f[name][win][:, :, :km], qf[name].numpy()[win][:, :, :km], near_zero=1e-14)
        assert e < 1e-11, (name, e)
    for name in tr:
        e = compare(tr[name][cw][:, :, :km], qt[name].numpy()[cw][:, :, :km], near_zero=1e-18)
        assert e < 1e-11, (name, e)


@pytest.mark.parametrize("cfg", [dict(hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5), dict(hord_dp=5, hord_tm=6, hord_vt=5, hord_mt=6),
                                 dict(d_con=0.0), dict(nord=2)])
def test_d_sw_other_namelists_match_oracle(lib, cfg):
    """d_sw with the other advection orders the reference supports (5; mixed 5 / 6), without dissipative heating and with a
    lower damping order, C48 x 79 against the oracle (whose PPM ord-5 / ord-6 and damping functions are pinned one by one
    against the reference, tools/crosscheck_oracle.py): covers the <5, ...> instantiations of the transport and kinetic-
    energy kernels that the ord-6 fixtures never reach."""
    from oracle import dgrid_sw
    from pace_amd import synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import get_column_namelist

    n, nz = 48, 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(lib, "cuda", metrics, n, nz)
    full = dict(DSW_CFG, **cfg)
    pcfg = DGridShallowWaterLagrangianDynamicsConfig(**full)
    colq = get_column_namelist(pcfg, env.qf)
    col = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v))[:nz] for k, v in colq.items()}
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"], cfg=full)
    g = oracle_grid(metrics, n, nz)
    st = dgrid_sw.DSWState(s["u"].shape)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, full, st, *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        scale = float(np.abs(a[k][dsw_window(k, n, nz)]).max())
        err = compare(a[k][dsw_window(k, n, nz)], out[k][dsw_window(k, n, nz)], near_zero=1e-12 * max(scale, 1e-300))
        assert err < 3.2e-10, (cfg, k, err)


def test_dynamical_core_two_remapping_steps_matches_reference_run(lib, tmp_path):
    """k_split = 2: see the emulated twin of this test."""
    from helpers import check_dycore

    fixes, outs = run_in_child("dycore_k2", tmp_path)
    check_dycore(fixes, outs, default=1e-9)


def test_dynamical_core_step_remapping_order_10_matches_reference_run(lib, tmp_path):
    """One whole DynamicalCore step with every remapping order 10 against the reference's run of that namelist."""
    from helpers import check_dycore_kord10

    fixes, outs = run_in_child("dycore_kord10", tmp_path)
    check_dycore_kord10(fixes, outs)


@pytest.mark.parametrize("variant", ["nord2", "dcon0", "skeb", "dddmp0"])
def test_d_sw_namelist_variants_match_reference_run(lib, variant):
    """d_sw with one namelist option changed against runs of the reference with that namelist (tools/make_golden_dsw_variants.py)
    at the Translate tolerance of D_SW."""
    from helpers import dsw_variant_fixture, run_d_sw_variant_fixture

    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, dsw_variant_fixture(variant)[3])
    errs = run_d_sw_variant_fixture(env, variant)
    assert max(errs.values()) < 3.2e-10, errs  # translate_d_sw.py:19


def test_d_sw_order5_matches_reference_run(lib):
    from helpers import run_d_sw_h5_fixture

    env = Env(lib, "cuda", golden("grid_c12_tile0.npz"), 12, len(golden("d_sw_h5_c12_tile0_call1.npz")["k_sel"]))
    assert run_d_sw_h5_fixture(env) < 3.2e-10  # translate_d_sw.py:19


@pytest.mark.parametrize("n", [48, 96])
def test_every_operator_of_the_loop_matches_oracle(lib, n):
    """Per-operator parity at C48 / C96 x 79 (BASELINE configurations 2 and 3): each operator of the acoustic loop body --
    d2a2c_vect, c_sw, updatedzc, riem_solver_c, p_grad_c, d_sw, updatedzd, riem_solver3 (last call), edge_pe, pk3_halo,
    compute_geopotential, nh_p_grad, ray_fast, del2cubed, apply_diffusive_heating -- runs on the oracle's input state for
    that operator and is compared on the window and at the tolerance of the reference's Translate test for it
    (tests/opchain.py lists them).  At C96 the LDS-tile kernels have workgroups that touch no tile edge."""
    import json
    import os

    from opchain import Chain, ProductOps, check_case

    chain = Chain(n, 79)
    ops = ProductOps(lib, "cuda", chain)
    report, seen = {}, []
    for case in chain.cases():
        check_case(ops, case, report=report)
        seen.append(case.name)
    assert len(seen) == 15
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, f"operator_errors_c{n}.json"), "w"), indent=1)


def _loop_child(n, nz, n_split, out_path, geometry="synthetic", timestep=None):
    import pickle

    import opchain
    from pace_amd import _lib

    got = opchain.product_loop(_lib.load(), "cuda", n, nz, n_split, timestep or 3.571 * n_split, geometry)
    with open(out_path, "wb") as f:
        pickle.dump(got, f)


@pytest.mark.parametrize("n,geometry", [(96, "synthetic"), (96, "sphere"), (192, "sphere")], ids=["c96-synthetic", "c96-sphere", "c192-sphere"])
def test_full_acoustic_loop_six_tiles_matches_oracle(lib, tmp_path, n, geometry):
    """BASELINE configurations 3 and 4 (minus the wire): the FULL acoustic loop (c_sw, updatedzc, riem_solver_c, p_grad_c, d_sw,
    updatedzd, riem_solver3, pe / pk3 halo, nh_p_grad, ray_fast, del2cubed, heating; every halo-update group) at C96 x 79 and at
    C192 x 79, six tiles resident on ONE device and joined by the cubed-sphere exchange (the seven grouped exchanges per substep
    of halo_updater.py:217-303 / dyn_core.py:720-942 between tiles of the same device: what six GPUs do over RCCL), every operator
    running on its predecessor's output, against oracle/dyn_core.py.  Tolerances: the reference's DynCore bound, 2e-6, for what the
    vertical solvers feed; masses, temperatures, pressures 1e-12."""
    import json
    import os
    import pickle
    import subprocess
    import sys

    import opchain

    nz, n_split = 79, 1
    # synthetic: six copies of pace_amd/synthetic.py's tile (dt from its own Courant number); sphere: the gnomonic cubed sphere
    # with the baroclinic case's state (tests/opchain.py six_tile_inputs_sphere) and the C96 namelist's acoustic substep
    timestep = 3.571 * n_split if geometry == "synthetic" else (112.5 if n <= 96 else 56.25) * n_split  # (the C96 / C192 namelists' acoustic substeps)
    out = os.path.join(str(tmp_path), "loop.pkl")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r}); "
            f"import test_gpu_parity as t; t._loop_child({n}, {nz}, {n_split}, {out!r}, {geometry!r}, {timestep!r})")
    child = subprocess.Popen([sys.executable, "-X", "faulthandler", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    ref = opchain.oracle_loop(n, nz, n_split, timestep, geometry)  # the oracle runs on the host while the device works
    so, se = child.communicate(timeout=900)
    assert child.returncode == 0, (child.returncode, so[-2000:], se[-4000:])
    with open(out, "rb") as f:
        got = pickle.load(f)
    detail = {}
    errs = opchain.loop_errors(ref, got, n, nz, detail=detail, geometry=geometry)
    out_dir = os.path.join(root, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(detail, open(os.path.join(out_dir, f"acoustic_loop_c{n}_{geometry}_gpu_errors.json"), "w"), indent=1)
    for k, e in errs.items():
        assert e < opchain.LOOP_TOL.get(k, 1e-9), (k, e)
        # the ABSOLUTE error, as a fraction of the field's magnitude, is bounded on both geometries: the floors of the relative
        # metric only mask that metric
        bound = opchain.LOOP_ABS_SPHERE if geometry == "sphere" else opchain.LOOP_ABS_SYNTHETIC
        assert detail[k]["max_abs_error_over_magnitude"] < bound, (k, detail[k])


def test_standalone_ppm_and_divergence_damping_match_oracle(lib):
    """The stand-alone operator classes XPiecewiseParabolic / YPiecewiseParabolic (orders 5, 6, 8) and DivergenceDamping at
    C96 x 20 against the oracle (reference bounds: 1e-14 and 1.4e-10; no transcendental is involved, so: exact)."""
    from opchain import check_standalone_operators

    check_standalone_operators(lib, "cuda", 96, 20, exact=True)


def test_dynamical_core_step_from_generated_inputs_matches_reference_run(lib, tmp_path):
    """The GPU twin of test_dynamical_core_step_from_generated_inputs_emulated: (a) generated grid + the reference run's
    initial state, (b) generated grid + generated baroclinic state -> one whole step on six tiles -> the reference run's
    output, (b) within the algorithm's own sensitivity to 1e-13 m/s of wind noise (helpers.GENERATED_TOL)."""
    from helpers import check_dycore, check_dycore_generated

    (fixes, outs), (fixes2, outs2) = run_in_child("dycore_generated", tmp_path)
    check_dycore(fixes, outs)
    check_dycore_generated(fixes2, outs2)


def test_d_sw_separate_outputs_equal_in_place(lib):
    """The GPU twin of test_d_sw_separate_outputs_equal_in_place_emulated at C96 x 12: d_sw called twice with the halos rewritten
    in between, the reference's in-place contract against `swap_scalar_storage` (one stream and with the wind half on the side
    stream) -- every output bit for bit over the whole storage -- and against `skip_dead_outputs` on every live output."""
    from test_emu_kernels import check_dsw_contract_variants, dsw_contract_variants

    check_dsw_contract_variants(dsw_contract_variants(gpu=True))


@pytest.mark.gpu
def test_d_sw_launch_structure_switches_are_bit_identical(lib):
    """The GPU twin of ..._emulated at C48 x 7: FiniteVolumeFluxPrep as one launch or three, the wind halo copy inside it or as a launch of
    its own, kinetic energy + vorticity as one launch or two -- every argument of d_sw bit for bit over the whole storage."""
    from test_emu_kernels import dsw_launch_structure_switches

    dsw_launch_structure_switches(gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,nz", [(192, 8), (96, 79)])
def test_c_sw_interior_tiles_equal_the_four_passes_on_the_device(n, nz):
    """c_sw on the device: k_csw_tile on the interior tiles (6 x 12 of them at C192) + the four passes on the band, beside each other
    on two streams, against the four passes over the whole plane on one stream (PACE_CSW_NO_TILES): the same bits in every output
    array, whole storage -- and again with the call split around the halo exchange (start_interior / __call__)."""
    import os

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core.stencils.c_sw import CGridShallowWaterDynamics

    lib = _lib.load()
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    names = ("delp", "pt", "u", "v", "w", "uc", "vc", "ua", "va", "ut", "vt", "divgd", "omga")

    def run(split):
        import torch

        f = {k: env.q3(s[k] if k in s else np.zeros_like(s["pt"])) for k in names}
        op = CGridShallowWaterDynamics(env.stencil_factory, env.qf, env.grid_data, nested=False, grid_type=0, nord=3)
        args = [f[k] for k in names] + [0.5 * s["dt"]]
        if split:
            op.start_interior(*args)
        delpc, ptc = op(*args)
        torch.cuda.synchronize()
        out = {k: f[k].numpy().copy() for k in names}
        out.update(delpc=delpc.numpy().copy(), ptc=ptc.numpy().copy())
        return out

    os.environ["PACE_CSW_NO_TILES"] = "1"
    try:
        ref = run(False)
    finally:
        del os.environ["PACE_CSW_NO_TILES"]
    for split in (False, True):
        got = run(split)
        for k in ref:
            assert np.array_equal(ref[k], got[k], equal_nan=True), (k, split, float(np.nanmax(np.abs(ref[k] - got[k]))))
    assert not np.array_equal(ref["uc"][10:n - 4, 10:n - 4, :nz], s["uc"][10:n - 4, 10:n - 4, :nz])
