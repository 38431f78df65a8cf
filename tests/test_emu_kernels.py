"""The HIP kernel SOURCES compiled for the CPU (tests/emu) and driven through the same host classes and
C ABI, checked against the reference-generated fixtures.  This is how kernel index logic / LDS tiling /
barrier structure is verified in the GPU-less container (and under ASan); it says nothing about the GPU
build, which the `-m gpu` tests cover.  CPU only."""
import os

import numpy as np
import pytest

from helpers import (DSW_ARGS, Env, build_emu_small, oracle_grid, acoustic_errors, check_tracer_outputs, run_acoustic_six_tiles, run_tracer_six_tiles, build_emu, column_for_levels, compare, dsw_window, expand_riem_fixture, golden, run_d_sw,
                     run_riem3, window)


@pytest.fixture(scope="module")
def emu_lib():
    from pace_amd import _lib

    return _lib.Library(build_emu())


@pytest.mark.parametrize("name,tile", [("d_sw_c12_tile0_call1.npz", 0), ("d_sw_c12_tile1_call3.npz", 1)])
def test_d_sw_kernels_emulated(emu_lib, name, tile):
    fix = golden(name)
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    env = Env(emu_lib, "cpu", golden(f"grid_c12_tile{tile}.npz"), 12, nk)
    out, _ = run_d_sw(env, column_for_levels(k_sel), {k: fix["in_" + k] for k in DSW_ARGS}, float(fix["dt"]),
                      ut0=fix["in_uc_contra"], vt0=fix["in_vc_contra"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        err = compare(fix["out_" + k][dsw_window(k, 12, nk)], out[k][dsw_window(k, 12, nk)])
        assert err < 3.2e-10, (k, err)


@pytest.mark.parametrize("name", ["riem_solver3_c12_tile0_call2.npz", "riem_solver3_c12_tile0_call3.npz"])
def test_riem_solver3_kernel_emulated(emu_lib, name):
    fix = golden(name)
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)
    out = run_riem3(env, expand_riem_fixture(fix), bool(fix["last_call"]), float(fix["dt"]), float(fix["ptop"]))
    for k in ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w"):
        nk = 79 if k in ("delz", "w") else 80
        err = compare(fix["out_" + k][:, :, :nk], out[k][3:15, 3:7, :nk], near_zero=1e-12)
        assert err < 5e-6, (k, err)  # overrides/standard.yaml:49-61


def test_riem_solver3_bad_column_does_not_come_back_finite(emu_lib):
    """the column solvers' own log / exp (k_riem3f.hip lean_log / lean_exp) answer what the library's do outside
    their domain: a column whose pressure thickness has gone negative must come back non-finite, not as a plausible number that
    the `finite` checks downstream would let through."""
    fix = golden("riem_solver3_c12_tile0_call2.npz")
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)
    inp = expand_riem_fixture(fix)
    inp["delp"][5, 5, :] = -np.abs(inp["delp"][5, 5, :])  # one bad column; its neighbours stay good
    out = run_riem3(env, inp, bool(fix["last_call"]), float(fix["dt"]), float(fix["ptop"]))
    assert not np.isfinite(out["ppe"][5, 5, :79]).all() or not np.isfinite(out["delz"][5, 5, :79]).all()
    assert np.isfinite(out["ppe"][6, 5, :79]).all() and np.isfinite(out["delz"][6, 5, :79]).all()


@pytest.mark.parametrize("n,nz,which", [(24, 5, "small"), (48, 2, "big")])
def test_c_sw_interior_tiles_equal_the_four_passes(emu_lib, emu_small_lib, n, nz, which):
    """c_sw's interior goes through k_csw_tile (one workgroup = all of c_sw for a tile of cells of one level, every intermediate
    in LDS), the band near the edges through the four passes.  With PACE_CSW_NO_TILES the four passes take the whole plane: the
    same bits in every output array, whole storage (4 x 3 tiles at C24: twelve tiles and every seam between them and the band;
    the product's 30 x 18 at C48) -- and the same again when the call is split around the halo exchange (start / finish)."""
    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.c_sw import CGridShallowWaterDynamics

    lib = emu_small_lib if which == "small" else emu_lib
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cpu", m, n, nz)
    names = ("delp", "pt", "u", "v", "w", "uc", "vc", "ua", "va", "ut", "vt", "divgd", "omga")

    def run(split):
        f = {k: env.q3(s[k] if k in s else np.zeros_like(s["pt"])) for k in names}
        op = CGridShallowWaterDynamics(env.stencil_factory, env.qf, env.grid_data, nested=False, grid_type=0, nord=3)
        args = [f[k] for k in names] + [0.5 * s["dt"]]
        if split:
            op.start_interior(*args)
        delpc, ptc = op(*args)
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
    # (and the tiles did run: the interior of the caller's uc differs from what pass B alone leaves there)
    assert not np.array_equal(ref["uc"][10:n - 4, 10:n - 4, :nz], s["uc"][10:n - 4, 10:n - 4, :nz])


def test_d_sw_outputs_supported_is_what_the_launcher_accepts(emu_lib):
    """the query and the launcher must agree.  pace_d_sw_outputs_supported sees the column namelist: damping orders
    nord_v / nord_w / nord_t above 2 (get_column_namelist never makes them: d_sw.py:633-683) mean no separate outputs, and the
    operator built on such a namelist runs in place -- the launcher is never handed outputs it would refuse."""
    import ctypes as C

    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist

    n, nz = 24, 4
    from pace_amd import synthetic

    m = synthetic.tile_metrics(n, nz)
    from helpers import build_emu_canon
    from pace_amd import _lib

    env = Env(_lib.Library(build_emu_canon()), "cpu", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg, swap_scalar_storage=True)
    lib = op.lib
    assert lib.cdll.pace_d_sw_outputs_supported(C.byref(op._geom), C.byref(op._col), C.byref(op._cfg)) == 3 and op._pingpong and op._wind_outputs
    col3 = {k: (v.numpy().copy() if hasattr(v, "numpy") else np.array(v, dtype=float)) for k, v in col.items()}
    col3["nord_v"] = np.full_like(col3["nord_v"], 3.0)
    op3 = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, {k: env.kq(v) for k, v in col3.items()}, False, False, cfg,
                                              swap_scalar_storage=True)
    assert lib.cdll.pace_d_sw_outputs_supported(C.byref(op3._geom), C.byref(op3._col), C.byref(op3._cfg)) == 0
    assert not op3._pingpong and not op3._wind_outputs
    # the older, column-blind queries still say what they said
    assert lib.cdll.pace_d_sw_pingpong_supported(C.byref(op3._geom), C.byref(op3._cfg)) == 1


def test_in_checkpoints_hold_the_state_before_the_call(emu_lib):
    """AcousticDynamics overlaps the u / v and uc / vc halo exchanges with the interior of c_sw's first pass and
    of d_sw's flux preparation; with a checkpointer attached those early starts are skipped, so that "C_SW-In" / "D_SW-In" hold
    what the reference's savepoints hold -- the fields BEFORE the call.  At the first substep's D_SW-In the area fluxes are still
    the zeros they were allocated with (an early flux preparation would have filled the interior box), at every D_SW-In mfxd
    equals what the previous D_SW-Out left, and the run ends with the same bits as the run without a checkpointer."""
    seen = [dict(n=0, xfx0=None, mfx_out=None, ok=True) for _ in range(6)]

    def make(t):
        rec = seen[t]

        def cp(name, **f):
            if name == "D_SW-In":
                if rec["n"] == 0:
                    rec["xfx0"] = float(np.abs(f["xfxd"].numpy()).max()) + float(np.abs(f["yfxd"].numpy()).max())
                elif rec["mfx_out"] is not None:
                    rec["ok"] = rec["ok"] and np.array_equal(f["mfxd"].numpy(), rec["mfx_out"])
                rec["n"] += 1
            elif name == "D_SW-Out":
                rec["mfx_out"] = f["mfxd"].numpy().copy()

        return cp

    _, with_cp = run_acoustic_six_tiles(emu_lib, "cpu", checkpointers=[make(t) for t in range(6)])
    _, without = run_acoustic_six_tiles(emu_lib, "cpu")
    for t in range(6):
        assert seen[t]["n"] == 2 and seen[t]["xfx0"] == 0.0 and seen[t]["ok"], seen[t]
        for k in with_cp[t]:
            assert np.array_equal(with_cp[t][k], without[t][k], equal_nan=True), (t, k)


def _legacy_solver_child(precision):
    """(child process, PACE_LEGACY_COLUMN_SOLVERS=1 in its environment: the switch is read once per process)"""
    import pickle
    import sys

    from pace_amd import _lib, synthetic
    from oracle import vertical

    lib = _lib.Library(build_emu() if precision == 64 else __import__("test_f32").build_emu_f32())
    out = {}
    # riem_solver3 on the reference-run fixture
    fix = golden("riem_solver3_c12_tile0_call3.npz")
    env = Env(lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)
    got = run_riem3(env, expand_riem_fixture(fix), bool(fix["last_call"]), float(fix["dt"]), float(fix["ptop"]))
    for k in ("delz", "zh", "p", "ppe", "pk3", "pk", "log_p_interface", "w"):
        nk = 79 if k in ("delz", "w") else 80
        ref = fix["out_" + k][:, :, :nk]
        out["riem3." + k] = float(np.abs(ref - got[k][3:15, 3:7, :nk]).max() / (np.abs(ref).max() + 1e-300))
    # riem_solver_c against the oracle on the synthetic state
    from pace_amd.fv3core.stencils.riem_solver_c import NonhydrostaticVerticalSolverCGrid

    n, nz = 12, 20
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cpu", m, n, nz)
    g = oracle_grid(m, n, nz)
    a = {k: s[k].copy() for k in ("cappa", "pt", "q_con", "delp", "zh", "w")}
    hs, ws3 = s["zs"] * 9.80665, np.zeros(s["zs"].shape)
    pef = np.zeros_like(s["pt"])
    q = {k: env.q3(v) for k, v in a.items()}
    qpef, qhs, qws = env.q3(pef), env.q2(hs), env.q2(ws3)
    op = NonhydrostaticVerticalSolverCGrid(env.stencil_factory, env.qf, 0.05)
    op(0.5 * s["dt"], q["cappa"], float(m["ptop"]), qhs, qws, q["pt"], q["q_con"], q["delp"], q["zh"], qpef, q["w"])
    vertical.riem_solver_c(g, 0.5 * s["dt"], a["cappa"], float(m["ptop"]), hs, ws3, a["pt"], a["q_con"], a["delp"], a["zh"], pef, a["w"], p_fac=0.05)
    W = (slice(2, 4 + n), slice(2, 4 + n), slice(0, nz + 1))
    for k, ref, dev in (("gz", a["zh"], q["zh"].numpy()), ("pef", pef, qpef.numpy())):
        out["riem_c." + k] = float(np.abs(ref[W] - dev[W]).max() / (np.abs(ref[W]).max() + 1e-300))
    pickle.dump(out, sys.stdout.buffer)


@pytest.mark.parametrize("precision", [64, 32])
def test_legacy_thread_per_column_solvers_emulated(precision):
    """The thread-per-column kernels of k_riem3.hip -- reached for more than 128 layers or with PACE_LEGACY_COLUMN_SOLVERS=1 --
    after their workspace became double in both builds: riem_solver3 against the reference-run fixture and
    riem_solver_c against the oracle, float64 library at the operator's bound, float32-storage library at float32 accuracy."""
    import pickle
    import subprocess
    import sys

    from helpers import ROOT

    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
            f"import test_emu_kernels as t; t._legacy_solver_child({precision})")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=900, env=dict(os.environ, PACE_LEGACY_COLUMN_SOLVERS="1"))
    assert p.returncode == 0, p.stderr[-3000:].decode()
    errs = pickle.loads(p.stdout)
    for k, e in errs.items():
        tol = 1e-10 if precision == 64 else (5e-3 if k in ("riem3.ppe", "riem3.w") else 2e-5)
        assert e < tol, (k, e, errs)


def dsw_contract_variants(gpu=False):
    """d_sw called TWICE on a synthetic tile -- the halos of delp / pt / w / q_con rewritten between the calls, as a halo update
    would -- by four operators: the reference's in-place contract; `swap_scalar_storage` (the scalar-phase kernel writes to
    spare buffers that are swapped into the caller's Quantities: the second call's output buffers are the first call's inputs,
    stale halos and all); the same with the wind half on the side stream; and `skip_dead_outputs`.  CPU: C24 with the 8 x 8
    tiling of the emulation build `emu-canon` -- the tiling class the scalar-phase kernel takes; gpu: C96 x 12, product library.
    Returns {variant: ({name: whole storage after the second call}, operator)}."""
    from helpers import build_emu_canon
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist

    n, nz = (96, 12) if gpu else (24, 5)
    lib = _lib.load() if gpu else _lib.Library(build_emu_canon())
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda" if gpu else "cpu", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    rng = np.random.default_rng(5)
    halo_noise = {k: 1.0 + 1e-3 * rng.uniform(-1, 1, s[k].shape) for k in ("delp", "pt", "w", "q_con")}
    out = {}
    for name, ctor, call in (("in_place", {}, {}), ("swapped", dict(swap_scalar_storage=True), {}),
                             ("swapped_overlapped", dict(swap_scalar_storage=True), dict(overlap_winds=True)),
                             ("skip_dead", dict(swap_scalar_storage=True), dict(skip_dead_outputs=True)),
                             # (on the GPU: the kinetic energy on the side stream next to vorticity + divergence damping, `ke +=
                             # damped vorticity` left to the fused kernel -- pace_d_sw_overlapped)
                             ("skip_dead_overlapped", dict(swap_scalar_storage=True), dict(skip_dead_outputs=True, overlap_winds=True))):
        op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, get_column_namelist(cfg, env.qf),
                                                False, False, cfg, **ctor)
        f = {k: env.q3(s[k]) for k in DSW_ARGS}
        for rep in range(2):
            op(*[f[k] for k in DSW_ARGS], float(s["dt"]), **call)
            op.join()
            if rep == 0:
                for k, noise in halo_noise.items():  # new halo values (the halo proper: 3 cells around the compute domain, the
                    a = f[k].numpy()                  # model's levels); compute domain untouched
                    b = a.copy()
                    b[:n + 6, :n + 6, :nz] = (a * noise)[:n + 6, :n + 6, :nz]
                    b[3:3 + n, 3:3 + n] = a[3:3 + n, 3:3 + n]
                    f[k].set(b)
                for k in ("delpc", "divgd", "uc", "vc"):  # (c_sw recomputes the work fields before every d_sw)
                    f[k].set(s[k])
        if gpu:
            import torch

            torch.cuda.synchronize()
        out[name] = ({k: f[k].numpy() for k in DSW_ARGS}, op)
    return out


def check_dsw_contract_variants(res):
    ref, op_ref = res["in_place"]
    assert not op_ref._pingpong
    for name in ("swapped", "swapped_overlapped"):
        got, op = res[name]
        assert op._pingpong, "the library must have taken the separate outputs here"
        for k in ref:  # every output of d_sw bit for bit, whole storage (halos included)
            assert np.array_equal(ref[k], got[k], equal_nan=True), (name, k)
    n = ref["delp"].shape[0] - 7
    for name in ("skip_dead", "skip_dead_overlapped"):
        got, _ = res[name]
        for k in ref:
            if k in ("delpc", "divgd", "uc", "vc"):  # (unspecified: include/pace_hip.h PACE_DSW_SKIP_DEAD_OUTPUTS)
                continue
            a, b = ref[k].copy(), got[k].copy()
            if k in ("delp", "pt", "w", "q_con"):
                # ... as are the 3 x 3 corner blocks of the scalars' halo: the full contract leaves the transport's in-place corner
                # copy there (fvtp2d.py:262-345), this one the values that came in
                for x in (a, b):
                    for ci in (slice(0, 3), slice(n + 3, n + 6)):
                        for cj in (slice(0, 3), slice(n + 3, n + 6)):
                            x[ci, cj] = 0.0
            assert np.array_equal(a, b, equal_nan=True), (name, k)


def test_d_sw_separate_outputs_equal_in_place_emulated():
    check_dsw_contract_variants(dsw_contract_variants())


def dsw_launch_structure_switches(gpu=False):
    """The launch structure of d_sw is switchable for A/B measurements: the flux preparation as one launch
    or three (+ the wind halo copy as a launch of its own), the kinetic energy and the vorticity as one launch or two.  Every
    combination must leave the same bits in every argument over the whole storage (the separate-outputs contract, so that the
    wind halo copy -- which moves between launches with the switches -- is compared too)."""
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist
    from helpers import build_emu_canon

    n, nz = (48, 7) if gpu else (24, 4)
    lib = _lib.load() if gpu else _lib.Library(build_emu_canon())
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda" if gpu else "cpu", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    res = {}
    # (PACE_KE_LEVELS: levels per thread of the kinetic energy / vorticity launch -- 2 on large tiles, 1 on small ones by itself)
    for sw in ((), ("PACE_FXADV_SPLIT",), ("PACE_KE_VORT_SPLIT",), ("PACE_FXADV_SPLIT", "PACE_KE_VORT_SPLIT"), ("PACE_KE_LEVELS",)):
        for k in sw:
            os.environ[k] = "2" if k == "PACE_KE_LEVELS" else "1"
        try:
            op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, get_column_namelist(cfg, env.qf),
                                                    False, False, cfg, swap_scalar_storage=True)
            f = {k: env.q3(s[k]) for k in DSW_ARGS}
            op(*[f[k] for k in DSW_ARGS], float(s["dt"]))
            op.join()
            if gpu:
                import torch

                torch.cuda.synchronize()
            assert op._pingpong
            res[sw] = {k: f[k].numpy().copy() for k in DSW_ARGS}
        finally:
            for k in sw:
                os.environ.pop(k, None)
    ref = res[()]
    for sw, got in res.items():
        for k in ref:
            assert np.array_equal(ref[k], got[k], equal_nan=True), (sw, k)


def test_d_sw_launch_structure_switches_are_bit_identical_emulated():
    dsw_launch_structure_switches()


def test_swapped_storage_is_detectable():
    """What holds something derived from a Quantity's storage across a d_sw call with `swap_scalar_storage` can tell that it went
    stale: `Quantity.generation` counts the swaps, a tensor taken from `.data` before the call no longer aliases the Quantity,
    and a halo update whose fields were swapped between start() and wait() refuses to unpack."""
    import torch

    from helpers import build_emu_canon
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist

    n, nz = 24, 3
    lib = _lib.Library(build_emu_canon())
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cpu", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    f = {k: env.q3(s[k]) for k in DSW_ARGS}
    # the reference's contract (the default): storage identity never changes
    op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg)
    alias, ptr, view = f["delp"].data, f["delp"].ptr, f["delp"].transpose(["z", "y", "x"][::-1])
    before = alias.clone()
    op(*[f[k] for k in DSW_ARGS], float(s["dt"]))
    assert f["delp"].generation == 0 and f["delp"].ptr == ptr and f["delp"].data.data_ptr() == alias.data_ptr()
    assert not torch.equal(alias, before) and torch.equal(view.data, f["delp"].data)  # updated in place, aliases follow
    # the extension: the storage is replaced, and that is visible
    op = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg,
                                            swap_scalar_storage=True)
    alias, ptr = f["pt"].data, f["pt"].ptr
    before = alias.clone()
    op(*[f[k] for k in DSW_ARGS], float(s["dt"]))
    assert f["pt"].generation == 1 and f["pt"].ptr != ptr
    assert torch.equal(alias, before), "the old buffer keeps the old values: an alias taken before the call is stale"
    # a halo update started on a field that is then swapped
    from pace_amd.util import CubedSphereCommunicator, CubedSpherePartitioner, NullComm

    cube = CubedSphereCommunicator(NullComm(rank=0, total_ranks=6), CubedSpherePartitioner(), device="cpu", lib=lib)
    a, b = env.qf.zeros(["x", "y", "z"], ""), env.qf.zeros(["x", "y", "z"], "")
    up = cube.get_scalar_halo_updater([env.qf.get_quantity_halo_spec(["x", "y", "z"])])
    up.start([a])
    a.swap_storage(b)
    with pytest.raises(RuntimeError, match="swapped"):
        up.wait()


def test_fvtp2d_kernel_emulated(emu_lib):
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport

    fix = golden("fvtp2d_c12_tile0_call8.npz")
    k_sel = golden("d_sw_c12_tile0_call1.npz")["k_sel"]
    nk = len(k_sel)
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, nk)
    col = column_for_levels(k_sel)
    op = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 6, nord=env.kq(col["nord_t"]),
                               damp_c=env.kq(col["damp_t"]))
    f = {k[3:]: env.q3(v) for k, v in fix.items() if k.startswith("in_")}
    fx, fy = env.q3(), env.q3()
    op(f["q"], f["crx"], f["cry"], f["x_area_flux"], f["y_area_flux"], fx, fy, x_mass_flux=f["x_mass_flux"],
       y_mass_flux=f["y_mass_flux"], mass=f["mass"])
    assert compare(fix["out_q_x_flux"][window(12, 1, 0, nk)], fx.numpy()[window(12, 1, 0, nk)]) < 1e-14
    assert compare(fix["out_q_y_flux"][window(12, 0, 1, nk)], fy.numpy()[window(12, 0, 1, nk)]) < 1e-14


def test_acoustic_dynamics_six_tiles_emulated(emu_lib):
    """One whole AcousticDynamics call (n_split = 2: c_sw, updatedzc, riem_solver_c, p_grad_c, d_sw, updatedzd, riem_solver3,
    pe/pk3 halo, nh_p_grad, ray_fast, del2cubed, heating and all eleven halo-update groups incl. the vector and
    interface ones) on the six C12 tiles, against the reference run's output (tools/make_golden_acoustic.py).
    Every horizontal kernel is bit-exact on its own inputs; what is left is the vertical solvers (exp / log of glibc here,
    numpy's SIMD loops in the reference run; lane-cooperative scans instead of sequential sweeps) carried through two substeps.
    The reference accepts 5e-6 for Riem_Solver3 on every backend (overrides/standard.yaml:49-61): that bound for what the
    solvers feed, per-variable bounds two to three orders above the measured errors for the rest (helpers.ACOUSTIC_TOL; 1e-12
    for masses, temperatures and pressures) -- the same as the GPU twin of this test."""
    fixes, outs = run_acoustic_six_tiles(emu_lib, "cpu")
    from helpers import ACOUSTIC_TOL, ACOUSTIC_TOL_DEFAULT

    for t in range(6):
        for k, e in acoustic_errors(fixes[t], outs[t]).items():
            assert e < ACOUSTIC_TOL.get(k, ACOUSTIC_TOL_DEFAULT), (t, k, e)


def test_acoustic_dynamics_variant_six_tiles_emulated(emu_lib):
    """The same call with nord = 2, d_con = 0 and all advection orders 5 (c_sw's divergence with nord, two damping passes,
    second-order del-n damping, no dissipative heating, the order-5 kernels everywhere) against the reference's run of that
    namelist (tools/make_golden_acoustic.py v2)."""
    from helpers import ACOUSTIC_TOL, ACOUSTIC_TOL_DEFAULT

    fixes, outs = run_acoustic_six_tiles(emu_lib, "cpu", variant="v2")
    for t in range(6):
        for k, e in acoustic_errors(fixes[t], outs[t]).items():
            assert e < ACOUSTIC_TOL.get(k, ACOUSTIC_TOL_DEFAULT), (t, k, e)


def test_tracer_advection_six_tiles_emulated(emu_lib):
    """TracerAdvection (monotone ord-8 PPM transport, sub-cycling, tracer halo updates) on the six C12 tiles against the
    reference's own run (tools/make_golden_tracer.py): bit for bit -- no transcendental is involved."""
    fixes, outs = run_tracer_six_tiles(emu_lib, "cpu")
    check_tracer_outputs(fixes, outs)


@pytest.fixture(scope="module")
def emu_small_lib():
    from pace_amd import _lib

    return _lib.Library(build_emu_small())


@pytest.mark.parametrize("name,tile", [("d_sw_c12_tile0_call1.npz", 0), ("d_sw_c12_tile1_call3.npz", 1)])
def test_d_sw_small_tiles_bit_exact(emu_small_lib, name, tile):
    """4 x 4 tiles: 3 x 3 workgroups per level at C12, the middle one runs the interior (no edge logic) variants of the
    transport / damping kernels; every tile seam is crossed.  Must equal the reference run bit for bit like the big tiles."""
    fix = golden(name)
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    env = Env(emu_small_lib, "cpu", golden(f"grid_c12_tile{tile}.npz"), 12, nk)
    out, _ = run_d_sw(env, column_for_levels(k_sel), {k: fix["in_" + k] for k in DSW_ARGS}, float(fix["dt"]),
                      ut0=fix["in_uc_contra"], vt0=fix["in_vc_contra"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        err = compare(fix["out_" + k][dsw_window(k, 12, nk)], out[k][dsw_window(k, 12, nk)])
        assert err < 3.2e-10, (k, err)


@pytest.mark.parametrize("cfg", [dict(), dict(hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5)])
def test_d_sw_canonical_edge_tiling_vs_oracle(cfg):
    """The tiling the production sizes run with: every tile edge on a workgroup-tile boundary, so the transport kernels take
    the canonical edge path (k_fvtp2d<..., CANON = true>, common.h ppm_run_canon: interior PPM form + three patched interface
    values in the first / last run).  8 x 8 tiles with runs of 3 at C24 (3 x 3 workgroups: corner, edge and interior
    variants), all of d_sw -- the five transport modes, hord 6 and 5 -- against the oracle, exactly."""
    from helpers import DSW_CFG, build_emu_canon
    from oracle import dgrid_sw
    from pace_amd import _lib, synthetic

    lib = _lib.Library(build_emu_canon())
    n, nz = 24, 5
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = column_for_levels(np.arange(nz))
    env = Env(lib, "cpu", metrics, n, nz)
    c = dict(DSW_CFG, **cfg)
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"], cfg=c)
    from helpers import oracle_grid

    g = oracle_grid(metrics, n, nz)
    st = dgrid_sw.DSWState(s["u"].shape)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, c, st, *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nz)
        assert np.array_equal(a[k][W], out[k][W]), (k, compare(a[k][W], out[k][W]))


def test_d_sw_c48_takes_the_16_x_24_tile_shape_vs_oracle(emu_lib):
    """C48 (BASELINE configuration 2) is a multiple of 16 and of 24 but not of 32: the lean transport / scalar-phase kernels are
    compiled a second time for a 16 x 24 tile (csrc/k_fvt16.hip).  All of d_sw at C48 x 3 with the default emulation build (its
    tile shapes are the product's), against the oracle, exactly; and the library really took the fused path."""
    from helpers import DSW_CFG, oracle_grid
    from oracle import dgrid_sw
    from pace_amd import synthetic

    n, nz = 48, 3
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = column_for_levels(np.arange(nz))
    env = Env(emu_lib, "cpu", metrics, n, nz)
    out, op = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"], cfg=DSW_CFG)
    import ctypes as C

    assert emu_lib.cdll.pace_d_sw_wind_outputs_supported(C.byref(op._geom), C.byref(op._cfg)) == 1
    g = oracle_grid(metrics, n, nz)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, DSW_CFG, dgrid_sw.DSWState(s["u"].shape), *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k != "zh":
            W = dsw_window(k, n, nz)
            assert np.array_equal(a[k][W], out[k][W]), (k, compare(a[k][W], out[k][W]))


@pytest.mark.parametrize("which", ["big", "small"])
def test_ord8_transport_emulated_vs_oracle(emu_lib, emu_small_lib, which):
    """Monotone (ord 8) PPM transport against the oracle, bit for bit, with both tilings."""
    from oracle import ppm_transport as tr
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport

    lib = emu_lib if which == "big" else emu_small_lib
    fix = golden("fvtp2d_c12_tile0_call8.npz")
    nk = len(golden("d_sw_c12_tile0_call1.npz")["k_sel"])
    metrics = golden("grid_c12_tile1.npz")
    env = Env(lib, "cpu", metrics, 12, nk)
    g = oracle_grid(metrics, 12, nk)
    op = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 8)
    f = {k[3:]: env.q3(v) for k, v in fix.items() if k.startswith("in_")}
    fx, fy = env.q3(), env.q3()
    op(f["q"], f["crx"], f["cry"], f["x_area_flux"], f["y_area_flux"], fx, fy, x_mass_flux=f["x_mass_flux"], y_mass_flux=f["y_mass_flux"])
    ofx, ofy = np.zeros_like(fix["in_q"]), np.zeros_like(fix["in_q"])
    tr.fvtp2d(g, fix["in_q"].copy(), fix["in_crx"], fix["in_cry"], fix["in_x_area_flux"], fix["in_y_area_flux"], ofx, ofy, 8,
              x_mass_flux=fix["in_x_mass_flux"], y_mass_flux=fix["in_y_mass_flux"])
    assert np.array_equal(ofx[window(12, 1, 0, nk)], fx.numpy()[window(12, 1, 0, nk)])
    assert np.array_equal(ofy[window(12, 0, 1, nk)], fy.numpy()[window(12, 0, 1, nk)])


@pytest.mark.parametrize("name", sorted(__import__("helpers").REMAP_CASES))
def test_map_single_kernels_emulated(emu_lib, name):
    """k_remap.hip through the host class MapSingle against the run of the reference's MapSingle: bit for bit (the three
    kernels contain no transcendental)."""
    from helpers import REMAP_KM, run_map_single

    d = golden("remap_c12.npz")
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, REMAP_KM)
    out = run_map_single(env, name, d)
    assert np.array_equal(out, d[name + "_out"][:, :, :REMAP_KM])


def test_fillz_kernel_emulated(emu_lib):
    """k_fillz through FillNegativeTracerValues against the reference run (three tracers in one launch): bit for bit."""
    from helpers import REMAP_KM

    from pace_amd.fv3core.stencils.fillz import FillNegativeTracerValues

    d = golden("remap_c12.npz")
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, REMAP_KM)

    def embed(a):
        full = np.full((19, 19, REMAP_KM + 1), np.nan)
        full[3:15, 3:15, :] = a
        return env.q3(full)

    names = ["qvapor", "qliquid", "qrain"]
    trs = {nm: embed(d[f"fillz{t}_in"]) for t, nm in enumerate(names)}
    FillNegativeTracerValues(env.stencil_factory, env.qf, 3, trs)(embed(d["fillz_dp"]), trs)
    for t, nm in enumerate(names):
        assert np.array_equal(trs[nm].numpy()[3:15, 3:15, :REMAP_KM], d[f"fillz{t}_out"][:, :, :REMAP_KM]), nm
    with pytest.raises(KeyError):
        FillNegativeTracerValues(env.stencil_factory, env.qf, 4, trs)


@pytest.mark.parametrize("kord", [9, 10])
def test_mapn_tracer_emulated_vs_oracle(emu_lib, kord):
    """MapNTracer: seven tracers through the batched remap kernels (kord 10: two groups, tracer 5 stays kord 9) + fillz,
    against the oracle's per-tracer map_single + fillz: bit for bit."""
    from helpers import REMAP_KM, run_mapn_tracer

    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, REMAP_KM)
    got, exp = run_mapn_tracer(env, golden("remap_c12.npz"), kord)
    for t, (g, e) in enumerate(zip(got, exp)):
        assert np.array_equal(g, e), t


def test_lagrangian_to_eulerian_order_10_emulated(emu_lib):
    """LagrangianToEulerian with every remapping order 10 on the inputs of the reference's own run with that namelist (negatives in
    four condensates): everything without exp / log bit for bit -- all tracers, winds, w, delz --; pt and pkz are mapped in
    log-pressure, and the kord 10 limiter amplifies the last-ulp difference of libm's log against numpy's (measured 5e-9)."""
    from helpers import check_l2e, l2e_k10_fixture, run_l2e

    d = l2e_k10_fixture()
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)
    worst = check_l2e(run_l2e(env, d, False, kord=10), d, False, 1e-14, loose={"pt": 1e-6, "pkz": 1e-6})
    for name in ("delp", "delz", "u", "v", "w", "q_con", "pe", "cappa", "ps", "tr_qvapor", "tr_qliquid", "tr_qrain", "tr_qice",
                 "tr_qsnow", "tr_qgraupel", "tr_qo3mr", "tr_qsgs_tke"):
        assert worst[name] == 0.0, name


@pytest.mark.parametrize("last_step", [False, True])
def test_lagrangian_to_eulerian_emulated(emu_lib, last_step):
    """The whole LagrangianToEulerian host sequence (k_l2e.hip + the remap / fillz kernels) against the run of the
    reference: exact for everything that involves no exp / log, 1e-14 for pt, peln, pk, pkz (libm vs numpy)."""
    from helpers import check_l2e, run_l2e

    d = golden("l2e_c12.npz")
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)
    worst = check_l2e(run_l2e(env, d, last_step), d, last_step, 1e-14)
    for name in ("delp", "delz", "u", "v", "w", "q_con", "pe", "cappa", "tr_qvapor", "tr_qsgs_tke", "ps"):
        if not last_step:
            assert worst[name] == 0.0, name


def test_neg_adj3_kernels_emulated(emu_lib):
    """AdjustNegativeTracerMixingRatio (k_fix_neg_water + the four column operators in one launch) against the run of the
    reference on a state full of negative mixing ratios: bit for bit."""
    from pace_amd.fv3core.stencils.neg_adj3 import AdjustNegativeTracerMixingRatio

    d = golden("negadj_c12.npz")
    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, 79)

    def embed(a):
        full = np.full((19, 19, 80), np.nan)
        full[3:15, 3:15, :] = a
        return env.q3(full)

    names = ["qvapor", "qliquid", "qrain", "qsnow", "qice", "qgraupel", "qcld"]
    f = {k: embed(d["in_" + k]) for k in names + ["pt", "delp"]}
    AdjustNegativeTracerMixingRatio(env.stencil_factory, env.qf, False, False)(*[f[k] for k in names], f["pt"], f["delp"])
    for k in names + ["pt"]:
        assert np.array_equal(f[k].numpy()[3:15, 3:15, :79], d["out_" + k][:, :, :79]), k
    with pytest.raises(NotImplementedError):
        AdjustNegativeTracerMixingRatio(env.stencil_factory, env.qf, True, False)


def test_dynamical_core_step_six_tiles_emulated(emu_lib):
    """One whole DynamicalCore.step_dynamics (fv_setup, pt adjustment, AcousticDynamics with n_split = 2, TracerAdvection of
    eight tracers, LagrangianToEulerian, omega + its halo update and hyperdiffusion, neg_adj3, CubedToLatLon with its vector
    halo update) on the six C12 tiles against the run of the reference's DynamicalCore (tools/make_golden_dycore.py)."""
    from helpers import check_dycore, run_dycore_six_tiles

    class Recorder:
        """Stands in for pace.util.Checkpointer: called with a savepoint name and the variables as keyword arguments."""

        def __init__(self):
            self.calls = []

        def __call__(self, savepoint_name, **kwargs):
            self.calls.append((savepoint_name, {k: (tuple(v.dims), tuple(v.shape)) for k, v in kwargs.items()}))

    recs = [Recorder() for _ in range(6)]
    fixes, outs = run_dycore_six_tiles(emu_lib, "cpu", checkpointers=recs)
    check_dycore(fixes, outs)
    # the reference's checkpoint call sites, in its order (fv_dynamics.py:436-575), with [x, z, y] views of pe / peln
    names = [c[0] for c in recs[0].calls]
    acoustic = ["C_SW-In", "C_SW-Out", "D_SW-In", "D_SW-Out"] * 2  # n_split = 2 (dyn_core.py:744-850)
    assert names == ["FVDynamics-In"] + acoustic + ["Tracer2D1L-In", "Tracer2D1L-Out", "Remapping-In", "Remapping-Out",
                                                     "FVDynamics-Out"]
    rin = dict(recs[0].calls)["Remapping-In"]
    assert rin["pe"] == (("x", "z_interface", "y"), (19, 80, 19)) and set(rin) >= {"pt", "delp", "peln", "cappa", "wsd", "dp1"}


@pytest.mark.parametrize("cfg", [dict(hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5), dict(hord_dp=5, hord_tm=6, hord_vt=5, hord_mt=6),
                                 dict(d_con=0.0), dict(nord=2)])
def test_d_sw_other_namelists_emulated_vs_oracle(emu_lib, cfg):
    """d_sw with advection order 5 (all / mixed with 6), without dissipative heating, with damping order 2: the emulated
    kernels against the oracle, bit for bit (the ord-6 fixtures never reach these template instantiations)."""
    from helpers import DSW_CFG

    from oracle import dgrid_sw
    from pace_amd import synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import get_column_namelist

    n, nz = 12, 10
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(emu_lib, "cpu", metrics, n, nz)
    full = dict(DSW_CFG, **cfg)
    colq = get_column_namelist(DGridShallowWaterLagrangianDynamicsConfig(**full), env.qf)
    col = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v))[:nz] for k, v in colq.items()}
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"], cfg=full)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(oracle_grid(metrics, n, nz), col, full, dgrid_sw.DSWState(s["u"].shape), *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        assert compare(a[k][dsw_window(k, n, nz)], out[k][dsw_window(k, n, nz)]) == 0.0, (cfg, k)


@pytest.mark.parametrize("nz", [32, 91, 127])
def test_other_level_counts_emulated_vs_oracle(emu_lib, nz):
    """Nothing in d_sw / riem_solver3 may assume the 79 layers of the baseline configuration: 32, 91 and 127 layers against
    the oracle (d_sw bit for bit, the column solver to the exp / log rounding)."""
    from helpers import DSW_CFG

    from oracle import dgrid_sw, vertical
    from pace_amd import synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import get_column_namelist

    n = 12
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(emu_lib, "cpu", metrics, n, nz)
    colq = get_column_namelist(DGridShallowWaterLagrangianDynamicsConfig(**DSW_CFG), env.qf)
    col = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v))[:nz] for k, v in colq.items()}
    out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
    g = oracle_grid(metrics, n, nz)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(g, col, DSW_CFG, dgrid_sw.DSWState(s["u"].shape), *[a[k] for k in DSW_ARGS], s["dt"])
    for k in DSW_ARGS:
        if k != "zh":
            assert compare(a[k][dsw_window(k, n, nz)], out[k][dsw_window(k, n, nz)]) == 0.0, k
    inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": a["q_con"], "delp": a["delp"],
           "pt": a["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
           "log_p_interface": s["peln"], "w": a["w"]}
    got = run_riem3(env, inp, False, s["dt"], metrics["ptop"])
    b = {k: v.copy() for k, v in inp.items()}
    vertical.riem_solver3(g, False, s["dt"], b["cappa"], metrics["ptop"], b["zs"], b["ws"], b["delz"], b["q_con"], b["delp"], b["pt"],
                          b["zh"], b["p"], b["ppe"], b["pk3"], b["pk"], b["log_p_interface"], b["w"], p_fac=0.05)
    for k in ("delz", "zh", "ppe", "pk3", "w"):
        nk = nz if k in ("delz", "w") else nz + 1
        assert compare(b[k][window(n, 0, 0, nk)], got[k][window(n, 0, 0, nk)], near_zero=1e-9 * float(np.abs(b[k]).max())) < 5e-6, k


@pytest.mark.parametrize("n", [13, 16, 40])
def test_riem_column_windows_emulated_vs_oracle(emu_lib, n):
    """The column solver's sixteen-column windows (k_riem3f.hip ColumnWindows) in each of their shapes: the whole row inside
    one window with only a head (13), head and tail side by side in one workgroup (16: the shape of C48 / C96 / C192), head, a
    whole window and a tail that do not fit together (40) -- riem_solver3 on the compute domain and riem_solver_c on
    compute + 1 (other window bounds), against the oracle."""
    from oracle import vertical
    from pace_amd import synthetic
    from pace_amd.fv3core.stencils.riem_solver_c import NonhydrostaticVerticalSolverCGrid

    nz = 33
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(emu_lib, "cpu", metrics, n, nz)
    g = oracle_grid(metrics, n, nz)
    inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": s["q_con"], "delp": s["delp"],
           "pt": s["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
           "log_p_interface": s["peln"], "w": s["w"]}
    got = run_riem3(env, inp, True, s["dt"], metrics["ptop"])
    b = {k: v.copy() for k, v in inp.items()}
    vertical.riem_solver3(g, True, s["dt"], b["cappa"], metrics["ptop"], b["zs"], b["ws"], b["delz"], b["q_con"], b["delp"], b["pt"],
                          b["zh"], b["p"], b["ppe"], b["pk3"], b["pk"], b["log_p_interface"], b["w"], p_fac=0.05)
    for k in ("delz", "zh", "ppe", "pk3", "w", "p", "pk", "log_p_interface"):
        nk = nz if k in ("delz", "w") else nz + 1
        assert compare(b[k][window(n, 0, 0, nk)], got[k][window(n, 0, 0, nk)], near_zero=1e-9 * float(np.abs(b[k]).max())) < 5e-6, k
        # nothing outside the compute domain is written
        outside = np.ones(b[k].shape, dtype=bool)
        outside[window(n, 0, 0, nk)] = False
        outside[:, :, nk:] = False
        assert np.array_equal(got[k][outside], inp[k][outside]), k
    # riem_solver_c: compute + 1
    solver = NonhydrostaticVerticalSolverCGrid(env.stencil_factory, env.qf, 0.05)
    a = {k: s[k].copy() for k in ("cappa", "pt", "q_con", "delp", "zh", "w")}
    f = {k: env.q3(v) for k, v in a.items()}
    hs = s["zs"] * 9.80665
    ws3 = np.ascontiguousarray(s["ws"])
    pef = env.q3(np.zeros_like(s["zh"]))
    solver(0.5 * s["dt"], f["cappa"], float(metrics["ptop"]), env.q2(hs), env.q2(ws3), f["pt"], f["q_con"], f["delp"], f["zh"], pef, f["w"])
    ref_pef = np.zeros_like(s["zh"])
    vertical.riem_solver_c(g, 0.5 * s["dt"], a["cappa"], float(metrics["ptop"]), hs, ws3, a["pt"], a["q_con"], a["delp"], a["zh"], ref_pef,
                           a["w"], p_fac=0.05)
    W = (slice(2, 4 + n), slice(2, 4 + n), slice(0, nz + 1))
    assert compare(ref_pef[W], pef.numpy()[W]) < 5e-14
    assert compare(a["zh"][W], f["zh"].numpy()[W]) < 5e-14


@pytest.mark.parametrize("order", [2, 4])
def test_c2l_and_preamble_kernels_emulated_vs_oracle(emu_lib, order):
    """pace_c2l_ord (both orders), pace_fv_setup_pt and pace_omega_from_w against oracle/dycore_parts.py on a synthetic tile:
    the wind transform bit for bit, the preamble to the rounding of exp / log."""
    import ctypes as C

    from oracle import constants as oc
    from oracle import dycore_parts
    from pace_amd import synthetic
    from pace_amd.fv3core.stencils._common import dptr
    from pace_amd.fv3core.stencils.fillz import pointer_table
    from helpers import dycore_condensates

    n, nz = 12, 10
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    env = Env(emu_lib, "cpu", metrics, n, nz)
    from pace_amd.util.grid import geom_struct

    geom = geom_struct(env.qf)
    gd = env.grid_data
    u, v, ua, va = env.q3(s["u"]), env.q3(s["v"]), env.q3(), env.q3()
    emu_lib.call("pace_c2l_ord", C.byref(geom), C.byref(gd.c_struct()), order, dptr(u), dptr(v), dptr(gd.a11), dptr(gd.a12),
                 dptr(gd.a21), dptr(gd.a22), dptr(ua), dptr(va), None)
    fn = dycore_parts.c2l_ord2 if order == 2 else dycore_parts.c2l_ord4
    rua, rva = fn(s["u"], s["v"], metrics["dx"], metrics["dy"], metrics["a11"], metrics["a12"], metrics["a21"], metrics["a22"], n, nz)
    h = 1 if order == 2 else 0
    w = (slice(3 - h, 3 + n + h), slice(3 - h, 3 + n + h), slice(0, nz))
    assert np.array_equal(ua.numpy()[w], rua[w]) and np.array_equal(va.numpy()[w], rva[w])
    if order == 4:
        return
    # compute_preamble + omega
    cond = dycore_condensates(0, s["delp"].shape)
    t = {k: np.abs(cond[k]) for k in ("qliquid", "qrain", "qice", "qsnow", "qgraupel")}
    t["qvapor"] = 0.01 * np.ones_like(s["delp"])
    qt = {k: env.q3(a) for k, a in t.items()}
    pt0 = s["pt"] * 300.0
    f = {k: env.q3(a) for k, a in (("pt", pt0), ("delp", s["delp"]), ("delz", s["delz"]), ("w", s["w"]))}
    q_con, pkz, cappa, dp1, omga = env.q3(), env.q3(), env.q3(), env.q3(), env.q3()
    water = pointer_table([qt[k] for k in ("qvapor", "qliquid", "qrain", "qsnow", "qice", "qgraupel")])
    emu_lib.call("pace_fv_setup_pt", C.byref(geom), water, dptr(q_con), dptr(pkz), dptr(f["pt"]), dptr(cappa), dptr(f["delp"]),
                 dptr(f["delz"]), dptr(dp1), None)
    emu_lib.call("pace_omega_from_w", C.byref(geom), dptr(f["delp"]), dptr(f["delz"]), dptr(f["w"]), dptr(omga), None)
    cw = (slice(3, 3 + n), slice(3, 3 + n), slice(0, nz))
    tw = {k: a[cw] for k, a in t.items()}
    gz, _, rpkz, rcappa, rdp1 = dycore_parts.fv_setup(tw, pt0[cw], s["delp"][cw], s["delz"][cw])
    assert np.array_equal(q_con.numpy()[cw], gz) and np.array_equal(cappa.numpy()[cw], rcappa) and np.array_equal(dp1.numpy()[cw], rdp1)
    assert compare(rpkz, pkz.numpy()[cw]) < 1e-14
    rpt = pt0[cw] * (1.0 + rdp1) * (1.0 - gz) / rpkz
    assert compare(rpt, f["pt"].numpy()[cw]) < 1e-14
    assert np.array_equal(omga.numpy()[cw], s["delp"][cw] / s["delz"][cw] * s["w"][cw])
    assert oc.ZVIR > 0


def _lone_dycore(emu_lib):
    """The reference's drop-in test setup (tests/main/fv3core/test_dycore_call.py:23-137) with what exists here: rank 0 of a
    six-rank NullComm (halo updates receive zeros), the C12 tile-0 metrics and state of the fixtures."""
    import datetime

    from helpers import acoustic_config, dycore_condensates
    from pace_amd.fv3core import DycoreState, DynamicalCore, DynamicalCoreConfig
    from pace_amd.util import CubedSphereCommunicator, NullComm

    n, nz = 12, 79
    fa, fd = golden("acoustic_c12_tile0.npz"), golden("dycore_c12_tile0.npz")
    env = Env(emu_lib, "cpu", {k[5:]: v for k, v in fa.items() if k.startswith("grid_")}, n, nz)
    cube = CubedSphereCommunicator(NullComm(rank=0, total_ranks=6, fill_value=0.0), device="cpu", lib=emu_lib)
    arrays = {k: fa["in_" + k] for k in "u v w delz delp pe pk peln phis uc vc ua va".split()}
    shape = arrays["delp"].shape
    for name, key in (("pt", "in_pt"), ("qvapor", "in_qvapor")):
        a = np.zeros(shape)
        a[3:3 + n, 3:3 + n, :] = fd[key]
        arrays[name] = a
    arrays["ps"] = fd["in_ps"]
    for name, f in dycore_condensates(0, shape).items():
        arrays[name] = f * (arrays["delp"] > 0)
    state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
    config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=225.0, k_split=1, n_split=1,
                                 acoustic_dynamics=acoustic_config(1))
    core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                         datetime.timedelta(seconds=225.0))
    return core, state, arrays


def _snapshot(state):
    from helpers import DYCORE_OUT

    return {k: getattr(state, k).numpy().copy() for k in DYCORE_OUT}


def test_dycore_call_is_deterministic_and_stateless(emu_lib):
    """The two properties the reference's own drop-in test checks (test_dycore_call.py:146-210): two identically initialised
    dycores on identical states give identical results, and a dycore called again on the same input gives the same output --
    it retains nothing between calls that influences the result."""
    core1, state1, arrays = _lone_dycore(emu_lib)
    core2, state2, _ = _lone_dycore(emu_lib)
    core1.step_dynamics(state1)
    first = _snapshot(state1)
    core2.step_dynamics(state2)
    second = _snapshot(state2)
    for k in first:
        assert np.array_equal(first[k], second[k], equal_nan=True), k
    # same dycore, the input restored
    from pace_amd.fv3core.initialization.dycore_state import _FIELDS

    for name in _FIELDS:
        q = getattr(state1, name)
        if name in arrays:
            q.set(arrays[name])
        else:
            q.data[...] = 0.0
    core1.step_dynamics(state1)
    third = _snapshot(state1)
    for k in first:
        assert np.array_equal(first[k], third[k], equal_nan=True), k


def test_dynamical_core_two_remapping_steps_emulated(emu_lib):
    """k_split = 2 (n_split = 1): two AcousticDynamics calls of which only the second is the end step, two
    LagrangianToEulerian calls of which only the second is the last -- against the reference's run of that configuration
    (tools/make_golden_dycore.py 1 2)."""
    from helpers import check_dycore, run_dycore_six_tiles

    fixes, outs = run_dycore_six_tiles(emu_lib, "cpu", prefix="dycore_k2_c12")
    check_dycore(fixes, outs, default=1e-9)


def test_dynamical_core_step_remapping_order_10_emulated(emu_lib):
    """One whole step with every remapping order 10 (kord_tm = -10, kord_tr = kord_wz = kord_mt = 10: the other limiter family of
    RemapProfile for temperature, tracers, w, delz and the winds) against the reference's run of that namelist
    (tools/make_golden_dycore.py 2 1 kord10), within what that limiter's discontinuity allows (helpers.KORD10_TOL; the operator
    on the reference's own inputs: test_lagrangian_to_eulerian_order_10_*)."""
    from helpers import check_dycore_kord10, run_dycore_six_tiles

    fixes, outs = run_dycore_six_tiles(emu_lib, "cpu", prefix="dycore_kord10_c12")
    check_dycore_kord10(fixes, outs)


def test_d_sw_order5_emulated_against_reference_run(emu_lib):
    """The <5, ...> transport / kinetic-energy kernels against a run of the reference with hord_* = 5: bit for bit."""
    from helpers import run_d_sw_h5_fixture

    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, len(golden("d_sw_h5_c12_tile0_call1.npz")["k_sel"]))
    assert run_d_sw_h5_fixture(env) == 0.0


@pytest.mark.parametrize("variant", ["nord2", "dcon0", "skeb", "dddmp0"])
def test_d_sw_namelist_variants_emulated_against_reference_run(emu_lib, variant):
    """d_sw with one namelist option changed (nord = 2: two damping passes in the fused divergence-damping kernel and second-order
    del-n damping; d_con = 0; do_skeb; dddmp = 0) against runs of the reference with that namelist: bit for bit."""
    from helpers import dsw_variant_fixture, run_d_sw_variant_fixture

    env = Env(emu_lib, "cpu", golden("grid_c12_tile0.npz"), 12, dsw_variant_fixture(variant)[3])
    errs = run_d_sw_variant_fixture(env, variant)
    assert max(errs.values()) == 0.0, errs


@pytest.mark.parametrize("which,n,nz", [("big", 12, 10), ("small", 24, 12)])
def test_operator_chain_emulated_vs_oracle(emu_lib, emu_small_lib, which, n, nz):
    """Every operator of the acoustic loop body, one at a time, on the oracle's inputs for that operator (tests/opchain.py):
    d2a2c_vect, c_sw, updatedzc, riem_solver_c, p_grad_c, d_sw, updatedzd, riem_solver3, edge_pe, pk3_halo,
    compute_geopotential, nh_p_grad, ray_fast, del2cubed, apply_diffusive_heating.  Under emulation everything without a
    transcendental must equal the oracle bit for bit (with 4 x 4 tiles at C24: interior workgroups and every tile seam)."""
    from opchain import Chain, ProductOps, check_case

    chain = Chain(n, nz)
    ops = ProductOps(emu_lib if which == "big" else emu_small_lib, "cpu", chain)
    exact = {"d2a2c_vect", "c_sw", "updatedzc", "p_grad_c", "d_sw", "updatedzd", "edge_pe", "compute_geopotential", "nh_p_grad",
             "ray_fast", "del2cubed"}
    seen = []
    for case in chain.cases():
        errs = check_case(ops, case)
        seen.append(case.name)
        if case.name in exact:
            assert all(e == 0.0 for e in errs.values()), (case.name, errs)
    assert len(seen) == 15


def test_acoustic_loop_six_synthetic_tiles_emulated_vs_oracle(emu_lib):
    """The whole AcousticDynamics call (two substeps, every operator fed by its predecessor, all halo-update groups) on six
    synthetic tiles against oracle/dyn_core.py -- the CPU twin of the C96 x 79 GPU test."""
    import opchain

    n, nz, n_split = 12, 10, 2
    ref = opchain.oracle_loop(n, nz, n_split, 3.5 * n_split)
    got = opchain.product_loop(emu_lib, "cpu", n, nz, n_split, 3.5 * n_split)
    detail = {}
    for k, e in opchain.loop_errors(ref, got, n, nz, detail=detail).items():
        assert e < opchain.LOOP_TOL.get(k, 1e-9), (k, e)
        assert detail[k]["max_abs_error_over_magnitude"] < opchain.LOOP_ABS_SYNTHETIC, (k, detail[k])


def test_standalone_ppm_and_divergence_damping_emulated_vs_oracle(emu_lib):
    """XPiecewiseParabolic, YPiecewiseParabolic (orders 5, 6, 8), DivergenceDamping and Sim1Solver as stand-alone classes (the reference
    tests them on their own: TranslateXPPM / YPPM / DivergenceDamping): bit for bit against the oracle."""
    from opchain import check_standalone_operators

    check_standalone_operators(emu_lib, "cpu", 12, 8, exact=True)


def test_dynamical_core_step_from_generated_inputs_emulated(emu_lib):
    """End to end with nothing reference-derived: grid metrics from pace_amd.util.gridgen, state from pace_amd's
    init_baroclinic_state, one whole DynamicalCore.step_dynamics on six tiles -- against the OUTPUT of the reference's run
    (which started from the reference's own MetricTerms and init_baroclinic_state), within the envelope the reference
    algorithm itself has for 1e-13 m/s of wind noise (helpers.GENERATED_TOL).  (Generated metrics + the reference run's
    initial state, at the usual DynamicalCore tolerances: tests/test_guard_pages.py.)"""
    from helpers import check_dycore_generated, run_dycore_six_tiles

    fixes, outs = run_dycore_six_tiles(emu_lib, "cpu", generated="all")
    check_dycore_generated(fixes, outs)


def test_two_emulation_libraries_in_one_process_do_not_share_symbols():
    """the emulation libraries used to export their kernels' emulated `__shared__` statics as STB_GNU_UNIQUE
    symbols, which the loader unifies process-wide even under RTLD_LOCAL -- `libpace_emu_canon.so` (8 x 8 tiles) and
    `libpace_emu.so` (64 x 16) then shared LDS arrays of different sizes, and a test that ran after another library had been
    loaded failed.  Now: `-fno-gnu-unique` and a version script that exports the C ABI only (pace_amd/csrc/exports.map).  Load
    the canonical-tiling library FIRST, then the default one, and run DelnFlux (whose kernel `k_delnflux<...>` keeps its tile in
    such a static) through both, in both orders of use: each equals the oracle bit for bit."""
    import subprocess

    from helpers import build_emu_canon
    from oracle import ppm_transport
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import column_namelist_arrays
    from pace_amd.fv3core.stencils.delnflux import DelnFlux

    paths = [build_emu_canon(), build_emu()]
    for p in paths:  # nothing but the C ABI is exported, nothing is GNU-unique
        syms = subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True, check=True).stdout.split("\n")
        kinds = {ln.split()[1] for ln in syms if len(ln.split()) == 3}
        names = [ln.split()[2] for ln in syms if len(ln.split()) == 3]
        assert "u" not in kinds and all(nm.startswith("pace_") for nm in names), p
    libs = [_lib.Library(p) for p in paths]
    n, nz = 24, 3
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    g = oracle_grid(m, n, nz)
    col = column_namelist_arrays(DGridShallowWaterLagrangianDynamicsConfig(), nz)
    fx0, fy0 = s["xfx"] * 0.5, s["yfx"] * 0.5
    ref_fx, ref_fy = fx0.copy(), fy0.copy()
    ppm_transport.delnflux(g, s["pt"].copy(), ref_fx, ref_fy, col["nord_v"], col["damp_vt"], float(m["da_min"]), mass=s["delp"])
    for lib in libs + libs[::-1]:
        env = Env(lib, "cpu", m, n, nz)
        q, fx, fy, mass = env.q3(s["pt"]), env.q3(fx0), env.q3(fy0), env.q3(s["delp"])
        op = DelnFlux(env.stencil_factory, env.qf, env.damping, env.grid_data.rarea, env.kq(col["nord_v"]), env.kq(col["damp_vt"]),
                      grid_data=env.grid_data)
        op(q, fx, fy, mass=mass)
        assert np.array_equal(fx.numpy()[window(n, 1, 0, nz)], ref_fx[window(n, 1, 0, nz)]), lib.path
        assert np.array_equal(fy.numpy()[window(n, 0, 1, nz)], ref_fy[window(n, 0, 1, nz)]), lib.path


def test_d_sw_halo_state_memory_form_equals_lds_form():
    """k_divdamp_halo_state (the halo of divgd / uc / vc as the reference's in-place divergence damping leaves it: TranslateD_SW's
    whole-storage windows) has two forms: the band's divergence planes in LDS (tiles up to C320), or in two scratch fields (larger
    tiles; PACE_DDH_MEM=1 forces it).  C24: a band with an inside.  Both against the oracle, whole storage, bit for bit."""
    from helpers import DSW_CFG
    from oracle import dgrid_sw
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from pace_amd.fv3core.stencils.d_sw import column_namelist_arrays

    n, nz = 24, 5
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    col = column_namelist_arrays(DGridShallowWaterLagrangianDynamicsConfig(), nz)
    a = {k: s[k].copy() for k in DSW_ARGS}
    dgrid_sw.d_sw(oracle_grid(m, n, nz), col, DSW_CFG, dgrid_sw.DSWState(s["u"].shape), *[a[k] for k in DSW_ARGS], s["dt"])
    env = Env(_lib.Library(build_emu()), "cpu", m, n, nz)
    for mem in (False, True):
        if mem:
            os.environ["PACE_DDH_MEM"] = "1"
        try:
            out, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
        finally:
            os.environ.pop("PACE_DDH_MEM", None)
        for k in ("divgd", "uc", "vc", "delpc", "delp", "pt", "w", "q_con"):
            assert np.array_equal(a[k][dsw_window(k, n, nz)], out[k][dsw_window(k, n, nz)]), (mem, k)


def test_fxadv_one_launch_equals_the_four_launches_and_the_oracle():
    """FiniteVolumeFluxPrep as ONE launch (the frame's stages and fluxes in one workgroup per level beside the interior's blocks;
    k_fxadv_fused) and as round 5's four launches (PACE_FXADV_SPLIT=1): both against the oracle, bit for bit, on the windows
    TranslateFxAdv compares (translate_fxadv.py:49-72) and on the fluxes' whole domains.  C12 and C24 (levels per thread 1 .. 8)."""
    from oracle import dgrid_sw
    from pace_amd import _lib, synthetic
    from pace_amd.fv3core.stencils.fxadv import FiniteVolumeFluxPrep

    for n, nz in ((12, 3), (24, 9)):
        m = synthetic.tile_metrics(n, nz)
        s = synthetic.acoustic_state(m, n, nz)
        ref = {k: np.zeros_like(s["pt"]) for k in ("crx", "cry", "xfx", "yfx", "ut", "vt")}
        dgrid_sw.fxadv(oracle_grid(m, n, nz), s["uc"], s["vc"], ref["crx"], ref["cry"], ref["xfx"], ref["yfx"], ref["ut"], ref["vt"], s["dt"])
        env = Env(_lib.Library(build_emu()), "cpu", m, n, nz)
        prep = FiniteVolumeFluxPrep(env.stencil_factory, env.grid_data)
        for split in (False, True):
            if split:
                os.environ["PACE_FXADV_SPLIT"] = "1"
            try:
                uc, vc = env.q3(s["uc"]), env.q3(s["vc"])
                out = {k: env.q3() for k in ref}
                prep(uc, vc, out["crx"], out["cry"], out["xfx"], out["yfx"], out["ut"], out["vt"], s["dt"])
            finally:
                os.environ.pop("PACE_FXADV_SPLIT", None)
            wins = {"crx": (slice(3, n + 4), slice(0, n + 6)), "xfx": (slice(3, n + 4), slice(0, n + 6)),
                    "cry": (slice(0, n + 6), slice(3, n + 4)), "yfx": (slice(0, n + 6), slice(3, n + 4)),
                    "ut": (slice(1, n + 6), slice(1, n + 5)), "vt": (slice(1, n + 5), slice(1, n + 6))}
            for k, (wi, wj) in wins.items():
                a, b = ref[k][wi, wj, :nz], out[k].numpy()[wi, wj, :nz]
                assert np.array_equal(a, b), (n, split, k, float(np.abs(a - b).max()))
