"""Shared test plumbing: golden loading, environment construction, the reference's comparison metric."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

import sys

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from pace_amd.tile import (DSW_ARGS, DSW_CFG, DSW_DEAD, RIEM_ARGS, Env, compare, dsw_live_window, dsw_window, run_d_sw, run_riem3,  # noqa: E402,F401
                           window)


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def column_for_levels(k_sel):
    col = golden("column_namelist_c12.npz")
    return {k: np.ascontiguousarray(v[np.asarray(k_sel)]) for k, v in col.items()}


def oracle_grid(metrics, n, nk):
    from oracle._np import Grid

    return Grid(n, nk, dict(metrics))


def _make(target):
    """`make <target>` under an exclusive file lock: several test processes (pytest -n) may ask for the same library at once."""
    import fcntl

    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-s", "-j4", target], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_emu():
    """Build tests/emu/libpace_emu.so (the kernel sources compiled for the CPU).  Test infrastructure."""
    _make("emu")
    return os.path.join(ROOT, "tests", "emu", "libpace_emu.so")


def build_emu_canon():
    """8 x 8 transport tiles, runs of 3: the canonical edge path (ppm_run_canon) at C16 / C24 (Makefile emu-canon)."""
    _make("emu-canon")
    return os.path.join(ROOT, "tests", "emu", "libpace_emu_canon.so")


def build_emu_small():
    """tests/emu/libpace_emu_small.so: the emulation build with 4 x 4 LDS tiles (interior code paths at C12)."""
    _make("emu-small")
    return os.path.join(ROOT, "tests", "emu", "libpace_emu_small.so")


def expand_riem_fixture(fix, n=12, nz=79):
    """The fixture stores 4 rows of columns; tile them over the compute domain (columns are independent)."""
    out = {}
    reps = n // 4
    for key, v in fix.items():
        if not key.startswith("in_"):
            continue
        name = key[3:]
        if v.ndim == 3:
            full = np.zeros((n + 7, n + 7, nz + 1))
            full[3 : 3 + n, 3 : 3 + n] = np.tile(v, (1, reps, 1))
        else:
            full = np.zeros((n + 7, n + 7))
            full[3 : 3 + n, 3 : 3 + n] = np.tile(v, (1, reps))
        out[name] = full
    return out


# ------------------------------------------------------------------------------------------------------------------
# whole-AcousticDynamics runs (6 tiles in one process, one thread per tile)
# ------------------------------------------------------------------------------------------------------------------
ACOUSTIC_OUT = "u v w delz delp pt pe pk peln q_con omga ua va uc vc mfxd mfyd cxd cyd diss_estd heat_source".split()


# (u, v: the meridional wind of an equatorial tile of this zonal-flow case is ~0.1 m/s with zero crossings; the full-field
# fixtures contain entries 1e-8 of that, where a relative metric measures the rounding of the last ulp of the 35 m/s zonal wind)
_BANDS = {"mfxd": 1e-6, "mfyd": 1e-6, "cxd": 1e-6, "cyd": 1e-6, "uc": 1e-8, "vc": 1e-8, "diss_estd": 1e-8, "w": 1e-5, "omga": 1e-5,
          "u": 1e-7, "v": 1e-7}


def acoustic_fixture(t):
    """tests/golden/acoustic_c12_tile<t>.npz + what the reference run left in the work fields uc / vc (a later addition kept
    in acoustic_c12_ucvc.npz, tools/make_golden_acoustic.py; TranslateDynCore compares them, translate_dyncore.py:84-85)."""
    fix = golden(f"acoustic_c12_tile{t}.npz")
    extra = golden("acoustic_c12_ucvc.npz")
    for k in ("uc", "vc"):
        fix["out_" + k], fix["col_" + k] = extra[f"out_{k}_tile{t}"], extra[f"col_{k}_tile{t}"]
    # tiles 0 (equatorial) and 2 (polar) also have their FULL output fields: every level of the compute window + the
    # staggered row / column (acoustic_c12_full.npz, a later addition)
    full = golden("acoustic_c12_full.npz")
    if t in full["tiles"]:
        for k in ACOUSTIC_OUT:
            fix["full_" + k] = full[f"out_{k}_tile{t}"]
    return fix


# Bounds of the whole-AcousticDynamics comparisons (GPU and emulation; reference metric, bands above): the reference's own DynCore
# bound, 2e-6 for EVERY variable (translate_dyncore.py:120) is the ceiling; what is measured far below it is held two to three
# orders above the measurement (GPU, profiles/r06_acoustic_c12_gpu_errors.json: masses / temperatures / pressures <= 1.1e-15, the
# A-grid winds 1.4e-11, the accumulators 1e-12, delz 4e-13, heat_source 3e-9, u / v 1.5e-8, omga 8e-8, vc 1.3e-7, w 4e-7).
# The ONE variable not held to 2e-6 is diss_estd (measured 2.2e-6 .. 2.9e-6): w's damping heating of THIS fixture amplifies an ulp of
# the column solver's operands by ~1e10.  That is a property of the loop, not of the kernels: the numpy oracle ITSELF moves by
# diss_estd 1.35e-6 / w 1.7e-7 when only its two column sums of the interface pressures are formed in another order (extended
# precision, rounded once per level: <= 1 ulp away; tests/test_oracle_golden.py::test_loop_conditioning_of_diss_estd), and the
# device forms those sums AND the two tridiagonal sweeps as parallel scans over sixteen lanes (k_riem3f.hip).  With the sequential
# legacy solvers (PACE_LEGACY_COLUMN_SOLVERS=1, bit-identical arithmetic order) the same test measures diss_estd 1.0e-7, w 5.6e-9 --
# exactly what the oracle gives with exp / log evaluated in extended precision.  Round 6 took the solvers' own exp / log from
# ~1 ulp to <= 0.55 ulp (2.9e-6 -> 2.2e-6); the rest is re-association, which no parallel column solver avoids.
ACOUSTIC_TOL = {"w": 2e-6, "omga": 2e-6, "diss_estd": 3e-6, "uc": 2e-6, "vc": 2e-6, "u": 1e-6, "v": 1e-6, "heat_source": 1e-6,
                "ua": 1e-8, "va": 1e-8, "delz": 1e-10, "mfxd": 1e-9, "mfyd": 1e-9, "cxd": 1e-9, "cyd": 1e-9}
ACOUSTIC_TOL_DEFAULT = 1e-12  # delp, pt, pe, pk, peln, q_con


ACOUSTIC_VARIANTS = {"v2": dict(nord=2, d_con=0.0, hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5)}  # tools/make_golden_acoustic.py


def acoustic_config(n_split, variant=None):
    """The baroclinic_c12 namelist values the fixture was generated with (tools/capture.py dycore_config); `variant`: the
    options changed in acoustic_c12_<variant>.npz."""
    from pace_amd.fv3core import AcousticDynamicsConfig, DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig

    o = dict(nord=3, d_con=1.0, hord_dp=6, hord_tm=6, hord_vt=6, hord_mt=6)
    o.update(ACOUSTIC_VARIANTS.get(variant, {}))
    dsw = DGridShallowWaterLagrangianDynamicsConfig(nord=o["nord"], d_con=o["d_con"], hord_dp=o["hord_dp"], hord_tm=o["hord_tm"],
                                                    hord_vt=o["hord_vt"], hord_mt=o["hord_mt"])
    return AcousticDynamicsConfig(n_split=n_split, k_split=1, nord=o["nord"], d_con=o["d_con"], rf_fast=True, rf_cutoff=3000.0, tau=10.0,
                                  p_fac=0.05, hord_tm=o["hord_tm"], delt_max=0.002, d_grid_shallow_water=dsw,
                                  riemann=RiemannConfig(p_fac=0.05))


def run_acoustic_tile(comm, lib, device, fix, n, nz, variant=None, checkpointer=None):
    """One tile's program: build the environment from the fixture, run one AcousticDynamics call."""
    import torch

    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.dyn_core import AcousticDynamics
    from pace_amd.util import CubedSphereCommunicator

    metrics = {k[5:]: v for k, v in fix.items() if k.startswith("grid_")}
    env = Env(lib, device, metrics, n, nz)
    cube = CubedSphereCommunicator(comm, device=device, lib=lib)
    state = DycoreState.init_from_numpy_arrays({k[3:]: v for k, v in fix.items() if k.startswith("in_") and k != "in_cappa"}, env.qf)
    n_split = int(fix["n_split"])
    wsd = env.q2()
    dyn = AcousticDynamics(cube, env.stencil_factory, env.qf, env.grid_data, env.damping, 0, False, False,
                           acoustic_config(n_split, variant), state.phis, wsd, state, checkpointer=checkpointer)
    dyn.cappa.set(fix["in_cappa"])
    dyn(state, timestep=float(fix["timestep"]), n_map=1)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    out = {k: getattr(state, k).numpy() for k in ACOUSTIC_OUT if k != "heat_source"}
    out["heat_source"] = dyn._heat_source.numpy()
    return out


def acoustic_variant_fixture(t, variant):
    """Inputs of the baseline fixture of tile t, outputs of the reference run with the variant's namelist."""
    fix = {k: v for k, v in acoustic_fixture(t).items() if not k.startswith(("out_", "col_", "full_"))}
    var = golden(f"acoustic_c12_{variant}.npz")
    for k in ACOUSTIC_OUT:
        fix["out_" + k], fix["col_" + k] = var[f"out_{k}_tile{t}"], var[f"col_{k}_tile{t}"]
    return fix


def run_acoustic_six_tiles(lib, device, n=12, nz=79, variant=None, checkpointers=None):
    from pace_amd.util import run_tiles

    fixes = [acoustic_variant_fixture(t, variant) if variant else acoustic_fixture(t) for t in range(6)]
    cps = checkpointers or [None] * 6
    return fixes, run_tiles(6, lambda comm: run_acoustic_tile(comm, lib, device, fixes[comm.Get_rank()], n, nz, variant,
                                                              checkpointer=cps[comm.Get_rank()]))


def acoustic_errors(fix, out, n=12):
    """max reference-metric error per variable on the fixture's level subset and full columns (compute domain + the
    staggered interface row/column)."""
    errs = {}
    ks = fix["k_sel"]
    for k in ACOUSTIC_OUT:
        full = out[k]
        di = 1 if k in ("v", "mfxd", "cxd", "uc") else 0
        dj = 1 if k in ("u", "mfyd", "cyd", "vc") else 0
        kk = [x for x in ks if x < (80 if k in ("pe", "pk", "peln") else 79)]
        idx = [list(ks).index(x) for x in kk]
        got = full[3 : 3 + n + di, 3 : 3 + n + dj][:, :, kk]
        ref = fix["out_" + k][: n + di, : n + dj][:, :, idx]
        # values that are zero by symmetry (e.g. the meridional mass flux on the equator row of an equatorial tile) come
        # out as rounding residue 13+ orders below the field's scale; the reference's translate tests skip those through
        # per-variable near_zero overrides (tests/savepoint/translate/overrides/standard.yaml)
        # (mass / Courant fluxes across the tile's symmetry line cancel to ~1e-7 of the field scale and carry the same
        # ABSOLUTE rounding error as every other row, hence the wider band for the four accumulators)
        # (uc / vc after the call are work-field leftovers: the C-grid winds on the sponge levels -- where the meridional one
        # is rounding residue on half of the tiles of this zonal-flow case -- and ~1e-29 elsewhere; the reference ignores them
        # below 1e-13 ABSOLUTE (overrides/baroclinic.yaml:12-20), here: below 1e-8 of the field's magnitude)
        # (diss_estd is a sum of dissipation terms of both signs: entries 1e-10 of the field's magnitude are residue)
        # (w and omga cross zero; the vertical solver's absolute error is ~1e-11 of their magnitude on every implementation
        # -- device or libm exp / log against numpy's, amplified by the tridiagonal solves -- so a RELATIVE bound of 5e-6 only
        # means something above ~1e-5 of the magnitude; tools/riem_check.py prints both measures)
        band = _BANDS.get(k, 1e-12)
        near_zero = band * float(np.abs(ref).max()) + 1e-300
        e = compare(ref, got, near_zero=near_zero)
        nk = 80 if k in ("pe", "pk", "peln") else 79
        cols = np.stack([full[i, j, :nk] for (i, j) in fix["cols"]])
        e = max(e, compare(fix["col_" + k][:, :nk], cols, near_zero=near_zero))
        if "full_" + k in fix:  # the whole field, all levels
            e = max(e, compare(fix["full_" + k][: n + di, : n + dj, :nk], full[3 : 3 + n + di, 3 : 3 + n + dj, :nk], near_zero=near_zero))
        errs[k] = e
    return errs


# ------------------------------------------------------------------------------------------------------------------
# TracerAdvection on six tiles
# ------------------------------------------------------------------------------------------------------------------
def run_tracer_tile(comm, lib, device, fix, metrics, n):
    import torch

    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport
    from pace_amd.fv3core.stencils.tracer_2d_1l import TracerAdvection
    from pace_amd.util import CubedSphereCommunicator

    nk = len(fix["k_sel"])
    env = Env(lib, device, metrics, n, nk)
    cube = CubedSphereCommunicator(comm, device=device, lib=lib)
    tracers = {"qvapor": env.q3(fix["in_qvapor"]), "q2": env.q3(fix["in_q2"])}
    f = {k: env.q3(fix["in_" + k]) for k in ("dp1", "mfxd", "mfyd", "cxd", "cyd")}
    transport = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 8)
    adv = TracerAdvection(env.stencil_factory, env.qf, transport, env.grid_data, cube, tracers)
    adv(tracers, f["dp1"], f["mfxd"], f["mfyd"], f["cxd"], f["cyd"])
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    out = {k: v.numpy() for k, v in tracers.items()}
    out.update({k: v.numpy() for k, v in f.items()})
    return out


def run_tracer_six_tiles(lib, device, n=12):
    from pace_amd.util import run_tiles

    fixes = [golden(f"tracer_c12_tile{t}.npz") for t in range(6)]
    metrics = [{k[5:]: v for k, v in golden(f"acoustic_c12_tile{t}.npz").items() if k.startswith("grid_")} for t in range(6)]
    outs = run_tiles(6, lambda comm: run_tracer_tile(comm, lib, device, fixes[comm.Get_rank()], metrics[comm.Get_rank()], n))
    return fixes, outs


def check_tracer_outputs(fixes, outs, n=12):
    for t in range(6):
        nk = len(fixes[t]["k_sel"])
        W = (slice(3, 3 + n), slice(3, 3 + n), slice(0, nk))
        for name in ("qvapor", "q2"):
            assert np.array_equal(outs[t][name][W], fixes[t]["out_" + name][W]), (t, name)
        assert np.array_equal(outs[t]["mfxd"][3 : 4 + n, 3 : 3 + n, :nk], fixes[t]["out_mfxd"][3 : 4 + n, 3 : 3 + n, :nk]), t
        assert np.array_equal(outs[t]["cyd"][3 : 3 + n, 3 : 4 + n, :nk], fixes[t]["out_cyd"][3 : 3 + n, 3 : 4 + n, :nk]), t


# ------------------------------------------------------------------------------------------------------------------
# Six-tile GPU runs in a child process (six tiles = six host threads sharing one device; run as coroutines, see
# pace_amd/util/comm.py).  Round 1 retried such a run when the child died from a signal ("seen twice in ~40 runs").  Round 2
# hunted for it: 240 consecutive runs on an MI355X (tools/abort_hunt.py: 140 acoustic, 60 whole-dycore, 40 tracer; fault
# handler on) without a single failure (profiles/r02_abort_hunt.json), so the retry is gone: a child that dies fails the test,
# and its return code + output are kept as an artefact (gpurun_out/child_failure_<what>.txt).
# ------------------------------------------------------------------------------------------------------------------
def _child_main(what, out_path, hard_exit=False):
    import pickle

    from pace_amd import _lib

    lib = _lib.load()
    if what == "acoustic":
        result = run_acoustic_six_tiles(lib, "cuda")
    elif what == "acoustic_v2":
        result = run_acoustic_six_tiles(lib, "cuda", variant="v2")
    elif what == "tracer":
        result = run_tracer_six_tiles(lib, "cuda")
    elif what == "dycore":
        result = run_dycore_six_tiles(lib, "cuda")
    elif what == "dycore_k2":
        result = run_dycore_six_tiles(lib, "cuda", prefix="dycore_k2_c12")
    elif what == "dycore_kord10":
        result = run_dycore_six_tiles(lib, "cuda", prefix="dycore_kord10_c12")
    elif what == "dycore_f32":
        result = run_dycore_six_tiles(_lib.load(32), "cuda")
    elif what == "dycore_c384_f32":
        result = (run_dycore_one_tile_synthetic(_lib.load(32), "cuda", 384, 91), run_dycore_one_tile_synthetic(_lib.load(32), "cuda", 384, 91))
    elif what == "dycore_generated":
        result = (run_dycore_six_tiles(lib, "cuda", generated="metrics"), run_dycore_six_tiles(lib, "cuda", generated="all"))
    else:
        raise ValueError(what)
    with open(out_path, "wb") as f:
        pickle.dump(result, f)
        f.flush()
        os.fsync(f.fileno())
    if hard_exit:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def run_in_child(what, tmp_path):
    import pickle

    out = os.path.join(str(tmp_path), f"{what}.pkl")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
            f"import helpers; helpers._child_main({what!r}, {out!r})")
    p = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        report = f"child run {what!r} failed (rc {p.returncode}):\n{p.stdout[-4000:]}\n{p.stderr[:3000]}\n...\n{p.stderr[-8000:]}"
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, f"child_failure_{what}.txt"), "w") as f:
                f.write(f"child run {what!r} failed (rc {p.returncode}):\n{p.stdout}\n{p.stderr}")
        raise RuntimeError(report)
    with open(out, "rb") as f:
        return pickle.load(f)


def run_dycore_one_tile_synthetic(lib, device, n, nz, n_split=2):
    """One DynamicalCore.step_dynamics of ONE tile at any size behind a lone-rank LoopbackComm (each halo receives what the tile
    itself sent to that neighbour), pace_amd/synthetic.py's balanced state + smooth condensates.  Returns the prognostic fields
    after the step as numpy arrays (compute domain)."""
    import datetime

    import torch

    from pace_amd import synthetic
    from pace_amd.fv3core import DynamicalCoreConfig
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.fv_dynamics import DynamicalCore
    from pace_amd.util import CubedSphereCommunicator, LoopbackComm
    from pace_amd.util import constants as c

    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    dt_atmos = float(s["dt"]) * n_split
    env = Env(lib, device, metrics, n, nz)
    cube = CubedSphereCommunicator(LoopbackComm(rank=0, total_ranks=6), device=device, lib=lib)
    arrays = {k: s[k] for k in "u v w delz delp pt pe uc vc ua va q_con".split()}
    with np.errstate(all="ignore"):
        arrays["peln"] = np.log(s["pe"])
        arrays["pk"] = np.exp(c.KAPPA * arrays["peln"])
    arrays["phis"] = c.GRAV * s["zs"]
    arrays["ps"] = s["pe"][:, :, nz]
    arrays["pt"] = s["pt"] * np.exp(c.KAPPA * np.log(1.0e5))
    arrays["qvapor"] = 0.01 * np.exp(-6.0 * (1.0 - s["pe"] / s["pe"][:, :, nz:])) * (s["delp"] > 0)
    for name, f in dycore_condensates(0, s["delp"].shape).items():
        arrays[name] = np.abs(f)
    state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
    config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=dt_atmos, k_split=1, n_split=n_split,
                                 acoustic_dynamics=acoustic_config(n_split))
    core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                         datetime.timedelta(seconds=dt_atmos))
    core.step_dynamics(state)
    if device != "cpu":
        torch.cuda.synchronize()
    W = (slice(3, 3 + n), slice(3, 3 + n), slice(0, nz))
    return {k: getattr(state, k).numpy()[W].copy() for k in "u v w delz delp pt pe pk peln q_con qvapor qliquid".split() if hasattr(state, k)}


# ---- vertical remapping (tests/golden/remap_c12.npz: a run of the reference's MapSingle, tools/make_golden_remap.py) ----
# name -> (kord, iv, source interfaces, target interfaces, bottom value, qmin)
REMAP_CASES = {
    "pt_k9_iv1": (9, 1, "pn1", "pn2", None, 184.0),
    "qv_k9_iv0": (9, 0, "pe1", "pe2", None, 0.0),
    "w_k9_ivm2": (9, -2, "pe1", "pe2", "wsd", 0.0),
    "delz_k9_iv1": (9, 1, "pe1", "pe2", None, 0.0),
    "u_k9_ivm1": (9, -1, "pe1_u", "pe2_u", None, 0.0),
    "pt_k9_iv1_s": (9, 1, "pn1s", "pn2", None, 184.0),
    "qv_k9_iv0_s": (9, 0, "pe1s", "pe2", None, 0.0),
    "w_k9_ivm2_s": (9, -2, "pe1s", "pe2", "wsd", 0.0),
    "u_k9_ivm1_s": (9, -1, "pe1s_u", "pe2_u", None, 0.0),
    "qv_k10_iv0_s": (10, 0, "pe1s", "pe2", None, 0.0),
    "qv_k10_iv0": (10, 0, "pe1", "pe2", None, 0.0),
    "pt_k10_iv1": (10, 1, "pn1", "pn2", None, 184.0),
}
REMAP_KM = 79


def run_map_single(env, name, d, n=12):
    """One REMAP_CASES entry through the host class MapSingle: the fixture's compute-window arrays are embedded in full
    fields (halo 3; everything outside the window is NaN so that any read from there shows).  Returns the window of
    the remapped field."""
    import torch

    from pace_amd.fv3core.stencils.map_single import MapSingle

    kord, iv, src, dst, qs, qmin = REMAP_CASES[name]
    ustag = name.startswith("u_")
    dims = ["x", "y_interface" if ustag else "y", "z"]
    nj = n + 1 if ustag else n

    def embed(a):
        full = np.full((n + 7, n + 7, REMAP_KM + 1), np.nan)
        full[3:3 + n, 3:3 + nj, :] = a
        return env.q3(full)

    q, p1, p2 = embed(d[name + "_in"]), embed(d[src]), embed(d[dst])
    qsq = None
    if qs is not None:
        full = np.full((n + 7, n + 7), np.nan)
        full[3:3 + n, 3:3 + n] = d[qs]
        qsq = env.q2(full)
    op = MapSingle(env.stencil_factory, env.qf, kord, iv, dims)
    op(q, p1, p2, qs=qsq, qmin=qmin)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    return q.numpy()[3:3 + n, 3:3 + nj, :REMAP_KM]


def run_mapn_tracer(env, d, kord, nq=7, fill=True, n=12):
    """MapNTracer (+ fillz) through the host classes on tracers made from the fixture's fields (one of them with negative
    values so that fillz acts), and the same through the oracle.  Returns (got, expected): lists of window arrays."""
    import torch

    from oracle import remapping
    from pace_amd.fv3core.stencils.fillz import tracer_variables
    from pace_amd.fv3core.stencils.mapn_tracer import MapNTracer

    km = REMAP_KM

    def embed(a):
        full = np.full((n + 7, n + 7, km + 1), np.nan)
        full[3:3 + n, 3:3 + n, :] = a
        return env.q3(full)

    base = [d["qv_k9_iv0_in"], d["fillz0_in"], d["fillz1_in"], d["pt_k9_iv1_in"] * 1e-3, d["qv_k9_iv0_in"] * 0.5 + 1e-4,
            d["fillz2_in"] * 0.1, d["w_k9_ivm2_in"], d["qv_k9_iv0_in"] * 2.0][:nq]
    names = tracer_variables[:nq]
    tracers = {nm: embed(a) for nm, a in zip(names, base)}
    pe1, pe2 = d["pe1s"], d["pe2"]
    dp2 = np.zeros_like(pe2)
    dp2[:, :, :km] = pe2[:, :, 1:] - pe2[:, :, :-1]
    op = MapNTracer(env.stencil_factory, env.qf, kord, nq, fill, tracers)
    op(embed(pe1), embed(pe2), embed(dp2), tracers)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    got = [tracers[nm].numpy()[3:3 + n, 3:3 + n, :km] for nm in names]
    exp = []
    for t, a in enumerate(base):
        q = a.copy()
        remapping.map_single(q, pe1, pe2, km, 9 if t == 5 else kord, 0)
        if fill:
            remapping.fillz(q, dp2, km)
        exp.append(q[:, :, :km])
    return got, exp


L2E_OUT3 = ("pt", "delp", "delz", "peln", "u", "v", "w", "q_con", "pkz", "pk", "pe", "cappa")


def l2e_k10_fixture():
    """l2e_c12.npz with the inputs / outputs of the reference's run with every remapping order 10 merged over it
    (tools/make_golden_l2e.py kord10: negatives in four condensates)."""
    d = golden("l2e_c12.npz")
    d.update(golden("l2e_k10_c12.npz"))
    return d


def run_l2e(env, d, last_step, n=12, km=79, kord=9):
    """LagrangianToEulerian through the host class on the reference-run fixture (tests/golden/l2e_c12.npz; its arrays are
    the [2:16, 2:16] window of the 19 x 19 storage).  Returns dict name -> full numpy array."""
    import torch

    from pace_amd.fv3core import RemappingConfig
    from pace_amd.fv3core.stencils.remapping import LagrangianToEulerian
    from pace_amd.util import constants as c

    def embed(a):
        if a.ndim == 3:
            full = np.full((n + 7, n + 7, km + 1), np.nan)
            full[2:16, 2:16, :] = a
            return env.q3(full)
        full = np.full((n + 7, n + 7), np.nan)
        full[2:16, 2:16] = a
        return env.q2(full)

    f = {k[3:]: embed(d[k]) for k in d if k.startswith("in_") and not k.startswith("in_tr_")}
    tracers = {k[6:]: embed(d[k]) for k in d if k.startswith("in_tr_")}
    ak, bk = env.kq(d["ak"]), env.kq(d["bk"])
    op = LagrangianToEulerian(env.stencil_factory, env.qf,
                              RemappingConfig(kord_tm=-kord, kord_tr=kord, kord_wz=kord, kord_mt=kord), None, 8, None, tracers)
    op(tracers, f["pt"], f["delp"], f["delz"], f["peln"], f["u"], f["v"], f["w"], f["cappa"], f["q_con"], f["qcld"], f["pkz"],
       f["pk"], f["pe"], f["phis"], f["ps"], f["wsd"], ak, bk, None, float(d["ptop"]), c.KAPPA, c.ZVIR, last_step, 0.0, 112.5)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    out = {k: v.numpy() for k, v in f.items()}
    out.update({"tr_" + k: v.numpy() for k, v in tracers.items()})
    return out


def check_l2e(out, d, last_step, tol, n=12, km=79, loose=None):
    """Compare with the fixture on the windows the reference writes; returns the worst metric per variable."""
    cw = (slice(3, 3 + n), slice(3, 3 + n))
    worst = {}

    def ref_full(a):
        full = np.full((n + 7, n + 7) + a.shape[2:], np.nan)
        full[2:16, 2:16] = a
        return full

    if last_step:
        e = compare(ref_full(d["out_last_pt"])[cw][:, :, :km], out["pt"][cw][:, :, :km])
        assert e < tol, ("pt(last)", e)
        return {"pt(last)": e}
    for key in d:
        if not key.startswith("out_") or key == "out_last_pt":
            continue
        name = key[4:]
        ref = ref_full(d[key])
        win = {"u": (slice(3, 3 + n), slice(3, 4 + n)), "v": (slice(3, 4 + n), slice(3, 3 + n))}.get(name, cw)
        if ref.ndim == 2:
            e = compare(ref[cw], out[name][cw])
        else:
            kk = km + 1 if name in ("pe", "peln", "pk") else km
            e = compare(ref[win][:, :, :kk], out[name][win][:, :, :kk], near_zero=1e-18)
        worst[name] = e
        assert e < (loose or {}).get(name, tol), (name, e)
    return worst


# ------------------------------------------------------------------------------------------------------------------
# DynamicalCore.step_dynamics on six tiles (tests/golden/dycore_c12_tile*.npz, tools/make_golden_dycore.py)
# ------------------------------------------------------------------------------------------------------------------
DYCORE_TRACERS = "qvapor qliquid qrain qice qsnow qgraupel qo3mr qsgs_tke qcld".split()
DYCORE_OUT = "u v w delz delp pt pe pk peln pkz q_con omga ua va mfxd mfyd cxd cyd".split() + DYCORE_TRACERS


def dycore_condensates(tile, shape):
    """The deterministic content the fixture generator gives the species the test case leaves at zero (the same function
    as tools/make_golden_dycore.py:condensates -- the generator asserts that its inputs equal it)."""
    i, j, k = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    base = 0.5 + 0.5 * np.sin(0.7 * i + 1.3 * j + 0.37 * k + tile)
    out = {}
    for n, (name, scale) in enumerate((("qliquid", 2e-4), ("qrain", 1e-4), ("qice", 5e-5), ("qsnow", 3e-5), ("qgraupel", 2e-5),
                                       ("qo3mr", 1e-6), ("qsgs_tke", 1e-2), ("qcld", 0.3))):
        f = scale * (0.2 + base * (0.5 + 0.5 * np.cos(0.9 * i - 0.4 * j + 0.11 * k + n)))
        neg = ((3 * i + 5 * j + 7 * k + n + tile) % 23) == 0
        if name not in ("qo3mr", "qsgs_tke"):
            f = np.where(neg, -0.3 * f, f)
        out[name] = f
    return out


def generated_inputs(n, nz):
    """Per tile (metrics, state arrays) from pace_amd's OWN grid and initial-state generators -- nothing reference-derived."""
    from pace_amd.fv3core.initialization.baroclinic import baroclinic_state_six_tiles
    from pace_amd.util import gridgen

    tiles = gridgen.tiles(n, nz)
    states = baroclinic_state_six_tiles(tiles, n, nz)
    out = []
    for t in range(6):
        m = {k: v for k, v in tiles[t].items() if k not in ("ee1", "ee2", "es1", "ew2")}
        s = {k: states[t][k] for k in "u v w delz delp pe pk peln phis uc vc ua va pt qvapor ps".split()}
        out.append((m, s))
    return out


def run_dycore_tile(comm, lib, device, fix_ac, fix_dy, n, nz, checkpointer=None, generated=None):
    """One tile's program: state = the acoustic fixture's inputs (identical to the dycore run's, as the generator asserts)
    with the temperature before the preamble, the vapour and the regenerated condensates; one step_dynamics."""
    import datetime

    import torch

    from pace_amd.fv3core import DynamicalCoreConfig
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.fv_dynamics import DynamicalCore
    from pace_amd.util import CubedSphereCommunicator

    tile = comm.Get_rank()
    if generated is not None and generated[tile][1] is not None:
        metrics, arrays = generated[tile]
        arrays = {k: v.copy() for k, v in arrays.items()}
        shape = arrays["delp"].shape
    else:
        metrics = {k[5:]: v for k, v in fix_ac.items() if k.startswith("grid_")}
        if generated is not None:
            metrics = generated[tile][0]
        arrays = {k: fix_ac["in_" + k] for k in "u v w delz delp pe pk peln phis uc vc ua va".split()}
        shape = arrays["delp"].shape
        pt = np.zeros(shape)
        pt[3:3 + n, 3:3 + n, :] = fix_dy["in_pt"]
        qv = np.zeros(shape)
        qv[3:3 + n, 3:3 + n, :] = fix_dy["in_qvapor"]
        arrays.update(pt=pt, qvapor=qv, ps=fix_dy["in_ps"])
    env = Env(lib, device, metrics, n, nz)
    cube = CubedSphereCommunicator(comm, device=device, lib=lib)
    for name, f in dycore_condensates(tile, shape).items():
        arrays[name] = f * (arrays["delp"] > 0)
    state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
    n_split = int(fix_dy["n_split"])
    k_split = int(fix_dy["k_split"]) if "k_split" in fix_dy else 1
    ac = acoustic_config(n_split)
    ac.k_split = k_split
    kord = int(fix_dy["kord"]) if "kord" in fix_dy else 9  # (dycore_kord10_c12_*: every remapping order 10)
    config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=float(fix_dy["timestep"]), k_split=k_split,
                                 n_split=n_split, acoustic_dynamics=ac, kord_tm=-kord, kord_tr=kord, kord_wz=kord, kord_mt=kord)
    core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                         datetime.timedelta(seconds=float(fix_dy["timestep"])), checkpointer=checkpointer)
    core.step_dynamics(state)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    out = {k: getattr(state, k).numpy() for k in DYCORE_OUT}
    out["ps"] = state.ps.numpy()
    return out


def run_dycore_six_tiles(lib, device, n=12, nz=79, checkpointers=None, prefix="dycore_c12", generated=False):
    from pace_amd.util import run_tiles

    fa = [golden(f"acoustic_c12_tile{t}.npz") for t in range(6)]
    fd = [golden(f"{prefix}_tile{t}.npz") for t in range(6)]
    if prefix == "dycore_c12":  # tiles 0 and 2 also have FULL output fields of nine state variables (a later addition)
        full = golden("dycore_c12_full.npz")
        for t in full["tiles"]:
            for k in full:
                if k.endswith(f"_tile{t}"):
                    fd[int(t)]["full_" + k[4:-6]] = full[k]
    cps = checkpointers or [None] * 6
    gen = generated_inputs(n, nz) if generated else None
    if generated == "metrics":  # pace_amd's grid generator, the reference run's initial state
        gen = [(m, None) for m, _ in gen]
    return fd, run_tiles(6, lambda comm: run_dycore_tile(comm, lib, device, fa[comm.Get_rank()], fd[comm.Get_rank()], n, nz,
                                                         cps[comm.Get_rank()], generated=gen))


def dycore_errors(fix, out, n=12):
    """max reference-metric error per variable on the fixture's level subset and full columns."""
    errs = {}
    ks = fix["k_sel"]
    for k in DYCORE_OUT:
        full = out[k]
        di = 1 if k in ("v", "mfxd", "cxd") else 0
        dj = 1 if k in ("u", "mfyd", "cyd") else 0
        nk = 80 if k in ("pe", "pk", "peln") else 79
        kk = [x for x in ks if x < nk]
        idx = [list(ks).index(x) for x in kk]
        got = full[3:3 + n + di, 3:3 + n + dj][:, :, kk]
        ref = fix["out_" + k][:n + di, :n + dj][:, :, idx]
        band = _BANDS.get(k, 1e-12)
        near_zero = band * float(np.abs(ref).max()) + 1e-300
        e = compare(ref, got, near_zero=near_zero)
        cols = np.stack([full[i, j, :nk] for (i, j) in fix["cols"]])
        e = max(e, compare(fix["col_" + k][:, :nk], cols, near_zero=near_zero))
        if "full_" + k in fix:  # the whole field, all levels
            e = max(e, compare(fix["full_" + k][:n + di, :n + dj, :nk], full[3:3 + n + di, 3:3 + n + dj, :nk], near_zero=near_zero))
        errs[k] = e
    errs["ps"] = compare(fix["out_ps"][:n, :n], out["ps"][3:3 + n, 3:3 + n])
    return errs


# (measured on the GPU, profiles/r02_dycore_c12_gpu_errors.json: pressures 1e-15, delp / pt / pkz / tracers <= 1.5e-13, delz 4e-13,
# accumulators 5e-10, u / v / va 1e-8, w 3e-7, omga 8e-7; default bound of check_dycore for everything not listed: 1e-9 -> 1e-11)
DYCORE_TOL = {"w": 2e-6, "omga": 2e-6, "u": 1e-6, "v": 1e-6, "ua": 1e-6, "va": 1e-6, "delz": 1e-10, "mfxd": 1e-7, "mfyd": 1e-7,
              "cxd": 1e-7, "cyd": 1e-7}


def check_dycore(fixes, outs, default=1e-11):
    """Tolerances: the acoustic loop's (2e-6 = the reference's own DynCore bound, translate_dyncore.py:120, see ACOUSTIC_TOL)
    carried through tracer advection, remapping (reference bound 2e-8) and the
    final adjustments; `default` for everything else (masses, temperatures, tracers, pressures): 1e-11 after one remapping step
    (measured <= 1.5e-13), 1e-9 for the k_split = 2 run (two remapping steps: the condensates reach 3e-10)."""
    worst = {}
    for t in range(6):
        for k, e in dycore_errors(fixes[t], outs[t]).items():
            worst[k] = max(worst.get(k, 0.0), e)
    for k, e in worst.items():
        assert e < DYCORE_TOL.get(k, default), (k, e)
    return worst


# Envelope of the fully generated run (tools/wind_noise_sensitivity.py): one step of the REFERENCE ALGORITHM (the oracle)
# moves by u, v 1.8e-4, w 6e-6, ua / va 1.4e-6, delp 6e-9 of the field's magnitude when 1e-13 m/s of noise is added to the
# winds of this zonal-flow state (upwind / limiter branches decided by wind components that are zero by symmetry); the
# generated winds differ from the reference's by up to 3e-13 m/s, so that is the accuracy an end-to-end comparison can have.
GENERATED_TOL = {"u": 1e-3, "v": 1e-3, "va": 1e-3, "ua": 1e-5, "w": 5e-5, "omga": 5e-5, "mfxd": 5e-6, "mfyd": 5e-6,
                 "cxd": 5e-6, "cyd": 5e-6}


# One whole step with every remapping order 10: the kord 10 limiter is discontinuous in its inputs -- the reference algorithm
# (oracle, bit-identical to the reference's LagrangianToEulerian on the same inputs) turns 1e-13 of input noise into 2e-6 (v),
# 1e-5 (w), 6e-7 (pt), 2e-4 (condensates) of each field's magnitude, where kord 9 leaves 1e-13 -- and its inputs here come out
# of the acoustic loop, whose vertical solver carries the device's exp / log.  Hence scaled-error bounds:
KORD10_TOL = {"u": 1e-4, "v": 1e-4, "ua": 1e-4, "va": 1e-4, "w": 1e-4, "omga": 1e-4, "pt": 1e-6, "pkz": 1e-6, "q_con": 1e-3,
              "qliquid": 1e-3, "qrain": 1e-3, "qice": 1e-3, "qsnow": 1e-3, "qsgs_tke": 1e-8}


def check_dycore_kord10(fixes, outs):
    worst = dycore_scaled_errors(fixes, outs)
    for k, e in worst.items():
        assert e < KORD10_TOL.get(k, 1e-10), (k, e)
    return worst


def check_dycore_generated(fixes, outs):
    worst = dycore_scaled_errors(fixes, outs)
    for k, e in worst.items():
        assert e < GENERATED_TOL.get(k, 5e-7), (k, e)
    return worst


def dycore_scaled_errors(fixes, outs, n=12):
    """max |got - ref| / max |ref| per variable over the six tiles (fixture's level subset)."""
    worst = {}
    for fix, out in zip(fixes, outs):
        ks = fix["k_sel"]
        for k in DYCORE_OUT:
            di = 1 if k in ("v", "mfxd", "cxd") else 0
            dj = 1 if k in ("u", "mfyd", "cyd") else 0
            nk = 80 if k in ("pe", "pk", "peln") else 79
            kk = [x for x in ks if x < nk]
            idx = [list(ks).index(x) for x in kk]
            got = out[k][3:3 + n + di, 3:3 + n + dj][:, :, kk]
            ref = fix["out_" + k][:n + di, :n + dj][:, :, idx]
            e = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300))
            worst[k] = max(worst.get(k, 0.0), e)
    return worst


def l2e_synthetic_case(n, km, seed=2):
    """Inputs of LagrangianToEulerian at any size: the synthetic balanced state with its Lagrangian surfaces deformed by a
    few per cent of a layer ... two layers (column by column), vapour + the deterministic condensates.  Returns
    (fields, tracers, ak, bk, ptop) as full numpy arrays."""
    from pace_amd import synthetic

    m = synthetic.tile_metrics(n, km)
    s = synthetic.acoustic_state(m, n, km)
    rng = np.random.default_rng(seed)
    ni = n + 7
    ak, bk = m["ak"], m["bk"]
    ptop = float(ak[0])
    # deform: redistribute the layer thicknesses, keep the column mass
    sig = np.linspace(0.0, 1.0, km + 1)
    amp = (2.0 / km) * rng.random((ni, ni))
    pe_e = s["pe"]
    ps = pe_e[:, :, km]
    frac = (pe_e - ptop) / (ps - ptop)[:, :, None]
    frac = frac + amp[:, :, None] * np.sin(2.0 * np.pi * sig)[None, None, :] * frac * (1.0 - frac) * 4.0
    frac[:, :, 0], frac[:, :, km] = 0.0, 1.0
    pe = ptop + (ps - ptop)[:, :, None] * frac
    f = {}
    f["pe"] = pe
    f["peln"] = np.log(pe)
    delp = np.zeros_like(pe)
    delp[:, :, :km] = pe[:, :, 1:] - pe[:, :, :-1]
    f["delp"] = delp
    for k in ("pt", "delz", "u", "v", "w"):
        f[k] = s[k].copy()
    f["pt"][:, :, :km] = s["pt"][:, :, :km] * 300.0  # a temperature-like magnitude (the operator takes logs of it)
    for k in ("cappa", "q_con", "pkz", "pk", "qcld"):
        f[k] = np.zeros_like(pe)
    f["pk"][:, :, km] = np.exp(0.2857 * np.log(pe[:, :, km]))
    f["ps"] = np.zeros((ni, ni))
    f["wsd"] = 0.01 * rng.standard_normal((ni, ni))
    f["phis"] = np.zeros((ni, ni))
    tracers = {"qvapor": 0.01 * np.exp(-4.0 * (1.0 - frac[:, :, :1] * 0 - np.minimum(frac, 1.0))) * (1 + 0.2 * rng.random(pe.shape))}
    tracers.update({k: v for k, v in dycore_condensates(0, pe.shape).items() if k != "qcld"})
    return f, tracers, ak, bk, ptop


DSW_VARIANTS = {  # tools/make_golden_dsw_variants.py: runs of the reference with one namelist option changed
    "h5": dict(hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5),
    "nord2": dict(nord=2),
    "dcon0": dict(d_con=0.0),
    "skeb": dict(do_skeb=True),
    "dddmp0": dict(dddmp=0.0),
}


def dsw_variant_fixture(variant):
    fix = golden(f"d_sw_{variant}_c12_tile0_call1.npz")
    k_sel = fix["k_sel"]
    cfg = dict(DSW_CFG, **DSW_VARIANTS[variant])
    col = {k[4:]: np.ascontiguousarray(v[np.asarray(k_sel)]) for k, v in fix.items() if k.startswith("col_")}
    return fix, cfg, col, len(k_sel)


def run_d_sw_variant_fixture(env, variant):
    """A run of the reference with one namelist option changed, through the host class; returns {field: error}."""
    fix, cfg, col, nk = dsw_variant_fixture(variant)
    out, _ = run_d_sw(env, col, {k: fix["in_" + k] for k in DSW_ARGS}, float(fix["dt"]), ut0=fix["in_uc_contra"],
                      vt0=fix["in_vc_contra"], cfg=cfg)
    return {k: compare(fix["out_" + k][dsw_window(k, 12, nk)], out[k][dsw_window(k, 12, nk)]) for k in DSW_ARGS if k != "zh"}


def run_d_sw_h5_fixture(env):
    return max(run_d_sw_variant_fixture(env, "h5").values())
