"""The float32-storage build (BASELINE configuration 5; libpace_hip_f32.so / tests/emu/libpace_emu_f32.so: the same kernel
sources with pace_real_t = float -- fields, metrics and K-tables in float32, arithmetic in registers in float64).

The reference's own 32-bit mode (PACE_FLOAT_PRECISION=32, dsl/pace/dsl/typing.py:24) has no fixtures and cannot run here, so
parity is against the float64 run of the reference (tests/golden) within what float32 STORAGE allows: every field is rounded
to 6e-8 of its value each time it is stored.  Bounds (max |got - ref| / max |ref| per variable):
  * one operator (d_sw): 2e-5;
  * one whole DynamicalCore step: 1e-5 for masses, pressures, temperatures; 1e-4 for the condensates (values ~1e-3 of the
    field's range); 2e-3 for the horizontal winds and 2e-2 for w / omga -- the algorithm amplifies 1e-13 of wind noise to
    2e-4 in one step of this zonal-flow case (tools/wind_noise_sensitivity.py), float32 rounding is a million times more.
"""
import os
import subprocess

import numpy as np
import pytest

from helpers import (DSW_ARGS, ROOT, Env, column_for_levels, dsw_window, dycore_scaled_errors, golden, run_d_sw,
                     run_dycore_six_tiles)

STEP_TOL = {"u": 2e-3, "v": 2e-3, "va": 2e-3, "ua": 2e-3, "w": 2e-2, "omga": 2e-2, "qliquid": 1e-4, "qrain": 1e-4, "qice": 1e-4,
            "qsnow": 1e-4, "qgraupel": 1e-4, "q_con": 1e-4}


def build_emu_f32():
    from helpers import _make

    _make("emu-f32")
    return os.path.join(ROOT, "tests", "emu", "libpace_emu_f32.so")


def check_d_sw(lib, device):
    fix = golden("d_sw_c12_tile0_call1.npz")
    k_sel = fix["k_sel"]
    nk = len(k_sel)
    env = Env(lib, device, golden("grid_c12_tile0.npz"), 12, nk)
    out, _ = run_d_sw(env, column_for_levels(k_sel), {k: fix["in_" + k] for k in DSW_ARGS}, float(fix["dt"]),
                      ut0=fix["in_uc_contra"], vt0=fix["in_vc_contra"])
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, 12, nk)
        assert out[k].dtype == np.float32
        e = float(np.abs(fix["out_" + k][W] - out[k][W]).max() / (np.abs(fix["out_" + k][W]).max() + 1e-300))
        assert e < 2e-5, (k, e)


def check_step(fixes, outs):
    worst = dycore_scaled_errors(fixes, outs)
    for k, e in worst.items():
        assert e < STEP_TOL.get(k, 1e-5), (k, e)
    return worst


@pytest.fixture(scope="module")
def emu_f32():
    from pace_amd import _lib

    lib = _lib.Library(build_emu_f32())
    assert lib.real_bytes == 4
    return lib


def test_f32_d_sw_emulated(emu_f32):
    check_d_sw(emu_f32, "cpu")


def test_f32_dynamical_core_step_emulated(emu_f32):
    """One whole DynamicalCore.step_dynamics, six tiles, float32 fields (acoustic loop, tracer advection, remapping, all halo
    exchanges with float32 messages) against the float64 run of the reference."""
    fixes, outs = run_dycore_six_tiles(emu_f32, "cpu")
    check_step(fixes, outs)


def test_f32_lean_kernels_c48_against_f64_emulated(emu_f32):
    """The fused scalar + wind kernel (csrc/fvt_core.h) with float32 fields -- BASELINE configuration 5's storage type; it was a
    float64-only kernel until round 5 -- at a size its tilings cover (C48: the 16 x 24 tile shape), all of d_sw against the SAME
    kernels of the float64 emulation build on the same synthetic state, to float32 storage accuracy."""
    import ctypes as C

    from helpers import DSW_CFG, build_emu
    from pace_amd import _lib, synthetic

    n, nz = 48, 2
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = {k: np.ascontiguousarray(v[:nz]) for k, v in golden("column_namelist_c12.npz").items()}
    outs = {}
    for name, lib in (("f64", _lib.Library(build_emu())), ("f32", emu_f32)):
        env = Env(lib, "cpu", metrics, n, nz)
        outs[name], op = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"], cfg=DSW_CFG)
        assert lib.cdll.pace_d_sw_wind_outputs_supported(C.byref(op._geom), C.byref(op._cfg)) == 1, name  # the fused path is the one that ran
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nz)
        a, b = outs["f64"][k][W], outs["f32"][k][W]
        e = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-300))
        # (the divergence damping's work fields are gradients of a divergence: differences of nearly equal numbers)
        assert e < (5e-3 if k in ("heat_source", "diss_est") else 2e-4 if k in ("uc", "vc", "divgd", "delpc") else 5e-5), (k, e)


def test_storage_type_mismatch_is_refused(emu_f32):
    """float64 fields handed to the float32 library (or the reverse) are an error, not a reinterpretation."""
    import torch

    from pace_amd import _lib
    from pace_amd.dsl import CompilationConfig, GridIndexing, StencilConfig, StencilFactory
    from pace_amd.fv3core.stencils.riem_solver_c import NonhydrostaticVerticalSolverCGrid
    from pace_amd.util import QuantityFactory, SubtileGridSizer

    sizer = SubtileGridSizer.from_tile_params(nx_tile=12, ny_tile=12, nz=8, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
    qf64 = QuantityFactory(sizer, device="cpu", dtype=torch.float64)
    sf = StencilFactory(StencilConfig(compilation_config=CompilationConfig()), GridIndexing.from_sizer_and_communicator(sizer, None),
                        lib=emu_f32, quantity_factory=qf64)
    with pytest.raises(_lib.PaceError):
        NonhydrostaticVerticalSolverCGrid(sf, qf64, 0.05)
    qf32 = QuantityFactory(sizer, device="cpu", dtype=torch.float32)
    assert qf32.row_stride % 32 == 0 and qf64.row_stride % 16 == 0  # 128-byte rows in both


@pytest.mark.gpu
def test_f32_d_sw_gpu():
    from pace_amd import _lib

    check_d_sw(_lib.load(32), "cuda")


@pytest.mark.gpu
def test_f32_dynamical_core_step_gpu(tmp_path):
    from helpers import run_in_child

    fixes, outs = run_in_child("dycore_f32", tmp_path)
    check_step(fixes, outs)


@pytest.mark.gpu
def test_f32_d_sw_and_riem3_c96_against_f64_gpu():
    """The float32-storage kernels at a size with interior workgroups and every tile seam (C96 x 79): d_sw and riem_solver3
    against the SAME kernels of the float64 library on the same synthetic state, to float32 storage accuracy."""
    from pace_amd import _lib, synthetic
    from pace_amd.tile import run_riem3

    n, nz = 96, 79
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    col = {k: np.ascontiguousarray(v[:nz]) for k, v in golden("column_namelist_c12.npz").items()}
    outs = {}
    for prec in (64, 32):
        env = Env(_lib.load(prec), "cuda", metrics, n, nz)
        d, _ = run_d_sw(env, col, {k: s[k] for k in DSW_ARGS}, s["dt"])
        inp = {"cappa": s["cappa"], "zs": s["zs"], "ws": s["ws"], "delz": s["delz"], "q_con": s["q_con"], "delp": s["delp"],
               "pt": s["pt"], "zh": s["zh"], "p": s["pe"], "ppe": s["ppe"], "pk3": s["pk3"], "pk": s["pk"],
               "log_p_interface": s["peln"], "w": s["w"]}
        r = run_riem3(env, inp, False, s["dt"], metrics["ptop"])
        outs[prec] = (d, r)
    for k in DSW_ARGS:
        if k == "zh":
            continue
        W = dsw_window(k, n, nz)
        a, b = outs[64][0][k][W], outs[32][0][k][W]
        e = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-300))
        # (measured: 4e-8 ... 2e-6; the dissipative heating and its estimate are differences of kinetic-energy-sized terms: 6e-4)
        assert e < (5e-3 if k in ("heat_source", "diss_est") else 2e-5), ("d_sw", k, e)
    for k, nk in (("delz", nz), ("zh", nz + 1), ("pk3", nz + 1), ("w", nz), ("ppe", nz + 1)):
        a, b = outs[64][1][k][3:3 + n, 3:3 + n, :nk], outs[32][1][k][3:3 + n, 3:3 + n, :nk]
        e = float(np.abs(a - b).max() / (np.abs(a).max() + 1e-300))
        # (measured: delz 1.5e-6, zh 1e-7, pk3 4e-8; the perturbation pressure and w come from small differences of large
        # pressures: 4e-5 and 7e-4 of their ranges)
        assert e < (5e-3 if k in ("ppe", "w") else 2e-5), ("riem_solver3", k, e)
