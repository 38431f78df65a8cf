"""Per-entry-point timing on the GPU box (HIP events on the launch stream), C192 x 79 by default.

    python tools/kbench.py [--lib path/to/libpace_hip.so] [--n 192] [--reps 20] [--only fvtp2d,riem3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--precision", type=int, default=64, choices=(64, 32), help="storage type of the fields: libpace_hip.so or libpace_hip_f32.so")
    args = ap.parse_args()
    from pace_amd.tile import DSW_ARGS, Env

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist
    from pace_amd.fv3core.stencils.delnflux import DelnFlux, DelnFluxNoSG
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport
    from pace_amd.fv3core.stencils.fxadv import FiniteVolumeFluxPrep
    from pace_amd.fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver

    lib = _lib.Library(args.lib) if args.lib else _lib.load(args.precision)
    n, nz = args.n, args.nz
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    col = get_column_namelist(DGridShallowWaterLagrangianDynamicsConfig(), env.qf)
    f = {k: env.q3(s[k]) for k in list(DSW_ARGS) + ["cappa", "delz", "pe", "ppe", "pk3", "pk", "peln"]}
    zs, ws = env.q2(s["zs"]), env.q2(s["ws"])
    ut, vt, fx, fy = env.q3(), env.q3(), env.q3(), env.q3()
    prep = FiniteVolumeFluxPrep(env.stencil_factory, env.grid_data)
    prep(f["uc"], f["vc"], f["crx"], f["cry"], f["xfx"], f["yfx"], ut, vt, s["dt"])
    tp = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 6)
    dn0 = DelnFluxNoSG(env.stencil_factory, env.damping, env.grid_data.rarea, col["nord_w"])
    dn2 = DelnFlux(env.stencil_factory, env.qf, env.damping, env.grid_data.rarea, col["nord_t"], col["damp_t"])
    dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False,
                                              DGridShallowWaterLagrangianDynamicsConfig())
    riem = NonhydrostaticVerticalSolver(env.stencil_factory, env.qf, RiemannConfig())
    cells = n * n * nz
    field_mb = (n + 1) * (n + 1) * nz * (args.precision // 8) / 1e6
    damp_w = torch.as_tensor(np.full(nz, 1.0e9), device="cuda")

    def restore():
        for k in f:
            f[k].set(s[k])

    import ctypes as C

    from pace_amd.fv3core.stencils._common import dptr

    def dsw_phase(mask):
        fields = [f[k] for k in DSW_ARGS]
        dsw.lib.call("pace_d_sw_phases", mask, C.byref(dsw._geom), C.byref(dsw._met), C.byref(dsw._col), C.byref(dsw._cfg),
                     dsw._workspace.data_ptr(), *[dptr(x) for x in fields], float(s["dt"]), dsw.stream())

    from pace_amd.fv3core.stencils.map_single import MapSingle

    rng = np.random.default_rng(1)
    sig = np.linspace(0.0, 1.0, nz + 1) ** 1.6
    ps = 1.0e5 * (1.0 + 0.02 * rng.random((n + 7, n + 7)))
    pe2_h = 300.0 + (ps - 300.0)[:, :, None] * sig[None, None, :]
    s1 = sig[None, None, :] + (1.5 / nz * rng.random((n + 7, n + 7)))[:, :, None] * np.sin(2.0 * np.pi * sig)[None, None, :]
    s1[:, :, 0], s1[:, :, nz] = 0.0, 1.0
    pe1_h = 300.0 + (ps - 300.0)[:, :, None] * s1
    rq, rp1, rp2 = env.q3(s["pt"]), env.q3(pe1_h), env.q3(pe2_h)
    remap = MapSingle(env.stencil_factory, env.qf, 9, 1, ["x", "y", "z"])

    cases = {
        "fxadv": (lambda: prep(f["uc"], f["vc"], f["crx"], f["cry"], f["xfx"], f["yfx"], ut, vt, s["dt"]), 8),
        "fvtp2d": (lambda: tp(f["pt"], f["crx"], f["cry"], f["xfx"], f["yfx"], fx, fy, x_mass_flux=f["mfx"], y_mass_flux=f["mfy"]), 9),
        "delnflux_nosg": (lambda: dn0(f["w"], fx, fy, damp_w, None), 3),
        "delnflux_mass": (lambda: dn2(f["pt"], fx, fy, mass=f["delp"]), 6),
        "riem3": (lambda: riem(False, s["dt"], f["cappa"], m["ptop"], zs, ws, f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"],
                               f["pe"], f["ppe"], f["pk3"], f["pk"], f["peln"], f["w"]), 13),
        "map_single": (lambda: remap(rq, rp1, rp2), 4),
        "dsw_scalars": (lambda: dsw_phase(2), 18),
        "dsw_winds": (lambda: dsw_phase(12), 20),
        "d_sw": (lambda: dsw(*[f[k] for k in DSW_ARGS], s["dt"]), 32),
    }
    only = [x for x in args.only.split(",") if x]
    print(f"{'case':16s} {'us':>10s} {'alg GB/s':>10s} {'%8TB/s':>8s}")
    for name, (fn, nfields) in cases.items():
        if only and name not in only:
            continue
        restore()
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            if name in ("d_sw", "riem3", "dsw_scalars", "dsw_winds"):
                restore()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = float(np.median(ts))
        gbs = nfields * field_mb * 1e6 / (us * 1e-6) / 1e9
        print(f"{name:16s} {us:10.1f} {gbs:10.1f} {100*gbs/8000:8.2f}")


if __name__ == "__main__":
    main()
