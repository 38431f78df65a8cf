"""Per-operator timing of one whole acoustic substep on ONE tile (no halo exchanges), C192 x 79 by default, HIP events
on the launch stream.  Inputs: the synthetic state, advanced through the sequence once so every operator sees the
fields its predecessor produced.

    python tools/acoustic_bench.py [--n 192] [--reps 10]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


PRECISION = 64


def build_ops(n, nz):
    """[(name, callable)] for one acoustic substep on one tile + the fields they work on."""
    from pace_amd.tile import DSW_ARGS, Env

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig
    from pace_amd.fv3core.stencils.c_sw import CGridShallowWaterDynamics
    from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist
    from pace_amd.fv3core.stencils.nh_p_grad import NonHydrostaticPressureGradient
    from pace_amd.fv3core.stencils.pk3_halo import PK3Halo
    from pace_amd.fv3core.stencils.ray_fast import RayleighDamping
    from pace_amd.fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver
    from pace_amd.fv3core.stencils.riem_solver_c import NonhydrostaticVerticalSolverCGrid
    from pace_amd.fv3core.stencils.updatedzc import UpdateGeopotentialHeightOnCGrid
    from pace_amd.fv3core.stencils.updatedzd import UpdateHeightOnDGrid

    lib = _lib.load(PRECISION)
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    sf, qf, gd, dc = env.stencil_factory, env.qf, env.grid_data, env.damping
    col = get_column_namelist(DGridShallowWaterLagrangianDynamicsConfig(), qf)
    names = list(DSW_ARGS) + ["cappa", "delz", "pe", "ppe", "pk3", "pk", "peln"]
    f = {k: env.q3(s[k]) for k in names}
    zs, ws3, wsd = env.q2(s["zs"]), env.q2(), env.q2()
    phis = env.q2(s["zs"] * 9.80665)
    ut, vt, gz, omga = env.q3(), env.q3(), env.q3(), env.q3()
    csw = CGridShallowWaterDynamics(sf, qf, gd, False, 0, 3)
    zc = UpdateGeopotentialHeightOnCGrid(sf, qf, gd.area, gd.dp_ref)
    rc = NonhydrostaticVerticalSolverCGrid(sf, qf, 0.05)
    # (as AcousticDynamics constructs and calls it: the transported scalars and the winds to spare buffers that are swapped in, the
    # divergence damping's dead work fields skipped)
    dsw = DGridShallowWaterLagrangianDynamics(sf, qf, gd, dc, col, False, False, DGridShallowWaterLagrangianDynamicsConfig(),
                                              swap_scalar_storage=True)
    zd = UpdateHeightOnDGrid(sf, qf, dc, gd, 0, 6, col)
    r3 = NonhydrostaticVerticalSolver(sf, qf, RiemannConfig())
    nh = NonHydrostaticPressureGradient(sf, qf, gd, 0)
    pk3h = PK3Halo(sf, qf)
    ray = RayleighDamping(sf, 3000.0, 10.0, False, quantity_factory=qf)
    dt = float(s["dt"])
    dt2 = 0.5 * dt
    ptop = float(m["ptop"])
    geom = csw._geom
    import ctypes as C

    def call(name, *a):
        lib.call(name, C.byref(geom), *a)

    st = csw.stream
    ops = [
        ("c_sw", lambda: csw(f["delp"], f["pt"], f["u"], f["v"], f["w"], f["uc"], f["vc"], f["ua"], f["va"], ut, vt, f["divgd"], omga, dt2)),
        ("copy zh->gz", lambda: call("pace_copy", f["zh"].ptr, gz.ptr, st())),
        ("updatedzc", lambda: zc(zs, ut, vt, gz, ws3, dt2)),
        ("riem_solver_c", lambda: rc(dt2, f["cappa"], ptop, phis, ws3, csw.ptc, f["q_con"], csw.delpc, gz, f["ppe"], omga)),
        ("p_grad_c", lambda: call("pace_p_grad_c", C.byref(csw._met), f["uc"].ptr, f["vc"].ptr, csw.delpc.ptr, f["ppe"].ptr, gz.ptr, dt2, st())),
        ("d_sw", lambda: dsw(*[f[k] if k != "delpc" else vt for k in DSW_ARGS], dt, skip_dead_outputs=True)),
        ("updatedzd", lambda: zd(zs, f["zh"], f["crx"], f["cry"], f["xfx"], f["yfx"], wsd, dt)),
        ("riem_solver3", lambda: r3(False, dt, f["cappa"], ptop, zs, wsd, f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"], f["pe"],
                                    f["ppe"], f["pk3"], f["pk"], f["peln"], f["w"])),
        ("pk3_halo", lambda: pk3h(f["pk3"], f["delp"], ptop, 287.05 / 1004.6)),
        ("compute_geopotential", lambda: call("pace_compute_geopotential", f["zh"].ptr, gz.ptr, st())),
        ("nh_p_grad", lambda: nh(f["u"], f["v"], f["ppe"], gz, f["pk3"], f["delp"], dt, ptop, 287.05 / 1004.6)),
        ("ray_fast", lambda: ray(f["u"], f["v"], f["w"], gd.dp_ref, gd.p, dt, ptop)),
    ]
    return ops, f, (ut, vt, gz, omga)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--precision", type=int, default=64, choices=(64, 32))
    args = ap.parse_args()
    global PRECISION
    PRECISION = args.precision
    n, nz = args.n, args.nz
    ops, f, (ut, vt, gz, omga) = build_ops(n, nz)
    snap = {}

    def save():
        for k, q in list(f.items()) + [("ut", ut), ("vt", vt), ("gz", gz), ("omga", omga)]:
            snap[k] = q._base.clone()

    def restore():
        for k, q in list(f.items()) + [("ut", ut), ("vt", vt), ("gz", gz), ("omga", omga)]:
            q._base.copy_(snap[k])

    save()
    for _, fn in ops:  # warm-up pass
        fn()
    torch.cuda.synchronize()
    nanfrac = float(torch.isnan(f["w"].data).double().mean())
    times = {k: [] for k, _ in ops}
    for _ in range(args.reps):
        restore()
        torch.cuda.synchronize()
        for name, fn in ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) * 1e3)
    # distinct 3-D fields per operator, each once per direction (the convention of SURVEY.md section 8d; 2-D / K fields ~ 0):
    #   c_sw: delp, pt, u, v, w -> delpc, ptc, uc, vc, ua, va, ut, vt, divgd, w3;  updatedzc: ut, vt, gz -> gz;
    #   riem_solver_c: cappa, ptc, q_con, delpc, gz, w3 -> gz, pef;  p_grad_c: uc, vc, delpc, pkc, gz -> uc, vc;
    #   d_sw: 16 -> 16;  updatedzd: zh, crx, cry, xfx, yfx -> zh;  riem_solver3: 7 -> 6;  nh_p_grad: u, v, pp, gz, pk3, delp -> u, v;
    #   pk3_halo / ray_fast touch halo rows / sponge levels only (no figure)
    fields = {"c_sw": 15, "copy zh->gz": 2, "updatedzc": 4, "riem_solver_c": 8, "p_grad_c": 7, "d_sw": 32, "updatedzd": 6, "riem_solver3": 13,
              "compute_geopotential": 2, "nh_p_grad": 8}
    item = 8 if PRECISION == 64 else 4
    cells = n * n * nz
    tot = tot_mb = 0.0
    print(f"single-tile acoustic substep, C{n} x {nz}  (NaN fraction in w after one pass: {nanfrac:.2e})")
    print(f"{'operator':24s} {'us':>10s} {'algorithmic MB':>15s} {'GB/s':>8s} {'% of 8 TB/s':>12s}")
    for name, _ in ops:
        us = float(np.median(times[name]))
        tot += us
        if name in fields:
            mb = fields[name] * item * cells / 1e6
            tot_mb += mb
            print(f"{name:24s} {us:10.1f} {mb:15.1f} {mb / us * 1e3:8.0f} {100 * mb / us * 1e3 / 8000:11.1f}%")
        else:
            print(f"{name:24s} {us:10.1f}")
    print(f"{'total':24s} {tot:10.1f} {tot_mb:15.1f} {tot_mb / tot * 1e3:8.0f} {100 * tot_mb / tot * 1e3 / 8000:11.1f}%   -> "
          f"{cells / tot * 1e6 / 1e9:.2f} G cell-updates/s for the whole loop body")


if __name__ == "__main__":
    main()
