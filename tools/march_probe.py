"""The wave-private marching transport kernel (pace_amd/csrc/k_march.hip, an experiment) against the LDS-tile kernel:
bit-exactness on the interior box and time per cell.   python tools/march_probe.py [--n 192]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    from pace_amd.tile import Env

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core.stencils.fvtp2d import FiniteVolumeTransport
    from pace_amd.fv3core.stencils.fxadv import FiniteVolumeFluxPrep

    lib = _lib.Library(args.lib) if args.lib else _lib.load()
    n, nz = args.n, args.nz
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    f = {k: env.q3(s[k]) for k in ("uc", "vc", "crx", "cry", "xfx", "yfx", "pt")}
    ut, vt = env.q3(), env.q3()
    FiniteVolumeFluxPrep(env.stencil_factory, env.grid_data)(f["uc"], f["vc"], f["crx"], f["cry"], f["xfx"], f["yfx"], ut, vt, s["dt"])
    tp = FiniteVolumeTransport(env.stencil_factory, env.qf, env.grid_data, env.damping, 0, 6)
    fx, fy, gx, gy = env.q3(), env.q3(), env.q3(), env.q3()
    geom, met = tp._geom, tp._met
    st = tp.stream
    pad = 6
    ib, nx = 3 + pad, n - 2 * pad
    jb, ny = 3 + pad, n - 2 * pad

    def tile_kernel():
        tp(f["pt"], f["crx"], f["cry"], f["xfx"], f["yfx"], fx, fy)

    def march():
        lib.call("pace_fvtp2d_march_probe", C.byref(geom), C.byref(met), f["pt"].ptr, f["crx"].ptr, f["cry"].ptr, f["xfx"].ptr,
                 f["yfx"].ptr, gx.ptr, gy.ptr, ib, nx, jb, ny, nz, st())

    tile_kernel()
    march()
    torch.cuda.synchronize()
    w = (slice(ib, ib + nx), slice(jb, jb + ny), slice(0, nz))
    a, b = fx.numpy()[w], gx.numpy()[w]
    c, d = fy.numpy()[w], gy.numpy()[w]
    print(f"box {nx} x {ny} x {nz}: fx identical {np.array_equal(a, b)} (max diff {np.abs(a - b).max():.3e}), "
          f"fy identical {np.array_equal(c, d)} (max diff {np.abs(c - d).max():.3e})")

    def timeit(fn):
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return float(np.median(ts))

    t_tile, t_march = timeit(tile_kernel), timeit(march)
    cells_tile, cells_box = n * n * nz, nx * ny * nz
    print(f"LDS-tile kernel : {t_tile:8.1f} us for {cells_tile} cells = {1e3 * t_tile / cells_tile:.4f} ns/cell")
    print(f"marching kernel : {t_march:8.1f} us for {cells_box} cells = {1e3 * t_march / cells_box:.4f} ns/cell "
          f"-> {(t_tile / cells_tile) / (t_march / cells_box):.2f}x per cell")


if __name__ == "__main__":
    main()
