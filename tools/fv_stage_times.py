"""Where does a workgroup of the fused transport kernel spend its time?  (experiment; needs `make prof`)

Runs pace_fvtp2d_update (k_fvtp2d<6,2,1>) at C192 x 79 from the instrumented library build/prof/libpace_prof.so and prints, for
one interior workgroup per level, the shader-clock cycles between consecutive stage boundaries (median over levels)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.tile import Env  # noqa: E402
from pace_amd.util.grid import geom_struct  # noqa: E402

NAMES = ["load q footprint", "fused del-n damping", "inner y PPM (+cry)", "q_i (+yfx, area)", "inner x PPM (+crx)",
         "q_j (+xfx, area)", "outer x PPM (+x mass flux, store)", "outer y PPM (+y mass flux, store)", "fluxes -> LDS",
         "cell update (+rarea, q, delp)"]


def main():
    n, nz = 192, 79
    lib = _lib.Library(os.path.join(ROOT, "build", "prof", "libpace_prof.so"))
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    f = {k: env.q3(s[k]) for k in ("pt", "delp", "uc", "vc")}
    crx, cry, xfx, yfx, ut, vt, fx, fy, out = (env.q3() for _ in range(9))
    geom = geom_struct(env.qf)
    met = env.grid_data.c_struct()
    lib.call("pace_fxadv", C.byref(geom), C.byref(met), f["uc"].ptr, f["vc"].ptr, crx.ptr, cry.ptr, xfx.ptr, yfx.ptr, ut.ptr, vt.ptr,
             float(s["dt"]), None)
    lib.call("pace_fvtp2d", C.byref(geom), C.byref(met), f["delp"].ptr, crx.ptr, cry.ptr, xfx.ptr, yfx.ptr, fx.ptr, fy.ptr, None,
             None, 6, nz, None)
    kdev = torch.as_tensor(np.concatenate([np.full(nz, (0.06 * m["da_min"]) ** 3), np.full(nz, 2.0)]), device="cuda")
    scratch = [env.q3(s["pt"]) for _ in range(12)]  # distinct inputs so that nothing is cache-warm
    torch.cuda.synchronize()
    host = (C.c_longlong * (256 * 16))()
    rows = []
    for rep in range(12):
        lib.call("pace_fvtp2d_update", C.byref(geom), C.byref(met), scratch[rep].ptr, crx.ptr, cry.ptr, xfx.ptr, yfx.ptr, fx.ptr,
                 fy.ptr, f["delp"].ptr, kdev.data_ptr(), kdev.data_ptr() + 8 * nz, 2, out.ptr, 6, nz, None)
        torch.cuda.synchronize()
        assert lib.cdll.pace_debug_fv_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(256, 16)[:nz, :11].astype(float)
        if rep >= 2:
            rows.append(np.diff(a, axis=1))
    d = np.concatenate(rows)
    med = np.median(d, axis=0)
    tot = med.sum()
    print(f"one interior workgroup of k_fvtp2d<6,2,1>, C{n} x {nz}: {tot:.0f} shader-clock cycles between first and last stamp")
    for i, v in enumerate(med):
        print(f"  {NAMES[i]:42s} -> {v:8.0f}  ({100 * v / tot:4.1f} %)")


if __name__ == "__main__":
    main()
