"""The flux preparation of d_sw timed in its parts (hipEvents, median of N): the whole (pace_d_sw_phases 1), the interior box alone
(16) and the frame alone (32), for the library in PACE_HIP_LIB (default: the product's) and, with PACE_FXADV_SPLIT=1, round 5's launches.

    python tools/fxadv_parts.py [--n 192] [--reps 30]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig  # noqa: E402
from pace_amd.fv3core.stencils._common import dptr  # noqa: E402
from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist  # noqa: E402
from pace_amd.tile import DSW_ARGS, Env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    n, nz = a.n, a.nz
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    lib = _lib.load()
    env = Env(lib, "cuda", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg)
    f = {k: env.q3(s[k]) for k in DSW_ARGS}

    def phase(mask):
        lib.call("pace_d_sw_phases", mask, C.byref(dsw._geom), C.byref(dsw._met), C.byref(dsw._col), C.byref(dsw._cfg),
                 dsw._workspace.data_ptr(), *[dptr(f[k]) for k in DSW_ARGS], float(s["dt"]), None)

    out = []
    for label, mask in (("whole", 1), ("interior box", 16), ("frame", 32)):
        ts = []
        for r in range(a.reps + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            phase(mask)
            e1.record()
            torch.cuda.synchronize()
            if r >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        out.append(f"{label} {np.median(ts):.1f} (min {np.min(ts):.1f})")
    print(f"C{n} fxadv us: " + "   ".join(out), "  split launches" if os.environ.get("PACE_FXADV_SPLIT") == "1" else "  one launch")


if __name__ == "__main__":
    main()
