"""Timeline of k_csw_tile's workgroups (start, phase boundaries, end on one 100 MHz clock for the device).  Development tool: needs
tools/build_prof.sh (build/var/prof/libpace_hip.so).  C192 x 79, synthetic state."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core.stencils.c_sw import CGridShallowWaterDynamics  # noqa: E402
from pace_amd.tile import Env  # noqa: E402

NB = 8192
PHASES = ["footprints -> LDS", "utmp, vtmp", "ua va uc vc ut vt", "divgd, ke, vorticity", "delp pt w -> LDS", "transport + update"]


def main():
    n, nz = 192, 79
    lib = _lib.Library(os.path.join(ROOT, "build", "var", "prof", "libpace_hip.so"))
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    names = ("delp", "pt", "u", "v", "w", "uc", "vc", "ua", "va", "ut", "vt", "divgd", "omga")
    f = {k: env.q3(s[k] if k in s else np.zeros_like(s["pt"])) for k in names}
    op = CGridShallowWaterDynamics(env.stencil_factory, env.qf, env.grid_data, nested=False, grid_type=0, nord=3)
    host = (C.c_longlong * (NB * 8))()
    for rep in range(3):
        op(*[f[k] for k in names], 0.5 * s["dt"])
        torch.cuda.synchronize()
        assert lib.cdll.pace_debug_csw_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(NB, 8).astype(float)
        a = a[a[:, 0] > 0]
        t0 = a[:, 0].min()
        start, end = (a[:, 0] - t0) / 100.0, (a[:, 7] - t0) / 100.0
        if rep == 0:
            continue
        st = np.diff(a[:, [0, 1, 2, 3, 4, 5, 7]], axis=1) / 100.0
        print(f"rep {rep}: {len(a)} workgroups, first start -> last end {end.max():.1f} us; lifetime median {np.median(end - start):.1f} us")
        print("   phases (median us): " + " | ".join(f"{p} {x:.2f}" for p, x in zip(PHASES, np.median(st, axis=0))))
        edges = np.arange(0.0, end.max() + 10.0, 10.0)
        print("   in flight every 10 us:", [int(((start <= t) & (end > t)).sum()) for t in edges])


if __name__ == "__main__":
    main()
