"""Timeline of k_divdamp_fused's workgroups: when each starts and ends (one 100 MHz clock for the device), how many run at once,
what the tile workgroups spend on their stages.  Development tool: needs tools/build_prof.sh (build/var/prof/libpace_hip.so).
C192 x 79, synthetic state, the configuration of bench.py (outputs swapped, dead outputs skipped)."""
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

NB = 8192


def main():
    n, nz = 192, 79
    lib = _lib.Library(os.path.join(ROOT, "build", "var", "prof", "libpace_hip.so"))
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg,
                                              swap_scalar_storage=True)
    dsw._cfg.flags = _lib.DSW_SKIP_DEAD_OUTPUTS
    copies = [{k: env.q3(s[k]) for k in DSW_ARGS} for _ in range(4)]

    def phase(mask, f):
        lib.call("pace_d_sw_phases", mask, C.byref(dsw._geom), C.byref(dsw._met), C.byref(dsw._col), C.byref(dsw._cfg),
                 dsw._workspace.data_ptr(), *[dptr(f[k]) for k in DSW_ARGS], float(s["dt"]), None)

    host = (C.c_longlong * (NB * 8))()
    for rep in range(4):
        f = copies[rep]
        phase(1, f)
        torch.cuda.synchronize()
        phase(14, f)  # everything after the flux preparation (the stamps are k_divdamp_fused's)
        torch.cuda.synchronize()
        assert lib.cdll.pace_debug_dd_prof(host) == 0
        a = np.frombuffer(host, dtype=np.int64).reshape(NB, 8).astype(float)
        used = a[:, 0] > 0
        a = a[used]
        t0 = a[:, 0].min()
        start, end = (a[:, 0] - t0) / 100.0, (a[:, 7] - t0) / 100.0  # us
        if rep == 0:
            continue
        print(f"rep {rep}: {len(a)} workgroups, first start -> last end {end.max():.1f} us; last start at {start.max():.1f} us")
        dur = end - start
        # kinds by duration signature: tiles have stamp 1
        tiles = a[:, 1] >= a[:, 0]
        tiles &= a[:, 1] > 0
        print(f"   tile workgroups {tiles.sum()}: duration median {np.median(dur[tiles]):.1f} us (p10 {np.percentile(dur[tiles], 10):.1f}, "
              f"p90 {np.percentile(dur[tiles], 90):.1f}); others {(~tiles).sum()}: median {np.median(dur[~tiles]):.1f} us, max {dur[~tiles].max():.1f}")
        st = np.diff(a[tiles][:, [0, 1, 2, 3, 7]], axis=1) / 100.0
        print("   tile stages (median us): footprint -> LDS %.2f | passes %.2f | vorticity -> LDS + tail %.2f | rest %.2f" % tuple(np.median(st, axis=0)))
        edges = np.arange(0.0, end.max() + 5.0, 5.0)
        conc = [int(((start <= t) & (end > t)).sum()) for t in edges]
        print("   workgroups in flight every 5 us:", conc)
        idx = np.where(used)[0]
        nsp, nstr = 3 * ((n + 1) ** 2 + 255) // 256, 2 * (-(-(n + 1) // 96) + -(-(n - 3) // 96)) * (nz - 3)  # (as launch_divergence_damping)
        for label, sel in (("sponge", idx < nsp), ("strips", (idx >= nsp) & (idx < nsp + nstr)), ("tiles", idx >= nsp + nstr)):
            if sel.sum() == 0:
                continue
            stg = np.diff(a[sel][:, [0, 1, 2, 3, 7]], axis=1) / 100.0 if label != "sponge" else None
            print(f"   {label:7s} {int(sel.sum()):5d} workgroups: duration median {np.median(dur[sel]):5.1f} us, p90 {np.percentile(dur[sel], 90):5.1f}, sum {dur[sel].sum() / 1e3:6.1f} ms"
                  + ("" if stg is None else "  | stages (median): load %.2f  passes %.2f  tail %.2f" % tuple(np.median(stg, axis=0)[:3])))
        order = np.argsort(idx)
        print("   start time of every 200th workgroup in launch order (us):", [round(float(x), 1) for x in start[order][::200]])


if __name__ == "__main__":
    main()
