"""How long the HOST takes to enqueue one substep (d_sw with the wind half on a side stream + riem_solver3), next to how long the
device takes to run it: python tools/host_overhead.py [--n 48]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pace_amd import _lib, synthetic  # noqa: E402
from pace_amd.fv3core import DGridShallowWaterLagrangianDynamicsConfig, RiemannConfig  # noqa: E402
from pace_amd.fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics, get_column_namelist  # noqa: E402
from pace_amd.fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver  # noqa: E402
from pace_amd.tile import DSW_ARGS, Env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    n, nz = a.n, a.nz
    lib = _lib.load()
    m = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(m, n, nz)
    env = Env(lib, "cuda", m, n, nz)
    cfg = DGridShallowWaterLagrangianDynamicsConfig()
    col = get_column_namelist(cfg, env.qf)
    dsw = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, col, False, False, cfg)
    riem = NonhydrostaticVerticalSolver(env.stencil_factory, env.qf, RiemannConfig())
    names = list(DSW_ARGS) + ["cappa", "delz", "pe", "ppe", "pk3", "pk", "peln"]
    f = {k: env.q3(s[k]) for k in names}
    zs, ws = env.q2(s["zs"]), env.q2(s["ws"])

    def step():
        dsw(*[f[k] for k in DSW_ARGS], s["dt"], overlap_winds=True)
        riem(False, s["dt"], f["cappa"], m["ptop"], zs, ws, f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"], f["pe"], f["ppe"], f["pk3"],
             f["pk"], f["peln"], f["w"])
        dsw.join()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"C{n}x{nz}: host enqueue {1e6 * (t1 - t0) / a.steps:.1f} us per step; enqueue + drain {1e6 * (t2 - t0) / a.steps:.1f} us per step")


if __name__ == "__main__":
    main()
