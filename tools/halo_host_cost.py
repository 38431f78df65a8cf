"""Host cost of the halo updates of the acoustic loop, measured with what one GPU can show: six tiles on ONE device (one Python
thread per tile, pace_amd.util.ThreadComm: the transfers are device-side copies), AcousticDynamics with n_split substeps.
Every HaloUpdater.start / wait (pack launch + post, wait + unpack launch: pace_amd/util/halo.py; reference
util/pace/util/halo_updater.py:217-303) is timed on the host -- CPU time of the calling thread (time.thread_time: not inflated by
the other five threads holding the interpreter) and wall time -- per updater, next to the wall time of the loop body.
What this does NOT show: the host cost of RCCL's grouped send / receive (TorchDistComm.exchange), which needs one process per GPU.

    python tools/halo_host_cost.py [--n 96] [--n_split 4]
"""
import argparse
import collections
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--n_split", type=int, default=4)
    a = ap.parse_args()
    import torch

    import opchain
    from helpers import Env, acoustic_config
    from pace_amd import _lib
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.dyn_core import AcousticDynamics
    from pace_amd.util import CubedSphereCommunicator, run_tiles
    from pace_amd.util import halo as halo_mod

    lib = _lib.load()
    n, nz, n_split = a.n, a.nz, a.n_split
    ms, tiles = opchain.loop_inputs(n, nz, "synthetic")
    timestep = 3.571 * n_split
    acc = collections.defaultdict(lambda: [0.0, 0.0, 0])  # (name, phase) -> cpu seconds, wall seconds, calls
    lock = threading.Lock()
    timing = {"on": False}
    names = {}

    def wrap(method, phase):
        orig = getattr(halo_mod.HaloUpdater, method)

        def timed(self, *args, **kw):
            c0, w0 = time.thread_time(), time.perf_counter()
            r = orig(self, *args, **kw)
            c1, w1 = time.thread_time(), time.perf_counter()
            if timing["on"] and threading.current_thread().name.endswith("tile0"):
                with lock:
                    e = acc[(names.get(id(self), f"updater {self._tag}"), phase)]
                    e[0] += c1 - c0
                    e[1] += w1 - w0
                    e[2] += 1
            return r

        setattr(halo_mod.HaloUpdater, method, timed)

    wrap("start", "start")
    wrap("wait", "wait")
    walls = {}

    def tile(comm):
        rank = comm.Get_rank()
        threading.current_thread().name = f"tile{rank}"
        env = Env(lib, "cuda", ms[rank], n, nz)
        cube = CubedSphereCommunicator(comm, device="cuda", lib=lib)
        state = DycoreState.init_from_numpy_arrays(tiles[rank][0], env.qf)
        dyn = AcousticDynamics(cube, env.stencil_factory, env.qf, env.grid_data, env.damping, 0, False, False, acoustic_config(n_split),
                               state.phis, env.q2(), state)
        dyn.cappa.set(tiles[rank][1])
        if rank == 0:
            for k, w in vars(dyn._halo_updaters).items():
                up = getattr(w, "_updater", None)
                if up is not None:
                    names[id(up)] = k
        dyn(state, timestep=timestep, n_map=1)  # warm-up
        torch.cuda.synchronize()
        comm.barrier() if hasattr(comm, "barrier") else None
        if rank == 0:
            timing["on"] = True
        t0 = time.perf_counter()
        dyn(state, timestep=timestep, n_map=1)
        torch.cuda.synchronize()
        walls[rank] = time.perf_counter() - t0
        return None

    run_tiles(6, tile)
    wall = max(walls.values())
    print(f"C{n} x {nz}, six tiles on one device, n_split = {n_split}: loop body {1e3 * wall / n_split:.3f} ms wall per substep (six tiles sharing the device)")
    print(f"{'updater':22s} {'calls':>6s} {'start cpu us':>13s} {'start wall us':>14s} {'wait cpu us':>12s} {'wait wall us':>13s}")
    tot_cpu = 0.0
    for k in sorted({k for k, _ in acc}):
        s, w = acc[(k, "start")], acc[(k, "wait")]
        tot_cpu += s[0] + w[0]
        print(f"{k:22s} {s[2]:6d} {1e6 * s[0] / max(1, s[2]):13.1f} {1e6 * s[1] / max(1, s[2]):14.1f} {1e6 * w[0] / max(1, w[2]):12.1f} {1e6 * w[1] / max(1, w[2]):13.1f}")
    print(f"host CPU time of all halo start / wait calls of one tile: {1e6 * tot_cpu / n_split:.0f} us per substep "
          f"= {100 * tot_cpu / wall:.1f} % of the loop body's wall time")


if __name__ == "__main__":
    main()
