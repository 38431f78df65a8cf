"""Timing of whole DynamicalCore steps on the GPU box: the six tiles of a cubed sphere resident on ONE device (one host
thread per tile, halo exchanges through ThreadComm), synthetic balanced state, fp64.

    python tools/dycore_bench.py [--n 192] [--nz 79] [--n-split 4] [--steps 2]

Prints the wall time per step for all six tiles (the device is shared, so per tile = / 6) and the split between the
acoustic loop, the tracer advection, the remapping and the rest, from synchronising timers.
"""
import argparse
import datetime
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


class SyncTimer:
    def __init__(self):
        self.t = {}

    def clock(self, name):
        timer = self

        class Ctx:
            def __enter__(self):
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()

            def __exit__(self, *a):
                torch.cuda.synchronize()
                timer.t[name] = timer.t.get(name, 0.0) + time.perf_counter() - self.t0
                return False

        return Ctx()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--n-split", type=int, default=4)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--precision", type=int, default=64, choices=(64, 32), help="storage type of the fields: libpace_hip.so or libpace_hip_f32.so")
    ap.add_argument("--single", action="store_true",
                    help="one tile only, behind a lone-rank LoopbackComm (each halo receives what the tile sent to that neighbour): the "
                         "device time of one tile's step without the thread rendezvous of the six-tile mode")
    args = ap.parse_args()
    if os.environ.get("PACE_BENCH_TRACE"):
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["PACE_BENCH_TRACE"]), exit=True)
    from helpers import Env, acoustic_config, dycore_condensates

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DynamicalCoreConfig
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.fv_dynamics import DynamicalCore
    from pace_amd.util import CubedSphereCommunicator, LoopbackComm, constants as c, run_tiles

    lib = _lib.load(args.precision)
    n, nz = args.n, args.nz
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    dt_atmos = float(s["dt"]) * args.n_split
    results = {}
    lock = threading.Lock()

    def program(comm):
        tile = comm.Get_rank()
        env = Env(lib, "cuda", metrics, n, nz)
        cube = CubedSphereCommunicator(comm, device="cuda", lib=lib)
        arrays = {k: s[k] for k in "u v w delz delp pt pe uc vc ua va q_con".split()}
        arrays["peln"] = np.log(s["pe"])
        arrays["pk"] = np.exp(c.KAPPA * arrays["peln"])
        arrays["phis"] = c.GRAV * s["zs"]
        arrays["ps"] = s["pe"][:, :, nz]
        arrays["pt"] = s["pt"] * np.exp(c.KAPPA * np.log(1.0e5))  # a temperature-like magnitude
        arrays["qvapor"] = 0.01 * np.exp(-6.0 * (1.0 - s["pe"] / s["pe"][:, :, nz:])) * (s["delp"] > 0)
        for name, f in dycore_condensates(tile, s["delp"].shape).items():
            arrays[name] = np.abs(f)
        state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
        config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=dt_atmos, k_split=1, n_split=args.n_split,
                                     acoustic_dynamics=acoustic_config(args.n_split))
        core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                             datetime.timedelta(seconds=dt_atmos))
        core.step_dynamics(state)  # warm-up
        torch.cuda.synchronize()
        comm.barrier()
        timer = SyncTimer()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            core.step_dynamics(state, timer)
        torch.cuda.synchronize()
        comm.barrier()
        wall = time.perf_counter() - t0
        with lock:
            results[tile] = (wall, dict(timer.t), float(np.isnan(state.w.numpy()).mean()), float(np.isnan(state.pt.numpy()).mean()))

    if args.single:
        program(LoopbackComm(rank=0, total_ranks=6))
        wall = results[0][0] / args.steps
        cells = n * n * nz
        print(f"C{n} x {nz}L, ONE tile (lone-rank LoopbackComm), n_split = {args.n_split}, k_split = 1, {args.steps} steps")
        print(f"wall per step: {1e3 * wall:9.2f} ms   ({cells * args.n_split / wall / 1e9:5.2f} G cell-updates/s counting the "
              f"acoustic substeps; the timers synchronise the device at every section boundary)")
    else:
        run_tiles(6, program)
        wall = max(r[0] for r in results.values()) / args.steps
        cells = 6 * n * n * nz
        print(f"C{n} x {nz}L, six tiles on one device, n_split = {args.n_split}, k_split = 1, {args.steps} steps")
        print(f"wall per step (six tiles): {1e3 * wall:9.2f} ms   = {1e3 * wall / 6:7.2f} ms per tile   "
              f"({cells * args.n_split / wall / 1e9:5.2f} G cell-updates/s counting the acoustic substeps)")
    t = results[0][1]
    tot = sum(t.values())
    for k, v in t.items():
        print(f"  {k:18s} {1e3 * v / args.steps:9.2f} ms of tile 0's wall time (threads interleave on the device) {100 * v / tot:5.1f} %")
    print(f"  NaN fraction after the run: w {results[0][2]:.2e}, pt {results[0][3]:.2e}")


if __name__ == "__main__":
    main()
