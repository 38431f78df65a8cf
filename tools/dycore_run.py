"""Whole-model run: DynamicalCore.step_dynamics on the six tiles of a cubed sphere, ONE PROCESS PER TILE over torch.distributed
(RCCL on GPUs, gloo with --cpu-emulation), synthetic balanced state, fp64.  Launch with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/dycore_run.py [--tile-size 192] [--nz 79] [--n-split 6] [--steps 5] [--warmup 1]

Rank 0 prints one JSON line: wall time per step (max over ranks, barrier + synchronize brackets) and the cell-updates/s it
amounts to (cells x acoustic substeps per step, all six tiles).
"""
import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile-size", dest="n", type=int, default=192)
    ap.add_argument("--nz", type=int, default=79)
    ap.add_argument("--n-split", type=int, default=6)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-emulation", action="store_true", help="gloo + the CPU emulation library (logic check only)")
    ap.add_argument("--precision", type=int, default=64, choices=(64, 32),
                    help="storage type of the fields: 32 = libpace_hip_f32.so (BASELINE configuration 5: --tile-size 384 --nz 91 --precision 32)")
    args = ap.parse_args()
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != 6:
        raise SystemExit("a cubed sphere has six tiles: launch six ranks")
    if args.cpu_emulation:
        dist.init_process_group(backend="gloo")
        device = "cpu"
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    from helpers import Env, acoustic_config, dycore_condensates

    from pace_amd import _lib, synthetic
    from pace_amd.fv3core import DynamicalCoreConfig
    from pace_amd.fv3core.initialization.dycore_state import DycoreState
    from pace_amd.fv3core.stencils.fv_dynamics import DynamicalCore
    from pace_amd.util import CubedSphereCommunicator, TorchDistComm, constants as c

    if args.cpu_emulation:
        lib = _lib.Library(os.path.join(ROOT, "tests", "emu", "libpace_emu_f32.so" if args.precision == 32 else "libpace_emu.so"))
    else:
        lib = _lib.load(args.precision)
    n, nz = args.n, args.nz
    metrics = synthetic.tile_metrics(n, nz)
    s = synthetic.acoustic_state(metrics, n, nz)
    dt_atmos = float(s["dt"]) * args.n_split
    env = Env(lib, device, metrics, n, nz)
    cube = CubedSphereCommunicator(TorchDistComm(), device=device, lib=lib)
    arrays = {k: s[k] for k in "u v w delz delp pt pe uc vc ua va q_con".split()}
    arrays["peln"] = np.log(s["pe"])
    arrays["pk"] = np.exp(c.KAPPA * arrays["peln"])
    arrays["phis"] = c.GRAV * s["zs"]
    arrays["ps"] = s["pe"][:, :, nz]
    arrays["pt"] = s["pt"] * np.exp(c.KAPPA * np.log(1.0e5))
    arrays["qvapor"] = 0.01 * np.exp(-6.0 * (1.0 - s["pe"] / s["pe"][:, :, nz:])) * (s["delp"] > 0)
    for name, f in dycore_condensates(rank, s["delp"].shape).items():
        arrays[name] = np.abs(f)
    state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
    config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=dt_atmos, k_split=1, n_split=args.n_split,
                                 acoustic_dynamics=acoustic_config(args.n_split))
    core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                         datetime.timedelta(seconds=dt_atmos))

    def barrier():
        dist.barrier()
        if device != "cpu":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        core.step_dynamics(state)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        core.step_dynamics(state)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    nan = float(np.isnan(state.w.numpy()).mean())
    if rank == 0:
        cells = 6 * n * n * nz
        print(json.dumps({"workload": f"DynamicalCore.step_dynamics, C{n}x{nz}L, six tiles, n_split={args.n_split}, k_split=1, " + ("fp64" if args.precision == 64 else "float32 fields"),
                          "n_gpus": 0 if args.cpu_emulation else 6, "steps": args.steps, "ms_per_step": 1e3 * elapsed / args.steps,
                          "cell_updates_per_s": cells * args.n_split * args.steps / elapsed, "nan_fraction_w": nan,
                          "transport": "gloo (CPU emulation)" if args.cpu_emulation else "RCCL, one process per GPU"}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
