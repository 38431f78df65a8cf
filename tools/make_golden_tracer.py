"""Generate tests/golden/tracer_c12_tile*.npz by RUNNING THE REFERENCE's TracerAdvection (tracer_2d_1l.py:171-392) in this
container: 6 tile ranks on threads, real halo exchanges, gtscript stencils executed by tools/gtinterp.py.

The mass fluxes / Courant numbers fed in are those one AcousticDynamics call of the reference accumulates (n_split = 2);
two tracers are advected: the model's specific humidity and a second, deterministic field built from it (the other
tracers of the baroclinic test state are identically zero).  The operator has no vertical coupling, so the fixture keeps
a subset of levels of every input and output.  Data only.
"""
import datetime
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
K_SEL = [0, 1, 2, 3, 4, 40, 78]
N, NZ = 12, 79


def ksub(a):
    out = np.zeros(a.shape[:2] + (len(K_SEL) + 1,))
    out[:, :, : len(K_SEL)] = a[:, :, K_SEL]
    return out


def main():
    import capture
    import pace.fv3core as fv3core
    import pace.util
    import refenv
    from pace.fv3core.stencils import fvtp2d, tracer_2d_1l
    from threadcomm import run_ranks

    config = capture.dycore_config(n_split=2, npx=N + 1, npz=NZ)

    def rank(comm):
        env = refenv.build_rank(comm, N, NZ)
        dycore = fv3core.DynamicalCore(
            comm=env.cube, grid_data=env.grid_data, stencil_factory=env.stencil_factory, quantity_factory=env.qf,
            damping_coefficients=env.damping, config=config, timestep=datetime.timedelta(seconds=config.dt_atmos),
            phis=env.state.phis, state=env.state)
        state = env.state
        dycore.compute_preamble(state, is_root_rank=comm.Get_rank() == 0)
        dycore._copy_stencil(state.delp, dycore._dp_initial)
        dycore.acoustic_dynamics(state, timestep=dycore._timestep / dycore._k_split, n_map=1)
        q2 = env.qf.zeros([pace.util.X_DIM, pace.util.Y_DIM, pace.util.Z_DIM], units="kg/kg")
        q2.data[:] = 0.5 * state.qvapor.data + 1.0e-6 * state.pt.data
        tracers = {"qvapor": state.qvapor, "q2": q2}
        transport = fvtp2d.FiniteVolumeTransport(stencil_factory=env.stencil_factory, quantity_factory=env.qf, grid_data=env.grid_data,
                                                 damping_coefficients=env.damping, grid_type=0, hord=8)
        adv = tracer_2d_1l.TracerAdvection(env.stencil_factory, env.qf, transport, env.grid_data, env.cube, tracers)
        snap = capture._snap
        args = dict(qvapor=state.qvapor, q2=q2, dp1=dycore._dp_initial, mfxd=state.mfxd, mfyd=state.mfyd, cxd=state.cxd, cyd=state.cyd)
        before = {k: snap(v) for k, v in args.items()}
        adv(tracers, dycore._dp_initial, state.mfxd, state.mfyd, state.cxd, state.cyd)
        after = {k: snap(v) for k, v in args.items()}
        return before, after

    out = run_ranks(6, rank)
    for t, (before, after) in enumerate(out):
        data = {"k_sel": np.array(K_SEL)}
        data.update({"in_" + k: ksub(v) for k, v in before.items()})
        data.update({"out_" + k: ksub(v) for k, v in after.items()})
        np.savez_compressed(os.path.join(GOLDEN, f"tracer_c12_tile{t}.npz"), **data)
    for f in sorted(os.listdir(GOLDEN)):
        if f.startswith("tracer"):
            print(f, os.path.getsize(os.path.join(GOLDEN, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
