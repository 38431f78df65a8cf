"""Generate tests/golden/d_sw_<variant>_c12_tile0_call1.npz by RUNNING THE REFERENCE's AcousticDynamics (six ranks on threads,
gtscript executed by tools/gtinterp.py) with one namelist option changed from the baseline_c12 configuration, capturing the
second d_sw call on tile 0 exactly as tools/make_golden.py does for the baseline namelist.  Data only: inputs, outputs, the
column namelist the reference derives for this configuration, the changed options.

    python tools/make_golden_dsw_variants.py h5 | nord2 | dcon0 | skeb | dddmp0

h5: advection orders 5 (the other fixtures are all order 6) . nord2: second-order-lower damping (two divergence-damping passes,
nord_v / nord_t / nord_w follow) . dcon0: no dissipative heating (d_con = 0) . skeb: do_skeb = True (the dissipation estimate
is kept) . dddmp0: no Smagorinsky term in the divergence damping (dddmp = 0)."""
import dataclasses
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
    if a.ndim != 3:
        return a
    out = np.zeros(a.shape[:2] + (len(K_SEL) + 1,))
    out[:, :, : len(K_SEL)] = a[:, :, K_SEL]
    return out


VARIANTS = {
    "h5": dict(hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5),
    "nord2": dict(nord=2),
    "dcon0": dict(d_con=0.0),
    "skeb": dict(do_skeb=True),
    "dddmp0": dict(dddmp=0.0),
}


def main(variant):
    import capture
    import pace.fv3core as fv3core
    import refenv
    from pace.fv3core.stencils import d_sw, fxadv
    from threadcomm import run_ranks

    config = dataclasses.replace(capture.dycore_config(n_split=2, npx=N + 1, npz=NZ), **VARIANTS[variant])
    rec = capture.Recorder(ranks=(0,))
    rec.instrument(d_sw.DGridShallowWaterLagrangianDynamics)
    rec.instrument(fxadv.FiniteVolumeFluxPrep)
    cols = {}

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
        if comm.Get_rank() == 0:
            col = d_sw.get_column_namelist(config.acoustic_dynamics.d_grid_shallow_water, env.qf)
            cols.update({k: np.array(v.data) for k, v in col.items()})
        return None

    run_ranks(6, rank)
    idx = 1
    r = rec.records[("rank0", "DGridShallowWaterLagrangianDynamics")][idx]
    fx = rec.records[("rank0", "FiniteVolumeFluxPrep")][idx]
    data = {"dt": r["in"]["dt"], "k_sel": np.array(K_SEL)}
    for k, v in r["in"].items():
        if isinstance(v, np.ndarray):
            data["in_" + k] = ksub(v)
    data["in_uc_contra"] = ksub(fx["in"]["uc_contra"])
    data["in_vc_contra"] = ksub(fx["in"]["vc_contra"])
    for k, v in r["out"].items():
        if isinstance(v, np.ndarray):
            data["out_" + k] = ksub(v)
    for k, v in cols.items():
        data["col_" + k] = v
    for k, v in VARIANTS[variant].items():
        data["cfg_" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLDEN, f"d_sw_{variant}_c12_tile0_call1.npz"), **data)
    print(variant, sorted(data)[:8], len(data))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "h5")
