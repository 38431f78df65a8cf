"""Generate tests/golden/l2e_c12.npz by RUNNING THE REFERENCE's LagrangianToEulerian (remapping.py:286-695) in this
container (gtscript executed by tools/gtinterp.py) on tile 0 of the C12 baroclinic state after one AcousticDynamics call
(n_split = 2), with do_sat_adj = False (the saturation adjustment is outside the scope of pace_amd), once as an
intermediate remapping step (last_step = False) and once as the last one (only pt differs: moist_pt_last_step instead
of the division by pkz).  Arrays keep the compute domain plus one halo cell ([2:16, 2:16] of the 19 x 19 storage: the
operator reads pe one cell to the south / west).  Data only.
"""
import copy
import datetime
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
N, NZ = 12, 79
W = slice(2, 16)
KORD10 = len(sys.argv) > 1 and sys.argv[1] == "kord10"


def main():
    import dataclasses

    import capture
    import pace.fv3core as fv3core
    import pace.util.constants as constants
    import refenv
    from pace.fv3core.stencils.remapping import LagrangianToEulerian
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
        if comm.Get_rank() != 0:
            return None
        rcfg = dataclasses.replace(config.remapping, do_sat_adj=False)
        if KORD10:  # "python tools/make_golden_l2e.py kord10": every remapping order 10, negatives sprinkled into four condensates
            rcfg = dataclasses.replace(rcfg, kord_tm=-10, kord_tr=10, kord_wz=10, kord_mt=10)
        names3 = ["pt", "delp", "delz", "peln", "u", "v", "w", "q_con", "pkz", "pk", "pe"]
        fields = {n: getattr(state, n) for n in names3}
        fields["cappa"] = dycore._cappa
        fields["qcld"] = state.qcld
        tracers = dycore.tracers
        # give the (all-zero) condensate species of the test case some content so that moist_cv does something
        rng = np.random.default_rng(5)
        for nm, s in (("qliquid", 2e-4), ("qrain", 1e-4), ("qice", 5e-5), ("qsnow", 3e-5), ("qgraupel", 2e-5), ("qo3mr", 1e-6),
                      ("qsgs_tke", 1e-2)):
            tracers[nm].data[:] = s * rng.random(tracers[nm].data.shape) * (np.asarray(state.qvapor.data) > 0)
            if KORD10 and nm in ("qliquid", "qrain", "qice", "qsnow"):
                neg = rng.random(tracers[nm].data.shape) < 0.05
                tracers[nm].data[:] = np.where(neg, -0.3 * np.asarray(tracers[nm].data), np.asarray(tracers[nm].data))
        out = {"ak": np.asarray(dycore._ak.data), "bk": np.asarray(dycore._bk.data), "ptop": np.float64(dycore._ptop),
               "pfull": np.asarray(dycore._pfull.data), "tracer_names": np.array(list(tracers.keys()))}
        saved = {n: np.array(q.data) for n, q in fields.items()}
        saved_tr = {n: np.array(q.data) for n, q in tracers.items()}
        saved2 = {"ps": np.array(state.ps.data), "wsd": np.array(dycore._wsd.data), "phis": np.array(state.phis.data)}
        for n, a in {**saved, **{"tr_" + k: v for k, v in saved_tr.items()}}.items():
            out["in_" + n] = a[W, W, :]
        for n, a in saved2.items():
            out["in_" + n] = a[W, W]
        for tag, last in (("mid", False), ("last", True)):
            for n, q in fields.items():
                q.data[:] = saved[n]
            for n, q in tracers.items():
                q.data[:] = saved_tr[n]
            state.ps.data[:] = saved2["ps"]
            l2e = LagrangianToEulerian(env.stencil_factory, env.qf, rcfg, env.grid_data.area_64, fv3core.stencils.fv_dynamics.NQ,
                                       dycore._pfull, tracers)
            l2e(tracers, state.pt, state.delp, state.delz, state.peln, state.u, state.v, state.w, dycore._cappa, state.q_con,
                state.qcld, state.pkz, state.pk, state.pe, state.phis, state.ps, dycore._wsd, dycore._ak, dycore._bk,
                dycore._dp_initial, dycore._ptop, constants.KAPPA, constants.ZVIR, last, config.consv_te,
                dycore._timestep / dycore._k_split)
            if tag == "mid":
                for n, q in fields.items():
                    out["out_" + n] = np.array(q.data)[W, W, :]
                for n, q in tracers.items():
                    out["out_tr_" + n] = np.array(q.data)[W, W, :]
                out["out_ps"] = np.array(state.ps.data)[W, W]
            else:
                out["out_last_pt"] = np.array(state.pt.data)[W, W, :]
        return out

    res = run_ranks(6, rank)[0]
    os.makedirs(GOLDEN, exist_ok=True)
    if KORD10:
        base = np.load(os.path.join(GOLDEN, "l2e_c12.npz"))
        slim = {}
        for k, v in res.items():
            if k.startswith("in_") and k in base.files and np.array_equal(base[k], v, equal_nan=True):
                continue  # same input as the baseline fixture
            slim[k] = v
        np.savez_compressed(os.path.join(GOLDEN, "l2e_k10_c12.npz"), **slim)
        print("kord 10:", sorted(k for k in slim if k.startswith("in_")), os.path.getsize(os.path.join(GOLDEN, "l2e_k10_c12.npz")) // 1024, "KB")
        return
    np.savez_compressed(os.path.join(GOLDEN, "l2e_c12.npz"), **res)
    for k, v in res.items():
        v = np.asarray(v)
        if v.dtype.kind == "f" and v.ndim >= 2:
            extra = ""
            if k.startswith("out_") and ("in_" + k[4:]) in res:
                a, b = v[1:13, 1:13], np.asarray(res["in_" + k[4:]])[1:13, 1:13]
                extra = f" max|out-in| (compute) = {float(np.nanmax(np.abs(a - b))):.3e}"
            print(k, v.shape, extra)


if __name__ == "__main__":
    main()
