"""Generate the committed fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

    python tools/make_golden.py            # writes tests/golden/*.npz
    python tools/make_golden.py --cache    # only (re)build /tmp/ref_capture.pkl for crosscheck_oracle.py

What runs: the reference's own Python (grid generation, baroclinic initial state, AcousticDynamics
orchestration, halo exchange) imported from /root/reference, six tile ranks on threads
(tools/threadcomm.py), with every gtscript stencil executed by tools/gtinterp.py because GT4Py itself
is not installable here.  Two AcousticDynamics calls of n_split=2 substeps at C12 x 79L; inputs and
outputs of each component call on ranks 0 and 1 are captured (tools/capture.py).

Fixtures are data only (inputs + expected outputs); horizontal operators are stored on a subset of
levels (they are level-independent given the per-level column parameters), column solvers on a subset
of columns.
"""
import os
import pickle
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
K_SEL = [0, 1, 2, 3, 4, 40, 78]
N, NZ = 12, 79


def build_capture(path="/tmp/ref_capture.pkl"):
    import capture
    from pace.fv3core.stencils import (a2b_ord4, c_sw, d2a2c_vect, d_sw, del2cubed, delnflux, divergence_damping, fvtp2d, fxadv,
                                       nh_p_grad, pk3_halo, ray_fast, riem_solver3, riem_solver_c, sim1_solver, updatedzc,
                                       updatedzd, xppm, yppm)

    rec = capture.Recorder(ranks=(0, 1))
    for cls in [d_sw.DGridShallowWaterLagrangianDynamics, c_sw.CGridShallowWaterDynamics,
                riem_solver3.NonhydrostaticVerticalSolver, riem_solver_c.NonhydrostaticVerticalSolverCGrid,
                updatedzc.UpdateGeopotentialHeightOnCGrid, updatedzd.UpdateHeightOnDGrid,
                nh_p_grad.NonHydrostaticPressureGradient, fvtp2d.FiniteVolumeTransport, fxadv.FiniteVolumeFluxPrep,
                delnflux.DelnFlux, delnflux.DelnFluxNoSG, divergence_damping.DivergenceDamping,
                a2b_ord4.AGrid2BGridFourthOrder, xppm.XPiecewiseParabolic, yppm.YPiecewiseParabolic,
                ray_fast.RayleighDamping, del2cubed.HyperdiffusionDamping, pk3_halo.PK3Halo,
                d2a2c_vect.DGrid2AGrid2CGridVectors, sim1_solver.Sim1Solver]:
        rec.instrument(cls)
    envs = capture.run_acoustic(nx=N, nz=NZ, n_split=2, recorder=rec, n_calls=2)
    out = {"records": dict(rec.records)}
    for r in (0, 1):
        e = envs[r]
        g = {}
        for name in dir(e.grid_data):
            if name.startswith("_"):
                continue
            try:
                v = getattr(e.grid_data, name)
            except Exception:  # noqa: BLE001
                continue
            s = capture._snap(v)
            if s is not None and not isinstance(s, str):
                g[name] = s
        for name in ["del6_u", "del6_v", "divg_u", "divg_v", "da_min", "da_min_c"]:
            g[name] = capture._snap(getattr(e.damping, name))
        out[f"grid{r}"] = g
        out[f"before{r}"] = e.before
        out[f"after{r}"] = e.after
    pickle.dump(out, open(path, "wb"))
    return out


def ksub(a):
    """levels K_SEL of a 3-D field + one zero spare level (the reference's nz+1 allocation)."""
    if not isinstance(a, np.ndarray) or a.ndim != 3:
        return a
    out = np.zeros(a.shape[:2] + (len(K_SEL) + 1,))
    out[:, :, : len(K_SEL)] = a[:, :, K_SEL]
    return out


def main():
    path = "/tmp/ref_capture.pkl"
    cap = pickle.load(open(path, "rb")) if os.path.exists(path) and "--fresh" not in sys.argv else build_capture(path)
    if "--cache" in sys.argv:
        return
    os.makedirs(GOLDEN, exist_ok=True)
    for r in (0, 1):
        g = {k: v for k, v in cap[f"grid{r}"].items() if isinstance(v, (float, int)) or (isinstance(v, np.ndarray) and v.ndim <= 2)}
        for k in ("edge_w", "edge_e"):
            g[k] = np.ascontiguousarray(g[k][0, :]) if g[k].ndim == 2 else g[k]
        np.savez_compressed(os.path.join(GOLDEN, f"grid_c12_tile{r}.npz"), **g)
    # column namelist values the reference derives for this config (d_sw.get_column_namelist)
    from capture import dycore_config
    import refenv
    from pace.fv3core.stencils import d_sw as rdsw

    env0 = refenv.build_all(N, NZ, with_state=False)[0]
    col = rdsw.get_column_namelist(dycore_config().acoustic_dynamics.d_grid_shallow_water, env0.qf)
    col_np = {k: np.array(v.data) for k, v in col.items()}
    np.savez_compressed(os.path.join(GOLDEN, "column_namelist_c12.npz"), **col_np)

    # D_SW: rank 0 call 1 (2nd substep of the 1st acoustic call), rank 1 call 3
    for rank, idx in ((0, 1), (1, 3)):
        rec = cap["records"][(f"rank{rank}", "DGridShallowWaterLagrangianDynamics")][idx]
        # uc_contra / vc_contra state the object carried in (from the previous call's FxAdv output)
        fx = cap["records"][(f"rank{rank}", "FiniteVolumeFluxPrep")][idx]
        data = {"dt": rec["in"]["dt"], "k_sel": np.array(K_SEL)}
        for k, v in rec["in"].items():
            if isinstance(v, np.ndarray):
                data["in_" + k] = ksub(v)
        data["in_uc_contra"] = ksub(fx["in"]["uc_contra"])
        data["in_vc_contra"] = ksub(fx["in"]["vc_contra"])
        for k, v in rec["out"].items():
            if isinstance(v, np.ndarray):
                data["out_" + k] = ksub(v)
        np.savez_compressed(os.path.join(GOLDEN, f"d_sw_c12_tile{rank}_call{idx}.npz"), **data)
    # Riem_Solver3: rank 0, calls 2 (last_call False) and 3 (True); columns j in [3, 7)
    JS = slice(3, 7)
    for idx in (2, 3):
        rec = cap["records"][("rank0", "NonhydrostaticVerticalSolver")][idx]
        data = {k: rec["in"][k] for k in ("last_call", "dt", "ptop")}
        for k, v in rec["in"].items():
            if isinstance(v, np.ndarray):
                data["in_" + k] = v[3:15, JS]
        for k, v in rec["out"].items():
            if isinstance(v, np.ndarray) and v.ndim == 3:
                data["out_" + k] = v[3:15, JS]
        np.savez_compressed(os.path.join(GOLDEN, f"riem_solver3_c12_tile0_call{idx}.npz"), **data)
    # FvTp2d: rank 0 calls 6..11 (second d_sw), level subset
    recs = cap["records"][("rank0", "FiniteVolumeTransport")]
    for idx in (6, 8):
        rec = recs[idx]
        data = {}
        for k, v in rec["in"].items():
            if isinstance(v, np.ndarray):
                data["in_" + k] = ksub(v)
        for k in ("q_x_flux", "q_y_flux"):
            data["out_" + k] = ksub(rec["out"][k])
        np.savez_compressed(os.path.join(GOLDEN, f"fvtp2d_c12_tile0_call{idx}.npz"), **data)
    for f in sorted(os.listdir(GOLDEN)):
        print(f, os.path.getsize(os.path.join(GOLDEN, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
