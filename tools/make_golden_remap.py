"""Generate tests/golden/remap_c12.npz by RUNNING THE REFERENCE's MapSingle (map_single.py:96-200, with RemapProfile,
remap_profile.py:566-681) in this container, gtscript executed by tools/gtinterp.py.

Inputs are the Lagrangian surfaces one AcousticDynamics call of the reference (n_split = 2) leaves on tile 0 of the C12
baroclinic state, remapped to the Eulerian reference coordinate ak + bk * ps exactly as LagrangianToEulerian does
(remapping.py:587-627): pt in log-pressure with qmin = t_min (iv = 1), a tracer (iv = 0), w with the surface boundary
value (iv = -2), delz (iv = 1), u on its staggered grid (iv = -1); kord 9 everywhere (the baseline configuration) and
kord 10 once.  Data only.
"""
import datetime
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
N, NZ = 12, 79


def main():
    import capture
    import pace.fv3core as fv3core
    import pace.util
    import refenv
    from pace.fv3core.stencils.map_single import MapSingle
    from pace.util import X_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM
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
        qf, sf = env.qf, env.stencil_factory
        ak = np.asarray(env.grid_data.ak.data if hasattr(env.grid_data.ak, "data") else env.grid_data.ak)
        bk = np.asarray(env.grid_data.bk.data if hasattr(env.grid_data.bk, "data") else env.grid_data.bk)
        delp = np.asarray(state.delp.data)
        ptop = float(ak[0])
        pe1 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        pe2 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        pe1.data[:, :, 0] = ptop
        for k in range(NZ):
            pe1.data[:, :, k + 1] = pe1.data[:, :, k] + delp[:, :, k]
        ps = pe1.data[:, :, NZ]
        for k in range(NZ + 1):
            pe2.data[:, :, k] = ak[k] + bk[k] * ps
        pn1 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="")
        pn2 = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="")
        pn1.data[:] = np.log(np.maximum(pe1.data, 1e-30))
        pn2.data[:] = np.log(np.maximum(pe2.data, 1e-30))
        out = {"ak": ak, "bk": bk, "pe1": np.array(pe1.data), "pe2": np.array(pe2.data), "pn1": np.array(pn1.data),
               "pn2": np.array(pn2.data)}

        def run(name, field, kord, iv, dims, a, b, qs=None, qmin=0.0):
            q = qf.zeros(dims, units="")
            q.data[:] = field
            out[name + "_in"] = np.array(q.data)
            ms = MapSingle(sf, qf, kord, iv, dims=dims)
            if qs is None:
                ms(q, a, b, qmin=qmin)
            else:
                ms(q, a, b, qs=qs, qmin=qmin)
            out[name + "_out"] = np.array(q.data)

        c3 = [X_DIM, Y_DIM, Z_DIM]
        run("pt_k9_iv1", np.asarray(state.pt.data), 9, 1, c3, pn1, pn2, qmin=184.0)
        run("qv_k9_iv0", np.asarray(state.qvapor.data), 9, 0, c3, pe1, pe2)
        wsd = qf.zeros([X_DIM, Y_DIM], units="m/s")
        wsd.data[:] = np.asarray(state.w.data)[:, :, NZ - 1] * 0.5
        out["wsd"] = np.array(wsd.data)
        run("w_k9_ivm2", np.asarray(state.w.data), 9, -2, c3, pe1, pe2, qs=wsd)
        run("delz_k9_iv1", np.asarray(state.delz.data), 9, 1, c3, pe1, pe2)
        run("u_k9_ivm1", np.asarray(state.u.data), 9, -1, [X_DIM, Y_INTERFACE_DIM, Z_DIM], pe1, pe2)
        # a strongly deformed Lagrangian coordinate (up to ~2.5 layers of displacement, varying from column to column), so
        # that target layers span several source layers and the search loop of map_single.py:65-67 iterates
        pe1s = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="Pa")
        pn1s = qf.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="")
        ii, jj = np.meshgrid(np.arange(pe1.data.shape[0]), np.arange(pe1.data.shape[1]), indexing="ij")
        amp = 0.03 * (0.25 + 0.75 * ((3 * ii + 5 * jj) % 7) / 6.0)
        sig = (pe2.data - ptop) / (ps - ptop)[:, :, None]
        pe1s.data[:] = ptop + (ps - ptop)[:, :, None] * (sig + amp[:, :, None] * np.sin(2.0 * np.pi * sig))
        pe1s.data[:, :, 0] = ptop
        pe1s.data[:, :, NZ] = ps
        pn1s.data[:] = np.log(np.maximum(pe1s.data, 1e-30))
        out["pe1s"], out["pn1s"] = np.array(pe1s.data), np.array(pn1s.data)
        run("pt_k9_iv1_s", np.asarray(state.pt.data), 9, 1, c3, pn1s, pn2, qmin=184.0)
        run("qv_k9_iv0_s", np.asarray(state.qvapor.data), 9, 0, c3, pe1s, pe2)
        run("w_k9_ivm2_s", np.asarray(state.w.data), 9, -2, c3, pe1s, pe2, qs=wsd)
        run("u_k9_ivm1_s", np.asarray(state.u.data), 9, -1, [X_DIM, Y_INTERFACE_DIM, Z_DIM], pe1s, pe2)
        run("qv_k10_iv0_s", np.asarray(state.qvapor.data), 10, 0, c3, pe1s, pe2)
        run("qv_k10_iv0", np.asarray(state.qvapor.data), 10, 0, c3, pe1, pe2)
        run("pt_k10_iv1", np.asarray(state.pt.data), 10, 1, c3, pn1, pn2, qmin=184.0)
        # FillNegativeTracerValues (fillz.py:120-163) on tracers with negative masses sprinkled in: isolated ones, runs,
        # at the top and at the bottom, small and large against the neighbours
        from pace.fv3core.stencils.fillz import FillNegativeTracerValues

        rng = np.random.default_rng(7)
        dp2 = qf.zeros(c3, units="Pa")
        dp2.data[:, :, :NZ] = pe2.data[:, :, 1:] - pe2.data[:, :, :-1]
        names = ["qvapor", "qliquid", "qrain"]
        trs = {n: qf.zeros(c3, units="kg/kg") for n in names}
        shp = trs["qvapor"].data.shape
        base = np.abs(np.asarray(state.qvapor.data)) + 1.0e-6
        for t, (n, frac, scale) in enumerate(zip(names, (0.08, 0.3, 0.6), (0.5, 2.0, 5.0))):
            f = base * (0.5 + rng.random(shp))
            neg = rng.random(shp) < frac
            f = np.where(neg, -scale * f * rng.random(shp), f)
            f[:, :, 0] = np.where(rng.random(shp[:2]) < 0.5, -np.abs(f[:, :, 0]), f[:, :, 0])
            f[:, :, NZ - 1] = np.where(rng.random(shp[:2]) < 0.5, -np.abs(f[:, :, NZ - 1]), f[:, :, NZ - 1])
            trs[n].data[:] = f
            out[f"fillz{t}_in"] = np.array(trs[n].data)
        out["fillz_dp"] = np.array(dp2.data)
        FillNegativeTracerValues(sf, qf, len(names), trs)(dp2, trs)
        for t, n in enumerate(names):
            out[f"fillz{t}_out"] = np.array(trs[n].data)
        return out

    res = run_ranks(6, rank)[0]
    # keep the compute domain only (MapSingle works on it alone; the halos of w / delz hold fill values)
    out = {}
    for k, v in res.items():
        v = np.asarray(v)
        if v.ndim == 3:
            nj = N + 1 if k.startswith("u_") else N
            v = v[3:3 + N, 3:3 + nj, :]
        elif v.ndim == 2:
            v = v[3:3 + N, 3:3 + N]
        out[k] = np.ascontiguousarray(v)
        if k in ("pe1", "pe2", "pe1s"):  # the same interfaces on u's (y-staggered) window
            out[k + "_u"] = np.ascontiguousarray(np.asarray(res[k])[3:3 + N, 3:3 + N + 1, :])
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, "remap_c12.npz"), **out)
    for k, v in out.items():
        extra = ""
        if k.endswith("_out"):
            extra = f" max|out-in| = {float(np.max(np.abs(v - out[k[:-4] + '_in']))):.3e}"
        print(k, v.shape, float(np.max(np.abs(v))), extra)


if __name__ == "__main__":
    main()
