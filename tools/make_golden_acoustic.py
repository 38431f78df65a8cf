"""Generate tests/golden/acoustic_c12_*.npz by RUNNING THE REFERENCE's AcousticDynamics in this container: 6 tile ranks on
threads (tools/threadcomm.py), real halo exchanges, every gtscript stencil executed by tools/gtinterp.py (see
make_golden.py for what that means).  One call of n_split = 2 substeps at C12 x 79L, k_split = 1 (so the call is the
last one: remap_step / end_step paths are exercised).

Per tile the fixture holds the grid metrics, the full state going in (+ cappa, which DynamicalCore's preamble computes),
and the state coming out on a level subset plus a few full columns.  Data only.
"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
K_SEL = [0, 1, 2, 3, 4, 40, 77, 78, 79]
COLS = [(3, 3), (8, 9), (14, 14), (3, 14)]
N, NZ, N_SPLIT = 12, 79, 2
STATE_IN = "u v w delz delp pt pe pk peln q_con omga ua va uc vc mfxd mfyd cxd cyd diss_estd phis".split()
STATE_OUT = "u v w delz delp pt pe pk peln q_con omga ua va mfxd mfyd cxd cyd diss_estd".split()
# what d_sw leaves in its work fields uc / vc (compared by TranslateDynCore, translate_dyncore.py:84-85): kept in a separate,
# later-added file (acoustic_c12_ucvc.npz) so that the six per-tile fixtures stay byte-identical
EXTRA_OUT = ["uc", "vc"]
FULL_TILES = [0, 2]
# `python tools/make_golden_acoustic.py <variant>`: the same call with several namelist options changed at once; only the outputs
# (level subset + columns, all six tiles), the options and the column namelist the reference derives are stored
# (acoustic_c12_<variant>.npz) -- the inputs are those of the baseline fixtures
VARIANTS = {"v2": dict(nord=2, d_con=0.0, hord_dp=5, hord_tm=5, hord_vt=5, hord_mt=5)}


def main():
    import capture
    import datetime
    import pace.fv3core as fv3core
    import refenv
    from threadcomm import run_ranks

    import dataclasses

    variant = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in VARIANTS else None
    config = capture.dycore_config(n_split=N_SPLIT, npx=N + 1, npz=NZ)
    if variant:
        config = dataclasses.replace(config, **VARIANTS[variant])
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
        snap = capture._snap
        before = {k: snap(getattr(state, k)) for k in STATE_IN}
        before["cappa"] = snap(dycore.acoustic_dynamics.cappa)
        dycore.acoustic_dynamics(state, timestep=dycore._timestep / dycore._k_split, n_map=1)
        after = {k: snap(getattr(state, k)) for k in STATE_OUT}
        after["heat_source"] = snap(dycore.acoustic_dynamics._heat_source)
        for k in EXTRA_OUT:
            after[k] = snap(getattr(state, k))
        grid = {}
        for name in dir(env.grid_data):
            if name.startswith("_"):
                continue
            try:
                v = getattr(env.grid_data, name)
            except Exception:  # noqa: BLE001
                continue
            s = snap(v)
            if s is not None and not isinstance(s, str):
                grid[name] = s
        for name in ["del6_u", "del6_v", "divg_u", "divg_v", "da_min", "da_min_c"]:
            grid[name] = snap(getattr(env.damping, name))
        if comm.Get_rank() == 0:
            from pace.fv3core.stencils import d_sw

            col = d_sw.get_column_namelist(config.acoustic_dynamics.d_grid_shallow_water, env.qf)
            cols.update({k: np.array(v.data) for k, v in col.items()})
        return grid, before, after, float(dycore._timestep / dycore._k_split)

    out = run_ranks(6, rank)
    if variant:
        data = {"k_sel": np.array(K_SEL), "cols": np.array(COLS), "timestep": out[0][3], "n_split": N_SPLIT}
        for k, v in VARIANTS[variant].items():
            data["cfg_" + k] = np.asarray(v)
        for k, v in cols.items():
            data["namelist_" + k] = v
        for t, (grid, before, after, timestep) in enumerate(out):
            ref_in = np.load(os.path.join(GOLDEN, f"acoustic_c12_tile{t}.npz"))
            for k, v in before.items():
                assert np.array_equal(ref_in["in_" + k], v, equal_nan=True), (t, k)  # same inputs as the baseline fixture
            for k, v in after.items():
                data[f"out_{k}_tile{t}"] = np.ascontiguousarray(v[3 : 3 + N + 1, 3 : 3 + N + 1][:, :, K_SEL])
                data[f"col_{k}_tile{t}"] = np.stack([v[i, j, :] for (i, j) in COLS])
        path = os.path.join(GOLDEN, f"acoustic_c12_{variant}.npz")
        np.savez_compressed(path, **data)
        print(variant, os.path.getsize(path) // 1024, "KB")
        return
    os.makedirs(GOLDEN, exist_ok=True)
    extra = {"k_sel": np.array(K_SEL), "cols": np.array(COLS)}
    for t, (grid, before, after, timestep) in enumerate(out):
        after = dict(after)
        for k in EXTRA_OUT:
            v = after.pop(k)
            extra[f"out_{k}_tile{t}"] = np.ascontiguousarray(v[3 : 3 + N + 1, 3 : 3 + N + 1][:, :, K_SEL])
            extra[f"col_{k}_tile{t}"] = np.stack([v[i, j, :] for (i, j) in COLS])
        path = os.path.join(GOLDEN, f"acoustic_c12_tile{t}.npz")
        if os.path.exists(path) and "--rewrite" not in sys.argv:
            # regression check of the tool chain: the committed fixture must come out of this run bit for bit
            old = np.load(path)
            for k, v in after.items():
                assert np.array_equal(old["out_" + k], v[3 : 3 + N + 1, 3 : 3 + N + 1][:, :, K_SEL], equal_nan=True), (t, k)
            for k, v in before.items():
                assert np.array_equal(old["in_" + k], v, equal_nan=True), (t, k)
            print("tile", t, "reproduces the committed fixture")
            continue
        g = {k: v for k, v in grid.items() if isinstance(v, (float, int)) or (isinstance(v, np.ndarray) and v.ndim <= 2)}
        for k in ("edge_w", "edge_e"):
            g[k] = np.ascontiguousarray(g[k][0, :]) if g[k].ndim == 2 else g[k]
        data = {"grid_" + k: v for k, v in g.items()}
        data.update({"in_" + k: v for k, v in before.items()})
        for k, v in after.items():
            data["out_" + k] = np.ascontiguousarray(v[3 : 3 + N + 1, 3 : 3 + N + 1][:, :, K_SEL])
            data["col_" + k] = np.stack([v[i, j, :] for (i, j) in COLS])
        data["k_sel"], data["cols"] = np.array(K_SEL), np.array(COLS)
        data["timestep"], data["n_split"] = timestep, N_SPLIT
        np.savez_compressed(os.path.join(GOLDEN, f"acoustic_c12_tile{t}.npz"), **data)
    np.savez_compressed(os.path.join(GOLDEN, "acoustic_c12_ucvc.npz"), **extra)
    # FULL output fields (every level of the compute window + the staggered row / column) for an equatorial and a polar tile:
    # the per-tile fixtures above hold a level subset and four columns only
    full = {}
    for t in FULL_TILES:
        after = dict(out[t][2])
        for k, v in after.items():
            full[f"out_{k}_tile{t}"] = np.ascontiguousarray(v[3 : 3 + N + 1, 3 : 3 + N + 1, :])
    np.savez_compressed(os.path.join(GOLDEN, "acoustic_c12_full.npz"), tiles=np.array(FULL_TILES), **full)
    for f in sorted(os.listdir(GOLDEN)):
        print(f, os.path.getsize(os.path.join(GOLDEN, f)) // 1024, "KB")


if __name__ == "__main__":
    main()
