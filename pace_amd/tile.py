"""One tile's worth of host objects (sizer, factories, grid data) and drivers for the two headline operators.

Used by bench.py, __graft_entry__.smoke(), the tools/ scripts and the tests: everything that needs "a tile at C<n> x nz on
device D" builds it through this module, so no product measurement imports test code.
"""
import numpy as np

DSW_ARGS = "delpc delp pt u v w uc vc ua va divgd mfx mfy cx cy crx cry xfx yfx q_con zh heat_source diss_est".split()
DSW_CFG = dict(hord_dp=6, hord_tm=6, hord_vt=6, hord_mt=6, dddmp=0.5, d4_bg=0.15, nord=3, d_con=1.0, do_skeb=False)
RIEM_ARGS = "cappa zs ws delz q_con delp pt zh p ppe pk3 pk log_p_interface w".split()


def compare(a, b, near_zero=0.0):
    """util/pace/util/testing/comparison.py:6-68: 2|a-b|/(|a|+|b|), NaN==NaN passes, optional
    near-zero escape.  Returns the max metric."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    diff = np.abs(a - b)
    denom = np.abs(a) + np.abs(b)
    with np.errstate(all="ignore"):
        rel = np.where(denom > 0, 2 * diff / np.where(denom > 0, denom, 1.0), 0.0)
    rel[both_nan] = 0.0
    rel[np.isnan(rel)] = np.inf
    if near_zero > 0:
        rel[(np.abs(a) < near_zero) & (np.abs(b) < near_zero)] = 0.0
    return float(rel.max()) if rel.size else 0.0


def window(n, di=0, dj=0, nk=None):
    return (slice(3, 3 + n + di), slice(3, 3 + n + dj), slice(0, nk))


def dsw_window(name, n, nk):
    """The output windows of TranslateD_SW, exactly (translate_d_sw.py:36-65 with the dictionaries of
    stencils/pace/stencils/testing/grid.py:288-382): `{}` = the whole domain, halo included ([isd, ied]^2 = N + 6 points each
    way): delpc, delp, pt, w, q_con, ua, va, heat_source, diss_est; x3d / y3d_domain_dict (one more point along the staggered axis):
    uc, v / vc, u; x3d_compute_domain_y / y3d_compute_domain_x: the Courant numbers and area fluxes; x3d / y3d_compute_dict: the mass
    fluxes; default_dict_buffer_2d: divgd.  This is the window of the FULL contract (pace_dsw_config_t.flags == 0)."""
    full = n + 6
    if name in ("uc", "v"):
        return (slice(0, full + 1), slice(0, full), slice(0, nk))
    if name in ("vc", "u"):
        return (slice(0, full), slice(0, full + 1), slice(0, nk))
    if name in ("xfx", "crx", "cx"):
        return (slice(3, 3 + n + 1), slice(0, full), slice(0, nk))
    if name in ("yfx", "cry", "cy"):
        return (slice(0, full), slice(3, 3 + n + 1), slice(0, nk))
    if name == "mfx":
        return window(n, 1, 0, nk)
    if name == "mfy":
        return window(n, 0, 1, nk)
    if name == "divgd":
        return (slice(0, full + 1), slice(0, full + 1), slice(0, nk))
    return (slice(0, full), slice(0, full), slice(0, nk))


DSW_DEAD = ("delpc", "divgd", "uc", "vc")  # unspecified under PACE_DSW_SKIP_DEAD_OUTPUTS (include/pace_hip.h)


def dsw_live_window(name, n, nk):
    """What a call with PACE_DSW_SKIP_DEAD_OUTPUTS specifies: as dsw_window, but the four transported scalars without the 3 x 3
    corner blocks of the halo (the in-place corner copies of the reference's transport are not replayed: the halo update that
    follows d_sw in AcousticDynamics overwrites them) -- their compute domain; DSW_DEAD not at all."""
    if name in ("delp", "pt", "w", "q_con"):
        return window(n, 0, 0, nk)
    return dsw_window(name, n, nk)


class Env:
    """Everything a test needs to call the host classes on one device."""

    def __init__(self, lib, device, metrics, n, nz):
        from .dsl import CompilationConfig, GridIndexing, StencilConfig, StencilFactory
        from .util import QuantityFactory, SubtileGridSizer
        from .util.grid import DampingCoefficients, GridData

        self.n, self.nz = n, nz
        self.sizer = SubtileGridSizer.from_tile_params(nx_tile=n, ny_tile=n, nz=nz, n_halo=3, extra_dim_lengths={}, layout=(1, 1))
        import torch

        # the storage type follows the library: float64 for libpace_hip.so, float32 for the _f32 build
        self.qf = QuantityFactory(self.sizer, device=device, dtype=torch.float32 if lib.real_bytes == 4 else torch.float64)
        self.grid_indexing = GridIndexing.from_sizer_and_communicator(self.sizer, None)
        self.stencil_factory = StencilFactory(StencilConfig(compilation_config=CompilationConfig()), self.grid_indexing, lib=lib,
                                              quantity_factory=self.qf)
        self.grid_data = GridData(self.qf, metrics)
        self.damping = DampingCoefficients(self.grid_data)

    def q3(self, array=None):
        q = self.qf.zeros(["x", "y", "z"], "")
        if array is not None:
            q.set(array)
        return q

    def q2(self, array=None):
        q = self.qf.zeros(["x", "y"], "")
        if array is not None:
            q.set(array)
        return q

    def kq(self, array):
        q = self.qf.zeros(["z"], "")
        a = np.zeros(self.nz + 1)
        a[: len(array)] = array[: self.nz + 1]
        q.set(a)
        return q


def run_d_sw(env, col, inputs, dt, ut0=None, vt0=None, cfg=None):
    """inputs: dict name -> numpy (N+7, N+7, nz+1).  Returns dict of numpy outputs."""
    import torch

    from .fv3core import DGridShallowWaterLagrangianDynamicsConfig
    from .fv3core.stencils.d_sw import DGridShallowWaterLagrangianDynamics

    config = DGridShallowWaterLagrangianDynamicsConfig(**(cfg or {}))
    colq = {k: env.kq(v) for k, v in col.items()}
    obj = DGridShallowWaterLagrangianDynamics(env.stencil_factory, env.qf, env.grid_data, env.damping, colq, nested=False,
                                              stretched_grid=False, config=config)
    if ut0 is not None:
        # uc_contra / vc_contra carried from the previous call live at the head of the workspace
        # (the workspace tensor is float64; the kernels carve it in the library's storage type)
        ws = obj._workspace.view(env.qf.real)
        tmp = env.q3(ut0)
        n = tmp._base.numel()
        ws[:n] = tmp._base.reshape(-1)
        tmp = env.q3(vt0)
        ws[n : 2 * n] = tmp._base.reshape(-1)
    f = {k: env.q3(inputs[k]) for k in DSW_ARGS}
    obj(*[f[k] for k in DSW_ARGS], dt)
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    return {k: f[k].numpy() for k in DSW_ARGS}, obj


def run_riem3(env, inputs, last_call, dt, ptop, p_fac=0.05):
    import torch

    from .fv3core import RiemannConfig
    from .fv3core.stencils.riem_solver3 import NonhydrostaticVerticalSolver

    obj = NonhydrostaticVerticalSolver(env.stencil_factory, env.qf, RiemannConfig(p_fac=p_fac))
    f = {k: (env.q2(inputs[k]) if inputs[k].ndim == 2 else env.q3(inputs[k])) for k in RIEM_ARGS}
    obj(last_call, dt, f["cappa"], ptop, f["zs"], f["ws"], f["delz"], f["q_con"], f["delp"], f["pt"], f["zh"], f["p"], f["ppe"],
        f["pk3"], f["pk"], f["log_p_interface"], f["w"])
    if env.qf.device.type == "cuda":
        torch.cuda.synchronize()
    return {k: f[k].numpy() for k in RIEM_ARGS}


# ------------------------------------------------------------------------------------------------------------------
# Inputs of the measured substep taken from the baroclinic test case itself (bench.py --state baroclinic)
# ------------------------------------------------------------------------------------------------------------------
RIEM_ONLY = ("cappa", "delz", "pe", "ppe", "pk3", "pk", "peln")


def baroclinic_substep_inputs(lib, device, n, nz, tile, dt_atmos=None, cache_dir=None):
    """What `d_sw` and `riem_solver3` are handed in the SECOND acoustic substep of the first DynamicalCore step of the
    Jablonowski-Williamson baroclinic case on the gnomonic cubed sphere C<n> x <nz>: grid from pace_amd.util.gridgen, initial
    state from pace_amd's init_baroclinic_state, the six tiles stepped together on `device` (one host thread per tile,
    halo exchanges through ThreadComm) and the fields of tile `tile` captured at the reference's "D_SW-In" checkpoint.

    Returns (metrics, fields, scalars): the tile's metric terms, numpy arrays for DSW_ARGS + RIEM_ONLY + zs + ws, and
    {"dt": acoustic substep, "ptop": ...}.  The result is cached as an .npz under `cache_dir` (the run is deterministic)."""
    import datetime
    import os

    path = None
    if cache_dir:
        path = os.path.join(cache_dir, f"pace_amd_baroclinic_c{n}x{nz}_tile{tile}_r{lib.real_bytes}.npz")
        if os.path.exists(path):
            d = dict(np.load(path))
            metrics = {k[2:]: d[k] for k in d if k.startswith("m_")}
            fields = {k[2:]: d[k] for k in d if k.startswith("f_")}
            return metrics, fields, {"dt": float(d["dt"]), "ptop": float(d["ptop"])}
    from .fv3core import (AcousticDynamicsConfig, DGridShallowWaterLagrangianDynamicsConfig, DynamicalCoreConfig, RiemannConfig)
    from .fv3core.initialization.baroclinic import baroclinic_state_six_tiles
    from .fv3core.initialization.dycore_state import DycoreState
    from .fv3core.stencils.fv_dynamics import DynamicalCore
    from .util import CubedSphereCommunicator, gridgen, run_tiles

    tiles = gridgen.tiles(n, nz)
    states = baroclinic_state_six_tiles(tiles, n, nz)
    n_split = 2
    if dt_atmos is None:
        dt_atmos = 2 * 225.0 * 48.0 / n / 2.0  # the C48 namelist's acoustic substep (225 s / 8 ... kept at Courant ~ C48's)
    captured = {}

    def program(comm):
        t = comm.Get_rank()
        if device != "cpu":
            import torch

            # the current device is per THREAD: a tile thread would otherwise launch on device 0's streams (rank r of a
            # multi-GPU run owns device r)
            d = torch.device(device)
            torch.cuda.set_device(d if d.index is not None else torch.device("cuda", torch.cuda.current_device()))
        metrics = {k: v for k, v in tiles[t].items() if k not in ("ee1", "ee2", "es1", "ew2")}
        env = Env(lib, device, metrics, n, nz)
        cube = CubedSphereCommunicator(comm, device=device, lib=lib)
        arrays = {k: states[t][k] for k in "u v w delz delp pe pk peln phis uc vc ua va pt qvapor ps".split()}
        state = DycoreState.init_from_numpy_arrays(arrays, env.qf)
        ac = AcousticDynamicsConfig(n_split=n_split, k_split=1, nord=3, d_con=1.0, rf_fast=True, rf_cutoff=3000.0, tau=10.0, p_fac=0.05,
                                    hord_tm=6, delt_max=0.002, d_grid_shallow_water=DGridShallowWaterLagrangianDynamicsConfig(),
                                    riemann=RiemannConfig(p_fac=0.05))
        config = DynamicalCoreConfig(npx=n + 1, npy=n + 1, npz=nz, dt_atmos=dt_atmos, k_split=1, n_split=n_split, acoustic_dynamics=ac)
        seen = {"n": 0}
        box = {}

        def checkpointer(name, **fields):
            if name != "D_SW-In":
                return
            seen["n"] += 1
            if seen["n"] != 2 or t != tile:
                return
            dyn = box["core"].acoustic_dynamics
            q = {"delpc": dyn._vt, "delp": state.delp, "pt": state.pt, "u": state.u, "v": state.v, "w": state.w, "uc": state.uc,
                 "vc": state.vc, "ua": state.ua, "va": state.va, "divgd": dyn._divgd, "mfx": state.mfxd, "mfy": state.mfyd,
                 "cx": state.cxd, "cy": state.cyd, "crx": dyn._crx, "cry": dyn._cry, "xfx": dyn._xfx, "yfx": dyn._yfx,
                 "q_con": state.q_con, "zh": dyn._zh, "heat_source": dyn._heat_source, "diss_est": state.diss_estd,
                 "cappa": dyn.cappa, "delz": state.delz, "pe": state.pe, "ppe": dyn._pkc, "pk3": dyn._pk3, "pk": state.pk,
                 "peln": state.peln, "zs": dyn._zs, "ws": dyn._wsd}
            if device != "cpu":
                import torch

                torch.cuda.synchronize()
            captured.update({k: v.numpy().astype(np.float64) for k, v in q.items()})

        core = DynamicalCore(cube, env.grid_data, env.stencil_factory, env.qf, env.damping, config, state.phis, state,
                             datetime.timedelta(seconds=dt_atmos), checkpointer=checkpointer)
        box["core"] = core
        core.step_dynamics(state)
        if device != "cpu":
            import torch

            torch.cuda.synchronize()
        return None

    run_tiles(6, program)
    missing = [k for k in list(DSW_ARGS) + list(RIEM_ONLY) + ["zs", "ws"] if k not in captured]
    if missing:
        raise RuntimeError(f"the D_SW-In checkpoint of tile {tile} did not deliver {missing}")
    bad = [k for k, v in captured.items() if not np.isfinite(v[3:3 + n, 3:3 + n]).all()]
    if bad:
        raise RuntimeError(f"non-finite values in the captured state: {bad}")
    metrics = {k: np.asarray(v) for k, v in tiles[tile].items() if k not in ("ee1", "ee2", "es1", "ew2")}
    scalars = {"dt": dt_atmos / n_split, "ptop": float(metrics["ptop"])}
    if path:  # written under a private name, then renamed: another rank may be looking for the same tile's file
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, dt=scalars["dt"], ptop=scalars["ptop"], **{"m_" + k: v for k, v in metrics.items()},
                 **{"f_" + k: v for k, v in captured.items()})
        os.replace(tmp, path)
    return metrics, captured, scalars
