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
    """Output windows of translate_d_sw.py:36-65 (x-interface / y-interface / centred variables)."""
    di = 1 if name in ("mfx", "cx", "crx", "xfx", "v", "delpc", "uc", "divgd") else 0
    dj = 1 if name in ("mfy", "cy", "cry", "yfx", "u", "delpc", "vc", "divgd") else 0
    return window(n, di, dj, nk)


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
        tmp = env.q3(ut0)
        n = tmp._base.numel()
        obj._workspace[:n] = tmp._base.reshape(-1)
        tmp = env.q3(vt0)
        obj._workspace[n : 2 * n] = tmp._base.reshape(-1)
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
