"""Run the reference's AcousticDynamics (6 ranks on threads, stencils executed by
tools/gtinterp) and capture inputs/outputs of every component call on chosen ranks.

Dev-container tool; see make_golden.py.
"""
import collections
import datetime
import inspect
import threading

import numpy as np

import refenv  # installs the shim
from threadcomm import run_ranks

import pace.fv3core as fv3core  # noqa: E402
import pace.util  # noqa: E402


def dycore_config(n_split=2, k_split=1, dt_atmos=225.0, npx=13, npz=79, do_sat_adj=True):
    return fv3core.DynamicalCoreConfig(
        layout=(1, 1), npx=npx, npy=npx, npz=npz, ntiles=6, nwat=6, dt_atmos=dt_atmos, a_imp=1.0, beta=0.0,
        consv_te=False, d2_bg=0.0, d2_bg_k1=0.2, d2_bg_k2=0.1, d4_bg=0.15, d_con=1.0, d_ext=0.0, dddmp=0.5,
        delt_max=0.002, do_sat_adj=do_sat_adj, do_vort_damp=True, fill=True, hord_dp=6, hord_mt=6, hord_tm=6,
        hord_tr=8, hord_vt=6, hydrostatic=False, k_split=k_split, ke_bg=0.0, kord_mt=9, kord_tm=-9, kord_tr=9,
        kord_wz=9, n_split=n_split, nord=3, p_fac=0.05, rf_fast=True, rf_cutoff=3000.0, tau=10.0, vtdm4=0.06,
        z_tracer=True, do_qa=True,
    )


def _snap(v):
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if hasattr(v, "dims") and hasattr(v, "data"):
        return np.array(v.data, copy=True)
    if isinstance(v, np.ndarray):
        return np.array(v, copy=True)
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    return None


class Recorder:
    def __init__(self, ranks=(0,)):
        self.ranks = {f"rank{r}" for r in ranks}
        self.records = collections.defaultdict(list)  # (rank, name) -> [ {in:{}, out:{}} ]
        self._lock = threading.Lock()

    def instrument(self, cls, name=None, method="__call__"):
        name = name or cls.__name__
        orig = getattr(cls, method)
        sig = inspect.signature(orig)
        rec = self

        def wrapped(obj, *args, **kwargs):
            tname = threading.current_thread().name
            if tname not in rec.ranks:
                return orig(obj, *args, **kwargs)
            bound = sig.bind(obj, *args, **kwargs)
            items = [(k, v) for k, v in bound.arguments.items() if k != "self"]
            entry = {"in": {k: _snap(v) for k, v in items}}
            out = orig(obj, *args, **kwargs)
            entry["out"] = {k: _snap(v) for k, v in items}
            with rec._lock:
                rec.records[(tname, name)].append(entry)
            return out

        setattr(cls, method, wrapped)


def run_acoustic(nx=12, nz=79, n_split=2, recorder=None, n_calls=1):
    """Returns per-rank (env, dycore, state_before, state_after) after one AcousticDynamics call."""
    config = dycore_config(n_split=n_split, npx=nx + 1, npz=nz)

    def rank(comm):
        env = refenv.build_rank(comm, nx, nz)
        dycore = fv3core.DynamicalCore(
            comm=env.cube, grid_data=env.grid_data, stencil_factory=env.stencil_factory, quantity_factory=env.qf,
            damping_coefficients=env.damping, config=config,
            timestep=datetime.timedelta(seconds=config.dt_atmos), phis=env.state.phis, state=env.state,
        )
        state = env.state
        dycore.compute_preamble(state, is_root_rank=comm.Get_rank() == 0)
        dycore._copy_stencil(state.delp, dycore._dp_initial)
        env.dycore = dycore
        env.before = {k: _snap(getattr(state, k)) for k in state.__dict__ if _snap(getattr(state, k)) is not None}
        for n in range(n_calls):
            dycore.acoustic_dynamics(state, timestep=dycore._timestep / dycore._k_split, n_map=1)
        env.after = {k: _snap(getattr(state, k)) for k in state.__dict__ if _snap(getattr(state, k)) is not None}
        env.config = config
        return env

    return run_ranks(6, rank)
