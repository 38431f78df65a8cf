"""DynamicalCore -- Fortran fv_dynamics (reference: fv3core/pace/fv3core/stencils/fv_dynamics.py:92-624)."""
from datetime import timedelta

from ...util import constants
from ...util.constants import X_DIM, Y_DIM, Z_DIM, Z_INTERFACE_DIM
from ...util.halo import WrappedHaloUpdater
from .._config import DynamicalCoreConfig
from . import fvtp2d, tracer_2d_1l
from ._common import Operator, check_layout, dptr
from .c2l_ord import CubedToLatLon
from .del2cubed import HyperdiffusionDamping
from .dyn_core import AcousticDynamics
from .fillz import pointer_table, tracer_variables
from .neg_adj3 import AdjustNegativeTracerMixingRatio
from .remapping import LagrangianToEulerian

# fv_dynamics.py:33-37: 8 is the only supported value
NQ = 8


from ...util._timing import NullTimer as _NullTimer  # noqa: E402


def fvdyn_temporaries(quantity_factory):
    """fv_dynamics.py:70-86"""
    tmps = {}
    for name in ["te_2d", "te0_2d", "wsd"]:
        tmps[name] = quantity_factory.zeros(dims=[X_DIM, Y_DIM], units="unknown")
    for name in ["dp1", "cvm"]:
        tmps[name] = quantity_factory.zeros(dims=[X_DIM, Y_DIM, Z_DIM], units="unknown")
    return tmps


class DynamicalCore(Operator):
    """Corresponds to fv_dynamics in original Fortran sources."""

    def __init__(self, comm, grid_data, stencil_factory, quantity_factory, damping_coefficients, config: DynamicalCoreConfig,
                 phis, state, timestep: timedelta, checkpointer=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        nested = False
        stretched_grid = False
        assert config.moist_phys, "fvsetup is only implemented for moist_phys=true"
        assert config.nwat == 6, "Only nwat=6 has been implemented and tested"
        self.comm_rank = comm.rank
        self.grid_data = grid_data
        self._da_min = damping_coefficients.da_min
        self.config = config
        tracer_transport = fvtp2d.FiniteVolumeTransport(stencil_factory, quantity_factory, grid_data, damping_coefficients,
                                                        config.grid_type, config.hord_tr)
        self.tracers = {}
        for name in tracer_variables[0:NQ]:
            self.tracers[name] = state.__dict__[name]
        temporaries = fvdyn_temporaries(quantity_factory)
        self._te_2d = temporaries["te_2d"]
        self._te0_2d = temporaries["te0_2d"]
        self._wsd = temporaries["wsd"]
        self._dp_initial = temporaries["dp1"]
        self._cvm = temporaries["cvm"]
        self.tracer_advection = tracer_2d_1l.TracerAdvection(stencil_factory, quantity_factory, tracer_transport, grid_data, comm,
                                                             self.tracers)
        self._ak = quantity_factory.zeros([Z_INTERFACE_DIM], units="Pa")
        self._bk = quantity_factory.zeros([Z_INTERFACE_DIM], units="")
        self._ak.set(grid_data.ak)
        self._bk.set(grid_data.bk)
        self._phis = phis
        self._ptop = float(grid_data.ptop)
        self._pfull = grid_data.p
        self.acoustic_dynamics = AcousticDynamics(comm, stencil_factory, quantity_factory, grid_data, damping_coefficients,
                                                  config.grid_type, nested, stretched_grid, config.acoustic_dynamics, self._phis,
                                                  self._wsd, state, checkpointer=checkpointer)
        self._hyperdiffusion = HyperdiffusionDamping(stencil_factory, quantity_factory, damping_coefficients, grid_data.rarea,
                                                     config.nf_omega)
        self._cubed_to_latlon = CubedToLatLon(state, stencil_factory, quantity_factory, grid_data, config.c2l_ord, comm)
        self._cappa = self.acoustic_dynamics.cappa
        if not (not config.inline_q and NQ != 0):
            raise NotImplementedError("tracer_2d not implemented, turn on z_tracer")
        self._adjust_tracer_mixing_ratio = AdjustNegativeTracerMixingRatio(stencil_factory, quantity_factory,
                                                                           check_negative=config.check_negative,
                                                                           hydrostatic=config.hydrostatic)
        self._lagrangian_to_eulerian_obj = LagrangianToEulerian(stencil_factory, quantity_factory, config.remapping,
                                                                getattr(grid_data, "area_64", None), NQ, self._pfull, self.tracers,
                                                                checkpointer=checkpointer)
        full_xyz_spec = quantity_factory.get_quantity_halo_spec([X_DIM, Y_DIM, Z_DIM], n_halo=self.grid_indexing.n_halo)
        self._omega_halo_updater = WrappedHaloUpdater(comm.get_scalar_halo_updater([full_xyz_spec]), state, ["omga"], comm=comm)
        self.call_checkpointer = checkpointer is not None
        self.checkpointer = checkpointer
        self._n_split = config.n_split
        self._k_split = config.k_split
        self._conserve_total_energy = config.consv_te
        self._timestep = timestep.total_seconds()

    # ---- the reference's checkpoint call sites (fv_dynamics.py:321-423): same names, same variables ----
    def _checkpoint_fvdynamics(self, state, tag: str):
        if self.call_checkpointer:
            self.checkpointer(f"FVDynamics-{tag}", u=state.u, v=state.v, w=state.w, delz=state.delz, va=state.va, uc=state.uc,
                              vc=state.vc, qvapor=state.qvapor)

    def _checkpoint_remapping_in(self, state):
        if self.call_checkpointer:
            self.checkpointer("Remapping-In", pt=state.pt, delp=state.delp, delz=state.delz,
                              peln=state.peln.transpose([X_DIM, Z_INTERFACE_DIM, Y_DIM]), u=state.u, v=state.v, w=state.w,
                              ua=state.ua, va=state.va, cappa=self._cappa, pk=state.pk,
                              pe=state.pe.transpose([X_DIM, Z_INTERFACE_DIM, Y_DIM]), phis=state.phis, te_2d=self._te0_2d,
                              ps=state.ps, wsd=self._wsd, omga=state.omga, dp1=self._dp_initial)

    def _checkpoint_remapping_out(self, state):
        if self.call_checkpointer:
            self.checkpointer("Remapping-Out", pt=state.pt, delp=state.delp, delz=state.delz,
                              peln=state.peln.transpose([X_DIM, Z_INTERFACE_DIM, Y_DIM]), u=state.u, v=state.v, w=state.w,
                              cappa=self._cappa, pkz=state.pkz, pk=state.pk,
                              pe=state.pe.transpose([X_DIM, Z_INTERFACE_DIM, Y_DIM]), dp1=self._dp_initial)

    def _checkpoint_tracer_advection_in(self, state):
        if self.call_checkpointer:
            self.checkpointer("Tracer2D1L-In", dp1=self._dp_initial, mfxd=state.mfxd, mfyd=state.mfyd, cxd=state.cxd,
                              cyd=state.cyd)

    def _checkpoint_tracer_advection_out(self, state):
        if self.call_checkpointer:
            self.checkpointer("Tracer2D1L-Out", dp1=self._dp_initial, mfxd=state.mfxd, mfyd=state.mfyd, cxd=state.cxd,
                              cyd=state.cyd)

    def step_dynamics(self, state, timer=None):
        """Step the model state forward by one timestep."""
        self._checkpoint_fvdynamics(state=state, tag="In")
        self._compute(state, timer if timer is not None else _NullTimer())
        self._checkpoint_fvdynamics(state=state, tag="Out")

    def compute_preamble(self, state, is_root_rank: bool):
        if self.config.hydrostatic:
            raise NotImplementedError("Hydrostatic is not implemented")
        if self._conserve_total_energy > 0:
            raise NotImplementedError("compute total energy is not implemented")
        if (not self.config.rf_fast) and self.config.tau != 0:
            raise NotImplementedError("Rayleigh_Super, called when rf_fast=False and tau !=0")
        if self.config.adiabatic and self.config.kord_tm > 0:
            raise NotImplementedError("unimplemented namelist options adiabatic with positive kord_tm")
        # fv_setup + the pt adjustment (fv_dynamics.py:446-487) in one pass
        water = pointer_table([state.qvapor, state.qliquid, state.qrain, state.qsnow, state.qice, state.qgraupel])
        check_layout(self._geom, state.q_con, state.pkz, state.pt, self._cappa, state.delp, state.delz, self._dp_initial)
        self.call("pace_fv_setup_pt", water, dptr(state.q_con), dptr(state.pkz), dptr(state.pt), dptr(self._cappa),
                  dptr(state.delp), dptr(state.delz), dptr(self._dp_initial), self.stream())

    def __call__(self, *args, **kwargs):
        return self.step_dynamics(*args, **kwargs)

    def _compute(self, state, timer):
        last_step = False
        self.compute_preamble(state, is_root_rank=self.comm_rank == 0)
        for k_split in range(self._k_split):
            n_map = k_split + 1
            last_step = k_split == self._k_split - 1
            self.call("pace_copy", dptr(state.delp), dptr(self._dp_initial), self.stream())
            with timer.clock("DynCore"):
                self.acoustic_dynamics(state, timestep=self._timestep / self._k_split, n_map=n_map)
            if self.config.z_tracer:
                with timer.clock("TracerAdvection"):
                    self._checkpoint_tracer_advection_in(state)
                    self.tracer_advection(self.tracers, self._dp_initial, state.mfxd, state.mfyd, state.cxd, state.cyd)
                    self._checkpoint_tracer_advection_out(state)
            else:
                raise NotImplementedError("z_tracer=False is not implemented")
            if self.grid_indexing.domain[2] > 4:
                with timer.clock("Remapping"):
                    self._checkpoint_remapping_in(state)
                    self._lagrangian_to_eulerian_obj(
                        self.tracers, state.pt, state.delp, state.delz, state.peln, state.u, state.v, state.w, self._cappa,
                        state.q_con, state.qcld, state.pkz, state.pk, state.pe, state.phis, state.ps, self._wsd, self._ak,
                        self._bk, self._dp_initial, self._ptop, constants.KAPPA, constants.ZVIR, last_step,
                        self._conserve_total_energy, self._timestep / self._k_split)
                    self._checkpoint_remapping_out(state)
                if last_step:
                    da_min = float(self._da_min)
                    if not self.config.hydrostatic:
                        self.call("pace_omega_from_w", dptr(state.delp), dptr(state.delz), dptr(state.w), dptr(state.omga),
                                  self.stream())
                    if self.config.nf_omega > 0:
                        self._omega_halo_updater.update()
                        self._hyperdiffusion(state.omga, 0.18 * da_min)
        self._adjust_tracer_mixing_ratio(state.qvapor, state.qliquid, state.qrain, state.qsnow, state.qice, state.qgraupel,
                                         state.qcld, state.pt, state.delp)
        self._cubed_to_latlon(state.u, state.v, state.ua, state.va)
