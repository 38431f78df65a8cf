"""AcousticDynamics -- Fortran dyn_core (reference: fv3core/pace/fv3core/stencils/dyn_core.py:174-970).

The sequencing, the eleven halo-updater groups and their start/wait placement are the reference's.  Every numerical
step is a class-level HIP entry point; the halo transfers run on RCCL's stream between ``start`` and ``wait`` while the
compute stream keeps launching (the reference begins every ``start`` with a device-wide synchronise,
halo_updater.py:223 -- nothing here does)."""
import ctypes as C
from typing import Dict

import numpy as np
import torch

from ... import _lib
from ...util import constants
from ...util.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM, Z_INTERFACE_DIM
from ...util.halo import WrappedHaloUpdater
from .._config import AcousticDynamicsConfig
from . import d_sw
from ._common import Operator, dptr
from .c_sw import CGridShallowWaterDynamics
from .del2cubed import HyperdiffusionDamping
from .nh_p_grad import NonHydrostaticPressureGradient
from .pk3_halo import PK3Halo
from .ray_fast import RayleighDamping
from .riem_solver3 import NonhydrostaticVerticalSolver
from .riem_solver_c import NonhydrostaticVerticalSolverCGrid
from .updatedzc import UpdateGeopotentialHeightOnCGrid
from .updatedzd import UpdateHeightOnDGrid

HUGE_R = 1.0e40


# The definition functions of the one-launch stencils of dyn_core.py:51-171.  In the reference these are gtscript bodies; here
# their bodies are the device kernels registered under the same identities (pace_amd/dsl/device_stencils.py), and -- exactly as
# in the reference -- AcousticDynamics builds FrozenStencils from them through its StencilFactory.
def zero_data(mfxd, mfyd, cxd, cyd, heat_source, diss_estd, first_timestep):
    """dyn_core.py:51-80"""


def gz_from_surface_height_and_thicknesses(zs, delz, gz):
    """dyn_core.py:83-96"""


def interface_pressure_from_toa_pressure_and_thickness(delp, pem, ptop):
    """dyn_core.py:99-112"""


def compute_geopotential(zh, gz):
    """dyn_core.py:115-117"""


def p_grad_c_stencil(rdxc, rdyc, uc, vc, delpc, pkc, gz, dt2):
    """dyn_core.py:120-171"""


def get_nk_heat_dissipation(config, npz: int) -> int:
    """dyn_core.py:174-189."""
    if config.convert_ke or config.vtdm4 > 1.0e-4:
        return npz
    if config.d2_bg_k1 < 1.0e-3:
        return 0
    if config.d2_bg_k2 < 1.0e-3:
        return 1
    return 2


def dyncore_temporaries(quantity_factory) -> Dict[str, object]:
    """dyn_core.py:192-218."""
    t = {}
    for name in ["ut", "vt", "pem", "pk3", "heat_source", "cappa"]:
        t[name] = quantity_factory.zeros([X_DIM, Y_DIM, Z_DIM], units="unknown")
    for name in ["gz", "pkc", "zh"]:
        t[name] = quantity_factory.zeros([X_DIM, Y_DIM, Z_INTERFACE_DIM], units="unknown")
    t["divgd"] = quantity_factory.zeros([X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM], units="unknown")
    t["ws3"] = quantity_factory.zeros([X_DIM, Y_DIM], units="unknown")
    for name in ["crx", "xfx"]:
        t[name] = quantity_factory.zeros([X_INTERFACE_DIM, Y_DIM, Z_DIM], units="unknown")
    for name in ["cry", "yfx"]:
        t[name] = quantity_factory.zeros([X_DIM, Y_INTERFACE_DIM, Z_DIM], units="unknown")
    return t


class AcousticDynamics(Operator):
    class _HaloUpdaters:
        """dyn_core.py:227-343."""

        def __init__(self, comm, grid_indexing, quantity_factory, state, cappa, gz, zh, divgd, heat_source, pkc):
            spec = quantity_factory.get_quantity_halo_spec
            n_halo = grid_indexing.n_halo
            xyz = spec([X_DIM, Y_DIM, Z_DIM], n_halo=n_halo)
            xyiz = spec([X_DIM, Y_INTERFACE_DIM, Z_DIM], n_halo=n_halo)
            xiyz = spec([X_INTERFACE_DIM, Y_DIM, Z_DIM], n_halo=n_halo)
            xyzi = spec([X_DIM, Y_DIM, Z_INTERFACE_DIM], n_halo=n_halo)
            xiyiz = spec([X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM], n_halo=n_halo)
            W = WrappedHaloUpdater
            self.q_con__cappa = W(comm.get_scalar_halo_updater([xyz] * 2), dict(q_con=state.q_con, cappa=cappa), ["q_con", "cappa"])
            self.delp__pt = W(comm.get_scalar_halo_updater([xyz] * 2), state, ["delp", "pt"])
            self.u__v = W(comm.get_vector_halo_updater([xyiz], [xiyz]), state, ["u"], ["v"])
            self.w = W(comm.get_scalar_halo_updater([xyz]), state, ["w"])
            self.gz = W(comm.get_scalar_halo_updater([xyzi]), {"gz": gz}, ["gz"])
            self.delp__pt__q_con = W(comm.get_scalar_halo_updater([xyz] * 3), state, ["delp", "pt", "q_con"])
            self.zh = W(comm.get_scalar_halo_updater([xyzi]), {"zh": zh}, ["zh"])
            self.divgd = W(comm.get_scalar_halo_updater([xiyiz]), {"divgd": divgd}, ["divgd"])
            self.heat_source = W(comm.get_scalar_halo_updater([xyz]), {"heat_source": heat_source}, ["heat_source"])
            two_pts = spec([X_DIM, Y_DIM, Z_INTERFACE_DIM], n_halo=2)
            self.pkc = W(comm.get_scalar_halo_updater([two_pts]), {"pkc": pkc}, ["pkc"])
            self.uc__vc = W(comm.get_vector_halo_updater([xiyz], [xyiz]), state, ["uc"], ["vc"])
            self.interface_uc__vc = W(None, state, ["u"], ["v"], comm=comm)

    def __init__(self, comm, stencil_factory, quantity_factory, grid_data, damping_coefficients, grid_type, nested,
                 stretched_grid, config: AcousticDynamicsConfig, phis, wsd, state, checkpointer=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        grid_indexing = stencil_factory.grid_indexing
        self.config = config
        assert config.d_ext == 0, "d_ext != 0 is not implemented"
        assert config.beta == 0, "beta != 0 is not implemented"
        assert not config.use_logp, "use_logp=True is not implemented"
        if config.hydrostatic:
            raise NotImplementedError("the hydrostatic configuration is not implemented")
        self.call_checkpointer = checkpointer is not None
        self.checkpointer = checkpointer
        self._da_min = damping_coefficients.da_min
        self.grid_data = grid_data
        self._ptop = grid_data.ptop
        self._pfull = grid_data.p
        self._wsd = wsd
        nz = grid_indexing.domain[2]
        self._nk_heat_dissipation = get_nk_heat_dissipation(config.d_grid_shallow_water, npz=nz)
        self.nonhydrostatic_pressure_gradient = NonHydrostaticPressureGradient(stencil_factory, quantity_factory=quantity_factory,
                                                                              grid_data=grid_data, grid_type=config.grid_type)
        t = dyncore_temporaries(quantity_factory)
        self._heat_source, self._divgd, self._gz, self._pkc, self._zh = t["heat_source"], t["divgd"], t["gz"], t["pkc"], t["zh"]
        self.cappa, self._ut, self._vt, self._pem, self._pk3 = t["cappa"], t["ut"], t["vt"], t["pem"], t["pk3"]
        self._crx, self._cry, self._xfx, self._yfx, self._ws3 = t["crx"], t["cry"], t["xfx"], t["yfx"], t["ws3"]
        # (1e40 overflows float32 storage: the largest finite value then)
        self._pk3.data[:] = min(HUGE_R, float(torch.finfo(self._pk3.data.dtype).max))
        column_namelist = d_sw.get_column_namelist(config.d_grid_shallow_water, quantity_factory=quantity_factory)
        self._dp_ref = grid_data.dp_ref
        self._zs = quantity_factory.zeros([X_DIM, Y_DIM], units="m")
        self._zs.data[:] = phis.data / constants.GRAV
        self.update_height_on_d_grid = UpdateHeightOnDGrid(stencil_factory, quantity_factory=quantity_factory,
                                                           damping_coefficients=damping_coefficients, grid_data=grid_data,
                                                           grid_type=grid_type, hord_tm=config.hord_tm,
                                                           column_namelist=column_namelist)
        self.vertical_solver = NonhydrostaticVerticalSolver(stencil_factory, quantity_factory=quantity_factory, config=config.riemann)
        self.vertical_solver_cgrid = NonhydrostaticVerticalSolverCGrid(stencil_factory, quantity_factory=quantity_factory,
                                                                       p_fac=config.p_fac)
        self.dgrid_shallow_water_lagrangian_dynamics = d_sw.DGridShallowWaterLagrangianDynamics(
            stencil_factory, quantity_factory=quantity_factory, grid_data=grid_data, damping_coefficients=damping_coefficients,
            column_namelist=column_namelist, nested=nested, stretched_grid=stretched_grid, config=config.d_grid_shallow_water,
            swap_scalar_storage=True)  # (this class owns the state between its halo updates: no stale aliases; d_sw.py docstring)
        self.cgrid_shallow_water_lagrangian_dynamics = CGridShallowWaterDynamics(
            stencil_factory, quantity_factory=quantity_factory, grid_data=grid_data, nested=nested, grid_type=config.grid_type,
            nord=config.nord)
        self.update_geopotential_height_on_c_grid = UpdateGeopotentialHeightOnCGrid(
            stencil_factory, quantity_factory=quantity_factory, area=grid_data.area, dp_ref=grid_data.dp_ref, grid_data=grid_data)
        self._do_del2cubed = self._nk_heat_dissipation != 0 and config.d_con > 1.0e-5
        if self._do_del2cubed:
            nf_ke = min(3, config.nord + 1)
            self._hyperdiffusion = HyperdiffusionDamping(stencil_factory, quantity_factory=quantity_factory,
                                                         damping_coefficients=damping_coefficients, rarea=grid_data.rarea, nmax=nf_ke)
        if config.rf_fast:
            self._rayleigh_damping = RayleighDamping(stencil_factory, rf_cutoff=config.rf_cutoff, tau=config.tau,
                                                     hydrostatic=config.hydrostatic, quantity_factory=quantity_factory)
        self._pk3_halo = PK3Halo(stencil_factory, quantity_factory)
        self._build_stencils(stencil_factory, grid_indexing)
        self._halo_updaters = AcousticDynamics._HaloUpdaters(comm, grid_indexing, quantity_factory, state, cappa=self.cappa,
                                                             gz=self._gz, zh=self._zh, divgd=self._divgd,
                                                             heat_source=self._heat_source, pkc=self._pkc)

    # ---- the one-launch dyn_core stencils: FrozenStencils from the factory, windows as in dyn_core.py:480-587 ----
    def _build_stencils(self, stencil_factory, grid_indexing):
        from . import basic_operations as basic
        from . import pe_halo, temperature_adjust

        sf = stencil_factory
        if getattr(sf, "quantity_factory", None) is None:
            sf.quantity_factory = self._qf  # a factory built with the reference's signature: the layout is this object's
        origin, domain = grid_indexing.get_origin_domain([X_DIM, Y_DIM, Z_INTERFACE_DIM], halos=(2, 2))
        self._compute_geopotential_stencil = sf.from_origin_domain(compute_geopotential, origin=origin, domain=domain)
        self._gz_from_surface_height_and_thickness = sf.from_origin_domain(
            gz_from_surface_height_and_thicknesses, origin=grid_indexing.origin_compute(),
            domain=grid_indexing.domain_compute(add=(0, 0, 1)))
        self._interface_pressure_from_toa_pressure_and_thickness = sf.from_origin_domain(
            interface_pressure_from_toa_pressure_and_thickness, origin=grid_indexing.origin_compute(add=(-1, -1, 0)),
            domain=grid_indexing.domain_compute(add=(2, 2, 0)))
        self._p_grad_c = sf.from_origin_domain(p_grad_c_stencil, origin=grid_indexing.origin_compute(),
                                               domain=grid_indexing.domain_compute(add=(1, 1, 0)),
                                               externals={"hydrostatic": self.config.hydrostatic})
        self._zero_data = sf.from_origin_domain(zero_data, origin=grid_indexing.origin_full(), domain=grid_indexing.domain_full())
        ax_offsets_pe = grid_indexing.axis_offsets(grid_indexing.origin_full(), grid_indexing.domain_full(add=(0, 0, 1)))
        self._edge_pe_stencil = sf.from_origin_domain(pe_halo.edge_pe, origin=grid_indexing.origin_full(),
                                                      domain=grid_indexing.domain_full(add=(0, 0, 1)), externals={**ax_offsets_pe},
                                                      skip_passes=("PruneKCacheFills",))
        if self._nk_heat_dissipation > 0:
            self._apply_diffusive_heating = sf.from_origin_domain(
                temperature_adjust.apply_diffusive_heating, origin=grid_indexing.origin_compute(),
                domain=grid_indexing.restrict_vertical(nk=self._nk_heat_dissipation).domain_compute())
        self._copy_stencil = sf.from_origin_domain(basic.copy_defn, origin=grid_indexing.origin_full(),
                                                   domain=grid_indexing.domain_full(add=(0, 0, 1)))

    def _get_da_min(self) -> float:
        return self._da_min

    # ----------------------------------------------------------------------------------------------------
    # ---- the reference's checkpoint call sites (dyn_core.py:608-669) ----
    def _checkpoint_csw(self, state, tag: str):
        if self.call_checkpointer:
            self.checkpointer(f"C_SW-{tag}", delpd=state.delp, ptd=state.pt, ud=state.u, vd=state.v, wd=state.w, ucd=state.uc,
                              vcd=state.vc, uad=state.ua, vad=state.va, utd=self._ut, vtd=self._vt, divgdd=self._divgd)

    def _checkpoint_dsw_in(self, state):
        if self.call_checkpointer:
            # delpc is a temporary and not a variable in D_SW savepoint
            self.checkpointer("D_SW-In", ucd=state.uc, vcd=state.vc, wd=state.w, delpcd=self._vt, delpd=state.delp, ud=state.u,
                              vd=state.v, ptd=state.pt, uad=state.ua, vad=state.va, zhd=self._zh, divgdd=self._divgd,
                              xfxd=self._xfx, yfxd=self._yfx, mfxd=state.mfxd, mfyd=state.mfyd)

    def _checkpoint_dsw_out(self, state):
        if self.call_checkpointer:
            self.dgrid_shallow_water_lagrangian_dynamics.join()  # the wind half may still be running on the side stream
            self.checkpointer("D_SW-Out", ucd=state.uc, vcd=state.vc, wd=state.w, delpcd=self._vt, delpd=state.delp, ud=state.u,
                              vd=state.v, ptd=state.pt, uad=state.ua, vad=state.va, divgdd=self._divgd, xfxd=self._xfx,
                              yfxd=self._yfx, mfxd=state.mfxd, mfyd=state.mfyd)

    def __call__(self, state, timestep: float, n_map=1):
        """dyn_core.py:670-970."""
        cfg = self.config
        end_step = n_map == cfg.k_split
        akap = constants.KAPPA
        dt_acoustic_substep = timestep / cfg.n_split
        dt2 = 0.5 * dt_acoustic_substep
        n_split = cfg.n_split
        halo = self._halo_updaters
        halo.q_con__cappa.start()
        halo.delp__pt.start()
        halo.u__v.start()
        halo.q_con__cappa.wait()
        self._zero_data(state.mfxd, state.mfyd, state.cxd, state.cyd, self._heat_source, state.diss_estd, n_map == 1)
        csw = self.cgrid_shallow_water_lagrangian_dynamics
        for it in range(n_split):
            remap_step = cfg.breed_vortex_inline or (it == n_split - 1)
            halo.w.start()
            if it == 0:
                self._gz_from_surface_height_and_thickness(self._zs, state.delz, self._gz)
                halo.gz.start()
            if it == 0:
                halo.delp__pt.wait()
            if it == n_split - 1 and end_step and cfg.use_old_omega:
                self._interface_pressure_from_toa_pressure_and_thickness(state.delp, self._pem, self._ptop)
            csw_args = (state.delp, state.pt, state.u, state.v, state.w, state.uc, state.vc, state.ua, state.va, self._ut, self._vt,
                        self._divgd, state.omga, dt2)
            # while the u / v (and w) strips travel: the part of c_sw's first pass that reads no halo value of u / v
            # (not with a checkpointer attached: the "C_SW-In" / "D_SW-In" savepoints must hold the state BEFORE the call, as the
            # reference's do -- the early starts already overwrite uad / vad, xfx / yfx / crx / cry and accumulate cx / cy)
            early = not self.call_checkpointer
            if early:
                csw.start_interior(*csw_args)
            halo.u__v.wait()
            halo.w.wait()
            self._checkpoint_csw(state, tag="In")
            csw(*csw_args)
            self._checkpoint_csw(state, tag="Out")
            if cfg.nord > 0:
                halo.divgd.start()
            if it == 0:
                halo.gz.wait()
                self._copy_stencil(self._gz, self._zh)
            else:
                self._copy_stencil(self._zh, self._gz)
            self.update_geopotential_height_on_c_grid(self._zs, self._ut, self._vt, self._gz, self._ws3, dt2)
            self.vertical_solver_cgrid(dt2, self.cappa, self._ptop, state.phis, self._ws3, csw.ptc, state.q_con, csw.delpc,
                                       self._gz, self._pkc, state.omga)
            self._p_grad_c(self.grid_data.rdxc, self.grid_data.rdyc, state.uc, state.vc, csw.delpc, self._pkc, self._gz, dt2)
            halo.uc__vc.start()
            # delpc of d_sw aliases vt, as in the reference (dyn_core.py:825)
            dsw_args = (self._vt, state.delp, state.pt, state.u, state.v, state.w, state.uc, state.vc, state.ua, state.va, self._divgd,
                        state.mfxd, state.mfyd, state.cxd, state.cyd, self._crx, self._cry, self._xfx, self._yfx, state.q_con,
                        self._zh, self._heat_source, state.diss_estd, dt_acoustic_substep)
            # while the uc / vc strips travel: the interior of d_sw's flux preparation (it reads no halo value of uc / vc)
            if early:
                self.dgrid_shallow_water_lagrangian_dynamics.start_flux_preparation(*dsw_args)
            if cfg.nord > 0:
                halo.divgd.wait()
            halo.uc__vc.wait()
            self._checkpoint_dsw_in(state)
            # delpc, divgd, uc, vc are work fields c_sw recomputes in the next substep: d_sw brings them to the reference's final
            # state only where something can see it -- after the last substep (TranslateDynCore compares uc / vc) or a checkpointer
            self.dgrid_shallow_water_lagrangian_dynamics(*dsw_args, overlap_winds=True,
                                                         skip_dead_outputs=(it != n_split - 1 and not self.call_checkpointer))
            self._checkpoint_dsw_out(state)
            # dyn_core.py:854 updates the halos of delp / pt / q_con right here.  Nothing before pk3_halo reads them (updatedzd works
            # on zh and the Courant numbers, the column solver on the compute domain's columns), so without checkpoints the
            # exchange is only POSTED here and waited for after the column solver -- it travels while the two run.
            if early:
                halo.delp__pt__q_con.start()
            else:
                halo.delp__pt__q_con.update()
            self.update_height_on_d_grid(surface_height=self._zs, height=self._zh, courant_number_x=self._crx,
                                         courant_number_y=self._cry, x_area_flux=self._xfx, y_area_flux=self._yfx, ws=self._wsd,
                                         dt=dt_acoustic_substep)
            self.vertical_solver(remap_step, dt_acoustic_substep, self.cappa, self._ptop, self._zs, self._wsd, state.delz,
                                 state.q_con, state.delp, state.pt, self._zh, state.pe, self._pkc, self._pk3, state.pk, state.peln,
                                 state.w)
            if early:
                halo.delp__pt__q_con.wait()
            halo.zh.start()
            halo.pkc.start()
            if remap_step:
                self._edge_pe_stencil(state.pe, state.delp, self._ptop)
            self._pk3_halo(self._pk3, state.delp, self._ptop, akap)
            halo.zh.wait()
            self._compute_geopotential_stencil(self._zh, self._gz)
            halo.pkc.wait()
            self.dgrid_shallow_water_lagrangian_dynamics.join()  # the overlapped wind half of d_sw
            self.nonhydrostatic_pressure_gradient(state.u, state.v, self._pkc, self._gz, self._pk3, state.delp, dt_acoustic_substep,
                                                  self._ptop, akap)
            if cfg.rf_fast:
                self._rayleigh_damping(u=state.u, v=state.v, w=state.w, dp=self._dp_ref, pfull=self._pfull,
                                       dt=dt_acoustic_substep, ptop=self._ptop)
            if it != n_split - 1:
                halo.u__v.start()
            elif cfg.grid_type < 4:
                halo.interface_uc__vc.interface()
        if self._do_del2cubed:
            halo.heat_source.update()
            cd = constants.CNST_0P20 * self._get_da_min()
            self._hyperdiffusion(self._heat_source, cd)
            delt_time_factor = abs(dt_acoustic_substep * cfg.delt_max)
            self._apply_diffusive_heating(state.delp, state.delz, self.cappa, self._heat_source, state.pt, delt_time_factor)
