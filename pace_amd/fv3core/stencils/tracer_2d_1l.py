"""TracerAdvection -- Fortran tracer_2d_1l (reference: fv3core/pace/fv3core/stencils/tracer_2d_1l.py:171-392)."""
import ctypes as C
import math
from typing import Dict

from ...util.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM
from ...util.halo import WrappedHaloUpdater
from ._common import Operator, check_layout, dptr
from .fvtp2d import FiniteVolumeTransport


class TracerAdvection(Operator):
    """Performs horizontal advection on tracers: sub-cycled (n_split = 3, the reference hard-codes cmax = 2.0) monotone-PPM
    transport of every tracer with the mass fluxes and Courant numbers accumulated over the acoustic substeps."""

    def __init__(self, stencil_factory, quantity_factory, transport: FiniteVolumeTransport, grid_data, comm,
                 tracers: Dict[str, object]):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self._tracer_count = len(tracers)
        self.grid_data = grid_data
        self._x_area_flux = quantity_factory.zeros([X_INTERFACE_DIM, Y_DIM, Z_DIM], units="unknown")
        self._y_area_flux = quantity_factory.zeros([X_DIM, Y_INTERFACE_DIM, Z_DIM], units="unknown")
        self._x_flux = quantity_factory.zeros([X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM], units="unknown")
        self._y_flux = quantity_factory.zeros([X_INTERFACE_DIM, Y_INTERFACE_DIM, Z_DIM], units="unknown")
        self._tmp_dp = quantity_factory.zeros([X_DIM, Y_DIM, Z_DIM], units="Pa")
        self.finite_volume_transport = transport
        spec = quantity_factory.get_quantity_halo_spec([X_DIM, Y_DIM, Z_DIM], n_halo=3)
        self._tracers_halo_updater = WrappedHaloUpdater(comm.get_scalar_halo_updater([spec] * self._tracer_count), tracers,
                                                        [t for t in tracers.keys()])

    def __call__(self, tracers, dp1, x_mass_flux, y_mass_flux, x_courant, y_courant):
        check_layout(self._geom, dp1, x_mass_flux, y_mass_flux, x_courant, y_courant, *tracers.values())
        met, st = C.byref(self._met), self.stream
        self.call("pace_tracer_flux_compute", met, dptr(x_courant), dptr(y_courant), dptr(self._x_area_flux),
                  dptr(self._y_area_flux), st())
        cmax_max_all_ranks = 2.0  # tracer_2d_1l.py:336-339
        n_split = math.floor(1.0 + cmax_max_all_ranks)
        if n_split > 1.0:
            self.call("pace_tracer_divide_fluxes", dptr(x_courant), dptr(self._x_area_flux), dptr(x_mass_flux), dptr(y_courant),
                      dptr(self._y_area_flux), dptr(y_mass_flux), int(n_split), st())
        self._tracers_halo_updater.update()
        dp2 = self._tmp_dp
        for it in range(n_split):
            last_call = it == n_split - 1
            self.call("pace_apply_mass_flux", met, dptr(dp1), dptr(x_mass_flux), dptr(y_mass_flux), dptr(dp2), st())
            for q in tracers.values():
                self.finite_volume_transport(q, x_courant, y_courant, self._x_area_flux, self._y_area_flux, self._x_flux,
                                             self._y_flux, x_mass_flux=x_mass_flux, y_mass_flux=y_mass_flux)
                self.call("pace_apply_tracer_flux", met, dptr(q), dptr(dp1), dptr(self._x_flux), dptr(self._y_flux), dptr(dp2), st())
            if not last_call:
                self._tracers_halo_updater.update()
                self.call("pace_swap_dp", dptr(dp1), dptr(dp2), st())
