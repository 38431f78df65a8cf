"""FiniteVolumeTransport (reference: fv3core/pace/fv3core/stencils/fvtp2d.py:122-346)."""
import ctypes as C

from ._common import Operator, check_layout, dptr
from .delnflux import DelnFlux


class FiniteVolumeTransport(Operator):
    def __init__(self, stencil_factory, quantity_factory, grid_data, damping_coefficients, grid_type: int, hord,
                 nord=None, damp_c=None):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        assert grid_type < 3
        if hord not in (5, 6, 8):
            raise NotImplementedError(f"hord={hord}: implemented on device are the unlimited PPM (5, 6) and the monotone one (8)")
        self._hord = int(hord)
        self._nlev = self.grid_indexing.domain[2]
        self._do_delnflux = (nord is not None) and (damp_c is not None)
        if self._do_delnflux:
            self.delnflux = DelnFlux(stencil_factory, quantity_factory, damping_coefficients, grid_data.rarea, nord, damp_c,
                                     grid_data=grid_data)

    def __call__(self, q, crx, cry, x_area_flux, y_area_flux, q_x_flux, q_y_flux, x_mass_flux=None, y_mass_flux=None,
                 mass=None):
        check_layout(self._geom, q, crx, cry, x_area_flux, y_area_flux, q_x_flux, q_y_flux, x_mass_flux, y_mass_flux, mass)
        self.call("pace_fvtp2d", C.byref(self._met), dptr(q), dptr(crx), dptr(cry), dptr(x_area_flux), dptr(y_area_flux),
                  dptr(q_x_flux), dptr(q_y_flux), dptr(x_mass_flux), dptr(y_mass_flux), self._hord, self._nlev, self.stream())
        if self._do_delnflux:
            self.delnflux(q, q_x_flux, q_y_flux, mass=mass)
