"""FiniteVolumeFluxPrep (reference: fv3core/pace/fv3core/stencils/fxadv.py:489-661)."""
import ctypes as C

from ._common import Operator, check_layout, dptr


class FiniteVolumeFluxPrep(Operator):
    """Same constructor and call signature as the reference class; runs pace_fxadv."""

    def __init__(self, stencil_factory, grid_data, quantity_factory=None):
        if quantity_factory is None:
            quantity_factory = grid_data._qf
        super().__init__(stencil_factory, quantity_factory, grid_data)

    def __call__(self, uc, vc, crx, cry, x_area_flux, y_area_flux, uc_contra, vc_contra, dt):
        check_layout(self._geom, uc, vc, crx, cry, x_area_flux, y_area_flux, uc_contra, vc_contra)
        self.call("pace_fxadv", C.byref(self._met), dptr(uc), dptr(vc), dptr(crx), dptr(cry), dptr(x_area_flux),
                  dptr(y_area_flux), dptr(uc_contra), dptr(vc_contra), float(dt), self.stream())
