"""CubedToLatLon -- Fortran c2l_ord2 (reference: stencils/pace/stencils/c2l_ord.py:115-196)."""
import ctypes as C

from ...util.constants import X_DIM, X_INTERFACE_DIM, Y_DIM, Y_INTERFACE_DIM, Z_DIM
from ...util.halo import WrappedHaloUpdater
from ._common import Operator, check_layout, dptr


class CubedToLatLon(Operator):
    """Interpolate D-grid to A-grid winds at latitude-longitude coordinates, 2nd or 4th order."""

    def __init__(self, state, stencil_factory, quantity_factory, grid_data, order: int, comm):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        self._a = [grid_data.a11, grid_data.a12, grid_data.a21, grid_data.a22]
        self._do_ord4 = order != 2
        n_halo = self.grid_indexing.n_halo
        spec_u = quantity_factory.get_quantity_halo_spec([X_DIM, Y_INTERFACE_DIM, Z_DIM], n_halo=n_halo)
        spec_v = quantity_factory.get_quantity_halo_spec([X_INTERFACE_DIM, Y_DIM, Z_DIM], n_halo=n_halo)
        self.u__v = WrappedHaloUpdater(comm.get_vector_halo_updater([spec_u], [spec_v]), state, ["u"], ["v"])

    def __call__(self, u, v, ua, va):
        """u, v: winds on the D-grid (in); ua, va: winds on the A-grid (out)."""
        check_layout(self._geom, u, v, ua, va)
        if self._do_ord4:
            self.u__v.update()
        self.call("pace_c2l_ord", C.byref(self._met), 4 if self._do_ord4 else 2, dptr(u), dptr(v), *[dptr(a) for a in self._a],
                  dptr(ua), dptr(va), self.stream())
