"""AGrid2BGridFourthOrder (reference: fv3core/pace/fv3core/stencils/a2b_ord4.py:509-761)."""
import ctypes as C

from ...util.constants import Z_DIM, Z_INTERFACE_DIM
from ._common import Operator, check_layout, dptr


class AGrid2BGridFourthOrder(Operator):
    def __init__(self, stencil_factory, quantity_factory, grid_data, grid_type, z_dim=Z_DIM, replace: bool = False):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        assert grid_type < 3
        self.replace = replace
        k0 = self.grid_indexing.origin[2]
        nk = self.grid_indexing.domain[2] + (1 if z_dim == Z_INTERFACE_DIM else 0)
        self._k0, self._k1 = k0, k0 + nk

    def __call__(self, qin, qout):
        check_layout(self._geom, qin, qout)
        self.call("pace_a2b_ord4", C.byref(self._met), dptr(qin), dptr(qout), self._k0, self._k1, int(self.replace), self.stream())
