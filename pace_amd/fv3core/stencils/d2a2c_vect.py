"""DGrid2AGrid2CGridVectors (reference: fv3core/pace/fv3core/stencils/d2a2c_vect.py:380-655)."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr


class DGrid2AGrid2CGridVectors(Operator):
    def __init__(self, stencil_factory, quantity_factory, grid_data, nested: bool, grid_type: int, dord4: bool):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        if grid_type >= 3:
            raise NotImplementedError("unimplemented grid_type >= 3")
        if nested:
            raise NotImplementedError("nested grids are not implemented")
        if not dord4:
            raise NotImplementedError("dord4 = False is not implemented")
        nbytes = self.lib.cdll.pace_c_sw_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, uc, vc, u, v, ua, va, utc, vtc):
        check_layout(self._geom, uc, vc, u, v, ua, va, utc, vtc)
        self.call("pace_d2a2c_vect", C.byref(self._met), self._workspace.data_ptr(), dptr(uc), dptr(vc), dptr(u), dptr(v),
                  dptr(ua), dptr(va), dptr(utc), dptr(vtc), self.stream())
