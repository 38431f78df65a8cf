"""NonHydrostaticPressureGradient (reference: fv3core/pace/fv3core/stencils/nh_p_grad.py:115-255)."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr


class NonHydrostaticPressureGradient(Operator):
    def __init__(self, stencil_factory, quantity_factory, grid_data, grid_type):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        nbytes = self.lib.cdll.pace_nh_p_grad_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, u, v, pp, gz, pk3, delp, dt: float, ptop: float, akap: float):
        check_layout(self._geom, u, v, pp, gz, pk3, delp)
        self.call("pace_nh_p_grad", C.byref(self._met), self._workspace.data_ptr(), dptr(u), dptr(v), dptr(pp), dptr(gz),
                  dptr(pk3), dptr(delp), float(dt), float(ptop), float(akap), self.stream())
