"""HyperdiffusionDamping (reference: fv3core/pace/fv3core/stencils/del2cubed.py:78-194)."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr


class HyperdiffusionDamping(Operator):
    def __init__(self, stencil_factory, quantity_factory, damping_coefficients, rarea, nmax: int):
        super().__init__(stencil_factory, quantity_factory, damping_coefficients._grid_data)
        self._nmax = int(nmax)
        nbytes = self.lib.cdll.pace_del2cubed_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, qdel, cd: float):
        check_layout(self._geom, qdel)
        self.call("pace_del2cubed", C.byref(self._met), self._workspace.data_ptr(), dptr(qdel), float(cd), self._nmax,
                  self.stream())
