"""UpdateGeopotentialHeightOnCGrid (reference: fv3core/pace/fv3core/stencils/updatedzc.py:120-207)."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr, host_column


class UpdateGeopotentialHeightOnCGrid(Operator):
    def __init__(self, stencil_factory, quantity_factory, area, dp_ref, grid_data=None):
        """``area`` is accepted for signature parity; the kernel reads it from the metrics table."""
        super().__init__(stencil_factory, quantity_factory, grid_data if grid_data is not None else _grid_of(area))
        nz = self.grid_indexing.domain[2]
        self._dp_ref = torch.as_tensor(host_column(dp_ref, nz), dtype=quantity_factory.real, device=quantity_factory.device)
        nbytes = self.lib.cdll.pace_updatedzc_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, zs, ut, vt, gz, ws, dt: float):
        check_layout(self._geom, ut, vt, gz)
        self.call("pace_updatedzc", C.byref(self._met), self._workspace.data_ptr(), self._dp_ref.data_ptr(), dptr(zs), dptr(ut),
                  dptr(vt), dptr(gz), dptr(ws), float(dt), self.stream())


def _grid_of(area):
    gd = getattr(area, "_grid_data", None)
    if gd is None:
        raise ValueError("pass grid_data= (or an area Quantity obtained from GridData)")
    return gd
