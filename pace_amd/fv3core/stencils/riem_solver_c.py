"""NonhydrostaticVerticalSolverCGrid (reference: fv3core/pace/fv3core/stencils/riem_solver_c.py:126-250)."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr


class NonhydrostaticVerticalSolverCGrid(Operator):
    def __init__(self, stencil_factory, quantity_factory, p_fac: float):
        super().__init__(stencil_factory, quantity_factory, None)
        self._p_fac = float(p_fac)
        nbytes = self.lib.cdll.pace_riem_solver_c_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, dt2: float, cappa, ptop: float, hs, ws, ptc, q_con, delpc, gz, pef, w3):
        check_layout(self._geom, cappa, ptc, q_con, delpc, gz, pef, w3)
        self.call("pace_riem_solver_c", self._workspace.data_ptr(), float(dt2), dptr(cappa), float(ptop), dptr(hs), dptr(ws),
                  dptr(ptc), dptr(q_con), dptr(delpc), dptr(gz), dptr(pef), dptr(w3), self._p_fac, self.stream())
