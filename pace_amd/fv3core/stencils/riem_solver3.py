"""NonhydrostaticVerticalSolver (reference: fv3core/pace/fv3core/stencils/riem_solver3.py:148-321)."""
import ctypes as C

import torch

from .._config import RiemannConfig
from ._common import Operator, check_layout, dptr


class NonhydrostaticVerticalSolver(Operator):
    def __init__(self, stencil_factory, quantity_factory, config: RiemannConfig):
        super().__init__(stencil_factory, quantity_factory, None)
        if config.a_imp <= 0.999:
            raise NotImplementedError("a_imp <= 0.999 is not implemented")
        if config.use_logp or config.beta != 0.0:
            raise NotImplementedError("use_logp / beta != 0 are not implemented")
        self._p_fac = config.p_fac
        nbytes = self.lib.cdll.pace_riem_solver3_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, last_call: bool, dt: float, cappa, ptop: float, zs, ws, delz, q_con, delp, pt, zh, p, ppe, pk3, pk,
                 log_p_interface, w):
        check_layout(self._geom, cappa, delz, q_con, delp, pt, zh, p, ppe, pk3, pk, log_p_interface, w)
        self.call("pace_riem_solver3", self._workspace.data_ptr(), int(bool(last_call)), float(dt), dptr(cappa), float(ptop),
                  dptr(zs), dptr(ws), dptr(delz), dptr(q_con), dptr(delp), dptr(pt), dptr(zh), dptr(p), dptr(ppe), dptr(pk3),
                  dptr(pk), dptr(log_p_interface), dptr(w), float(self._p_fac), self.stream())
