"""DivergenceDamping (reference: fv3core/pace/fv3core/stencils/divergence_damping.py:254-632)."""
import ctypes as C

import numpy as np
import torch

from ._common import Operator, check_layout, dptr, host_column


class DivergenceDamping(Operator):
    """A large section in Fortran's d_sw that applies divergence damping.  Inside pace_amd's d_sw the same launches run as part
    of the fused sequence; this class is the operator on its own (five launches, reference: ~45)."""

    def __init__(self, stencil_factory, quantity_factory, grid_data, damping_coefficients, nested: bool, stretched_grid: bool,
                 dddmp, d4_bg, nord: int, grid_type, nord_col, d2_bg):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        assert not nested, "nested not implemented"
        assert grid_type < 3, "Not implemented, grid_type>=3, specifically smag_corner"
        if stretched_grid:
            raise NotImplementedError("stretched_grid")
        nz = self.grid_indexing.domain[2]
        self._dddmp, self._d4_bg, self._nord = float(dddmp), float(d4_bg), int(nord)
        self._nord_col = np.ascontiguousarray(host_column(nord_col, nz))
        self._d2_bg = torch.as_tensor(host_column(d2_bg, nz), dtype=quantity_factory.real, device=quantity_factory.device)
        nbytes = self.lib.cdll.pace_divergence_damping_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, u, v, va, damped_rel_vort_bgrid, ua, divg_d, vc, uc, delpc, ke, rel_vort_agrid, dt: float):
        check_layout(self._geom, u, v, va, damped_rel_vort_bgrid, ua, divg_d, vc, uc, delpc, ke, rel_vort_agrid)
        self.call("pace_divergence_damping", C.byref(self._met), self._workspace.data_ptr(), dptr(u), dptr(v), dptr(va),
                  dptr(damped_rel_vort_bgrid), dptr(ua), dptr(divg_d), dptr(vc), dptr(uc), dptr(delpc), dptr(ke),
                  dptr(rel_vort_agrid), float(dt), self._nord_col.ctypes.data_as(C.POINTER(C.c_double)), self._d2_bg.data_ptr(),
                  self._dddmp, self._d4_bg, self._nord, self.stream())
