"""CGridShallowWaterDynamics (reference: fv3core/pace/fv3core/stencils/c_sw.py:483-766)."""
import ctypes as C

import torch

from ...util.constants import X_DIM, Y_DIM, Z_DIM
from ._common import Operator, check_layout, dptr


class CGridShallowWaterDynamics(Operator):
    """Fortran c_sw: 4 HIP passes (csrc/k_csw.hip) instead of 24 stencil launches.  ``delpc`` and ``ptc`` are
    public attributes exactly as in the reference (c_sw.py:497-502; read by dyn_core.py:795-800)."""

    def __init__(self, stencil_factory, quantity_factory, grid_data, nested: bool, grid_type: int, nord: int):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        if nested:
            raise NotImplementedError("nested grids are not implemented")
        if grid_type >= 3:
            raise NotImplementedError("grid_type >= 3 is not implemented")
        self._nord = int(nord)
        self.delpc = quantity_factory.zeros([X_DIM, Y_DIM, Z_DIM], units="Pa")
        self.ptc = quantity_factory.zeros([X_DIM, Y_DIM, Z_DIM], units="K")
        nbytes = self.lib.cdll.pace_c_sw_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def _args(self, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2):
        check_layout(self._geom, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga)
        return (C.byref(self._met), self._workspace.data_ptr(), dptr(self.delpc), dptr(self.ptc), dptr(delp), dptr(pt), dptr(u),
                dptr(v), dptr(w), dptr(uc), dptr(vc), dptr(ua), dptr(va), dptr(ut), dptr(vt), dptr(divgd), dptr(omga), float(dt2),
                self._nord, self.stream())

    _started = False

    def start_interior(self, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2: float):
        """An extension for overlapping the u / v halo exchange in front of c_sw with compute (dyn_core.py:744-745): the points
        of the first pass (the D-grid winds interpolated to the A grid) that read no halo value of u / v are computed now; the
        following ``__call__`` (same arguments, after ``u__v.wait()``) does the rest.  Same results bit for bit."""
        self.lib.call("pace_c_sw_part", 1, C.byref(self._geom), *self._args(delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2))
        self._started = True

    def __call__(self, delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2: float):
        args = self._args(delp, pt, u, v, w, uc, vc, ua, va, ut, vt, divgd, omga, dt2)
        if self._started:
            self._started = False
            self.lib.call("pace_c_sw_part", 2, C.byref(self._geom), *args)
        else:
            self.call("pace_c_sw", *args)
        return self.delpc, self.ptc  # (c_sw.py:766: the reference returns its two internal fields)
