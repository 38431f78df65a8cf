"""DelnFlux / DelnFluxNoSG / calc_damp (reference: fv3core/pace/fv3core/stencils/delnflux.py:21-38,945-1261)."""
import ctypes as C

import numpy as np
import torch

from ...util import constants as c
from ._common import Operator, check_layout, dptr, expand_externals, host_column


def calc_damp(damp_c, da_min: float, nord):
    """delnflux.py:21-38 -> host numpy array (damp_c * da_min) ** (nord + 1)."""
    d = host_column(damp_c, len(damp_c) if not hasattr(damp_c, "dims") else damp_c.shape[0])
    n = host_column(nord, len(d))
    return (d * da_min) ** (n + 1)


class DelnFluxNoSG(Operator):
    def __init__(self, stencil_factory, damping_coefficients, rarea, nord, nk=None, grid_data=None, quantity_factory=None):
        grid_data = grid_data or damping_coefficients._grid_data
        quantity_factory = quantity_factory or grid_data._qf
        super().__init__(stencil_factory, quantity_factory, grid_data)
        nz = self.grid_indexing.domain[2]
        self._nk = nz if nk is None else nk
        nord_h = host_column(nord, nz)
        if not all(n in (0, 2, 3) for n in nord_h):
            raise NotImplementedError("nord must have values 0, 2, or 3")
        self._nmax = int(nord_h.max())
        if self._nmax > 3:
            raise ValueError("nord must be less than 3")
        if self._nk <= 3:
            raise NotImplementedError("nk must be more than 3 for DelnFluxNoSG")
        per_level = expand_externals(nord_h, self._nk)
        self._nord_dev = torch.as_tensor(per_level, dtype=quantity_factory.real, device=quantity_factory.device)
        self._damp_dev = torch.zeros(self._nk, dtype=quantity_factory.real, device=quantity_factory.device)

    def _damp(self, damp_c):
        if torch.is_tensor(damp_c) and damp_c.device == self._damp_dev.device:
            return damp_c
        a = host_column(damp_c, min(self._nk, len(damp_c) if not hasattr(damp_c, "dims") else damp_c.shape[0]))
        per = np.full(self._nk, a[-1])
        per[: len(a)] = a
        self._damp_dev.copy_(torch.as_tensor(per, dtype=self._damp_dev.dtype))
        return self._damp_dev

    def __call__(self, q, fx2, fy2, damp_c, d2, mass=None):
        """d2 (the damped copy of q) is not produced: nothing on the acoustic path reads it
        (d_sw.py:1032-1041 'output value for tmp_wk here is never used')."""
        check_layout(self._geom, q, fx2, fy2)
        damp = self._damp(damp_c)
        self.call("pace_delnflux_nosg", C.byref(self._met), dptr(q), dptr(fx2), dptr(fy2), damp.data_ptr(),
                  self._nord_dev.data_ptr(), self._nmax, 0 if mass is None else 1, self._nk, self.stream())


class DelnFlux(Operator):
    def __init__(self, stencil_factory, quantity_factory, damping_coefficients, rarea, nord_col, damp_c, grid_data=None):
        grid_data = grid_data or damping_coefficients._grid_data
        super().__init__(stencil_factory, quantity_factory, grid_data)
        nz = self.grid_indexing.domain[2]
        self._nk = nz
        damp_h = host_column(damp_c, nz)
        self._no_compute = bool((damp_h <= 1e-4).all())
        if (not self._no_compute) and (damp_h[:-1] <= 1e-4).any():
            raise NotImplementedError("damp_c currently must be always greater than 10^-4 for delnflux")
        nord_h = expand_externals(host_column(nord_col, nz), nz)
        self._nmax = int(nord_h.max())
        fac = (damp_h * damping_coefficients.da_min) ** (nord_h + 1)
        self._nord_dev = torch.as_tensor(nord_h, dtype=quantity_factory.real, device=quantity_factory.device)
        self._damp_dev = torch.as_tensor(fac, dtype=quantity_factory.real, device=quantity_factory.device)

    def __call__(self, q, fx, fy, d2=None, mass=None):
        if self._no_compute:
            return fx, fy
        check_layout(self._geom, q, fx, fy, mass)
        self.call("pace_delnflux", C.byref(self._met), dptr(q), dptr(fx), dptr(fy), dptr(mass), self._damp_dev.data_ptr(),
                  self._nord_dev.data_ptr(), self._nmax, self._nk, self.stream())
        return fx, fy
