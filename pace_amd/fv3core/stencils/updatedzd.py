"""UpdateHeightOnDGrid (reference: fv3core/pace/fv3core/stencils/updatedzd.py:129-356)."""
import ctypes as C

import numpy as np
import torch

from ... import _lib
from ._common import Operator, check_layout, dptr, expand_externals, host_column


def cubic_spline_interpolation_constants(dp0):
    """updatedzd.py:129-154 on the host; dp0 = dp_ref[:nz]."""
    dp0 = np.asarray(dp0, dtype=np.float64)
    nz = dp0.shape[0]
    gk, beta, gamma = np.zeros(nz), np.zeros(nz), np.zeros(nz)
    gk[0] = dp0[1] / dp0[0]
    beta[0] = gk[0] * (gk[0] + 0.5)
    gamma[0] = (1.0 + gk[0] * (gk[0] + 1.5)) / beta[0]
    gk[1:] = dp0[:-1] / dp0[1:]
    for i in range(1, nz):
        beta[i] = 2.0 + 2.0 * gk[i] - gamma[i - 1]
        gamma[i] = gk[i] / beta[i]
    return gk, beta, gamma


class UpdateHeightOnDGrid(Operator):
    """Fortran updatedzd: spline to interfaces (1 launch for the four fields), fvtp2d and delnflux on nz+1 levels
    (the d_sw kernels), one column kernel for the flux application + ws + monotonicity sweep."""

    def __init__(self, stencil_factory, quantity_factory, damping_coefficients, grid_data, grid_type: int, hord_tm: int,
                 column_namelist):
        super().__init__(stencil_factory, quantity_factory, grid_data)
        nz = self.grid_indexing.domain[2]
        damp_vt = host_column(column_namelist["damp_vt"], nz)
        if (damp_vt <= 1e-5).any():
            raise NotImplementedError("damp <= 1e-5 in column_namelist is untested")
        self._hord_tm = int(hord_tm)
        gk, beta, gamma = cubic_spline_interpolation_constants(host_column(grid_data.dp_ref, nz))
        nord = expand_externals(host_column(column_namelist["nord_v"], nz), nz + 1)
        damp = np.zeros(nz + 1)
        damp[:nz] = damp_vt  # the K-field has nz+1 entries, the last one is the allocator's zero
        self._k_dev = torch.as_tensor(np.concatenate([gk, beta, gamma, damp, nord]), dtype=quantity_factory.real, device=quantity_factory.device)
        base, sz = self._k_dev.data_ptr(), quantity_factory.itemsize
        k = _lib.UpdatedzdK()
        k.gk, k.beta, k.gamma = base, base + nz * sz, base + 2 * nz * sz
        k.damp, k.nord = base + 3 * nz * sz, base + (3 * nz + nz + 1) * sz
        # scalars the interpolation stencil derives from the K-fields (updatedzd.py:180-192)
        k.xt1_top = 2.0 * gk[0] * (gk[0] + 1.0)
        g = gk[nz - 1]
        k.a_bot = 1.0 + g * (g + 1.5)
        k.xt1_bot = 2.0 * g * (g + 1.0)
        k.xt2_bot = g * (g + 0.5) - k.a_bot * gamma[nz - 1]
        k.nmax = int(nord.max())
        self._k = k
        nbytes = self.lib.cdll.pace_updatedzd_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, surface_height, height, courant_number_x, courant_number_y, x_area_flux, y_area_flux, ws, dt: float):
        check_layout(self._geom, height, courant_number_x, courant_number_y, x_area_flux, y_area_flux)
        self.call("pace_updatedzd", C.byref(self._met), self._workspace.data_ptr(), C.byref(self._k), dptr(surface_height),
                  dptr(height), dptr(courant_number_x), dptr(courant_number_y), dptr(x_area_flux), dptr(y_area_flux), dptr(ws),
                  float(dt), self._hord_tm, self.stream())
