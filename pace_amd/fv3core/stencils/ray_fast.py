"""RayleighDamping (reference: fv3core/pace/fv3core/stencils/ray_fast.py:144-206)."""
import ctypes as C

from ._common import Operator, check_layout, dptr, host_column


class RayleighDamping(Operator):
    def __init__(self, stencil_factory, rf_cutoff, tau, hydrostatic, quantity_factory=None):
        if quantity_factory is None:
            raise ValueError("pace_amd needs quantity_factory= to know the field layout")
        super().__init__(stencil_factory, quantity_factory, None)
        self._rf_cutoff, self._tau, self._hydrostatic = float(rf_cutoff), float(tau), bool(hydrostatic)
        self._host_columns = {}  # id(K-field) -> host copy: a device -> host copy synchronises the stream, so it is made once

    def _host(self, field, nz):
        """dp_ref and pfull are constants of the grid (grid/helper.py:306-530): converted on first use, then reused, so
        no call after the first touches the device outside the kernel launch."""
        key = id(field)
        hit = self._host_columns.get(key)
        if hit is None or hit[0] is not field:
            hit = (field, host_column(field, nz))
            self._host_columns[key] = hit
        return hit[1]

    def __call__(self, u, v, w, dp, pfull, dt: float, ptop: float):
        check_layout(self._geom, u, v, w)
        nz = self.grid_indexing.domain[2]
        dp_h, pf_h = self._host(dp, nz), self._host(pfull, nz)
        dptr_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        self.call("pace_ray_fast", dptr(u), dptr(v), dptr(w), dptr_(dp_h), dptr_(pf_h), float(dt), float(ptop), self._rf_cutoff,
                  self._tau, int(self._hydrostatic), self.stream())
