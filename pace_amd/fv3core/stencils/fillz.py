"""FillNegativeTracerValues -- Fortran fillz (reference: fv3core/pace/fv3core/stencils/fillz.py:120-163)."""
import ctypes as C
from typing import Dict

from ._common import Operator, check_layout, dptr

# dsl/pace/dsl/gt4py_utils.py:24-34
tracer_variables = ["qvapor", "qliquid", "qrain", "qice", "qsnow", "qgraupel", "qo3mr", "qsgs_tke", "qcld"]


def pointer_table(quantities):
    arr = (C.c_void_p * len(quantities))()
    for n, q in enumerate(quantities):
        arr[n] = dptr(q)
    return arr


class FillNegativeTracerValues(Operator):
    """Fix tracer values to prevent negative masses (all tracers in one launch)."""

    def __init__(self, stencil_factory, quantity_factory, nq: int, tracers: Dict[str, object]):
        super().__init__(stencil_factory, quantity_factory)
        self._nq = int(nq)
        self._names = [name for name in tracer_variables[0:self._nq]]
        for name in self._names:
            tracers[name]  # KeyError as in the reference's constructor (fillz.py:147-149)

    def __call__(self, dp2, tracers: Dict[str, object]):
        """dp2 (in): pressure thickness of atmospheric layer; tracers (inout): tracers to fix negative masses in."""
        qs = [tracers[name] for name in self._names]
        check_layout(self._geom, dp2, *qs)
        self.call("pace_fillz", pointer_table(qs), len(qs), dptr(dp2), self.stream())
