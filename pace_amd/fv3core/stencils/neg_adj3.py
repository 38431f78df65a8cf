"""AdjustNegativeTracerMixingRatio -- Fortran neg_adj3 (reference: fv3core/pace/fv3core/stencils/neg_adj3.py:296-420)."""
from ._common import Operator, check_layout, dptr
from .fillz import pointer_table


class AdjustNegativeTracerMixingRatio(Operator):
    """Adjust tracer mixing ratios to fix negative values."""

    def __init__(self, stencil_factory, quantity_factory, check_negative: bool, hydrostatic: bool):
        super().__init__(stencil_factory, quantity_factory)
        if check_negative:
            raise NotImplementedError("Unimplemented namelist value check_negative=True")
        if hydrostatic:
            raise NotImplementedError("Unimplemented namelist hydrostatic=True")

    def __call__(self, qvapor, qliquid, qrain, qsnow, qice, qgraupel, qcld, pt, delp):
        """qvapor .. qcld, pt (inout); delp (in)."""
        check_layout(self._geom, qvapor, qliquid, qrain, qsnow, qice, qgraupel, qcld, pt, delp)
        self.call("pace_neg_adj3", pointer_table([qvapor, qliquid, qrain, qsnow, qice, qgraupel]), dptr(qcld), dptr(pt),
                  dptr(delp), self.stream())
