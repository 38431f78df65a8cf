"""PK3Halo (reference: fv3core/pace/fv3core/stencils/pk3_halo.py:36-69)."""
from ._common import Operator, check_layout, dptr


class PK3Halo(Operator):
    def __init__(self, stencil_factory, quantity_factory):
        super().__init__(stencil_factory, quantity_factory, None)

    def __call__(self, pk3, delp, ptop: float, akap: float):
        check_layout(self._geom, pk3, delp)
        self.call("pace_pk3_halo", dptr(pk3), dptr(delp), float(ptop), float(akap), self.stream())
