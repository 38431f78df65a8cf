"""MapSingle -- Fortran map_single (reference: fv3core/pace/fv3core/stencils/map_single.py:96-200; its RemapProfile,
remap_profile.py:566-681, is part of the same three-kernel launch sequence here: pace_amd/csrc/k_remap.hip)."""
import ctypes as C
from typing import Optional, Sequence

import torch

from ...util.constants import X_INTERFACE_DIM, Y_INTERFACE_DIM
from ._common import Operator, check_layout, dptr


class MapSingle(Operator):
    """Remaps one field from the deformed Lagrangian layers to the Eulerian reference layers.

    ``kord``: 9 or 10 (the reference supports up to 10; < 9 is not implemented here).  ``mode`` is the reference's
    ``iv``: -2 vertical velocity, -1 winds, 0 positive-definite tracers, 1 everything else.  ``dims``: the dimensions of
    the field (staggered fields own one more row / column)."""

    def __init__(self, stencil_factory, quantity_factory, kord: int, mode: int, dims: Sequence[str]):
        super().__init__(stencil_factory, quantity_factory)
        if abs(kord) > 10:
            raise AssertionError(f"kord {kord} not implemented.")  # remap_profile.py:596
        if abs(kord) < 9:
            raise NotImplementedError(f"kord {kord}: pace_amd implements the kord 9 and 10 profiles")
        self._kord, self._mode = int(kord), int(mode)
        self._xstag = int(X_INTERFACE_DIM in dims)
        self._ystag = int(Y_INTERFACE_DIM in dims)
        nbytes = self.lib.cdll.pace_map_single_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)

    def __call__(self, q1, pe1, pe2, qs: Optional[object] = None, qmin: float = 0.0):
        """q1 (inout): the field, remapped in place; pe1 (in): Lagrangian interface pressures (or log-pressures);
        pe2 (in): Eulerian ones; qs (in): bottom boundary value (mode -2); qmin (in): lower bound used by the kord 9
        extremum filter."""
        check_layout(self._geom, q1, pe1, pe2)
        if self._mode == -2 and qs is None:
            raise ValueError("mode -2 (vertical velocity) needs the bottom boundary value qs")
        self.call("pace_map_single", self._workspace.data_ptr(), dptr(q1), dptr(pe1), dptr(pe2), dptr(qs), float(qmin),
                  self._kord, self._mode, self._xstag, self._ystag, self.stream())
        return q1
