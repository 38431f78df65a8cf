"""MapNTracer -- Fortran mapn_tracer (reference: fv3core/pace/fv3core/stencils/mapn_tracer.py:13-82)."""
import ctypes as C
from typing import Dict

import torch

from ._common import Operator, check_layout, dptr
from .fillz import FillNegativeTracerValues, pointer_table, tracer_variables


class MapNTracer(Operator):
    """Remaps the tracer species onto the Eulerian grid and optionally fills negative values.  The tracers share the
    source and target coordinates, so they go through the remapping kernels together (one launch sequence per distinct
    kord: the reference pins tracer 5 to kord 9, mapn_tracer.py:36-37)."""

    def __init__(self, stencil_factory, quantity_factory, kord: int, nq: int, fill: bool, tracers: Dict[str, object]):
        super().__init__(stencil_factory, quantity_factory)
        self._nq = int(nq)
        kord_tracer = [kord] * self._nq
        if self._nq > 5:
            kord_tracer[5] = 9
        else:
            raise IndexError("list assignment index out of range")  # the reference's kord_tracer[5] = 9 with nq <= 5
        for k in kord_tracer:
            if abs(k) > 10:
                raise AssertionError(f"kord {k} not implemented.")
            if abs(k) < 9:
                raise NotImplementedError(f"kord {k}: pace_amd implements the kord 9 and 10 profiles")
        self._groups = {}
        for n, k in enumerate(kord_tracer):
            self._groups.setdefault(abs(k), []).append(n)
        nmax = max(len(v) for v in self._groups.values())
        nbytes = self.lib.cdll.pace_mapn_tracer_workspace_bytes(C.byref(self._geom), nmax)
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=quantity_factory.device)
        self._fill_negative_tracers = bool(fill)
        if fill:
            self._fillz = FillNegativeTracerValues(stencil_factory, quantity_factory, self._nq, tracers)

    def __call__(self, pe1, pe2, dp2, tracers: Dict[str, object]):
        """pe1 (in): Lagrangian pressure levels; pe2 (in): Eulerian pressure levels; dp2 (in): difference in pressure
        between Eulerian levels; tracers (inout): tracers to be remapped.  Assumes the minimum value is 0 for each tracer."""
        names = tracer_variables[0:self._nq]
        qs = [tracers[q] for q in names]
        check_layout(self._geom, pe1, pe2, dp2, *qs)
        for kord, members in self._groups.items():
            group = [qs[n] for n in members]
            self.call("pace_mapn_tracer", self._workspace.data_ptr(), pointer_table(group), len(group), dptr(pe1), dptr(pe2),
                      int(kord), self.stream())
        if self._fill_negative_tracers is True:
            self._fillz(dp2, tracers)
