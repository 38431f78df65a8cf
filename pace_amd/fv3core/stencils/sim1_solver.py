"""Sim1Solver (reference: fv3core/pace/fv3core/stencils/sim1_solver.py:144-219) as a class of its own.

Inside NonhydrostaticVerticalSolver / NonhydrostaticVerticalSolverCGrid the same arithmetic is part of the fused column
kernel (pace_riem_solver3 / pace_riem_solver_c); this class serves callers -- and tests -- that use the solver directly."""
import ctypes as C

import torch

from ._common import Operator, check_layout, dptr


def sim1_solver(w, delta_mass, gamma, dz, potential_temperature, pm, pe, pem, ws, cp3, dt, t1g, rdt, p_fac):
    """sim1_solver.py:20-141: the definition function of the stencil Sim1Solver builds.  Body: pace_sim1_solver."""


class Sim1Solver(Operator):
    """Fortran name is sim1_solver.  Namelist: p_fac -- safety factor for minimum nonhydrostatic pressures."""

    def __init__(self, stencil_factory, p_fac: float, n_halo: int):
        qf = getattr(stencil_factory, "quantity_factory", None)
        if qf is None:
            raise ValueError("Sim1Solver needs the field layout: build the StencilFactory with quantity_factory=...")
        super().__init__(stencil_factory, qf, None)
        self._pfac = float(p_fac)
        self._n_halo = int(n_halo)
        nbytes = self.lib.cdll.pace_sim1_solver_workspace_bytes(C.byref(self._geom))
        self._workspace = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=qf.device)

    def __call__(self, dt: float, gamma, cp3, pe, delta_mass, pm, pem, w, dz, potential_temperature, ws):
        """gamma, cp3, delta_mass, pm, pem, potential_temperature, ws (2-D) in; pe out; w, dz inout."""
        check_layout(self._geom, gamma, cp3, pe, delta_mass, pm, pem, w, dz, potential_temperature)
        self.call("pace_sim1_solver", self._workspace.data_ptr(), self._n_halo, float(dt), self._pfac, dptr(gamma), dptr(cp3),
                  dptr(pe), dptr(delta_mass), dptr(pm), dptr(pem), dptr(w), dptr(dz), dptr(potential_temperature), dptr(ws),
                  self.stream())
