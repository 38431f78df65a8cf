"""YPiecewiseParabolic (reference: fv3core/pace/fv3core/stencils/yppm.py:290-355)."""
from .xppm import _PiecewiseParabolic


def compute_y_flux(q, courant, dya, yflux):
    """yppm.py:269-287 (see xppm.compute_x_flux)."""


class YPiecewiseParabolic(_PiecewiseParabolic):
    """Fortran name is yppm."""

    _axis = 1

    def __init__(self, stencil_factory, dya, grid_type: int, jord, origin, domain):
        super().__init__(stencil_factory, dya, grid_type, jord, origin, domain)
