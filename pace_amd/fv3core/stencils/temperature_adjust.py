"""fv3core/pace/fv3core/stencils/temperature_adjust.py."""


def apply_diffusive_heating(delp, delz, cappa, heat_source, pt, delt_time_factor):
    """temperature_adjust.py:8-43: pt += sign(min(|dT|, limit)) / pkz from the dissipative heat source."""
