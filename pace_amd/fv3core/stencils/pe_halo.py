"""fv3core/pace/fv3core/stencils/pe_halo.py."""


def edge_pe(pe, delp, ptop):
    """pe_halo.py:6-34: interface pressure in the one-cell ring around the compute domain (FORWARD scan of delp)."""
