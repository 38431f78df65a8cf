"""fv3core/pace/fv3core/stencils/basic_operations.py: the definition functions the acoustic path builds stencils from.  Bodies:
the device kernels registered under these identities (pace_amd/dsl/device_stencils.py)."""


def copy_defn(q_in, q_out):
    """basic_operations.py:7-15: q_out = q_in."""
