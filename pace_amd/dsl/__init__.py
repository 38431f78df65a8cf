from .stencil import (  # noqa: F401
    CompilationConfig,
    FrozenStencil,
    GridIndexing,
    StencilConfig,
    StencilFactory,
    get_stencils_with_varied_bounds,
    register_stencil,
)


from . import device_stencils  # noqa: F401,E402  (fills the registry)


def orchestrate(*args, **kwargs):
    """No-op: the reference's DaCe whole-program orchestration hook (dsl/pace/dsl/dace/orchestration.py:439)."""
    return None


def dace_inhibitor(func):
    return func
