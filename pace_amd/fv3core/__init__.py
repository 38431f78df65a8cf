from ._config import (  # noqa: F401
    AcousticDynamicsConfig,
    DGridShallowWaterLagrangianDynamicsConfig,
    DynamicalCoreConfig,
    RemappingConfig,
    RiemannConfig,
)
from .initialization.dycore_state import DycoreState  # noqa: F401,E402
from .stencils.fv_dynamics import DynamicalCore  # noqa: F401,E402
