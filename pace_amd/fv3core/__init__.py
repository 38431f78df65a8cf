from ._config import (  # noqa: F401
    AcousticDynamicsConfig,
    DGridShallowWaterLagrangianDynamicsConfig,
    DynamicalCoreConfig,
    RemappingConfig,
    RiemannConfig,
)
