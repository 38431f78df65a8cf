from .constants import *  # noqa: F401,F403
from .quantity import Quantity, QuantityFactory, SubtileGridSizer  # noqa: F401
from .comm import LoopbackComm, NullComm, ThreadComm, TorchDistComm, run_tiles  # noqa: F401,E402
from .halo import CubedSphereCommunicator, HaloUpdater, QuantityHaloSpec, WrappedHaloUpdater  # noqa: F401,E402
from .partitioner import CubedSpherePartitioner, RingPartitioner, TilePartitioner  # noqa: F401,E402
from ._timing import KernelTimes, NullTimer, Timer  # noqa: F401,E402
