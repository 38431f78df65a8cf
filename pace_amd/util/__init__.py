from .constants import *  # noqa: F401,F403
from .quantity import Quantity, QuantityFactory, SubtileGridSizer  # noqa: F401
