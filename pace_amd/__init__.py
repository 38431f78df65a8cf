"""pace_amd -- MI355X-native FV3 acoustic substep behind ai2cm/pace's stencil/Quantity API.

Only the hot path named in BASELINE.json is implemented.  All numerics run in
libpace_hip.so (hand-written HIP for gfx950); importing this package does not require a GPU, calling
any operator does.
"""
__version__ = "0.1.0"
