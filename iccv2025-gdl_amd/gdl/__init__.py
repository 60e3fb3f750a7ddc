"""gdl -- MI355X-native DGL audio-visual training step (host side over libgdl_hip.so).

Nothing here computes on the CPU: every operator is a hand-written gfx950 kernel behind the
C ABI of include/gdl_hip.h, and importing a compute entry point without the built library
raises.
"""
from . import _lib  # noqa: F401
from ._lib import GdlError  # noqa: F401
