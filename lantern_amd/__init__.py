"""lantern_amd -- MI355X-native implementation of LANTERN's relaxed speculative-decoding
verify/accept loop (hot path only).  HIP kernels behind the C-ABI of
include/lantern_hip.h, bound with ctypes; no CPU fallback."""
from . import _lib  # noqa: F401
from ._lib import LanternError, build  # noqa: F401

__all__ = ["LanternError", "build", "ops"]
