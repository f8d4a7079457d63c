"""frostnet_amd -- MI355X-native FrostNet QAT hot path (hand-written HIP for gfx950 behind the reference's
nn.Module / optimizer surface).  See DESIGN.md."""
from ._lib import LIB_PATH, SYMBOLS, load_library  # noqa: F401

__all__ = ["load_library", "LIB_PATH", "SYMBOLS"]
