"""afr-mi355x: MI355X-native training hot path of chenglou/ai-font-renderer (model.py fwd+bwd+AdamW).

Only light, pure-host modules are imported here; the HIP engine (engine.py, which loads
csrc/libafr.so through ctypes) is imported on first use and fails loudly when the library or a
GPU is missing -- there is no CPU fallback in the product path.
"""
from . import config, synth  # noqa: F401
from .config import SheetConfig, GlyphConfig, WORKLOADS, flat_layout  # noqa: F401

__version__ = "0.1.0"
