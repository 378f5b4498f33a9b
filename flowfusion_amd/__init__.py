"""flowfusion_amd: MI355X-native sampling / log-density path of Cosmo-Pop/flowfusion.

``flowfusion_amd.diffusion`` and ``flowfusion_amd.flow`` mirror the reference modules
``flowfusion.diffusion`` / ``flowfusion.flow``; the solves run in libflowfusion_amd.so
(hand-written HIP for gfx950, C ABI in include/flowfusion_amd.h).
"""
from . import diffusion, flow  # noqa: F401

__all__ = ["diffusion", "flow"]
__version__ = "0.4.0"
