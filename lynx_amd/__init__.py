"""
lynx_amd -- MI355X-native drop-in for the beam-tracking hot path of jank324/lynx:
`Segment.track()` on `ParticleBeam` / `ParameterBeam` (re-exports as lynx/__init__.py:1-19,
restricted to the elements on the path).  Compute runs in hand-written HIP kernels
(`lynx_amd/csrc`) behind the C ABI in `include/lynx_hip.h`; there is no CPU fallback.
"""

from . import config
from .accelerator import *  # noqa: F401,F403  (the element classes and Segment)
from .accelerator import __all__ as _elements
from .particles import Beam, ParameterBeam, ParticleBeam

__all__ = ["config", "Beam", "ParameterBeam", "ParticleBeam", *_elements]
__version__ = "0.1.0"
