"""
lynx_amd -- MI355X-native drop-in for the beam-tracking hot path of jank324/lynx:
`Segment.track()` on `ParticleBeam` / `ParameterBeam` (re-exports as lynx/__init__.py:1-19,
restricted to the elements on the path).  Compute runs in hand-written HIP kernels
(`lynx_amd/csrc`) behind the C ABI in `include/lynx_hip.h`; there is no CPU fallback.
"""

from . import config  # noqa: F401
from .accelerator import (  # noqa: F401
    BPM,
    Aperture,
    Cavity,
    CustomTransferMap,
    Dipole,
    Drift,
    Element,
    HorizontalCorrector,
    Marker,
    Quadrupole,
    RBend,
    Screen,
    Segment,
    Solenoid,
    Undulator,
    VerticalCorrector,
)
from .particles import Beam, ParameterBeam, ParticleBeam  # noqa: F401

__version__ = "0.1.0"
