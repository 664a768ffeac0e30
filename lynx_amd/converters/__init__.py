"""
Import paths of the reference's file-format helpers (`lynx.converters`, `lynx.latticejson`).
Only the converters that need nothing but NumPy exist: ASTRA distributions here, LatticeJSON
in `lynx_amd.latticejson`; Ocelot / Bmad / NX tables need third-party packages and lattice
parsers that are outside the tracking path.
"""

from . import astra  # noqa: F401
