"""`lynx.converters.astra` by its reference name: `from_astrabeam(path)` (converters/astra.py:8-62)."""

from ..io.astra import read_astra


def from_astrabeam(path: str):
    """(particles (N, 6) in (x, xp, y, yp, s, delta), reference energy in eV, macro-particle charges in C)."""
    return read_astra(path)
