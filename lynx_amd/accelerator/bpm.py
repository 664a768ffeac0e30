"""lynx/accelerator/bpm.py:24-80 and lynx/accelerator/marker.py:22-64."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _rep


class BPM(Element):
    """
    Beam Position Monitor (BPM) in a particle accelerator.

    :param is_active: If `True` the BPM records the beam position `[mu_x, mu_y]` in `reading`.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_IDENTITY

    def __init__(self, is_active: bool = False, name: Optional[str] = None) -> None:
        super().__init__(name=name)
        self.is_active = is_active
        self.reading = None

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    @property
    def _host_barrier(self) -> bool:
        return bool(self.is_active)

    def _observe(self, incoming) -> None:
        """bpm.py:48-54."""
        from ..particles.beam import Beam

        if incoming is Beam.empty:
            self.reading = None
        else:
            self.reading = np.stack([np.asarray(incoming.mu_x), np.asarray(incoming.mu_y)])

    def track(self, incoming):
        """Record the reading (also when inactive, as bpm.py:48-58 does) and return a copy."""
        from ..particles.beam import Beam
        from ..particles.parameter_beam import ParameterBeam
        from ..particles.particle_beam import ParticleBeam

        if incoming is not Beam.empty and not isinstance(incoming, (ParameterBeam, ParticleBeam)):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        self._observe(incoming)
        return incoming if incoming is Beam.empty else incoming._shallow_copy()

    def broadcast(self, shape: tuple) -> Element:
        new_bpm = self.__class__(is_active=self.is_active, name=self.name)
        new_bpm.length = _rep(self.length, shape)
        return new_bpm

    def split(self, resolution) -> list:
        return [self]


class Marker(Element):
    """General Marker / Monitor element (identity map)."""

    _kind = _ffi.KIND_IDENTITY

    def track(self, incoming):
        return incoming  # marker.py:37-40

    def broadcast(self, shape: tuple) -> Element:
        new_marker = self.__class__(name=self.name)
        new_marker.length = _rep(self.length, shape)
        return new_marker

    @property
    def is_skippable(self) -> bool:
        return True

    def split(self, resolution) -> list:
        return [self]
