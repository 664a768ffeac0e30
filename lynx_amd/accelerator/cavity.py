"""lynx/accelerator/cavity.py:23-361."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class Cavity(Element):
    """
    Accelerating cavity in a particle accelerator.

    :param length: Length in meters.
    :param voltage: Voltage of the cavity in volts.
    :param phase: Phase of the cavity in degrees.
    :param frequency: Frequency of the cavity in Hz.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_CAVITY

    def __init__(self, length, voltage=None, phase=None, frequency=None, name: Optional[str] = None,
                 device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        z = lambda v: _arr(v, dtype) if v is not None else np.zeros_like(self.length)  # noqa: E731
        self.voltage = z(voltage)
        self.phase = z(phase)
        self.frequency = z(frequency)

    def _param_rows(self, dtype):
        return [self.length, self.voltage, self.phase, self.frequency]

    @property
    def is_active(self) -> bool:
        return bool(np.any(np.asarray(self.voltage) != 0))

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def transfer_map(self, energy) -> np.ndarray:
        """`_cavity_rmatrix` (cavity.py:248-325) regardless of whether the cavity is on."""
        from .. import engine

        energy = np.asarray(energy)
        dtype = energy.dtype if energy.dtype in (np.float32, np.float64) else self.dtype
        return engine.cavity_rmatrix(self, energy, dtype)

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(length=_rep(self.length, shape), voltage=_rep(self.voltage, shape),
                              phase=_rep(self.phase, shape), frequency=_rep(self.frequency, shape),
                              name=self.name, dtype=self.length.dtype)

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length", "voltage", "phase", "frequency"]

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(length={repr(self.length)}, voltage={repr(self.voltage)}, "
                f"phase={repr(self.phase)}, frequency={repr(self.frequency)}, name={repr(self.name)})")
