"""lynx/accelerator/quadrupole.py:24-131."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class Quadrupole(Element):
    """
    Quadrupole magnet in a particle accelerator.

    :param length: Length in meters.
    :param k1: Strength of the quadrupole in rad/m.
    :param misalignment: Misalignment vector of the quadrupole in x- and y-directions.
    :param tilt: Tilt angle of the quadrupole in x-y plane [rad]. pi/4 for skew-quadrupole.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_QUADRUPOLE

    def __init__(self, length, k1=None, misalignment=None, tilt=None, name: Optional[str] = None,
                 device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        self.k1 = _arr(k1, dtype) if k1 is not None else np.zeros_like(self.length)
        self.misalignment = (_arr(misalignment, dtype) if misalignment is not None
                             else np.zeros((*self.length.shape, 2), dtype=dtype))
        self.tilt = _arr(tilt, dtype) if tilt is not None else np.zeros_like(self.length)

    def _param_rows(self, dtype):
        mis = np.asarray(self.misalignment)
        return [self.length, self.k1, self.tilt, mis[..., 0], mis[..., 1]]

    def _static_flags(self) -> int:
        f = 0
        if np.any(np.asarray(self.tilt) != 0):  # track_methods.py:101
            f |= _ffi.FLAG_TILT
        if not np.all(np.asarray(self.misalignment) == 0):  # quadrupole.py:75
            f |= _ffi.FLAG_MISALIGNED
        return f

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(length=_rep(self.length, shape), k1=_rep(self.k1, shape),
                              misalignment=_rep(self.misalignment, (*shape, 1)), tilt=_rep(self.tilt, shape),
                              name=self.name, dtype=self.length.dtype)

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def is_active(self) -> bool:
        return bool(np.any(np.asarray(self.k1) != 0))

    def split(self, resolution) -> list:
        split_elements = []
        remaining = float(np.asarray(self.length).reshape(-1)[0])
        resolution = float(np.asarray(resolution).reshape(-1)[0])
        while remaining > 0:
            split_elements.append(Quadrupole(np.array([min(resolution, remaining)]), self.k1,
                                             misalignment=self.misalignment, dtype=self.length.dtype))
            remaining -= resolution
        return split_elements

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length", "k1", "misalignment", "tilt"]

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(length={repr(self.length)}, k1={repr(self.k1)}, "
                f"misalignment={repr(self.misalignment)}, tilt={repr(self.tilt)}, name={repr(self.name)})")
