"""lynx/accelerator/dipole.py:24-234 and lynx/accelerator/rbend.py:20-80."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class Dipole(Element):
    """
    Dipole magnet (by default a sector bending magnet).

    :param length: Length in meters.
    :param angle: Deflection angle in rad.
    :param e1: The angle of inclination of the entrance face [rad].
    :param e2: The angle of inclination of the exit face [rad].
    :param tilt: Tilt of the magnet in x-y plane [rad].
    :param fringe_integral: Fringe field integral (of the enterance face).
    :param fringe_integral_exit: Fringe field integral of the exit face (default: same).
    :param gap: The magnet gap [m].
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_DIPOLE

    def __init__(self, length, angle=None, e1=None, e2=None, tilt=None, fringe_integral=None,
                 fringe_integral_exit=None, gap=None, name: Optional[str] = None, device=None,
                 dtype=np.float32):
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        z = lambda v: _arr(v, dtype) if v is not None else np.zeros_like(self.length)  # noqa: E731
        self.angle = z(angle)
        self.gap = z(gap)
        self.tilt = z(tilt)
        self.fringe_integral = z(fringe_integral)
        self.fringe_integral_exit = (self.fringe_integral if fringe_integral_exit is None
                                     else _arr(fringe_integral_exit, dtype))
        self.e1 = z(e1)
        self.e2 = z(e2)

    @property
    def hx(self) -> np.ndarray:
        """Curvature angle/length, 0 where length == 0 (dipole.py:96-102)."""
        length = np.asarray(self.length)
        angle = np.broadcast_to(np.asarray(self.angle), length.shape)
        value = np.zeros_like(length)
        nz = length != 0
        value[nz] = angle[nz] / length[nz]
        return value

    def _param_rows(self, dtype):
        return [self.length, self.angle, self.e1, self.e2, self.tilt, self.fringe_integral,
                self.fringe_integral_exit, self.gap]

    def _static_flags(self) -> int:
        return _ffi.FLAG_THICK if np.any(np.asarray(self.length) != 0.0) else 0  # dipole.py:119

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def is_active(self):
        return bool(np.any(np.asarray(self.angle) != 0))

    def broadcast(self, shape: tuple) -> Element:
        new = Dipole(length=_rep(self.length, shape), angle=_rep(self.angle, shape), e1=_rep(self.e1, shape),
                     e2=_rep(self.e2, shape), tilt=_rep(self.tilt, shape),
                     fringe_integral=_rep(self.fringe_integral, shape),
                     fringe_integral_exit=_rep(self.fringe_integral_exit, shape), gap=_rep(self.gap, shape),
                     name=self.name, dtype=self.length.dtype)
        new.__class__ = self.__class__  # an RBend's e1/e2 already hold the angle/2 shift
        return new

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length", "angle", "e1", "e2", "tilt", "fringe_integral",
                                            "fringe_integral_exit", "gap"]

    def __repr__(self):
        return (f"{self.__class__.__name__}(length={repr(self.length)}, angle={repr(self.angle)}, "
                f"e1={repr(self.e1)},e2={repr(self.e2)},tilt={repr(self.tilt)},"
                f"fringe_integral={repr(self.fringe_integral)},"
                f"fringe_integral_exit={repr(self.fringe_integral_exit)},gap={repr(self.gap)},"
                f"name={repr(self.name)})")


class RBend(Dipole):
    """Rectangular bending magnet: a `Dipole` with e1 += angle/2, e2 += angle/2 (rbend.py:79-80)."""

    def __init__(self, length, angle=None, e1=None, e2=None, tilt=None, fringe_integral=None,
                 fringe_integral_exit=None, gap=None, name: Optional[str] = None, device=None,
                 dtype=np.float32):
        super().__init__(length=length, angle=angle, e1=e1, e2=e2, tilt=tilt, fringe_integral=fringe_integral,
                         fringe_integral_exit=fringe_integral_exit, gap=gap, name=name, device=device,
                         dtype=dtype)
        self.e1 = self.e1 + self.angle / 2
        self.e2 = self.e2 + self.angle / 2
