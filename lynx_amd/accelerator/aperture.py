"""
`Aperture` (lynx/accelerator/aperture.py:23-153) as a parameter holder so that lattices
containing one load.  An inactive aperture is an identity element; an active one drops
particles, which changes N per batch sample -- out of scope of the fixed-shape streaming
path and refused loudly.
"""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _rep


class Aperture(Element):
    """
    Physical aperture.

    :param x_max: half size horizontal offset in [m]
    :param y_max: half size vertical offset in [m]
    :param shape: "rectangular" or "elliptical".
    :param is_active: If the aperture actually blocks particles.
    """

    _kind = _ffi.KIND_IDENTITY

    def __init__(self, x_max=None, y_max=None, shape: str = "rectangular", is_active: bool = True,
                 name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.x_max = np.asarray(x_max, dtype=dtype) if x_max is not None else np.asarray(np.inf, dtype=dtype)
        self.y_max = np.asarray(y_max, dtype=dtype) if y_max is not None else np.asarray(np.inf, dtype=dtype)
        self.shape = shape
        self.is_active = is_active
        self.lost_particles = None

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def track(self, incoming):
        if self.is_active:
            raise NotImplementedError("an active Aperture changes the particle count per sample; "
                                      "lynx_amd does not build ragged particle loss")
        return incoming

    def broadcast(self, shape: tuple) -> Element:
        new = self.__class__(x_max=_rep(self.x_max, shape), y_max=_rep(self.y_max, shape), shape=self.shape,
                             is_active=self.is_active, name=self.name)
        new.length = _rep(self.length, shape)
        return new

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["x_max", "y_max", "shape", "is_active"]
