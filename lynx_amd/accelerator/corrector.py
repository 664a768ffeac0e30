"""lynx/accelerator/horizontal_corrector.py:23-110 and vertical_corrector.py:23-109."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class _Corrector(Element):
    def __init__(self, length, angle=None, name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        self.angle = _arr(angle, dtype) if angle is not None else np.zeros_like(self.length)

    def _param_rows(self, dtype):
        return [self.length, self.angle]

    def broadcast(self, shape: tuple) -> Element:
        # NB the reference repeats `length` only and leaves `angle` as it is
        # (horizontal_corrector.py:69-72); it then broadcasts when the map is built.
        return self.__class__(length=_rep(self.length, shape), angle=self.angle, name=self.name,
                              dtype=self.length.dtype)

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def is_active(self) -> bool:
        return bool(np.any(np.asarray(self.angle) != 0))

    def split(self, resolution) -> list:
        split_elements = []
        total = float(np.asarray(self.length).reshape(-1)[0])
        remaining = total
        resolution = float(np.asarray(resolution).reshape(-1)[0])
        while remaining > 0:
            length = min(resolution, remaining)
            split_elements.append(self.__class__(np.array([length]), self.angle * length / total,
                                                 dtype=self.length.dtype))
            remaining -= resolution
        return split_elements

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length", "angle"]

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(length={repr(self.length)}, angle={repr(self.angle)}, "
                f"name={repr(self.name)})")


class HorizontalCorrector(_Corrector):
    """
    Horizontal corrector magnet: a drift with a thin kick `angle` in x' (map entry [1, 6]).

    :param length: Length in meters.
    :param angle: Particle deflection angle in the horizontal plane in rad.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_HCOR


class VerticalCorrector(_Corrector):
    """
    Vertical corrector magnet: a drift with a thin kick `angle` in y' (map entry [3, 6]).

    :param length: Length in meters.
    :param angle: Particle deflection angle in the vertical plane in rad.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_VCOR
