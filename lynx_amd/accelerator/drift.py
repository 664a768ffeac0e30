"""lynx/accelerator/drift.py:21-88."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class Drift(Element):
    """
    Drift section in a particle accelerator (linear map incl. R56 = -L / (beta^2 gamma^2)).

    :param length: Length in meters.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_DRIFT

    def __init__(self, length, name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)

    def _param_rows(self, dtype):
        return [self.length]

    def transfer_map(self, energy) -> np.ndarray:
        energy = np.asarray(energy)
        assert energy.shape == self.length.shape, (  # drift.py:45-47
            f"Beam shape {energy.shape} does not match element shape {self.length.shape}"
        )
        return super().transfer_map(energy)

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(length=_rep(self.length, shape), name=self.name, dtype=self.length.dtype)

    @property
    def is_skippable(self) -> bool:
        return True

    def split(self, resolution) -> list:
        split_elements = []
        remaining = float(np.asarray(self.length).reshape(-1)[0])
        resolution = float(np.asarray(resolution).reshape(-1)[0])
        while remaining > 0:
            split_elements.append(Drift(np.array([min(resolution, remaining)]), dtype=self.length.dtype))
            remaining -= resolution
        return split_elements

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length"]

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(length={repr(self.length)})"
