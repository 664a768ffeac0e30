"""lynx/accelerator/solenoid.py:23-145 and lynx/accelerator/undulator.py:22-95."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class Solenoid(Element):
    """
    Solenoid magnet (A. W. Chao p. 74).

    :param length: Length in meters.
    :param k: Normalised strength B0 / (2 B rho).
    :param misalignment: Misalignment vector of the solenoid in x- and y-directions.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_SOLENOID

    def __init__(self, length=None, k=None, misalignment=None, name: Optional[str] = None, device=None,
                 dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        self.k = _arr(k, dtype) if k is not None else np.zeros_like(self.length)
        self.misalignment = (_arr(misalignment, dtype) if misalignment is not None
                             else np.zeros((*self.length.shape, 2), dtype=dtype))

    def _param_rows(self, dtype):
        mis = np.asarray(self.misalignment)
        return [self.length, self.k, mis[..., 0], mis[..., 1]]

    def _static_flags(self) -> int:
        return 0 if np.all(np.asarray(self.misalignment) == 0) else _ffi.FLAG_MISALIGNED  # solenoid.py:98

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(length=_rep(self.length, shape), k=_rep(self.k, shape),
                              misalignment=_rep(self.misalignment, (*shape, 1)), name=self.name,
                              dtype=self.length.dtype)

    @property
    def is_active(self) -> bool:
        return bool(np.any(np.asarray(self.k) != 0))

    @property
    def is_skippable(self) -> bool:
        return True

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length", "k", "misalignment"]

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(length={repr(self.length)}, k={repr(self.k)}, "
                f"misalignment={repr(self.misalignment)}, name={repr(self.name)})")


class Undulator(Element):
    """
    Undulator: behaves like a drift section (with R56 = +L / gamma^2 as the reference spells it).

    :param length: Length in meters.
    :param is_active: Currently has no effect.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_UNDULATOR

    def __init__(self, length, is_active: bool = False, name: Optional[str] = None, device=None,
                 dtype=np.float32) -> None:
        super().__init__(name=name)
        self.length = _arr(length, dtype)
        self.is_active = is_active

    def _param_rows(self, dtype):
        return [self.length]

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(length=_rep(self.length, shape), is_active=self.is_active, name=self.name,
                              dtype=self.length.dtype)

    @property
    def is_skippable(self) -> bool:
        return True

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["length"]

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(length={repr(self.length)}, is_active={repr(self.is_active)}, name={repr(self.name)})"
