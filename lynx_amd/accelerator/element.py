"""
Base class of the lattice elements: the reference's `Element`
(lynx/accelerator/element.py:23-149) as a plain Python parameter holder.  `track` and
`transfer_map` run on the GPU through `lynx_amd.engine`.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _ffi, engine
from ..utils import UniqueNameGenerator

generate_unique_name = UniqueNameGenerator(prefix="unnamed_element")


def _arr(value, dtype):
    return np.asarray(value, dtype=dtype)


def _rep(value, shape):
    """`Tensor.repeat(shape)` as the reference's `broadcast` uses it (e.g. drift.py:64-65)."""
    value = np.asarray(value)
    return np.tile(value, tuple(shape))


class Element:
    """
    Base class for elements of particle accelerators.

    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_IDENTITY
    _host_barrier = False
    _version = 0
    length = np.zeros((1,), dtype=np.float32)

    def __init__(self, name: Optional[str] = None) -> None:
        self.name = name if name is not None else generate_unique_name()

    def __setattr__(self, key, value):
        # every parameter change invalidates the packed lattice programs this element is in
        object.__setattr__(self, key, value)
        if not key.startswith("_"):
            object.__setattr__(self, "_version", self._version + 1)

    # -- what the kernels need ---------------------------------------------------------------
    def _param_rows(self, dtype) -> list:
        """Parameter arrays in the kernel's fixed per-kind order (include/lynx_hip.h)."""
        return []

    def _static_flags(self) -> int:
        return 0

    @property
    def dtype(self):
        return np.asarray(self.length).dtype

    # -- reference API -----------------------------------------------------------------------
    def transfer_map(self, energy) -> np.ndarray:
        """
        The element's 7x7 transfer map for state (x, x', y, y', s, delta, 1)
        (lynx/accelerator/element.py:37-59), batched over `energy.shape`.
        """
        energy = np.asarray(energy)
        dtype = energy.dtype if energy.dtype in (np.float32, np.float64) else self.dtype
        return engine.transfer_map(self, [self], energy, dtype, raw=True)

    def track(self, incoming):
        """Track a `ParameterBeam` or `ParticleBeam` through the element (element.py:61-94)."""
        return engine.track(self, [self], incoming, raw=True)

    def forward(self, incoming):
        return self.track(incoming)

    __call__ = forward

    def broadcast(self, shape: tuple) -> "Element":
        raise NotImplementedError

    @property
    def is_skippable(self) -> bool:
        raise NotImplementedError

    @property
    def defining_features(self) -> list:
        return []

    def split(self, resolution) -> list:
        raise NotImplementedError

    def plot(self, ax, s: float) -> None:
        raise NotImplementedError("plotting is out of scope of lynx_amd (matplotlib cosmetics)")

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(name={repr(self.name)})"
