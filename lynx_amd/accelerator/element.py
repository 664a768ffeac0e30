"""
Lattice elements as *parameter holders with a schema*.

The reference spells every element class out by hand (lynx/accelerator/element.py:23-149 and
one module per kind).  Here a class only declares

* `_kind`      -- the `lynx_kind` the kernels dispatch on (include/lynx_hip.h),
* `_batched`   -- its per-sample parameters as `(attribute, trailing components)`,
* `_row`       -- the order of those parameters in the kind's parameter row of the C ABI,
* `_settings`  -- non-batched attributes that are part of the element's identity,

and the base class derives packing, `broadcast`, `split`, `defining_features` and `repr` from
that.  `track` / `transfer_map` run on the GPU through `lynx_amd.engine`; nothing here computes.
"""

from __future__ import annotations

import copy
import itertools
from typing import Optional

import numpy as np

from .. import _ffi, engine
from ..device import frozen

_anonymous = itertools.count()

# EPOCH is bumped by every public attribute write on any element: while it stands still,
# nothing the packed lattice programs depend on can have changed, and `engine` reuses them in
# O(1).  STRUCTURE is bumped only by writes that can change how a lattice is partitioned into
# kernel launches -- attributes of elements whose skippability is not a class constant (cavity
# voltage, `is_active` of BPM / Screen / Aperture) and anything set on a Segment -- so that the
# optimisation-loop pattern (write a magnet strength, track, repeat) keeps its plan.
EPOCH = [0]
STRUCTURE = [0]


def generate_unique_name() -> str:
    """Names of elements constructed without one: unnamed_element_0, _1, ... (lynx/utils.py:1-11)."""
    return f"unnamed_element_{next(_anonymous)}"


def _arr(value, dtype):
    return np.asarray(value, dtype=dtype)


def _rep(value, shape):
    """`Tensor.repeat(shape)` as the reference's `broadcast` uses it (e.g. drift.py:64-65)."""
    return np.tile(np.asarray(value), tuple(shape))


def _float_dtype(energy, fallback):
    return energy.dtype if energy.dtype in (np.float32, np.float64) else fallback


class Element:
    """
    Base class for elements of particle accelerators.

    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_IDENTITY
    _batched: tuple = ()
    _row: tuple = ()
    _settings: tuple = ()
    _kept_on_broadcast: tuple = ()  # parameters `broadcast` passes on unrepeated
    _transient: tuple = ("_lattice_cache", "_plan")  # never shared between an element and its copies
    _host_barrier = False
    _skippable: Optional[bool] = None
    _version = 0
    length = np.zeros((1,), dtype=np.float32)

    def __init__(self, name: Optional[str] = None) -> None:
        self.name = name if name is not None else generate_unique_name()

    def __setattr__(self, key, value):
        # every parameter change invalidates the packed lattice programs this element is in;
        # arrays are kept as private read-only copies, so that an in-place write (`quad.k1[0] = 5`),
        # which no version counter could see, raises instead of leaving a stale device copy behind
        if isinstance(value, np.ndarray) and not key.startswith("_"):
            value = frozen(value)
        object.__setattr__(self, key, value)
        if not key.startswith("_"):
            object.__setattr__(self, "_version", self._version + 1)
            EPOCH[0] += 1
            if self._skippable is None:
                STRUCTURE[0] += 1

    def _adopt(self, dtype, length, **given) -> None:
        """
        Store the batched parameters: `length` fixes the batch shape, every parameter left
        `None` becomes zeros of that shape (plus its trailing components).
        """
        self.length = _arr(length, dtype)
        tails = dict(self._batched)
        for attr, value in given.items():
            if value is None:
                value = np.zeros((*self.length.shape, *((tails[attr],) if tails[attr] else ())), dtype=dtype)
            setattr(self, attr, _arr(value, dtype))

    # -- what the kernels need ---------------------------------------------------------------
    def _param_rows(self, dtype) -> list:
        """Parameter arrays in the kernel's fixed per-kind order (include/lynx_hip.h)."""
        tails = dict(self._batched)
        rows = []
        for attr in self._row:
            value = np.asarray(getattr(self, attr))
            if tails.get(attr):
                rows.extend(value[..., i] for i in range(tails[attr]))
            else:
                rows.append(value)
        return rows

    @classmethod
    def _row_names(cls) -> list:
        """Names of the parameter-row slots; a vector parameter `v` contributes `v_x`, `v_y`."""
        tails = dict(cls._batched)
        return [name for attr in cls._row
                for name in ([f"{attr}_{axis}" for axis in "xyz"[: tails[attr]]] if tails.get(attr) else [attr])]

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
        return engine.transfer_map(self, [self], energy, _float_dtype(energy, self.dtype), raw=True)

    def track(self, incoming):
        """Track a `ParameterBeam` or `ParticleBeam` through the element (element.py:61-94)."""
        return engine.track(self, [self], incoming, raw=True)

    def forward(self, incoming):
        return self.track(incoming)

    __call__ = forward

    def _twin(self) -> "Element":
        """Same class, name and parameters; none of the transient state."""
        twin = copy.copy(self)
        for key in self._transient:
            twin.__dict__.pop(key, None)
        return twin

    def broadcast(self, shape: tuple) -> "Element":
        """Repeat the batched parameters `shape` times (`Tensor.repeat` semantics, e.g. drift.py:64-65)."""
        twin = self._twin()
        for attr, tail in (("length", 0), *self._batched):
            if attr in self._kept_on_broadcast:
                continue
            object.__setattr__(twin, attr, frozen(_rep(getattr(self, attr), (*shape, *((1,) if tail else ()))),
                                                  copy=False))
        return twin

    @property
    def is_skippable(self) -> bool:
        if self._skippable is None:
            raise NotImplementedError
        return self._skippable

    @property
    def defining_features(self) -> list:
        names = [attr for attr, _ in self._batched]
        if "length" in self._row:
            names.insert(0, "length")
        return names + list(self._settings)

    def split(self, resolution) -> list:
        """Elements without a lengthwise split return themselves (e.g. dipole.py:196-199)."""
        return [self]

    def _slices(self, resolution):
        """Lengths of the pieces of `split`: `resolution` each, the remainder last (drift.py:71-78)."""
        remaining = float(np.asarray(self.length).reshape(-1)[0])
        resolution = float(np.asarray(resolution).reshape(-1)[0])
        while remaining > 0:
            yield np.array([min(resolution, remaining)], dtype=self.dtype)
            remaining -= resolution

    def plot(self, ax, s: float) -> None:
        raise NotImplementedError("plotting is out of scope of lynx_amd (matplotlib cosmetics)")

    def __repr__(self) -> str:
        shown = [f"{attr}={getattr(self, attr)!r}" for attr in self.defining_features]
        return f"{type(self).__name__}({', '.join(shown + [f'name={self.name!r}'])})"
