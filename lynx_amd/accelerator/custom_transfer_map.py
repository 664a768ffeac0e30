"""lynx/accelerator/custom_transfer_map.py:22-116."""

from typing import Optional

import numpy as np

from .. import _ffi
from .element import Element, _arr, _rep


class CustomTransferMap(Element):
    """This element can represent any custom transfer map."""

    _kind = _ffi.KIND_CUSTOM

    def __init__(self, transfer_map, length=None, name: Optional[str] = None, device=None,
                 dtype=np.float32) -> None:
        super().__init__(name=name)
        transfer_map = np.asarray(transfer_map)
        assert transfer_map.shape[-2:] == (7, 7)
        self._transfer_map = _arr(transfer_map, dtype)
        self.length = (_arr(length, dtype) if length is not None
                       else np.zeros(transfer_map.shape[:-2], dtype=dtype))

    @classmethod
    def from_merging_elements(cls, elements: list, incoming_beam) -> "CustomTransferMap":
        """
        Combine the transfer maps of successive skippable elements into one map
        (custom_transfer_map.py:48-85): tm = M_n ... M_1, at the incoming beam's energy.
        """
        from .. import engine

        assert all(element.is_skippable for element in elements), (
            "Combining the elements in a Segment that is not skippable will result in"
            " incorrect tracking results."
        )
        energy = np.asarray(incoming_beam.energy)
        dtype = incoming_beam.dtype
        holder = Element(name="merge")
        tm = engine.transfer_map(holder, list(elements), energy, dtype)
        combined_length = sum(np.asarray(element.length) for element in elements)
        combined_name = "combined_" + "_".join(element.name for element in elements)
        return cls(tm, length=combined_length, dtype=dtype, name=combined_name)

    def _param_rows(self, dtype):
        tm = np.asarray(self._transfer_map)
        flat = tm.reshape(*tm.shape[:-2], 49)
        return [flat[..., i] for i in range(49)]

    def transfer_map(self, energy) -> np.ndarray:
        return self._transfer_map

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(_rep(self._transfer_map, (*shape, 1, 1)), length=_rep(self.length, shape),
                              name=self.name, dtype=self._transfer_map.dtype)

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["transfer_map"]

    def split(self, resolution) -> list:
        return [self]

    def __repr__(self):
        return (f"{self.__class__.__name__}(transfer_map={repr(self._transfer_map)}, "
                f"length={repr(self.length)}, name={repr(self.name)})")
