"""
`Segment`: ordered list of elements = the lattice (lynx/accelerator/segment.py:32-539).
`track` hands the whole lattice to one fused GPU launch per stretch between host-side
barriers (active BPMs); plotting and the file-format constructors are out of scope.
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from .. import engine
from .diagnostics import Marker
from .magnets import CustomTransferMap, Drift
from .element import EPOCH, STRUCTURE, Element


class Segment(Element):
    """
    Segment of a particle accelerator consisting of several elements.

    :param elements: List of elements that describe an accelerator (section).
    :param name: Unique identifier of the element.
    """

    def __init__(self, elements: list, name: Optional[str] = None) -> None:
        super().__init__(name=name)
        object.__setattr__(self, "elements", list(elements))
        for element in self.elements:
            # Elements are reachable as attributes by name; duplicates become a list
            # (segment.py:45-54).
            if element.name in self.__dict__:
                if isinstance(self.__dict__[element.name], list):
                    self.__dict__[element.name].append(element)
                else:
                    self.__dict__[element.name] = [self.__dict__[element.name], element]
            else:
                self.__dict__[element.name] = element

    def __setattr__(self, key, value):
        object.__setattr__(self, key, value)
        EPOCH[0] += 1
        STRUCTURE[0] += 1

    @property
    def _version(self):  # a segment's version is the version of what it contains
        return sum(e._version for e in self.elements)

    def subcell(self, start: str, end: str) -> "Segment":
        """Extract a subcell `[start, end]` from this segment (segment.py:56-68)."""
        subcell = []
        is_in_subcell = False
        for element in self.elements:
            if element.name == start:
                is_in_subcell = True
            if is_in_subcell:
                subcell.append(element)
            if element.name == end:
                break
        return self.__class__(subcell)

    def flattened(self) -> "Segment":
        """All sub-segments resolved into one top-level segment (segment.py:70-82)."""
        flattened_elements = []
        for element in self.elements:
            if isinstance(element, Segment):
                flattened_elements += element.flattened().elements
            else:
                flattened_elements.append(element)
        return Segment(elements=flattened_elements, name=self.name)

    def transfer_maps_merged(self, incoming_beam, except_for: Optional[list] = None) -> "Segment":
        """
        Segment where runs of skippable elements are merged into `CustomTransferMap`s
        (segment.py:84-132).  The beam is tracked along to know the energy at each run.
        """
        if except_for is None:
            except_for = []
        merged_elements = []
        skippable_elements = []
        tracked_beam = incoming_beam
        for element in self.elements:
            if element.is_skippable and element.name not in except_for:
                skippable_elements.append(element)
            else:
                if len(skippable_elements) == 1:
                    merged_elements.append(skippable_elements[0])
                    tracked_beam = skippable_elements[0].track(tracked_beam)
                elif len(skippable_elements) > 1:
                    merged_elements.append(
                        CustomTransferMap.from_merging_elements(skippable_elements, incoming_beam=tracked_beam))
                    tracked_beam = merged_elements[-1].track(tracked_beam)
                skippable_elements = []
                merged_elements.append(element)
                tracked_beam = element.track(tracked_beam)
        if len(skippable_elements) > 0:
            merged_elements.append(
                CustomTransferMap.from_merging_elements(skippable_elements, incoming_beam=tracked_beam))
        return Segment(elements=merged_elements, name=self.name)

    def without_inactive_markers(self, except_for: Optional[list] = None) -> "Segment":
        """segment.py:134-159 (removes every Marker not named in `except_for`)."""
        if except_for is None:
            except_for = []
        return Segment(
            elements=[e for e in self.elements if not isinstance(e, Marker) or e.name in except_for],
            name=self.name)

    def without_inactive_zero_length_elements(self, except_for: Optional[list] = None) -> "Segment":
        """segment.py:161-187."""
        if except_for is None:
            except_for = []
        return Segment(
            elements=[e for e in self.elements
                      if np.all(np.asarray(e.length) > 0.0)
                      or (hasattr(e, "is_active") and e.is_active) or e.name in except_for],
            name=self.name)

    def inactive_elements_as_drifts(self, except_for: Optional[list] = None) -> "Segment":
        """segment.py:189-218."""
        if except_for is None:
            except_for = []
        return Segment(
            elements=[
                (e if (hasattr(e, "is_active") and e.is_active) or np.all(np.asarray(e.length) == 0.0)
                 or e.name in except_for else Drift(e.length, dtype=np.asarray(e.length).dtype))
                for e in self.elements],
            name=self.name)

    @classmethod
    def from_lattice_json(cls, filepath: str) -> "Segment":
        """Load a LatticeJSON file written by `to_lattice_json` or by the reference (segment.py:220-228)."""
        from ..io.latticejson import load_segment

        return load_segment(filepath)

    def to_lattice_json(self, filepath: str, title: Optional[str] = None,
                        info: str = "This is a placeholder lattice description") -> None:
        """Save the segment as LatticeJSON (segment.py:230-247)."""
        from ..io.latticejson import save_segment

        save_segment(self, filepath, title, info)

    @property
    def is_skippable(self) -> bool:
        return all(element.is_skippable for element in self.elements)

    @property
    def length(self) -> np.ndarray:
        lengths = np.broadcast_arrays(*[np.asarray(element.length) for element in self.elements])
        return np.sum(np.stack(lengths, axis=0), axis=0)

    @property
    def dtype(self):
        return np.asarray(self.elements[0].length).dtype if self.elements else np.dtype(np.float32)

    def transfer_map(self, energy):
        """Product M_n ... M_1 of the element maps (segment.py:329-338); None if not skippable."""
        if not self.is_skippable:
            return None
        energy = np.asarray(energy)
        dtype = energy.dtype if energy.dtype in (np.float32, np.float64) else self.dtype
        return engine.transfer_map(self, self.elements, energy, dtype)

    def track(self, incoming):
        """segment.py:340-356."""
        return engine.track(self, self.elements, incoming)

    def forward(self, incoming):
        return self.track(incoming)

    __call__ = forward

    def broadcast(self, shape: tuple) -> Element:
        return self.__class__(elements=[element.broadcast(shape) for element in self.elements], name=self.name)

    def split(self, resolution) -> list:
        return [s for element in self.elements for s in element.split(resolution)]

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(elements={self.elements!r}, name={repr(self.name)})"
