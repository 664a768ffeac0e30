"""
`Segment`: ordered list of elements = the lattice (lynx/accelerator/segment.py:32-539).
`track` hands the whole lattice to one fused GPU launch per stretch between host-side
barriers (active BPMs); plotting and the file-format constructors are out of scope.
"""

from __future__ import annotations

import itertools
from typing import Optional

import numpy as np

from .. import engine
from .diagnostics import Marker
from .magnets import CustomTransferMap, Drift
from .element import EPOCH, STRUCTURE, Element


class ElementList(list):
    """
    `Segment.elements`: a list that tells the engine when it has been changed in place.  The tracking plan of a
    segment is remembered for as long as the global STRUCTURE counter stands still AND the segment holds the same
    list object (`engine.plan`); without this, `segment.elements.append(...)` could only be noticed by comparing the
    identities of all elements on every `track` -- 5 us of a 25 us call on a 128-element lattice.
    """

    __slots__ = ()

    @staticmethod
    def _changed():
        EPOCH[0] += 1
        STRUCTURE[0] += 1


def _mutator(name):
    plain = getattr(list, name)

    def method(self, *args, **kwargs):
        result = plain(self, *args, **kwargs)
        ElementList._changed()
        return result

    method.__name__ = name
    return method


for _name in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove",
              "reverse", "sort", "clear"):
    setattr(ElementList, _name, _mutator(_name))


class Segment(Element):
    """
    Segment of a particle accelerator consisting of several elements.

    :param elements: List of elements that describe an accelerator (section).
    :param name: Unique identifier of the element.
    """

    def __init__(self, elements: list, name: Optional[str] = None) -> None:
        super().__init__(name=name)
        object.__setattr__(self, "elements", ElementList(elements))
        # name -> elements carrying it, in lattice order; `segment.<name>` resolves through
        # `__getattr__` to the element, or to the list of them when the name is not unique
        # (behaviour of segment.py:45-54)
        index: dict = {}
        for member in self.elements:
            index.setdefault(member.name, []).append(member)
        object.__setattr__(self, "_by_name", index)

    def __getattr__(self, key):
        # only reached when normal lookup fails: properties and methods of the class win
        found = self.__dict__.get("_by_name", {}).get(key)
        if found is None:
            raise AttributeError(f"{type(self).__name__!r} object has no attribute {key!r}")
        return found[0] if len(found) == 1 else found

    def __setattr__(self, key, value):
        if key == "elements" and type(value) is not ElementList:
            value = ElementList(value)
        object.__setattr__(self, key, value)
        EPOCH[0] += 1
        STRUCTURE[0] += 1

    @property
    def _version(self):  # a segment's version is the version of what it contains
        return sum(e._version for e in self.elements)

    def subcell(self, start: str, end: str) -> "Segment":
        """
        The stretch from the first element named `start` through the first element named `end`
        (behaviour of segment.py:56-68: the scan stops at the first `end` wherever it is, so an
        `end` in front of `start` gives an empty segment; a missing `end` runs to the last element).
        """
        names = [member.name for member in self.elements]
        stop = names.index(end) if end in names else len(names) - 1
        if start not in names[: stop + 1]:
            return self.__class__([])
        return self.__class__(self.elements[names.index(start): stop + 1])

    def _leaves(self):
        """Non-segment elements in lattice order, nested segments opened up."""
        for member in self.elements:
            if isinstance(member, Segment):
                yield from member._leaves()
            else:
                yield member

    def flattened(self) -> "Segment":
        """All sub-segments resolved into one top-level segment (behaviour of segment.py:70-82)."""
        return Segment(elements=list(self._leaves()), name=self.name)

    def transfer_maps_merged(self, incoming_beam, except_for: Optional[list] = None) -> "Segment":
        """
        Every maximal stretch of skippable elements not named in `except_for` replaced by one
        `CustomTransferMap` (behaviour of segment.py:84-132).  The beam is carried along only for
        its energy at each stretch.  As in the reference a one-element stretch stays what it is,
        except at the very end of the lattice, where it is wrapped too.
        """
        protected = set(except_for or ())
        groups = [(mergeable, list(members)) for mergeable, members in itertools.groupby(
            self.elements, key=lambda el: el.is_skippable and el.name not in protected)]
        out, beam = [], incoming_beam
        for position, (mergeable, members) in enumerate(groups):
            if mergeable and (len(members) > 1 or position == len(groups) - 1):
                members = [CustomTransferMap.from_merging_elements(members, incoming_beam=beam)]
            out.extend(members)
            if not mergeable or position < len(groups) - 1:  # a merged tail has nothing downstream
                for member in members:
                    beam = member.track(beam)
        return Segment(elements=out, name=self.name)

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
