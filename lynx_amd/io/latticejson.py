"""
LatticeJSON persistence of a `Segment` (format written by lynx/latticejson.py:69-105:
`"version": "cheetah-0.6"`, `elements` = name -> [class name, parameters],
`lattices` = segment name -> element names, `root`).  Files written by the reference load
here and vice versa.
"""

from __future__ import annotations

import json
from typing import Optional

import numpy as np

FORMAT_VERSION = "cheetah-0.6"


def _plain(value):
    """NumPy values -> JSON-serialisable lists / scalars (latticejson.py:9-19)."""
    if isinstance(value, np.ndarray):
        return value.tolist()
    if isinstance(value, np.generic):
        return value.item()
    return value


def _describe(segment, elements: dict, lattices: dict) -> None:
    """Depth-first walk collecting element definitions and the cell of every (sub)segment."""
    from ..accelerator.segment import Segment

    cell = []
    for element in segment.elements:
        if isinstance(element, Segment):
            _describe(element, elements, lattices)
        else:
            params = {feature: _plain(getattr(element, "_transfer_map" if feature == "transfer_map" else feature))
                      for feature in element.defining_features}
            elements[element.name] = [type(element).__name__, params]
        cell.append(element.name)
    lattices[segment.name] = cell


def _dumps(document: dict, indent: int = 4) -> str:
    """Indent the two outer levels only, one element per line (layout of latticejson.py:108-126)."""
    pad = " " * indent
    lines = ["{"]
    keys = list(document)
    for ki, key in enumerate(keys):
        value = document[key]
        comma = "," if ki + 1 < len(keys) else ""
        if isinstance(value, dict):
            lines.append(f"{pad}{json.dumps(key)}: {{")
            inner = list(value.items())
            for ii, (name, item) in enumerate(inner):
                lines.append(f"{pad * 2}{json.dumps(name)}: {json.dumps(item)}{',' if ii + 1 < len(inner) else ''}")
            lines.append(f"{pad}}}{comma}")
        else:
            lines.append(f"{pad}{json.dumps(key)}: {json.dumps(value)}{comma}")
    lines.append("}")
    return "\n".join(lines) + "\n"


def save_segment(segment, filename: str, title: Optional[str] = None,
                 info: str = "This is a placeholder lattice description") -> None:
    """`Segment.to_lattice_json` (latticejson.py:69-105)."""
    if title is None:
        title = segment.name if segment.name is not None else "Unnamed Lattice"
    elements: dict = {}
    lattices: dict = {}
    _describe(segment, elements, lattices)
    document = {"version": FORMAT_VERSION, "title": title, "info": info,
                "root": segment.name if segment.name is not None else "cell", "elements": elements,
                "lattices": lattices}
    with open(filename, "w") as f:
        f.write(_dumps(document))


def _build(name: str, document: dict):
    import lynx_amd

    if name in document["lattices"]:
        return lynx_amd.Segment(elements=[_build(child, document) for child in document["lattices"][name]], name=name)
    class_name, params = document["elements"][name]
    cls = getattr(lynx_amd, class_name, None)
    if cls is None:
        raise ValueError(f"LatticeJSON element {name!r}: class {class_name!r} is not available in lynx_amd")
    kwargs = {key: (value if isinstance(value, (str, bool)) else np.asarray(value, dtype=np.float32))
              for key, value in params.items()}  # latticejson.py:129-138
    return cls(name=name, **kwargs)


def load_segment(filename: str):
    """`Segment.from_lattice_json` (latticejson.py:177-189)."""
    with open(filename, "r") as f:
        document = json.load(f)
    return _build(document["root"], document)
