"""`lynx.latticejson` by its reference names (latticejson.py:69-189), over `lynx_amd.io.latticejson`."""

from typing import Optional

from .io.latticejson import load_segment, save_segment


def save_cheetah_model(segment, filename: str, title: Optional[str] = None,
                       info: str = "This is a placeholder lattice description") -> None:
    save_segment(segment, filename, title=title, info=info)


def load_cheetah_model(filename: str):
    return load_segment(filename)
