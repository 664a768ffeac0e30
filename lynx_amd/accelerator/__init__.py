"""Lattice description: elements (schema over `Element`), diagnostics, and `Segment`."""

from .diagnostics import BPM, Aperture, Marker, Screen
from .element import Element
from .magnets import (Cavity, CustomTransferMap, Dipole, Drift, HorizontalCorrector, Quadrupole, RBend, Solenoid,
                      Undulator, VerticalCorrector)
from .segment import Segment

__all__ = ["Aperture", "BPM", "Cavity", "CustomTransferMap", "Dipole", "Drift", "Element", "HorizontalCorrector",
           "Marker", "Quadrupole", "RBend", "Screen", "Segment", "Solenoid", "Undulator", "VerticalCorrector"]
