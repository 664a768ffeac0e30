"""
The public helpers of `lynx/track_methods.py`, evaluated by the same GPU map builders the
tracking kernels use (`lynx_amd/csrc/lynx_maps.hpp`).  Inputs are array-likes, outputs
NumPy arrays of shape (*batch, 7, 7).
"""

from __future__ import annotations

from typing import Optional

import numpy as np

from . import _ffi, engine
from .accelerator.element import Element

REST_ENERGY = 510998.9506917531  # electron rest energy in eV (track_methods.py:9-11)


class _Helper(Element):
    """Parameter holder for one helper map (never part of a user's lattice)."""

    is_skippable = True

    def __init__(self, kind, rows, flags=0):
        super().__init__(name="track_methods_helper")
        self._kind = kind
        self._rows = rows
        self._flags = flags

    def _param_rows(self, dtype):
        return self._rows

    def _static_flags(self) -> int:
        return self._flags


def _dtype_of(*arrays):
    for a in arrays:
        a = np.asarray(a)
        if a.dtype in (np.float32, np.float64):
            return a.dtype
    return np.dtype(np.float32)


def rotation_matrix(angle) -> np.ndarray:
    """Rotate the transfer map in the x-y plane (track_methods.py:14-34)."""
    angle = np.asarray(angle)
    dtype = _dtype_of(angle)
    el = _Helper(_ffi.KIND_ROTATION, [angle.astype(dtype)])
    return engine.transfer_map(el, [el], np.zeros(angle.shape, dtype), dtype, raw=True)


def base_rmatrix(length, k1, hx, tilt: Optional[np.ndarray] = None, energy: Optional[np.ndarray] = None) -> np.ndarray:
    """
    Universal transfer matrix of a beamline element (track_methods.py:37-105).

    :param length: Length of the element in m.
    :param k1: Quadrupole strength in 1/m**2.
    :param hx: Curvature (1/radius) of the element in 1/m.
    :param tilt: Rotation of the element relative to the longitudinal axis in rad.
    :param energy: Beam energy in eV.
    """
    length = np.asarray(length)
    dtype = _dtype_of(length)
    shape = length.shape
    tilt = np.zeros(shape, dtype) if tilt is None else np.asarray(tilt, dtype)
    energy = np.zeros(shape, dtype) if energy is None else np.asarray(energy, dtype)
    rows = [length.astype(dtype), np.asarray(k1, dtype), np.asarray(hx, dtype), tilt]
    flags = _ffi.FLAG_TILT if np.any(tilt != 0) else 0  # :101, whole batch
    el = _Helper(_ffi.KIND_BASE_RMATRIX, rows, flags)
    return engine.transfer_map(el, [el], np.broadcast_to(energy, shape), dtype, raw=True)


def misalignment_matrix(misalignment) -> tuple:
    """Shift maps of a misaligned element: (R_entry, R_exit) (track_methods.py:108-122)."""
    misalignment = np.asarray(misalignment)
    dtype = _dtype_of(misalignment)
    batch = misalignment.shape[:-1]
    mx, my = misalignment[..., 0].astype(dtype), misalignment[..., 1].astype(dtype)
    out = []
    for sign in (-1.0, 1.0):
        el = _Helper(_ffi.KIND_MISALIGNMENT, [mx, my, np.full(batch, sign, dtype)])
        out.append(engine.transfer_map(el, [el], np.zeros(batch, dtype), dtype, raw=True))
    return out[0], out[1]
