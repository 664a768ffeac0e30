"""
Diagnostic screen (lynx/accelerator/screen.py:22-271).  An active screen swallows the beam
(`track` returns `Beam.empty`) and renders it on read: 2-D histogram of (x, y) for a
`ParticleBeam`, bivariate-normal density for a `ParameterBeam` -- both on the GPU
(`lynx_histogram2d`, `lynx_gaussian_image`).  Plotting is out of scope.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _ffi
from ..device import Dual, dtype_code, get_runtime
from .element import Element, _rep


class Screen(Element):
    """
    Diagnostic screen in a particle accelerator.

    :param resolution: Resolution of the camera sensor looking at the screen, (width, height) px.
    :param pixel_size: Size of a pixel on the screen in meters, (width, height).
    :param binning: Binning used by the camera.
    :param misalignment: Misalignment of the screen in meters, (x, y) per batch sample.
    :param is_active: If `True` the screen is in the beam path and records it.
    :param name: Unique identifier of the element.
    """

    _kind = _ffi.KIND_IDENTITY

    def __init__(self, resolution=None, pixel_size=None, binning=None, misalignment=None, is_active: bool = False,
                 name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        dtype = np.dtype(dtype)
        self.resolution = np.asarray(resolution if resolution is not None else (1024, 1024), dtype=dtype)
        self.pixel_size = np.asarray(pixel_size if pixel_size is not None else (1e-3, 1e-3), dtype=dtype)
        self.binning = np.asarray(binning if binning is not None else 1, dtype=dtype)
        self.misalignment = np.asarray(misalignment if misalignment is not None else (0.0, 0.0), dtype=dtype)
        self.length = np.zeros(self.misalignment.shape[:-1], dtype=dtype)
        self.is_active = is_active
        self._read_beam = None
        self._cached_reading = None

    # -- geometry (screen.py:86-120) -----------------------------------------------------------
    @property
    def effective_resolution(self) -> np.ndarray:
        return self.resolution / self.binning

    @property
    def effective_pixel_size(self) -> np.ndarray:
        return self.pixel_size * self.binning

    @property
    def extent(self) -> np.ndarray:
        half = self.resolution * self.pixel_size / 2
        return np.stack([-half[0], half[0], -half[1], half[1]])

    @property
    def pixel_bin_edges(self) -> tuple:
        e = self.extent
        dt = self.resolution.dtype
        return (np.linspace(e[0], e[1], int(self.effective_resolution[0]) + 1, dtype=dt),
                np.linspace(e[2], e[3], int(self.effective_resolution[1]) + 1, dtype=dt))

    # -- tracking ------------------------------------------------------------------------------
    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    @property
    def _host_barrier(self) -> bool:
        return bool(self.is_active)

    _swallows_beam = True  # engine.track: everything behind an active screen sees Beam.empty

    def _observe(self, incoming) -> None:
        """screen.py:126-139: park a misalignment-corrected copy of the beam."""
        from ..particles.beam import Beam
        from ..particles.parameter_beam import ParameterBeam

        if incoming is Beam.empty:
            self.set_read_beam(incoming)
            return
        mis = np.asarray(self.misalignment, dtype=incoming.dtype)
        copy = incoming._shallow_copy()
        if np.any(mis != 0):
            if isinstance(incoming, ParameterBeam):
                mu = np.array(incoming._mu)
                mu[..., 0] -= mis[..., 0]
                mu[..., 2] -= mis[..., 1]
                copy._mu_d = Dual(mu)
            else:
                # the reference subtracts the y misalignment from coordinate 1 (x'), not from
                # y (screen.py:134-135); kept as it is
                host = np.array(incoming._particles.host())
                host[..., 0] -= mis[..., None, 0]
                host[..., 1] -= mis[..., None, 1]
                copy._particles = Dual(host)
                copy._moments = None
        self.set_read_beam(copy)

    def track(self, incoming):
        from ..particles.beam import Beam

        if self.is_active:
            self._observe(incoming)
            return Beam.empty
        return incoming

    def get_read_beam(self):
        return self._read_beam

    def set_read_beam(self, value) -> None:
        self._read_beam = value
        self._cached_reading = None

    # -- read-out (screen.py:143-216) ----------------------------------------------------------
    @property
    def reading(self) -> np.ndarray:
        if self._cached_reading is not None:
            return self._cached_reading
        from ..particles.beam import Beam
        from ..particles.parameter_beam import ParameterBeam
        from ..particles.particle_beam import ParticleBeam

        beam = self.get_read_beam()
        nx, ny = int(self.effective_resolution[0]), int(self.effective_resolution[1])
        if beam is Beam.empty or beam is None:
            image = np.zeros((*self.misalignment.shape[:-1], ny, nx), dtype=self.resolution.dtype)
        elif isinstance(beam, ParticleBeam):
            rt = get_runtime()
            dtype = beam.dtype
            B = int(np.prod(beam.batch_shape, dtype=np.int64))
            xe, ye = (rt.to_device(np.ascontiguousarray(e.astype(dtype))) for e in self.pixel_bin_edges)
            out = rt.empty((*beam.batch_shape, ny, nx), np.int32)
            rt.check(rt.lib.lynx_histogram2d(rt.ctx, dtype_code(dtype), B, beam.num_particles,
                                             C.c_void_p(beam._particles.device(rt).ptr), C.c_void_p(xe.ptr),
                                             C.c_void_p(ye.ptr), nx, ny, C.c_void_p(out.ptr)))
            image = out.numpy().astype(dtype)
        elif isinstance(beam, ParameterBeam):
            rt = get_runtime()
            dtype = beam.dtype
            B = int(np.prod(beam.batch_shape, dtype=np.int64))
            e = self.extent.astype(dtype)
            step = (self.pixel_size * self.binning).astype(dtype)
            xs = np.arange(e[0], e[1], step[0], dtype=dtype)
            ys = np.arange(e[2], e[3], step[1], dtype=dtype)
            out = rt.empty((*beam.batch_shape, len(xs), len(ys)), dtype)
            rt.check(rt.lib.lynx_gaussian_image(rt.ctx, dtype_code(dtype), B, C.c_void_p(beam._mu_d.device(rt).ptr),
                                                C.c_void_p(beam._cov_d.device(rt).ptr),
                                                C.c_void_p(rt.to_device(xs).ptr), C.c_void_p(rt.to_device(ys).ptr),
                                                len(xs), len(ys), C.c_void_p(out.ptr)))
            image = out.numpy()
        else:
            raise TypeError(f"Read beam is of invalid type {type(beam)}")
        self._cached_reading = image
        return image

    def broadcast(self, shape: tuple) -> Element:
        new = self.__class__(resolution=self.resolution, pixel_size=self.pixel_size, binning=self.binning,
                             misalignment=_rep(self.misalignment, (*shape, 1)), is_active=self.is_active,
                             name=self.name, dtype=self.resolution.dtype)
        new.length = _rep(self.length, shape)
        return new

    def split(self, resolution) -> list:
        return [self]

    @property
    def defining_features(self) -> list:
        return super().defining_features + ["resolution", "pixel_size", "binning", "misalignment", "is_active"]

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(resolution={repr(self.resolution)}, pixel_size={repr(self.pixel_size)}, "
                f"binning={repr(self.binning)}, misalignment={repr(self.misalignment)}, "
                f"is_active={repr(self.is_active)}, name={repr(self.name)})")
