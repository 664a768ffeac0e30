"""
Elements that observe the beam instead of transforming it: `BPM`, `Marker`, `Screen`,
`Aperture` (lynx/accelerator/bpm.py:24-80, marker.py:22-64, screen.py:22-271,
aperture.py:23-153).  All are identity elements for the kernels; an active BPM or Screen is a
host-side barrier of `engine.track`.  An active screen swallows the beam (`track` returns
`Beam.empty`) and renders it on read: exact 2-D histogram of (x, y) for a `ParticleBeam`,
bivariate-normal density for a `ParameterBeam`, both on the GPU (`lynx_histogram2d`,
`lynx_gaussian_image`).  Plotting is out of scope.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _ffi
from ..device import Dual, dtype_code, get_runtime
from .element import Element


def _beam_types():
    from ..particles.beam import Beam
    from ..particles.parameter_beam import ParameterBeam
    from ..particles.particle_beam import ParticleBeam

    return Beam, ParameterBeam, ParticleBeam


class Marker(Element):
    """General Marker / Monitor element (identity map)."""

    _skippable = True

    def track(self, incoming):
        return incoming  # marker.py:37-40

    def __repr__(self) -> str:
        return f"{type(self).__name__}(name={self.name!r})"


class BPM(Element):
    """
    Beam Position Monitor (BPM) in a particle accelerator.

    :param is_active: If `True` the BPM records the beam position `[mu_x, mu_y]` in `reading`.
    :param name: Unique identifier of the element.
    """

    _transient = Element._transient + ("reading",)
    reading = None

    def __init__(self, is_active: bool = False, name: Optional[str] = None) -> None:
        super().__init__(name=name)
        self.is_active = is_active
        self.reading = None

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    @property
    def _host_barrier(self) -> bool:
        return bool(self.is_active)

    def _observe(self, incoming) -> None:
        """bpm.py:48-54."""
        if incoming is _beam_types()[0].empty:
            self.reading = None
        else:
            self.reading = np.stack([np.asarray(incoming.mu_x), np.asarray(incoming.mu_y)])

    def track(self, incoming):
        """Record the reading (also when inactive, as bpm.py:48-58 does) and return a copy."""
        Beam, ParameterBeam, ParticleBeam = _beam_types()
        if incoming is not Beam.empty and not isinstance(incoming, (ParameterBeam, ParticleBeam)):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        self._observe(incoming)
        return incoming if incoming is Beam.empty else incoming._shallow_copy()

    def __repr__(self) -> str:
        return f"{type(self).__name__}(name={self.name!r})"


class Aperture(Element):
    """
    Physical aperture, kept as a parameter holder so that lattices containing one load.  An
    inactive aperture is an identity element; an active one drops particles, which changes N
    per batch sample -- out of scope of the fixed-shape streaming path
    and refused loudly.

    :param x_max: half size horizontal offset in [m]
    :param y_max: half size vertical offset in [m]
    :param shape: "rectangular" or "elliptical".
    :param is_active: If the aperture actually blocks particles.
    """

    _batched = (("x_max", 0), ("y_max", 0))
    _settings = ("shape", "is_active")

    def __init__(self, x_max=None, y_max=None, shape: str = "rectangular", is_active: bool = True,
                 name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.x_max = np.asarray(np.inf if x_max is None else x_max, dtype=dtype)
        self.y_max = np.asarray(np.inf if y_max is None else y_max, dtype=dtype)
        self.shape = shape
        self.is_active = is_active
        self.lost_particles = None

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def track(self, incoming):
        if self.is_active:
            raise NotImplementedError("an active Aperture changes the particle count per sample; "
                                      "lynx_amd does not build ragged particle loss")
        return incoming


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

    _batched = (("misalignment", 2),)
    _settings = ("resolution", "pixel_size", "binning", "is_active")
    _transient = Element._transient + ("_read_beam", "_cached_reading")
    _read_beam = None
    _cached_reading = None

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
        Beam, ParameterBeam, _ = _beam_types()
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
                host = np.array(incoming.particles)
                host[..., 0] -= mis[..., None, 0]
                host[..., 1] -= mis[..., None, 1]
                copy._particles = Dual(host)
                copy._batch = None
                copy._moments = None
        self.set_read_beam(copy)

    def track(self, incoming):
        if self.is_active:
            self._observe(incoming)
            return _beam_types()[0].empty
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
        Beam, ParameterBeam, ParticleBeam = _beam_types()
        beam = self.get_read_beam()
        if isinstance(beam, ParticleBeam):
            beam = beam.materialized()
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

    @property
    def defining_features(self) -> list:
        return ["resolution", "pixel_size", "binning", "misalignment", "is_active"]
