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

    _transient = Element._transient + ("_reading", "_pending_reading")
    _reading = None
    _pending_reading = None

    def __init__(self, is_active: bool = False, name: Optional[str] = None) -> None:
        super().__init__(name=name)
        self.is_active = is_active
        self.reading = None

    @property
    def reading(self):
        """`stack([mu_x, mu_y])` of the last beam that entered the active BPM (bpm.py:50-54), shape (2, *batch).
        After a ParticleBeam pass it still sits in HBM (the streaming kernel summed x and y on the way) and
        is copied to the host here, on first look."""
        pending = self.__dict__.get("_pending_reading")
        if pending is not None:
            obs, k, batch_shape, dtype = pending
            host = obs.numpy()[:, k, :].reshape(*batch_shape, 2)
            object.__setattr__(self, "_reading", np.stack([host[..., 0], host[..., 1]]).astype(dtype))
            object.__setattr__(self, "_pending_reading", None)
        return self._reading

    @reading.setter
    def reading(self, value):
        object.__setattr__(self, "_pending_reading", None)
        object.__setattr__(self, "_reading", value)

    def _reading_from(self, obs, k, batch_shape, dtype) -> None:
        object.__setattr__(self, "_pending_reading", (obs, k, tuple(batch_shape), np.dtype(dtype)))

    @property
    def _fusable_observer(self) -> bool:
        return bool(self.is_active)

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
    Physical aperture (lynx/accelerator/aperture.py:23-153).  An active aperture removes the
    particles of a `ParticleBeam` outside |x| < x_max, |y| < y_max (or the ellipse); a
    `ParameterBeam` passes unchanged.  The mask, the survivor count and the order-preserving
    compaction run on the GPU (`lynx_aperture_mask`, `lynx_aperture_compact`).

    Particle loss makes the particle count depend on the sample, so a batch is accepted as long
    as no sample loses a particle (the ARES lattice's three apertures are infinite) and a single
    sample may lose any number; the reference's boolean-mask indexing flattens the batch
    dimension in that case (aperture.py:87), here the leading batch dimension of size 1 stays.

    :param x_max: half size horizontal offset in [m]
    :param y_max: half size vertical offset in [m]
    :param shape: "rectangular" or "elliptical".
    :param is_active: If the aperture actually blocks particles.
    """

    _batched = (("x_max", 0), ("y_max", 0))
    _settings = ("shape", "is_active")
    _transient = Element._transient + ("lost_particles", "lost_particle_charges")
    lost_particles = None
    lost_particle_charges = None

    def __init__(self, x_max=None, y_max=None, shape: str = "rectangular", is_active: bool = True,
                 name: Optional[str] = None, device=None, dtype=np.float32) -> None:
        super().__init__(name=name)
        self.x_max = np.asarray(np.inf if x_max is None else x_max, dtype=dtype)
        self.y_max = np.asarray(np.inf if y_max is None else y_max, dtype=dtype)
        self.shape = shape
        self.is_active = is_active

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    @property
    def _host_barrier(self) -> bool:
        return bool(self.is_active)

    def _observe(self, incoming) -> None:
        pass

    def _transform(self, incoming):
        """What an active aperture does to the beam (aperture.py:69-108)."""
        Beam, _, ParticleBeam = _beam_types()
        if not isinstance(incoming, ParticleBeam):
            return incoming  # only particle beams are clipped (aperture.py:70-72)
        assert np.all(self.x_max >= 0) and np.all(self.y_max >= 0)
        assert self.shape in ["rectangular", "elliptical"], f"Unknown aperture shape {self.shape}"
        beam = incoming.materialized()
        rt = get_runtime()
        dtype, batch, n = beam.dtype, beam.batch_shape, beam.num_particles
        B = int(np.prod(batch, dtype=np.int64))
        limits = []
        for value in (self.x_max, self.y_max):
            value = np.asarray(value, dtype=dtype)
            limits.append(value.reshape(1) if value.size == 1 else np.ascontiguousarray(np.broadcast_to(value, batch)).reshape(B))
        stride = 0 if limits[0].size == 1 and limits[1].size == 1 else 1
        if stride:
            limits = [np.ascontiguousarray(np.broadcast_to(v, (B,))) for v in limits]
        chunks = (n + 1023) // 1024
        mask = rt.empty((*batch, n), np.uint8)
        counts = rt.empty((B, chunks), np.int32)
        offsets = rt.empty((B, chunks), np.int64)
        totals = rt.empty((B,), np.int64)
        particles = beam._particles.device(rt)
        ptr = lambda a: C.c_void_p(a.ptr)  # noqa: E731
        x_dev, y_dev = rt.to_device(limits[0]), rt.to_device(limits[1])  # named: alive until the call is enqueued
        rt.check(rt.lib.lynx_aperture_mask(rt.ctx, dtype_code(dtype), B, n, ptr(particles), ptr(x_dev),
                                           ptr(y_dev), stride, int(self.shape == "elliptical"),
                                           ptr(mask), ptr(counts), ptr(offsets), ptr(totals)))
        survivors = totals.numpy()
        if np.all(survivors == n):  # nothing lost: the beam goes on as it is
            self.lost_particles = np.zeros((0, 7), dtype=dtype)
            self.lost_particle_charges = np.zeros((0,), dtype=dtype)
            return beam._shallow_copy()
        if B != 1:
            raise NotImplementedError(
                f"Aperture {self.name!r}: {int(np.sum(n - survivors))} particles lost in a batch of {B} samples; the "
                "particle count would differ between samples (the reference's mask indexing flattens the batch here)")
        k = int(survivors[0])
        keep = mask.numpy().reshape(-1).astype(bool)
        charges = np.asarray(beam.particle_charges).reshape(-1)
        if k == 0:
            self.lost_particles, self.lost_particle_charges = np.asarray(beam.particles).reshape(n, 7), charges
            return Beam.empty  # aperture.py:106-107
        kept, lost = rt.empty((*batch, k, 7), dtype), rt.empty((n - k, 7), dtype)
        rt.check(rt.lib.lynx_aperture_compact(rt.ctx, dtype_code(dtype), n, ptr(particles), ptr(mask), ptr(offsets),
                                              ptr(kept), ptr(lost)))
        self.lost_particles, self.lost_particle_charges = lost, charges[~keep]
        return ParticleBeam(kept, beam.energy, particle_charges=charges[keep].reshape(*batch, k), dtype=dtype)

    def track(self, incoming):
        return self._transform(incoming) if self.is_active else incoming


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
                copy._mu_d = Dual(mu, owned=True)
            else:
                # the reference subtracts the y misalignment from coordinate 1 (x'), not from
                # y (screen.py:134-135); kept as it is
                host = np.array(incoming.particles)
                host[..., 0] -= mis[..., None, 0]
                host[..., 1] -= mis[..., None, 1]
                copy._particles = Dual(host, owned=True)
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
            xs_dev, ys_dev = rt.to_device(xs), rt.to_device(ys)  # named: alive until the call is enqueued
            rt.check(rt.lib.lynx_gaussian_image(rt.ctx, dtype_code(dtype), B, C.c_void_p(beam._mu_d.device(rt).ptr),
                                                C.c_void_p(beam._cov_d.device(rt).ptr),
                                                C.c_void_p(xs_dev.ptr), C.c_void_p(ys_dev.ptr),
                                                len(xs), len(ys), C.c_void_p(out.ptr)))
            image = out.numpy()
        else:
            raise TypeError(f"Read beam is of invalid type {type(beam)}")
        self._cached_reading = image
        return image

    @property
    def defining_features(self) -> list:
        return ["resolution", "pixel_size", "binning", "misalignment", "is_active"]
