"""
`ParticleBeam`: particles (…, N, 7) resident in HBM (lynx/particles/particle_beam.py:13-855).

The particle array is only copied to the host when somebody asks for it; all moment
properties come from ONE fused reduction pass on the GPU (`lynx_moments`, or the epilogue
of the tracking kernel), not from 14 separate reductions as in the reference.
Samplers run on the host with NumPy (they are not on the hot path).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _ffi, config
from ..device import DeviceArray, Dual, dtype_code, get_runtime
from .beam import Beam, _batch_args


def _tri(i: int, j: int) -> int:
    """Index of cov[i, j] (i <= j < 6) in a moment record (include/lynx_hip.h)."""
    return 7 + i * 6 - (i * (i - 1)) // 2 + (j - i)


class ParticleBeam(Beam):
    """
    Beam of charged particles, where each particle is simulated.

    :param particles: 7-dimensional particle vectors, shape (*batch, N, 7); 7th entry is 1.
    :param energy: Energy of the beam in eV, shape `batch`.
    :param particle_charges: Charges of the macroparticles in C, shape (*batch, N).
    """

    def __init__(self, particles, energy, particle_charges=None, device=None, dtype=np.float32) -> None:
        dtype = np.dtype(dtype)
        shape = particles.shape
        assert shape[-2] > 0 and shape[-1] == 7, "Particle vectors must be 7-dimensional."
        if isinstance(particles, DeviceArray):
            assert particles.dtype == dtype
            p = Dual(dev=particles)
        else:
            p = Dual(np.ascontiguousarray(np.asarray(particles, dtype=dtype)))
        charges = None if particle_charges is None else np.asarray(particle_charges, dtype=dtype)
        self._init_raw(p, Dual(np.asarray(energy, dtype=dtype)), charges, dtype)

    def _init_raw(self, particles: Dual, energy: Dual, charges, dtype, moments: Dual | None = None,
                  batch: tuple | None = None):
        self._particles, self._energy, self._charges = particles, energy, charges
        self.dtype = np.dtype(dtype)
        self._moments = moments
        # Lazily broadcast beam: ONE stored beam (storage batch of size 1) standing for `batch`
        # identical samples.  `Segment.track` then reads it once per sample out of the caches
        # (LYNX_TRACK_SHARED_INPUT) instead of streaming `batch` physical copies from HBM.
        self._batch = batch

    def _shallow_copy(self):
        out = ParticleBeam.__new__(ParticleBeam)
        out._init_raw(self._particles, self._energy, self._charges, self.dtype, self._moments, self._batch)
        return out

    @property
    def is_shared(self) -> bool:
        return self._batch is not None

    def materialized(self) -> "ParticleBeam":
        """The same beam with its particles physically repeated (what the reference's `broadcast` returns)."""
        if not self.is_shared:
            return self
        return self.__class__(particles=np.ascontiguousarray(self.particles), energy=self.energy,
                              particle_charges=None if self._charges is None else np.ascontiguousarray(
                                  self.particle_charges), dtype=self.dtype)

    def _full(self, stored: np.ndarray, tail: int) -> np.ndarray:
        """Read-only view of a stored per-beam array as the logical (*batch, ...) array."""
        if not self.is_shared:
            return stored
        return np.broadcast_to(stored.reshape(stored.shape[stored.ndim - tail:]), (*self._batch, *stored.shape[stored.ndim - tail:]))

    # -- containers --------------------------------------------------------------------------
    @property
    def batch_shape(self):
        return tuple(self._particles.shape[:-2]) if self._batch is None else self._batch

    @property
    def particles(self):
        """The (*batch, N, 7) array: a `DeviceArray` while it lives in HBM only."""
        if self.is_shared:
            return self._full(self._particles.host(), 2)
        return self._particles._host if self._particles._host is not None else self._particles._dev

    @property
    def energy(self) -> np.ndarray:
        return self._energy.host()

    @property
    def particle_charges(self) -> np.ndarray:
        if self._charges is None:  # particle_beam.py:41-43: zeros
            return np.broadcast_to(np.zeros((), dtype=self.dtype), (*self.batch_shape, self.num_particles))
        return self._full(self._charges, 1)

    @property
    def total_charge(self) -> np.ndarray:
        return np.sum(self.particle_charges, axis=-1)

    @property
    def num_particles(self) -> int:
        return int(self._particles.shape[-2])

    def __len__(self) -> int:
        return self.num_particles

    # -- constructors ------------------------------------------------------------------------
    @classmethod
    def from_parameters(cls, num_particles=None, mu_x=None, mu_y=None, mu_xp=None, mu_yp=None, sigma_x=None,
                        sigma_y=None, sigma_xp=None, sigma_yp=None, sigma_s=None, sigma_p=None, cor_x=None,
                        cor_y=None, cor_s=None, energy=None, total_charge=None, device=None,
                        dtype=np.float32, seed=None) -> "ParticleBeam":
        """
        Random 6-D Gaussian beam (particle_beam.py:47-178).  `seed` makes it reproducible
        (the reference draws from torch's global RNG).
        """
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(mu_x=mu_x, mu_xp=mu_xp, mu_y=mu_y, mu_yp=mu_yp, sigma_x=sigma_x,
                                    sigma_xp=sigma_xp, sigma_y=sigma_y, sigma_yp=sigma_yp, sigma_s=sigma_s,
                                    sigma_p=sigma_p, cor_x=cor_x, cor_y=cor_y, cor_s=cor_s, energy=energy,
                                    total_charge=total_charge), dtype)
        d = lambda k, v: g.get(k, np.full(shape, v, dtype=dtype))  # noqa: E731
        n = int(np.asarray(num_particles).reshape(-1)[0]) if num_particles is not None else 100_000
        total_charge = d("total_charge", 0.0)
        mean = np.stack([d("mu_x", 0.0), d("mu_xp", 0.0), d("mu_y", 0.0), d("mu_yp", 0.0),
                         np.zeros(shape, dtype), np.zeros(shape, dtype)], axis=-1)
        cov = np.zeros((*shape, 6, 6), dtype=dtype)
        cov[..., 0, 0] = d("sigma_x", 175e-9) ** 2
        cov[..., 0, 1] = cov[..., 1, 0] = d("cor_x", 0.0)
        cov[..., 1, 1] = d("sigma_xp", 2e-7) ** 2
        cov[..., 2, 2] = d("sigma_y", 175e-9) ** 2
        cov[..., 2, 3] = cov[..., 3, 2] = d("cor_y", 0.0)
        cov[..., 3, 3] = d("sigma_yp", 2e-7) ** 2
        cov[..., 4, 4] = d("sigma_s", 1e-6) ** 2
        cov[..., 4, 5] = cov[..., 5, 4] = d("cor_s", 0.0)
        cov[..., 5, 5] = d("sigma_p", 1e-6) ** 2

        rng = np.random.default_rng(seed)
        particles = np.ones((*shape, n, 7), dtype=dtype)
        flat = particles.reshape(-1, n, 7)
        for i, (m, c) in enumerate(zip(mean.reshape(-1, 6).astype(np.float64),
                                       cov.reshape(-1, 6, 6).astype(np.float64))):
            # x = m + L z with L L^T = c (block structure keeps this exact for cor = 0)
            w, v = np.linalg.eigh(c)
            L = v * np.sqrt(np.maximum(w, 0.0))
            flat[i, :, :6] = (m + rng.standard_normal((n, 6)) @ L.T).astype(dtype)
        charges = (np.ones((*shape, n), dtype=dtype) * total_charge[..., None] / n
                   if np.any(total_charge != 0) else None)
        beam = cls(particles, d("energy", 1e8), particle_charges=charges, dtype=dtype)
        if charges is None:
            beam._zero_charge_shape = shape
        return beam

    @classmethod
    def from_twiss(cls, num_particles=None, beta_x=None, alpha_x=None, emittance_x=None, beta_y=None,
                   alpha_y=None, emittance_y=None, energy=None, sigma_s=None, sigma_p=None, cor_s=None,
                   total_charge=None, device=None, dtype=np.float32, seed=None) -> "ParticleBeam":
        """particle_beam.py:180-264."""
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(beta_x=beta_x, alpha_x=alpha_x, emittance_x=emittance_x, beta_y=beta_y,
                                    alpha_y=alpha_y, emittance_y=emittance_y, energy=energy, sigma_s=sigma_s,
                                    sigma_p=sigma_p, cor_s=cor_s, total_charge=total_charge), dtype)
        d = lambda k, v: g.get(k, np.full(shape, v, dtype=dtype))  # noqa: E731
        n = num_particles if num_particles is not None else 1_000_000
        beta_x, alpha_x, emittance_x = d("beta_x", 0.0), d("alpha_x", 0.0), d("emittance_x", 0.0)
        beta_y, alpha_y, emittance_y = d("beta_y", 0.0), d("alpha_y", 0.0), d("emittance_y", 0.0)
        with np.errstate(all="ignore"):
            sigma_x = np.sqrt(beta_x * emittance_x)
            sigma_xp = np.sqrt(emittance_x * (1 + alpha_x**2) / beta_x)
            sigma_y = np.sqrt(beta_y * emittance_y)
            sigma_yp = np.sqrt(emittance_y * (1 + alpha_y**2) / beta_y)
        z = np.full(shape, 0.0, dtype)
        return cls.from_parameters(num_particles=n, mu_x=z, mu_xp=z, mu_y=z, mu_yp=z, sigma_x=sigma_x,
                                   sigma_xp=sigma_xp, sigma_y=sigma_y, sigma_yp=sigma_yp,
                                   sigma_s=d("sigma_s", 1e-6), sigma_p=d("sigma_p", 1e-6),
                                   energy=d("energy", 1e8), cor_s=d("cor_s", 0.0),
                                   cor_x=-emittance_x * alpha_x, cor_y=-emittance_y * alpha_y,
                                   total_charge=d("total_charge", 0.0), dtype=dtype, seed=seed)

    @classmethod
    def uniform_3d_ellipsoid(cls, num_particles=None, radius_x=None, radius_y=None, radius_s=None, sigma_xp=None,
                             sigma_yp=None, sigma_p=None, energy=None, total_charge=None, device=None,
                             dtype=np.float32, seed=None) -> "ParticleBeam":
        """
        Particles uniformly distributed inside an ellipsoid in (x, y, s) (waterbag), Gaussian in
        the momenta (particle_beam.py:266-385; rejection sampling as there).
        """
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(radius_x=radius_x, radius_y=radius_y, radius_s=radius_s, sigma_xp=sigma_xp,
                                    sigma_yp=sigma_yp, sigma_p=sigma_p, energy=energy, total_charge=total_charge),
                               dtype)
        n = int(num_particles) if num_particles is not None else 1_000_000
        rx, ry, rs = (g.get(k, np.full(shape, 1e-3, dtype)) for k in ("radius_x", "radius_y", "radius_s"))
        rng = np.random.default_rng(seed)
        xyz = np.empty((int(np.prod(shape)), n, 3), dtype=dtype)
        for i, (a, b, c) in enumerate(zip(rx.reshape(-1), ry.reshape(-1), rs.reshape(-1))):
            filled = 0
            while filled < n:
                cand = (rng.random((n, 3)) - 0.5) * 2 * np.array([a, b, c])
                inside = cand[(cand[:, 0] / a) ** 2 + (cand[:, 1] / b) ** 2 + (cand[:, 2] / c) ** 2 < 1]
                take = min(n - filled, len(inside))
                xyz[i, filled:filled + take] = inside[:take]
                filled += take
        beam = cls.from_parameters(num_particles=n, mu_xp=np.full(shape, 0.0, dtype), mu_yp=np.full(shape, 0.0, dtype),
                                   sigma_xp=g.get("sigma_xp"), sigma_yp=g.get("sigma_yp"), sigma_p=g.get("sigma_p"),
                                   energy=g.get("energy"), total_charge=g.get("total_charge"), dtype=dtype, seed=seed)
        host = np.array(beam._particles.host())
        xyz = xyz.reshape(*shape, n, 3)
        host[..., 0], host[..., 2], host[..., 4] = xyz[..., 0], xyz[..., 1], xyz[..., 2]
        beam._particles = Dual(host, owned=True)
        return beam

    @classmethod
    def make_linspaced(cls, num_particles=None, mu_x=None, mu_y=None, mu_xp=None, mu_yp=None, sigma_x=None,
                       sigma_y=None, sigma_xp=None, sigma_yp=None, sigma_s=None, sigma_p=None, energy=None,
                       total_charge=None, device=None, dtype=np.float32) -> "ParticleBeam":
        """Beam of *n* linspaced particles (particle_beam.py:387-543)."""
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(mu_x=mu_x, mu_xp=mu_xp, mu_y=mu_y, mu_yp=mu_yp, sigma_x=sigma_x,
                                    sigma_xp=sigma_xp, sigma_y=sigma_y, sigma_yp=sigma_yp, sigma_s=sigma_s,
                                    sigma_p=sigma_p, energy=energy, total_charge=total_charge), dtype)
        d = lambda k, v: g.get(k, np.full(shape, v, dtype=dtype))  # noqa: E731
        n = int(num_particles) if num_particles is not None else 10
        total_charge = d("total_charge", 0.0)
        lo_hi = [(d("mu_x", 0.0), d("sigma_x", 175e-9)), (d("mu_xp", 0.0), d("sigma_xp", 2e-7)),
                 (d("mu_y", 0.0), d("sigma_y", 175e-9)), (d("mu_yp", 0.0), d("sigma_yp", 2e-7)),
                 (np.zeros(shape, dtype), d("sigma_s", 0.0)), (np.zeros(shape, dtype), d("sigma_p", 0.0))]
        particles = np.ones((shape[0], n, 7), dtype=dtype)
        for c, (m, s) in enumerate(lo_hi):
            particles[:, :, c] = np.stack([np.linspace(mi - si, mi + si, n, dtype=dtype)
                                           for mi, si in zip(m, s)], axis=0)
        charges = np.ones((shape[0], n), dtype=dtype) * total_charge.reshape(-1, 1) / n
        return cls(particles=particles, energy=d("energy", 1e8), particle_charges=charges, dtype=dtype)

    @classmethod
    def from_astra(cls, path: str, device=None, dtype=np.float32) -> "ParticleBeam":
        """Load an ASTRA particle distribution (particle_beam.py:563-578)."""
        from ..io.astra import read_astra

        particles, energy, charges = read_astra(path)
        p7 = np.ones((1, particles.shape[0], 7))
        p7[0, :, :6] = particles
        return cls(particles=p7, energy=np.array([energy]), particle_charges=charges[None, :], dtype=dtype)

    @classmethod
    def synthetic(cls, batch_shape, num_particles: int, mu=None, sigma=None, energy=1e8, seed: int = 0,
                  dtype=np.float32) -> "ParticleBeam":
        """
        Seeded uncorrelated 6-D Gaussian beam generated directly in HBM (`lynx_fill_gaussian`).
        Not in the reference: it is how large benchmark beams are made without a host copy.
        """
        dtype = np.dtype(dtype)
        rt = get_runtime()
        batch_shape = tuple(batch_shape)
        B = int(np.prod(batch_shape, dtype=np.int64))
        mu = np.zeros(6) if mu is None else np.asarray(mu, dtype=np.float64)
        sigma = (np.array([175e-9, 2e-7, 175e-9, 2e-7, 1e-6, 1e-6]) if sigma is None
                 else np.asarray(sigma, dtype=np.float64))
        arr = rt.empty((*batch_shape, num_particles, 7), dtype)
        dbl = C.c_double * 6
        rt.check(rt.lib.lynx_fill_gaussian(rt.ctx, dtype_code(dtype), B, num_particles, dbl(*mu), dbl(*sigma),
                                           C.c_uint64(seed), C.c_void_p(arr.ptr)))
        out = cls.__new__(cls)
        out._init_raw(Dual(dev=arr), Dual(np.full(batch_shape, energy, dtype=dtype)), None, dtype)
        return out

    def transformed_to(self, mu_x=None, mu_y=None, mu_xp=None, mu_yp=None, sigma_x=None, sigma_y=None,
                       sigma_xp=None, sigma_yp=None, sigma_s=None, sigma_p=None, energy=None,
                       total_charge=None, device=None, dtype=None) -> "ParticleBeam":
        """Affine rescale of the particles to new moments (particle_beam.py:580-715)."""
        dtype = np.dtype(dtype) if dtype is not None else self.dtype
        names = ["x", "xp", "y", "yp", "s", "p"]
        given = dict(mu_x=mu_x, mu_xp=mu_xp, mu_y=mu_y, mu_yp=mu_yp, sigma_x=sigma_x, sigma_xp=sigma_xp,
                     sigma_y=sigma_y, sigma_yp=sigma_yp, sigma_s=sigma_s, sigma_p=sigma_p, energy=energy,
                     total_charge=total_charge)
        shape = self.mu_x.shape
        assert all(np.asarray(v).shape == shape for v in given.values() if v is not None), (
            "Arguments must have the same shape.")
        pick = lambda k: np.asarray(given[k], dtype) if given[k] is not None else getattr(self, k)  # noqa: E731
        zeros = np.zeros(shape, dtype)
        new_mu = np.stack([pick("mu_x"), pick("mu_xp"), pick("mu_y"), pick("mu_yp"), zeros, zeros], axis=-1)
        new_sigma = np.stack([pick("sigma_" + n) for n in names], axis=-1)
        old_mu = np.stack([self.mu_x, self.mu_xp, self.mu_y, self.mu_yp, zeros, zeros], axis=-1)
        old_sigma = np.stack([getattr(self, "sigma_" + n) for n in names], axis=-1)
        host = np.asarray(self.particles)
        phase_space = (host[..., :6] - old_mu[..., None, :]) / old_sigma[..., None, :] * new_sigma[..., None, :] \
            + new_mu[..., None, :]
        particles = np.ones_like(host)
        particles[..., :6] = phase_space
        if total_charge is None:
            charges = self._charges
        else:
            charges = (np.ones(host.shape[:-1], dtype=dtype)
                       * np.asarray(total_charge, dtype)[..., None] / host.shape[-2])
        return self.__class__(particles=particles, energy=pick("energy"), particle_charges=charges, dtype=dtype)

    def broadcast(self, shape: tuple) -> "ParticleBeam":
        """
        particle_beam.py:838-843 repeats the particles `shape` times (`Tensor.repeat`).  A single
        beam (storage batch of size 1) is broadcast lazily instead -- same logical shapes and
        values, `is_shared` is True and nothing is copied; `materialized()` gives the repeated
        array.  `config.lazy_broadcast = False` restores the physical repeat.
        """
        stored = self._particles.shape[:-2]
        if config.lazy_broadcast and not self.is_shared and int(np.prod(stored, dtype=np.int64)) == 1:
            d = max(len(stored), len(shape))
            batch = tuple(a * b for a, b in zip((1,) * (d - len(stored)) + tuple(stored),
                                                (1,) * (d - len(shape)) + tuple(shape)))
            out = self.__class__.__new__(self.__class__)
            out._init_raw(self._particles, Dual(np.tile(self.energy, shape)), self._charges, self.dtype, self._moments, batch)
            return out
        if self.is_shared:
            return self.materialized().broadcast(shape)
        return self.__class__(
            particles=np.tile(self._particles.host(), (*shape, 1, 1)), energy=np.tile(self.energy, shape),
            particle_charges=None if self._charges is None else np.tile(self._charges, (*shape, 1)),
            dtype=self.dtype)

    # -- coordinates -------------------------------------------------------------------------
    def _coordinate(self, c: int):
        return np.asarray(self.particles)[..., c]

    def _set_coordinate(self, c: int, value):
        host = np.array(self.particles)  # a shared beam becomes a physical one when written to
        host[..., c] = value
        self._particles = Dual(host, owned=True)
        self._batch = None
        if self._charges is not None:
            self._charges = np.ascontiguousarray(np.broadcast_to(self._charges, host.shape[:-1]))
        self._moments = None

    xs = property(lambda self: self._coordinate(0), lambda self, v: self._set_coordinate(0, v))
    xps = property(lambda self: self._coordinate(1), lambda self, v: self._set_coordinate(1, v))
    ys = property(lambda self: self._coordinate(2), lambda self, v: self._set_coordinate(2, v))
    yps = property(lambda self: self._coordinate(3), lambda self, v: self._set_coordinate(3, v))
    ss = property(lambda self: self._coordinate(4), lambda self, v: self._set_coordinate(4, v))
    ps = property(lambda self: self._coordinate(5), lambda self, v: self._set_coordinate(5, v))

    # -- moments: one fused GPU pass (particle_beam.py:736-836) ---------------------------------
    def moment_record(self, covariance: bool = False) -> np.ndarray:
        """
        (*batch, 36) float64 record, layout in include/lynx_hip.h.  By default it holds what the
        reference's properties need (means, the six variances, cov(x, x'), cov(y, y'); the other
        covariance entries are NaN); `covariance=True` makes sure the whole 6x6 triangle is there,
        at the price of one more pass over the beam if it was not accumulated before.
        """
        have = self._moments
        if have is not None and covariance and not np.all(have.host()[..., 34] == 1.0):
            have = None
        if have is None:
            rt = get_runtime()
            p = self._particles.device(rt)
            stored = tuple(self._particles.shape[:-2])  # a shared beam has one stored sample
            B = int(np.prod(stored, dtype=np.int64))
            rec = rt.empty((*stored, _ffi.MOMENT_STRIDE), np.float64)
            rt.check(rt.lib.lynx_moments(rt.ctx, dtype_code(self.dtype), B, self.num_particles,
                                         C.c_void_p(p.ptr), C.c_void_p(rec.ptr), 1 if covariance else 0))
            self._moments = Dual(dev=rec)
        return self._full(self._moments.host(), 1)

    def covariance(self) -> np.ndarray:
        """Biased 6x6 covariance matrix of the particles, (*batch, 6, 6) float64 (one pass on the GPU)."""
        rec = self.moment_record(covariance=True)
        out = np.empty((*rec.shape[:-1], 6, 6))
        for i in range(6):
            for j in range(i, 6):
                out[..., i, j] = out[..., j, i] = rec[..., _tri(i, j)]
        return out

    def _mean(self, c: int) -> np.ndarray:
        return self.moment_record()[..., c].astype(self.dtype)

    def _std(self, c: int) -> np.ndarray:
        rec = self.moment_record()
        n = rec[..., 35]
        ddof = config.std_ddof
        with np.errstate(all="ignore"):
            return np.sqrt(rec[..., _tri(c, c)] * (n / (n - ddof))).astype(self.dtype)

    def _cov(self, i: int, j: int) -> np.ndarray:
        return self.moment_record()[..., _tri(i, j)].astype(self.dtype)

    mu_x = property(lambda self: self._mean(0))
    mu_xp = property(lambda self: self._mean(1))
    mu_y = property(lambda self: self._mean(2))
    mu_yp = property(lambda self: self._mean(3))
    mu_s = property(lambda self: self._mean(4))
    mu_p = property(lambda self: self._mean(5))
    sigma_x = property(lambda self: self._std(0))
    sigma_xp = property(lambda self: self._std(1))
    sigma_y = property(lambda self: self._std(2))
    sigma_yp = property(lambda self: self._std(3))
    sigma_s = property(lambda self: self._std(4))
    sigma_p = property(lambda self: self._std(5))
    # biased mean of centred products (particle_beam.py:825-836)
    sigma_xxp = property(lambda self: self._cov(0, 1))
    sigma_yyp = property(lambda self: self._cov(2, 3))

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(n={self.num_particles}, mu_x={self.mu_x!r}, mu_xp={self.mu_xp!r},"
                f" mu_y={self.mu_y!r}, mu_yp={self.mu_yp!r}, sigma_x={self.sigma_x!r},"
                f" sigma_xp={self.sigma_xp!r}, sigma_y={self.sigma_y!r}, sigma_yp={self.sigma_yp!r},"
                f" sigma_s={self.sigma_s!r}, sigma_p={self.sigma_p!r}, energy={self.energy!r})"
                f" total_charge={self.total_charge!r})")
