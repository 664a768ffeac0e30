"""`ParameterBeam`: mu (…,7) + cov (…,7,7) (lynx/particles/parameter_beam.py:13-444)."""

from __future__ import annotations

import numpy as np

from ..device import Dual
from .beam import Beam, _batch_args


class ParameterBeam(Beam):
    """
    Beam of charged particles described by its first and second moments.

    :param mu: Mu vector of the beam, shape (*batch, 7).
    :param cov: Covariance matrix of the beam, shape (*batch, 7, 7).
    :param energy: Energy of the beam in eV.
    :param total_charge: Total charge of the beam in C.
    """

    def __init__(self, mu, cov, energy, total_charge=None, device=None, dtype=np.float32) -> None:
        dtype = np.dtype(dtype)
        total_charge = total_charge if total_charge is not None else np.asarray([0.0], dtype=dtype)
        self._init_raw(Dual(np.asarray(mu, dtype=dtype)), Dual(np.asarray(cov, dtype=dtype)),
                       Dual(np.asarray(energy, dtype=dtype)), np.asarray(total_charge, dtype=dtype), dtype)

    def _init_raw(self, mu: Dual, cov: Dual, energy: Dual, total_charge, dtype):
        self._mu_d, self._cov_d, self._energy = mu, cov, energy
        self.total_charge = total_charge
        self.dtype = np.dtype(dtype)

    def _shallow_copy(self):
        out = ParameterBeam.__new__(ParameterBeam)
        out._init_raw(self._mu_d, self._cov_d, self._energy, self.total_charge, self.dtype)
        return out

    @property
    def batch_shape(self):
        return tuple(self._mu_d.shape[:-1])

    @property
    def _mu(self) -> np.ndarray:
        return self._mu_d.host()

    @property
    def _cov(self) -> np.ndarray:
        return self._cov_d.host()

    @property
    def energy(self) -> np.ndarray:
        return self._energy.host()

    @classmethod
    def from_parameters(cls, mu_x=None, mu_xp=None, mu_y=None, mu_yp=None, sigma_x=None, sigma_xp=None,
                        sigma_y=None, sigma_yp=None, sigma_s=None, sigma_p=None, cor_x=None, cor_y=None,
                        cor_s=None, energy=None, total_charge=None, device=None,
                        dtype=np.float32) -> "ParameterBeam":
        """parameter_beam.py:47-144."""
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(mu_x=mu_x, mu_xp=mu_xp, mu_y=mu_y, mu_yp=mu_yp, sigma_x=sigma_x,
                                    sigma_xp=sigma_xp, sigma_y=sigma_y, sigma_yp=sigma_yp, sigma_s=sigma_s,
                                    sigma_p=sigma_p, cor_x=cor_x, cor_y=cor_y, cor_s=cor_s, energy=energy,
                                    total_charge=total_charge), dtype)
        d = lambda k, v: g.get(k, np.full(shape, v, dtype=dtype))  # noqa: E731
        mu = np.stack([d("mu_x", 0.0), d("mu_xp", 0.0), d("mu_y", 0.0), d("mu_yp", 0.0),
                       np.full(shape, 0.0, dtype), np.full(shape, 0.0, dtype), np.full(shape, 1.0, dtype)],
                      axis=-1)
        cov = np.zeros((*shape, 7, 7), dtype=dtype)
        cov[..., 0, 0] = d("sigma_x", 175e-9) ** 2
        cov[..., 0, 1] = cov[..., 1, 0] = d("cor_x", 0.0)
        cov[..., 1, 1] = d("sigma_xp", 2e-7) ** 2
        cov[..., 2, 2] = d("sigma_y", 175e-9) ** 2
        cov[..., 2, 3] = cov[..., 3, 2] = d("cor_y", 0.0)
        cov[..., 3, 3] = d("sigma_yp", 2e-7) ** 2
        cov[..., 4, 4] = d("sigma_s", 1e-6) ** 2
        cov[..., 4, 5] = cov[..., 5, 4] = d("cor_s", 0.0)
        cov[..., 5, 5] = d("sigma_p", 1e-6) ** 2
        return cls(mu=mu, cov=cov, energy=d("energy", 1e8), total_charge=d("total_charge", 0.0), dtype=dtype)

    @classmethod
    def from_twiss(cls, beta_x=None, alpha_x=None, emittance_x=None, beta_y=None, alpha_y=None,
                   emittance_y=None, sigma_s=None, sigma_p=None, cor_s=None, energy=None, total_charge=None,
                   device=None, dtype=np.float32) -> "ParameterBeam":
        """parameter_beam.py:146-232."""
        dtype = np.dtype(dtype)
        g, shape = _batch_args(dict(beta_x=beta_x, alpha_x=alpha_x, emittance_x=emittance_x, beta_y=beta_y,
                                    alpha_y=alpha_y, emittance_y=emittance_y, sigma_s=sigma_s, sigma_p=sigma_p,
                                    cor_s=cor_s, energy=energy, total_charge=total_charge), dtype)
        d = lambda k, v: g.get(k, np.full(shape, v, dtype=dtype))  # noqa: E731
        beta_x, alpha_x, emittance_x = d("beta_x", 1.0), d("alpha_x", 0.0), d("emittance_x", 7.1971891e-13)
        beta_y, alpha_y, emittance_y = d("beta_y", 1.0), d("alpha_y", 0.0), d("emittance_y", 7.1971891e-13)
        assert np.all(beta_x > 0), "Beta function in x direction must be larger than 0 everywhere."
        assert np.all(beta_y > 0), "Beta function in y direction must be larger than 0 everywhere."
        return cls.from_parameters(
            sigma_x=np.sqrt(emittance_x * beta_x), sigma_xp=np.sqrt(emittance_x * (1 + alpha_x**2) / beta_x),
            sigma_y=np.sqrt(emittance_y * beta_y), sigma_yp=np.sqrt(emittance_y * (1 + alpha_y**2) / beta_y),
            sigma_s=d("sigma_s", 1e-6), sigma_p=d("sigma_p", 1e-6), energy=d("energy", 1e8),
            cor_s=d("cor_s", 0.0), cor_x=-emittance_x * alpha_x, cor_y=-emittance_y * alpha_y,
            total_charge=d("total_charge", 0.0), dtype=dtype)

    @classmethod
    def from_astra(cls, path: str, device=None, dtype=np.float32) -> "ParameterBeam":
        """Moments of an ASTRA particle distribution (parameter_beam.py:255-276)."""
        from ..io.astra import read_astra

        particles, energy, charges = read_astra(path)
        mu = np.ones(7)
        mu[:6] = particles.mean(axis=0)
        cov = np.zeros((7, 7))
        cov[:6, :6] = np.cov(particles.T)
        return cls(mu=mu[None], cov=cov[None], energy=np.array([energy]), total_charge=np.array([charges.sum()]),
                   dtype=dtype)

    def transformed_to(self, mu_x=None, mu_xp=None, mu_y=None, mu_yp=None, sigma_x=None, sigma_xp=None,
                       sigma_y=None, sigma_yp=None, sigma_s=None, sigma_p=None, energy=None,
                       total_charge=None, device=None, dtype=None) -> "ParameterBeam":
        """parameter_beam.py:286-369."""
        dtype = np.dtype(dtype) if dtype is not None else self.dtype
        given = dict(mu_x=mu_x, mu_xp=mu_xp, mu_y=mu_y, mu_yp=mu_yp, sigma_x=sigma_x, sigma_xp=sigma_xp,
                     sigma_y=sigma_y, sigma_yp=sigma_yp, sigma_s=sigma_s, sigma_p=sigma_p, energy=energy,
                     total_charge=total_charge)
        shape = self.mu_x.shape
        assert all(np.asarray(v).shape == shape for v in given.values() if v is not None), (
            "Arguments must have the same shape.")
        args = {k: (v if v is not None else getattr(self, k)) for k, v in given.items()}
        return self.__class__.from_parameters(**args, dtype=dtype)

    def broadcast(self, shape: tuple) -> "ParameterBeam":
        """parameter_beam.py:427-433 (`Tensor.repeat`)."""
        return self.__class__(mu=np.tile(self._mu, (*shape, 1)), cov=np.tile(self._cov, (*shape, 1, 1)),
                              energy=np.tile(self.energy, shape), total_charge=np.tile(self.total_charge, shape),
                              dtype=self.dtype)

    def _sigma(self, i) -> np.ndarray:
        return np.sqrt(np.maximum(self._cov[..., i, i], self.dtype.type(1e-20)))  # parameter_beam.py:376

    mu_x = property(lambda self: self._mu[..., 0])
    mu_xp = property(lambda self: self._mu[..., 1])
    mu_y = property(lambda self: self._mu[..., 2])
    mu_yp = property(lambda self: self._mu[..., 3])
    mu_s = property(lambda self: self._mu[..., 4])
    mu_p = property(lambda self: self._mu[..., 5])
    sigma_x = property(lambda self: self._sigma(0))
    sigma_xp = property(lambda self: self._sigma(1))
    sigma_y = property(lambda self: self._sigma(2))
    sigma_yp = property(lambda self: self._sigma(3))
    sigma_s = property(lambda self: self._sigma(4))
    sigma_p = property(lambda self: self._sigma(5))
    sigma_xxp = property(lambda self: self._cov[..., 0, 1])
    sigma_yyp = property(lambda self: self._cov[..., 2, 3])
