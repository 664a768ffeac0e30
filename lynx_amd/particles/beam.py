"""Abstract beam and the derived Twiss properties (lynx/particles/beam.py:11-324)."""

from __future__ import annotations

import numpy as np

ELECTRON_MASS_EV = 510998.95069  # beam.py:8


class Beam:
    empty = "I'm an empty beam!"

    @property
    def parameters(self) -> dict:
        return {k: getattr(self, k) for k in ("mu_x", "mu_xp", "mu_y", "mu_yp", "sigma_x", "sigma_xp",
                                              "sigma_y", "sigma_yp", "sigma_s", "sigma_p", "energy")}

    @property
    def relativistic_gamma(self) -> np.ndarray:
        return self.energy / self.energy.dtype.type(ELECTRON_MASS_EV)  # beam.py:241-243

    @property
    def relativistic_beta(self) -> np.ndarray:
        gamma = self.relativistic_gamma
        beta = np.ones_like(gamma)  # beam.py:245-251
        pos = np.abs(gamma) > 0
        beta[pos] = np.sqrt(1 - 1 / (gamma[pos] ** 2))
        return beta

    def _emittance(self, sigma, sigma_p, cross) -> np.ndarray:
        tiny = np.finfo(sigma.dtype).tiny
        return np.sqrt(np.maximum(sigma**2 * sigma_p**2 - cross**2, tiny))

    @property
    def emittance_x(self) -> np.ndarray:
        """Emittance of the beam in x direction in m*rad (beam.py:262-270)."""
        return self._emittance(self.sigma_x, self.sigma_xp, self.sigma_xxp)

    @property
    def normalized_emittance_x(self) -> np.ndarray:
        return self.emittance_x * self.relativistic_beta * self.relativistic_gamma

    @property
    def beta_x(self) -> np.ndarray:
        return self.sigma_x**2 / self.emittance_x

    @property
    def alpha_x(self) -> np.ndarray:
        return -self.sigma_xxp / self.emittance_x

    @property
    def emittance_y(self) -> np.ndarray:
        """Emittance of the beam in y direction in m*rad (beam.py:287-295)."""
        return self._emittance(self.sigma_y, self.sigma_yp, self.sigma_yyp)

    @property
    def normalized_emittance_y(self) -> np.ndarray:
        return self.emittance_y * self.relativistic_beta * self.relativistic_gamma

    @property
    def beta_y(self) -> np.ndarray:
        return self.sigma_y**2 / self.emittance_y

    @property
    def alpha_y(self) -> np.ndarray:
        return -self.sigma_yyp / self.emittance_y

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(mu_x={self.mu_x}, mu_xp={self.mu_xp}, mu_y={self.mu_y},"
                f" mu_yp={self.mu_yp}, sigma_x={self.sigma_x}, sigma_xp={self.sigma_xp},"
                f" sigma_y={self.sigma_y}, sigma_yp={self.sigma_yp}, sigma_s={self.sigma_s},"
                f" sigma_p={self.sigma_p}, energy={self.energy}), total_charge={self.total_charge})")


def _batch_args(named: dict, dtype):
    """Shape rule of every beam constructor (e.g. parameter_beam.py:69-94)."""
    given = {k: np.asarray(v, dtype=dtype) for k, v in named.items() if v is not None}
    shape = next(iter(given.values())).shape if given else (1,)
    if len(given) > 1:
        assert all(v.shape == shape for v in given.values()), "Arguments must have the same shape."
    return given, shape
