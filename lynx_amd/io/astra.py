"""
ASTRA particle-distribution reader (lynx/converters/astra.py:8-62, itself after Ocelot's
astra2ocelot).  Pure NumPy, host side: it prepares the arrays a beam is constructed from.

ASTRA columns: x y z px py pz clock charge[nC] index status; the first row is the reference
particle, all others are relative to it in z and pz; momenta in eV/c.
"""

from __future__ import annotations

import numpy as np

ELECTRON_MASS_EV = 510998.95069  # converters/astra.py:5


def read_astra(path: str):
    """
    :return: (particles (N, 6) in (x, x', y, y', s, delta), reference energy in eV,
        macro-particle charges in C)
    """
    table = np.loadtxt(path)
    table = table[table[:, 9] > 0]  # drop lost particles (status <= 0)
    pz_ref = table[0, 5]
    x, y, z = table[:, 0].copy(), table[:, 1].copy(), table[:, 2].copy()
    px, py, dpz = table[:, 3], table[:, 4], table[:, 5].copy()
    z[0] = 0.0    # the reference particle defines the origin in z ...
    dpz[0] = 0.0  # ... and in pz

    gamma_ref = np.sqrt((pz_ref / ELECTRON_MASS_EV) ** 2 + 1.0)
    energy = gamma_ref * ELECTRON_MASS_EV
    beta_ref = np.sqrt(1.0 - gamma_ref**-2)

    momentum = np.stack([px, py, dpz + pz_ref], axis=1)
    gamma = np.sqrt(1.0 + np.sum(momentum * momentum, axis=1) / ELECTRON_MASS_EV**2)
    beta = np.sqrt(1.0 - gamma**-2)
    direction = momentum / np.linalg.norm(momentum, axis=1, keepdims=True)

    # project every particle onto the reference plane z = 0 along its own direction
    cdt = -z / (beta * direction[:, 2])
    particles = np.zeros((table.shape[0], 6))
    particles[:, 0] = x + beta * direction[:, 0] * cdt
    particles[:, 1] = px / pz_ref
    particles[:, 2] = y + beta * direction[:, 1] * cdt
    particles[:, 3] = py / pz_ref
    particles[:, 4] = cdt
    particles[:, 5] = (gamma / gamma_ref - 1.0) / beta_ref
    charges = np.abs(table[:, 7]) * 1e-9  # nC -> C
    return particles, energy, charges
