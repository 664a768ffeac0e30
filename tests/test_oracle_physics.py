"""
A second anchor for the oracle, independent of the reference's code: textbook linear optics.

SURVEY.md section 8c: the reference's own tests hold no absolute numbers for dipoles, correctors, tilts,
misalignments or the cavity matrix ("parity unpinned" rows A1, A2, A6, A7, A9).  The oracle follows the
reference line by line; these tests check the same maps against closed forms every accelerator-optics text
derives (sector bend, thick quadrupole, thin edge wedge, Rosenzweig-Serafini cavity), so that a misreading of
the reference would have to coincide with a physics error to go unnoticed.  float64, CPU only.
"""

import numpy as np
import pytest

from oracle import lynx_oracle as o

ENERGY = np.array([1.0e9, 6.0e6])  # ultra-relativistic and gamma ~ 12


def _gamma_beta(energy):
    gamma = energy / o.REST_ENERGY
    return gamma, np.sqrt(1.0 - 1.0 / gamma**2)


def _f(v):
    return np.full(ENERGY.shape, v, dtype=np.float64)


def test_drift_is_the_free_space_map():
    L = 1.7
    T = o.element_transfer_map(o.Drift(_f(L)), ENERGY, np.float64)
    gamma, beta = _gamma_beta(ENERGY)
    expect = np.tile(np.eye(7), (2, 1, 1))
    expect[:, 0, 1] = expect[:, 2, 3] = L
    expect[:, 4, 5] = -L / (beta**2 * gamma**2)  # path length vs. momentum: velocity term only
    assert np.allclose(T, expect, rtol=1e-13, atol=1e-16)


@pytest.mark.parametrize("k1", [4.2, -4.2])
def test_thick_quadrupole_closed_form(k1):
    L = 0.3
    T = o.element_transfer_map(o.Quadrupole(_f(L), k1=_f(k1)), ENERGY, np.float64)
    w = np.sqrt(abs(k1))
    foc = np.array([[np.cos(w * L), np.sin(w * L) / w], [-w * np.sin(w * L), np.cos(w * L)]])
    defoc = np.array([[np.cosh(w * L), np.sinh(w * L) / w], [w * np.sinh(w * L), np.cosh(w * L)]])
    xb, yb = (foc, defoc) if k1 > 0 else (defoc, foc)
    for b in range(2):
        assert np.allclose(T[b, :2, :2], xb, rtol=1e-12)
        assert np.allclose(T[b, 2:4, 2:4], yb, rtol=1e-12)
        assert np.isclose(np.linalg.det(T[b, :2, :2]), 1.0, rtol=1e-12)
        assert np.isclose(np.linalg.det(T[b, 2:4, 2:4]), 1.0, rtol=1e-12)
        assert np.allclose(T[b, :4, 4:6], 0.0) and np.allclose(T[b, :4, 6], 0.0)  # no dispersion, no kick


def test_sector_bend_closed_form():
    L, angle = 0.8, 0.25
    rho = L / angle
    T = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle)), ENERGY, np.float64)
    _, beta = _gamma_beta(ENERGY)
    c, s = np.cos(angle), np.sin(angle)
    for b in range(2):
        assert np.allclose(T[b, :2, :2], [[c, rho * s], [-s / rho, c]], rtol=1e-12)
        # a drift vertically -- up to the reference's k1 = 0 -> 1e-12 substitution (track_methods.py:67-68)
        assert np.allclose(T[b, 2:4, 2:4], [[1.0, L], [0.0, 1.0]], rtol=1e-12, atol=2e-12)
        # dispersion of a sector magnet: D = rho (1 - cos), D' = sin, per unit delta = dp/p / beta convention
        assert np.isclose(T[b, 0, 5], rho * (1 - c) / beta[b], rtol=1e-12)
        assert np.isclose(T[b, 1, 5], s / beta[b], rtol=1e-12)
        # symplectic partner terms of the path length
        assert np.isclose(T[b, 4, 0], s / beta[b], rtol=1e-12)
        assert np.isclose(T[b, 4, 1], rho * (1 - c) / beta[b], rtol=1e-12)


def test_dipole_edges_are_thin_wedges():
    L, angle, e1, e2 = 0.8, 0.25, 0.11, -0.07
    h = angle / L
    body = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle)), ENERGY, np.float64)
    full = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle), e1=_f(e1), e2=_f(e2)), ENERGY, np.float64)

    def wedge(e):  # horizontally defocusing by h tan e, vertically focusing by the same (no fringe integral)
        W = np.eye(7)
        W[1, 0] = h * np.tan(e)
        W[3, 2] = -h * np.tan(e)
        return W

    for b in range(2):
        assert np.allclose(full[b], wedge(e2) @ body[b] @ wedge(e1), rtol=1e-12, atol=1e-15)
        assert np.isclose(np.linalg.det(full[b, :2, :2]), 1.0, rtol=1e-12)
        assert np.isclose(np.linalg.det(full[b, 2:4, 2:4]), 1.0, rtol=1e-12)


def test_fringe_field_reduces_the_vertical_edge_angle():
    L, angle, e1, fint, gap = 0.8, 0.25, 0.11, 0.45, 0.03
    h = angle / L
    T0 = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle), e1=_f(e1)), ENERGY, np.float64)
    T1 = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle), e1=_f(e1), fringe_integral=_f(fint),
                                         fringe_integral_exit=_f(0.0), gap=_f(gap)), ENERGY, np.float64)
    # SLAC-75 / MAD: psi = K g h (1 + sin^2 e) / cos e ; vertical kick -h tan(e - psi); horizontal unchanged
    psi = fint * gap * h * (1 + np.sin(e1) ** 2) / np.cos(e1)
    W0, W1 = np.eye(7), np.eye(7)
    W0[1, 0] = W1[1, 0] = h * np.tan(e1)
    W0[3, 2] = -h * np.tan(e1)
    W1[3, 2] = -h * np.tan(e1 - psi)
    for b in range(2):
        assert np.allclose(T1[b], T0[b] @ np.linalg.inv(W0) @ W1, rtol=1e-11, atol=1e-14)


def test_rbend_is_a_sector_bend_with_half_angle_wedges():
    L, angle = 0.8, 0.25
    r = o.element_transfer_map(o.RBend(_f(L), angle=_f(angle)), ENERGY, np.float64)
    s = o.element_transfer_map(o.Dipole(_f(L), angle=_f(angle), e1=_f(angle / 2), e2=_f(angle / 2)), ENERGY, np.float64)
    assert np.array_equal(r, s)
    # the textbook property of a rectangular magnet: no horizontal focusing at all
    rho = L / angle
    for b in range(2):
        assert np.allclose(r[b, :2, :2], [[1.0, rho * np.sin(angle)], [0.0, 1.0]], rtol=1e-12, atol=5e-12)  # cancellation


def test_correctors_are_drifts_with_a_kick():
    L, angle = 0.1, 3e-4
    d = o.element_transfer_map(o.Drift(_f(L)), ENERGY, np.float64)
    h = o.element_transfer_map(o.HorizontalCorrector(_f(L), angle=_f(angle)), ENERGY, np.float64)
    v = o.element_transfer_map(o.VerticalCorrector(_f(L), angle=_f(angle)), ENERGY, np.float64)
    eh, ev = d.copy(), d.copy()
    eh[:, 1, 6] = angle
    ev[:, 3, 6] = angle
    assert np.array_equal(h, eh) and np.array_equal(v, ev)


def test_tilt_is_a_rotation_of_the_transverse_plane():
    L, k1, tilt = 0.3, 4.2, 0.3
    plain = o.element_transfer_map(o.Quadrupole(_f(L), k1=_f(k1)), ENERGY, np.float64)
    tilted = o.element_transfer_map(o.Quadrupole(_f(L), k1=_f(k1), tilt=_f(tilt)), ENERGY, np.float64)
    c, s = np.cos(tilt), np.sin(tilt)
    R = np.eye(7)  # (x, y) -> (x cos + y sin, -x sin + y cos), same for the slopes
    R[0, 0] = R[1, 1] = R[2, 2] = R[3, 3] = c
    R[0, 2] = R[1, 3] = s
    R[2, 0] = R[3, 1] = -s
    for b in range(2):
        assert np.allclose(tilted[b], R.T @ plain[b] @ R, rtol=1e-12, atol=1e-15)
    # a quadrupole rolled by 45 degrees is a skew quadrupole: no x-x focusing term of first order in k1 L
    skew = o.element_transfer_map(o.Quadrupole(_f(1e-3), k1=_f(k1), tilt=_f(np.pi / 4)), ENERGY, np.float64)
    assert np.all(np.abs(skew[:, 1, 0]) < 1e-8) and np.allclose(skew[:, 1, 2], -k1 * 1e-3, rtol=1e-5)


def test_misalignment_shifts_the_axis():
    L, k1 = 0.3, 4.2
    dx, dy = 2e-4, -1e-4
    mis = np.tile(np.array([dx, dy]), (2, 1))
    T = o.element_transfer_map(o.Quadrupole(_f(L), k1=_f(k1), misalignment=mis), ENERGY, np.float64)
    plain = o.element_transfer_map(o.Quadrupole(_f(L), k1=_f(k1)), ENERGY, np.float64)
    # a particle on the magnet's own axis, parallel to it, stays there
    z = np.array([dx, 0.0, dy, 0.0, 0.0, 0.0, 1.0])
    for b in range(2):
        out = T[b] @ z
        assert np.allclose(out[:4], [dx, 0.0, dy, 0.0], atol=1e-18)
        # and any other particle sees the plain magnet in shifted coordinates
        p = np.array([1e-3, 2e-4, -5e-4, 1e-4, 0.0, 0.0, 1.0])
        shifted = p - np.array([dx, 0, dy, 0, 0, 0, 0])
        assert np.allclose((T[b] @ p)[:4], (plain[b] @ shifted)[:4] + [dx, 0, dy, 0], rtol=1e-12, atol=1e-18)


def test_cavity_matrix_damps_adiabatically():
    """Rosenzweig-Serafini standing-wave cavity: the transverse 2x2 block has determinant E_i / E_f
    (adiabatic damping of the normalised emittance), equal in both planes; on crest the energy gain is V."""
    L, V, f = 1.0377, 2.0e7, 1.3e9
    for phase in (0.0, 12.0, -30.0):
        spec = o.Cavity(_f(L), voltage=_f(V), phase=_f(phase), frequency=_f(f))
        energy = np.array([1.0e8, 6.0e6])
        T = o.element_transfer_map(spec, energy, np.float64)
        e_out = energy + V * np.cos(np.deg2rad(phase))
        for b in range(2):
            assert np.isclose(np.linalg.det(T[b, :2, :2]), energy[b] / e_out[b], rtol=1e-10)
            assert np.array_equal(T[b, :2, :2], T[b, 2:4, 2:4])
            assert np.allclose(T[b, :4, 4:], 0.0) and np.allclose(T[b, 4:6, :4], 0.0)  # s, delta couple only with each other
        beam = o.particle_beam(np.zeros((2, 4, 7)) + np.array([0, 0, 0, 0, 0, 0, 1.0]), energy, np.float64)
        out = o.element_track(spec, beam, np.float64)
        assert np.allclose(out["energy"], e_out, rtol=1e-14)
        # the reference particle (s = 0, delta = 0) stays the reference particle
        assert np.allclose(out["particles"][..., :6], 0.0, atol=1e-18)


def test_cavity_off_crest_chirps_the_bunch():
    """A particle ahead of (behind) the reference sees a different phase: to first order
    delta_out - delta_in E_i/E_f ~ -(V k sin(phi) / E_f) s for a relativistic beam -- sign and size of the chirp."""
    L, V, f, phase = 1.0377, 2.0e7, 1.3e9, -20.0
    energy = np.array([1.0e9])
    spec = o.Cavity(np.array([L]), voltage=np.array([V]), phase=np.array([phase]), frequency=np.array([f]))
    s0 = 1e-5
    P = np.zeros((1, 2, 7))
    P[..., 6] = 1.0
    P[0, 1, 4] = s0
    out = o.element_track(spec, o.particle_beam(P, energy, np.float64), np.float64)
    k = 2 * np.pi * f / 299792458.0
    e_out = energy[0] + V * np.cos(np.deg2rad(phase))
    # cos(-s k + phi) - cos(phi) ~ s k sin(phi): with phi < 0 a particle at s > 0 loses energy relative to the reference
    expect = V * s0 * k * np.sin(np.deg2rad(phase)) / e_out
    got = out["particles"][0, 1, 5] - out["particles"][0, 0, 5]
    assert np.isclose(got, expect, rtol=1e-3), (got, expect)
