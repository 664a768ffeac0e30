"""
Pins the oracle (oracle/lynx_oracle.py) against every known answer the reference holds for
this path.  These run on the CPU.
"""

import numpy as np
import pytest

from oracle import lynx_oracle as o


def test_kat1_parameter_beam_twiss_and_drift():
    """try_batched.ipynb: printed tensors (float32 torch, 5 significant digits)."""
    b = o.parameter_beam_from_twiss(
        dtype=np.float32, beta_x=[61.47503078, 99.0], alpha_x=[-1.21242463, -0.9],
        emittance_x=[7.1971891e-13, 5e-13], beta_y=[35.41897281, 60.0], alpha_y=[0.66554622, 0.5],
        emittance_y=[3.5866484e-15, 1e-15], energy=[150e6, 14.6e9])
    m = o.beam_moments(b)
    assert np.allclose(m["sigma_x"], [6.6517e-06, 7.0356e-06], rtol=1e-4)
    assert np.allclose(m["sigma_xxp"], [8.7260e-13, 4.5e-13], rtol=1e-4)
    assert np.allclose(m["relativistic_gamma"], [293.5427, 28571.4863], rtol=1e-6)
    # the rest of what the notebook prints for this beam (try_batched.ipynb cells 1, 14-41)
    printed = dict(sigma_xp=[1.7005e-07, 9.5611e-08], sigma_y=[3.5642e-07, 2.4495e-07], sigma_yp=[1.2088e-08, 4.5644e-09],
                   emittance_x=[7.1972e-13, 5.0000e-13], emittance_y=[3.5866e-15, 1.0000e-15],
                   normalized_emittance_x=[2.1127e-10, 1.4286e-08], normalized_emittance_y=[1.0528e-12, 2.8571e-11],
                   beta_x=[61.4750, 99.0000], alpha_x=[-1.2124, -0.9000], beta_y=[35.4190, 60.0000],
                   alpha_y=[0.6655, 0.5000], sigma_yyp=[-2.3871e-15, -5.0000e-16], relativistic_beta=[1.0, 1.0])
    for key, value in printed.items():
        assert np.allclose(m[key], value, rtol=1e-4), key
    out = o.element_track(o.Drift(np.array([1.0, 2.0], dtype=np.float32)), b)
    m2 = o.beam_moments(out)
    assert np.allclose(m2["sigma_x"], [6.7837e-06, 7.1650e-06], rtol=1e-4)
    assert np.allclose(m2["sigma_y"], [3.4987e-07, 2.4100e-07], rtol=1e-4)
    assert np.allclose(m2["sigma_xp"], m["sigma_xp"]) and np.allclose(m2["sigma_yp"], m["sigma_yp"])


@pytest.mark.parametrize("dtype,tol", [(np.float32, 3e-4), (np.float64, 2e-4)])
def test_kat2_composition_of_1051_maps(dtype, tol):
    """docs/examples/optimize_speed.ipynb:47-67 lattice; merged map printed at :270, drift at :287."""
    f = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    cell = [o.Quadrupole(f(0.1), k1=f(4.2)), o.Drift(f(0.2)), o.Quadrupole(f(0.1), k1=f(-4.2)), o.Drift(f(0.2)),
            o.Marker(), o.Quadrupole(f(0.1), k1=f(0.0)), o.Drift(f(0.2))]
    elements = [o.Drift(f(0.3))] + cell * 150
    energy = f(107315902.44394557)  # tests/test_astra_import.py:22
    tm = o.segment_transfer_map(elements, energy, dtype)[0]
    printed = {(0, 0): 1.3122, (0, 1): -3.4577, (1, 0): 0.18828, (1, 1): 0.26594, (2, 2): 0.30360,
               (2, 3): -3.2559, (3, 2): 0.18828, (3, 3): 1.2746}
    for (i, j), v in printed.items():
        assert abs(tm[i, j] - v) < tol * max(1.0, abs(v)), (i, j, tm[i, j], v)
    assert abs(tm[4, 5] - (-3.0678e-3)) < 1e-6
    assert np.array_equal(tm[6], [0, 0, 0, 0, 0, 0, 1]) and np.array_equal(tm[5], [0, 0, 0, 0, 0, 1, 0])
    drift = o.element_transfer_map(o.Drift(f(0.3)), energy, dtype)[0]
    assert np.isclose(drift[4, 5], -6.8021e-06, rtol=1e-4)  # pins -L/(beta^2 gamma^2), gamma = E/REST_ENERGY
    length = sum(float(e["length"][0]) for e in elements if "length" in e)
    assert abs(length - 135.2991) < 1e-3


def test_kat3_custom_transfer_map_passthrough():
    """tests/test_vectorized.py:371-392"""
    tm = np.array([[[1.0, 4.0e-02, 0, 0, 0, 0, 0], [0, 1.0, 0, 0, 0, 0, 1.0e-05], [0, 0, 1.0, 4.0e-02, 0, 0, 0],
                    [0, 0, 0, 1.0, 0, 0, 0], [0, 0, 0, 0, 1.0, -4.6422e-07, 0], [0, 0, 0, 0, 0, 1.0, 0],
                    [0, 0, 0, 0, 0, 0, 1.0]]], dtype=np.float32)
    spec = o.CustomTransferMap(tm, length=np.array([0.4], dtype=np.float32))
    assert np.array_equal(o.element_transfer_map(spec, np.array([1e8], np.float32)), tm)
    assert np.array_equal(o.segment_transfer_map([spec], np.array([1e8], np.float32))[0], tm[0])


def test_kat4_cavity_bmad_twiss():
    """tests/test_compare_ocelot.py:627-654: Bmad-confirmed Twiss behind the cavity."""
    b = o.parameter_beam_from_twiss(
        dtype=np.float64, beta_x=[5.91253677], alpha_x=[3.55631308], emittance_x=[3.494768647122823e-09],
        beta_y=[5.91253677], alpha_y=[3.55631308], emittance_y=[3.497810737006068e-09], energy=[6e6])
    c = o.Cavity(np.array([1.0377]), voltage=np.array([0.01815975e9]), frequency=np.array([1.3e9]),
                 phase=np.array([0.0]))
    out = o.element_track(c, b, np.float64)
    m = o.beam_moments(out)
    assert np.isclose(m["beta_x"][0], 0.23847352510683092, rtol=1e-7)
    assert np.isclose(m["alpha_x"][0], -1.0160687592932345, rtol=1e-7)
    assert np.isclose(m["beta_y"][0], 0.23847352512430994, rtol=1e-7)
    assert np.isclose(m["alpha_y"][0], -1.0160687593664295, rtol=1e-7)
    tm = o.cavity_rmatrix(c, np.array([6e6]), np.float64)[0]
    Ei, Ef = 6e6 / o.ELECTRON_MASS_EV, (6e6 + 0.01815975e9) / o.ELECTRON_MASS_EV
    assert np.isclose(np.linalg.det(tm[:2, :2]), Ei / Ef, rtol=1e-12)
    assert out["energy"][0] == 6e6 + 0.01815975e9


def test_kat5_beam_construction():
    """tests/test_parameter_beam.py:7-38,77-98"""
    kw = dict(mu_x=[1e-5], mu_xp=[1e-7], mu_y=[2e-5], mu_yp=[2e-7], sigma_x=[1.75e-7], sigma_xp=[2e-7],
              sigma_y=[1.75e-7], sigma_yp=[2e-7], sigma_s=[0.000001], sigma_p=[0.000001], cor_x=[0.0], cor_y=[0.0],
              cor_s=[0.0], energy=[1e7])
    m = o.beam_moments(o.parameter_beam_from_parameters(**kw))
    for k, v in kw.items():
        if k.startswith(("mu_", "sigma_")) or k == "energy":
            assert np.isclose(m[k], v[0]), k
    b = o.parameter_beam_from_twiss(beta_x=[5.91253676811640894], alpha_x=[3.55631307633660354],
                                    emittance_x=[3.494768647122823e-09], beta_y=[5.91253676811640982],
                                    alpha_y=[2e-7], emittance_y=[3.497810737006068e-09], energy=[6e6])
    m = o.beam_moments(b)
    assert np.isclose(m["beta_x"], 5.91253676811640894) and np.isclose(m["alpha_x"], 3.55631307633660354)
    assert np.isclose(m["emittance_x"], 3.494768647122823e-09) and np.isclose(m["beta_y"], 5.91253676811640982)
    assert np.isclose(m["alpha_y"], 2e-7, atol=1e-6) and np.isclose(m["emittance_y"], 3.497810737006068e-09)
    # defaults of from_parameters (parameter_beam.py:97-113)
    d = o.parameter_beam_from_parameters()
    assert d["mu"].shape == (1, 7) and d["cov"].shape == (1, 7, 7) and d["mu"][0, 6] == 1
    assert np.isclose(np.sqrt(d["cov"][0, 0, 0]), 175e-9) and d["energy"][0] == 1e8


def test_constants_follow_the_reference_definitions():
    assert np.isclose(o.REST_ENERGY, 510998.9506917531, rtol=1e-12)  # track_methods.py:9-11
    assert o.ELECTRON_MASS_EV == 510998.95069  # cavity.py:20


def test_relational_invariants_of_the_reference_suite():
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    E = f(1e8)
    # tests/test_quadrupole.py:6-22  k1 = 0 quadrupole behaves like a drift
    q = o.element_transfer_map(o.Quadrupole(f(1.0), k1=f(0.0)), E)
    d = o.element_transfer_map(o.Drift(f(1.0)), E)
    assert np.allclose(q[0, :4, :4], d[0, :4, :4], atol=1e-6)
    # tests/test_dipole.py:6-22  angle = 0 dipole behaves like a drift
    dp = o.element_transfer_map(o.Dipole(f(1.0), angle=f(0.0)), E)
    assert np.allclose(dp[0, :4, :4], d[0, :4, :4], atol=1e-6)
    # tests/test_quadrupole.py:77-98  tilt pi/4 == 5pi/4 != pi/2
    t = o.element_transfer_map(o.Quadrupole(np.full(3, 0.5, np.float32), k1=np.ones(3, np.float32),
                                            tilt=np.array([np.pi / 4, np.pi / 2, 5 * np.pi / 4], np.float32)),
                               np.full(3, 1e9, np.float32))
    assert np.allclose(t[0], t[2], atol=1e-6) and not np.allclose(t[0], t[1], atol=1e-3)
    # partition (segment.py:344-351)
    els = [o.Drift(f(1)), o.BPM(), o.Cavity(f(1), voltage=f(1e6)), o.Marker(), o.Drift(f(1)), o.BPM(is_active=True),
           o.Cavity(f(1), voltage=f(0.0))]
    assert [k for k, _ in o.partition(els)] == ["run", "single", "run", "single", "run"]


def test_whole_batch_branches():
    """`if any(...)` predicates act on the whole batch (dipole.py:119, cavity.py:128,164)."""
    dtype = np.float64
    mixed = o.element_transfer_map(o.Dipole(np.array([0.0, 0.5]), angle=np.array([0.01, 0.02])), np.array([1e8, 1e8]), dtype)
    alone = o.element_transfer_map(o.Dipole(np.array([0.0]), angle=np.array([0.01])), np.array([1e8]), dtype)
    assert mixed[0, 2, 6] == 0.0 and alone[0, 2, 6] == 0.01
    # V = 0 row of an active cavity: NaN in r12 like the reference (cavity.py:269)
    tm = o.cavity_rmatrix(o.Cavity(np.array([1.0, 1.0]), voltage=np.array([0.0, 1e7]), phase=np.zeros(2),
                                   frequency=np.full(2, 1.3e9)), np.array([1e8, 1e8]), dtype)
    assert np.isnan(tm[0, 0, 1]) and not np.isnan(tm[1]).any()
    with pytest.raises(AssertionError):  # cavity.py:260
        o.cavity_rmatrix(o.Cavity(np.array([1.0]), voltage=np.array([1e6])), np.array([0.0]), dtype)
