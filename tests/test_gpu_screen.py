"""
Screen read-out on the GPU (lynx_histogram2d, lynx_gaussian_image) against the oracle
(numpy.histogramdd / bivariate normal density), and the swallow-the-beam semantics of an
active screen inside a Segment (reference tests/test_screen.py, screen.py:126-216).
"""

import numpy as np
import pytest

from oracle import lynx_oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lx(built_library):
    import lynx_amd

    lynx_amd.device.get_runtime()
    return lynx_amd


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_particle_beam_histogram_is_exact(lx, dtype):
    res, px, binning = (200, 120), (3.5e-6, 2.5e-6), 2
    P = o.gaussian_particles((3,), 50_000, seed=1, dtype=dtype, mu=[5e-5, 0, -3e-5, 0, 0, 0],
                             sigma=[1.2e-4, 1e-5, 0.9e-4, 1e-5, 1e-5, 1e-3])
    edges = o.screen_bin_edges(res, px, binning, dtype)
    P[0, :50, 0] = edges[0][np.arange(50) % len(edges[0])]   # values exactly on bin edges (incl. the last one)
    P[0, :50, 2] = edges[1][np.arange(50) % len(edges[1])]
    screen = lx.Screen(resolution=res, pixel_size=px, binning=binning, is_active=True, dtype=dtype,
                       misalignment=np.zeros((3, 2)))
    out = screen.track(lx.ParticleBeam(P, np.full(3, 1e8), dtype=dtype))
    assert out is lx.Beam.empty
    image = screen.reading
    ref = o.screen_reading_particles(P, res, px, binning, dtype)
    assert image.shape == (3, 60, 100) == ref.shape
    assert np.array_equal(image, ref)  # integer counts, bit-exact bin assignment
    assert 0 < image.sum() <= 3 * 50_000
    assert screen.reading is image  # cached (screen.py:145-146)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_parameter_beam_gaussian_image(lx, dtype):
    res, px, binning = (160, 96), (2e-5, 2e-5), 4
    beam = lx.ParameterBeam.from_parameters(mu_x=np.array([1e-4, -2e-4]), mu_y=np.array([5e-5, 0.0]),
                                            sigma_x=np.array([2e-4, 3e-4]), sigma_y=np.array([1e-4, 2e-4]),
                                            dtype=dtype)
    screen = lx.Screen(resolution=res, pixel_size=px, binning=binning, is_active=True, dtype=dtype,
                       misalignment=np.zeros((2, 2)))
    assert screen.track(beam) is lx.Beam.empty
    image = screen.reading
    ref = o.screen_reading_parameters(beam._mu, beam._cov, res, px, binning, dtype)
    assert image.shape == ref.shape == (2, 40, 24)
    assert np.max(np.abs(image - ref)) <= (2e-4 if dtype == np.float32 else 1e-10) * ref.max()


def test_active_screen_inside_a_segment(lx):
    """reference tests/test_screen.py / tests/test_speed.py: ARES-style use."""
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    screen = lx.Screen(resolution=(64, 64), pixel_size=(2e-5, 2e-5), is_active=True, name="SCR")
    bpm = lx.BPM(name="B", is_active=True)
    seg = lx.Segment([lx.Drift(f(1.0)), lx.Quadrupole(f(0.2), k1=f(2.0)), screen, lx.Drift(f(1.0)), bpm])
    beam = lx.ParticleBeam.from_parameters(num_particles=20_000, sigma_x=f(1e-4), sigma_y=f(1e-4), seed=3)
    assert seg.track(beam) is lx.Beam.empty and bpm.reading is None  # the BPM behind the screen sees no beam
    assert screen.reading.shape == (1, 64, 64) and screen.reading.sum() > 10_000
    screen.is_active = False
    out = seg.track(beam)
    assert out is not lx.Beam.empty and bpm.reading.shape == (2, 1)
    empty = lx.Screen(resolution=(8, 6), is_active=True)
    assert np.array_equal(empty.reading, np.zeros((6, 8)))
