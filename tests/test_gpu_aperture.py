"""
Active `Aperture` on the GPU (lynx_aperture_mask / lynx_aperture_compact) against the oracle's
mask (reference lynx/accelerator/aperture.py:69-108): which particles survive, their order,
the lost ones, and the behaviour inside a Segment.  The reference has no test for this element.
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
@pytest.mark.parametrize("shape", ["rectangular", "elliptical"])
@pytest.mark.parametrize("n", [1, 1023, 1024, 50_001])
def test_single_sample_loses_particles_in_order(lx, dtype, shape, n):
    P = o.gaussian_particles((1,), n, seed=n, dtype=dtype, sigma=[1e-3, 1e-4, 2e-3, 1e-4, 1e-5, 1e-3])
    P[0, 0, 0] = 1e-3  # exactly on the rectangular limit: lost (strict inequality, aperture.py:79)
    charges = np.arange(n, dtype=dtype)[None] * 1e-15
    aperture = lx.Aperture(x_max=np.array([1e-3], dtype), y_max=np.array([1.5e-3], dtype), shape=shape, dtype=dtype)
    beam = lx.ParticleBeam(P, np.array([1e8], dtype), particle_charges=charges, dtype=dtype)
    keep = o.aperture_mask(P, np.array([1e-3], dtype), np.array([1.5e-3], dtype), shape)[0]
    out = aperture.track(beam)
    if keep.sum() == 0:
        assert out is lx.Beam.empty
        return
    assert out.num_particles == keep.sum() and out.batch_shape == (1,)
    assert np.array_equal(np.asarray(out.particles)[0], P[0][keep])          # survivors, original order
    assert np.array_equal(np.asarray(aperture.lost_particles), P[0][~keep])   # the rest, original order
    assert np.array_equal(out.particle_charges[0], charges[0][keep])
    assert np.array_equal(aperture.lost_particle_charges, charges[0][~keep])
    assert np.array_equal(out.energy, beam.energy)
    assert np.isclose(out.total_charge[0], charges[0][keep].sum())


def test_parameter_beam_and_inactive_aperture_pass_through(lx):
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    beam = lx.ParameterBeam.from_parameters(sigma_x=f(1e-2))
    assert lx.Aperture(x_max=f(1e-6), y_max=f(1e-6)).track(beam) is beam  # aperture.py:70-72
    pbeam = lx.ParticleBeam.from_parameters(num_particles=1000, sigma_x=f(1e-2), seed=0)
    assert lx.Aperture(x_max=f(1e-6), y_max=f(1e-6), is_active=False).track(pbeam) is pbeam


def test_batches_are_accepted_as_long_as_nothing_is_lost(lx):
    f = lambda v: np.full((2, 3), v, dtype=np.float32)  # noqa: E731
    beam = lx.ParticleBeam.from_parameters(num_particles=5000, sigma_x=np.array([1e-4], np.float32), seed=1).broadcast((2, 3))
    wide = lx.Aperture(x_max=f(np.inf), y_max=f(np.inf))
    out = wide.track(beam)
    assert np.array_equal(np.asarray(out.particles), np.asarray(beam.particles)) and wide.lost_particles.shape == (0, 7)
    narrow = lx.Aperture(x_max=f(1e-4), y_max=f(1.0))
    with pytest.raises(NotImplementedError, match="particles lost in a batch"):
        narrow.track(beam)


def test_aperture_inside_a_segment(lx):
    """ARES-style stretch: infinite active apertures (as in ARESlatticeStage3v1_9.json) and a real one."""
    f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
    P = o.gaussian_particles((1,), 20_000, seed=4, dtype=np.float32, sigma=[2e-4, 1e-5, 2e-4, 1e-5, 1e-5, 1e-3])
    beam = lx.ParticleBeam(P, f(1e8))
    inf = np.array([np.inf], np.float32)
    collimator = lx.Aperture(x_max=f(3e-4), y_max=f(3e-4), shape="elliptical", name="COL")
    seg = lx.Segment([lx.Aperture(x_max=inf, y_max=inf, name="ARLISLHG1"), lx.Drift(f(0.5)),
                      lx.Quadrupole(f(0.122), k1=f(4.2)), collimator, lx.Drift(f(0.3)),
                      lx.Solenoid(f(0.09), k=f(0.0)), lx.Screen(is_active=False), lx.Drift(f(0.2))])
    assert not seg.is_skippable
    out = seg.track(beam)
    # oracle: track to the collimator, clip, track on
    head = o.segment_track([o.Drift(f(0.5)), o.Quadrupole(f(0.122), k1=f(4.2))], o.particle_beam(P, f(1e8)), np.float32)
    keep = o.aperture_mask(head["particles"], f(3e-4), f(3e-4), "elliptical")[0]
    tail = o.segment_track([o.Drift(f(0.3)), o.Solenoid(f(0.09), k=f(0.0)), o.Drift(f(0.2))],
                           o.particle_beam(head["particles"][:, keep], f(1e8)), np.float32)
    got = np.asarray(out.particles)
    assert 0 < keep.sum() < 20_000 and got.shape == tail["particles"].shape
    assert np.allclose(got, tail["particles"], rtol=2e-5, atol=1e-9)
    assert collimator.lost_particles.shape == (20_000 - keep.sum(), 7)
    assert np.isclose(out.sigma_x[0], tail["particles"][0, :, 0].std(ddof=1), rtol=1e-4)
    # a ParameterBeam goes through the same lattice untouched by the apertures
    pout = seg.track(lx.ParameterBeam.from_parameters(sigma_x=f(2e-4)))
    assert pout.sigma_x.shape == (1,)
