"""
Gradient-based tuning of the ARES experimental-area magnets -- the scenario of the reference's
docs/examples/gradientbased.ipynb: three quadrupoles and two correctors are the parameters,
the loss is the mean squared (mu_x, sigma_x, mu_y, sigma_y) of the beam on the screen, the
optimiser is Adam (written out in NumPy; the reference used torch.optim).

    python examples/gradient_based_tuning.py          # needs an MI355X and the built library
"""

import numpy as np

import lynx_amd as lx
import lynx_amd.grad as grad

f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731

MAGNETS = (("AREAMQZM1", "k1"), ("AREAMQZM2", "k1"), ("AREAMCVM1", "angle"), ("AREAMQZM3", "k1"), ("AREAMCHM1", "angle"))
OBSERVED = ("mu_x", "sigma_x", "mu_y", "sigma_y")


def ares_ea():
    """AREASOLA1 ... AREABSCR1 of docs/examples/ARESlatticeStage3v1_9.json."""
    return lx.Segment([
        lx.Marker(name="AREASOLA1"), lx.Drift(f(0.17504)),
        lx.Quadrupole(f(0.122), k1=f(0.0), name="AREAMQZM1"), lx.Drift(f(0.428)),
        lx.Quadrupole(f(0.122), k1=f(0.0), name="AREAMQZM2"), lx.Drift(f(0.204)),
        lx.VerticalCorrector(f(0.02), angle=f(0.0), name="AREAMCVM1"), lx.Drift(f(0.204)),
        lx.Quadrupole(f(0.122), k1=f(0.0), name="AREAMQZM3"), lx.Drift(f(0.179)),
        lx.HorizontalCorrector(f(0.02), angle=f(0.0), name="AREAMCHM1"), lx.Drift(f(0.45)),
        lx.Screen(resolution=(2448, 2040), pixel_size=(3.5488e-06, 2.5003e-06), name="AREABSCR1"),
    ])


def tune(segment, beam, steps=100, scale=2e-3, lr=(0.5, 0.5, 2e-4, 0.5, 2e-4), target=np.zeros(4)):
    """Adam on the five settings; returns the loss history (loss = mse of observed / scale)."""
    m, v, history = np.zeros(5), np.zeros(5), []
    for t in range(1, steps + 1):
        vjp = grad.track_vjp(segment, beam)
        out = vjp.outgoing
        observed = np.array([float(getattr(out, name)[0]) for name in OBSERVED])
        residual = (observed - target) / scale
        history.append(float(np.mean(residual**2)))
        bars = {name: 2.0 * r / (4 * scale) for name, r in zip(OBSERVED, residual)}  # d mse / d observed
        g = vjp(**bars)
        gradient = np.array([float(g[getattr(segment, name)][attr][0]) for name, attr in MAGNETS])
        m = 0.9 * m + 0.1 * gradient
        v = 0.999 * v + 0.001 * gradient**2
        update = np.asarray(lr) * (m / (1 - 0.9**t)) / (np.sqrt(v / (1 - 0.999**t)) + 1e-12)
        for (name, attr), delta in zip(MAGNETS, update):
            element = getattr(segment, name)
            setattr(element, attr, (getattr(element, attr) - delta).astype(np.float32))
    return history


if __name__ == "__main__":
    segment = ares_ea()
    segment.AREAMQZM1.k1, segment.AREAMQZM2.k1, segment.AREAMQZM3.k1 = f(5.0), f(-5.0), f(5.0)
    segment.AREAMCVM1.angle, segment.AREAMCHM1.angle = f(1e-3), f(-1e-3)
    beam = lx.ParticleBeam.from_parameters(num_particles=100_000, sigma_x=f(1.75e-4), sigma_xp=f(3.7e-6),
                                           sigma_y=f(1.75e-4), sigma_yp=f(3.7e-6), sigma_s=f(8e-6), sigma_p=f(2.3e-3),
                                           energy=f(1.07e8), seed=0)
    history = tune(segment, beam)
    print(f"loss {history[0]:.4g} -> {history[-1]:.4g} after {len(history)} Adam steps")
    for name, attr in MAGNETS:
        print(f"  {name}.{attr} = {float(getattr(getattr(segment, name), attr)[0]):+.5g}")
