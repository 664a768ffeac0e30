"""Round 3: config 5's shape, particles: GPU vs float32 oracle vs float64 oracle, coordinate by coordinate (the numbers
next to the s / delta tolerances of the float32 particle tests)."""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import lynx_amd as lx  # noqa: E402
from oracle import lynx_oracle as o  # noqa: E402
from tests.helpers import make_lattice, rel_err  # noqa: E402

B, N = 16, 10_000
rng = np.random.default_rng(4)
f = lambda v: np.full(B, v)  # noqa: E731
desc = []
for _ in range(8):
    desc += [("drift", dict(length=f(0.3))),
             ("quadrupole", dict(length=f(0.1), k1=rng.uniform(-5, 5, B), misalignment=rng.normal(0, 1e-4, (B, 2)))),
             ("drift", dict(length=f(0.3))),
             ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B), frequency=f(1.3e9)))]
for sigma_s in (1e-4, 1e-5):
    sigma = [1e-4, 1e-5, 1e-4, 1e-5, sigma_s, 1e-3]
    elements, s32 = make_lattice(desc, np.float32, lx)
    _, s64 = make_lattice([(k, {a: np.asarray(v, dtype=np.float32).astype(np.float64) for a, v in kw.items()}) for k, kw in desc], np.float64)
    P = o.gaussian_particles((B,), N, seed=3, dtype=np.float32, sigma=sigma)
    out = lx.Segment(elements).track(lx.ParticleBeam(P, np.full(B, 6e6, np.float32), dtype=np.float32))
    got = np.asarray(out.particles)
    r32 = o.segment_track(s32, o.particle_beam(P, np.full(B, 6e6, np.float32), np.float32), np.float32)["particles"]
    r64 = o.segment_track(s64, o.particle_beam(P.astype(np.float64), np.full(B, 6e6), np.float64), np.float64)["particles"]
    print("sigma_s", sigma_s)
    for c in range(6):
        print("  coordinate %d: gpu-oracle32 %.2e  gpu-oracle64 %.2e  oracle32-oracle64 %.2e" % (
            c, rel_err(got[..., c], r32[..., c]), rel_err(got[..., c], r64[..., c]), rel_err(r32[..., c], r64[..., c])))
