"""Latency of the small configurations (BASELINE configs 1 and 2) and of a huge ParameterBeam batch."""
import json
import time

import numpy as np

import lynx_amd as lx
from lynx_amd.device import get_runtime

rt = get_runtime()
f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731


def ares():
    return lx.Segment([lx.BPM(name="B1"), lx.Drift(f(1.0)), lx.BPM(name="B2"), lx.Drift(f(1.0)),
                       lx.VerticalCorrector(f(0.3), angle=f(3.142e-3)), lx.Drift(f(0.2)),
                       lx.HorizontalCorrector(f(0.3), angle=f(1e-4)), lx.Drift(f(7.0)),
                       lx.HorizontalCorrector(f(0.3), angle=f(-1e-4)), lx.Drift(f(0.05)), lx.BPM(name="B3")])


def timed(fn, n):
    for _ in range(20):
        fn()
    rt.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    rt.sync()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e6, out


res = {}
seg = ares()
pb = lx.ParameterBeam.from_parameters()
res["c1_parameter_beam_B1_us"], out = timed(lambda: seg.track(pb), 2000)
_ = out.sigma_x
res["c1_parameter_beam_B1_with_readback_us"], _ = timed(lambda: seg.track(pb).sigma_x, 500)
beam = lx.ParticleBeam.from_parameters(num_particles=100_000, seed=0)
res["c2_particle_beam_100k_us"], _ = timed(lambda: seg.track(beam), 2000)
res["c2_particle_beam_100k_with_sigma_x_us"], _ = timed(lambda: seg.track(beam).sigma_x, 500)
shape = (3, 100_000)
big = seg.broadcast(shape)
pbb = lx.ParameterBeam.from_parameters().broadcast(shape)
res["c1_parameter_beam_3x100k_settings_us"], _ = timed(lambda: big.track(pbb), 200)
res["c1_settings_per_second"] = 3e5 / (res["c1_parameter_beam_3x100k_settings_us"] * 1e-6)
# optimisation-loop pattern (README.md:60, docs/examples/gradientbased.ipynb): one magnet
# setting is written before every track
k = [0.0]


def retune_and_track(b):
    k[0] += 0.01
    seg.elements[4].angle = f(3e-3 + 1e-6 * k[0])
    return seg.track(b)


res["c1_parameter_beam_B1_setting_changed_every_track_us"], _ = timed(lambda: retune_and_track(pb), 2000)
res["c2_particle_beam_100k_setting_changed_every_track_us"], _ = timed(lambda: retune_and_track(beam), 2000)

# the lattice of docs/examples/optimize_speed.ipynb:47-67 (1051 elements here): ParameterBeam,
# B = 1 and B = 1000, unoptimised and with transfer maps merged -- the only timings the
# reference publishes (PyTorch Cheetah, unstated CPU): 138 ms, 440 us, 1.9 ms for batch 1000
def speed_lattice(shape):
    r = lambda v: np.full(shape, v, dtype=np.float32)  # noqa: E731
    els = [lx.Drift(r(0.3))]
    for _ in range(150):
        els += [lx.Quadrupole(r(0.1), k1=r(4.2)), lx.Drift(r(0.2)), lx.Quadrupole(r(0.1), k1=r(-4.2)), lx.Drift(r(0.2)),
                lx.Marker(), lx.Quadrupole(r(0.1), k1=r(0.0)), lx.Drift(r(0.2))]
    return lx.Segment(els)


for B in (1, 1000):
    seg = speed_lattice((B,))
    pbB = lx.ParameterBeam.from_parameters(energy=np.full((B,), 1.0732e8, np.float32))
    res[f"speed_lattice_{len(seg.elements)}_elements_B{B}_us"], _ = timed(lambda: seg.track(pbB).sigma_x, 200)
    merged = seg.transfer_maps_merged(incoming_beam=pbB)
    res[f"speed_lattice_merged_{len(merged.elements)}_elements_B{B}_us"], _ = timed(lambda: merged.track(pbB).sigma_x, 500)
print(json.dumps({k: round(v, 2) for k, v in res.items()}))
