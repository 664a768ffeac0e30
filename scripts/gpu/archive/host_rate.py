"""Is the pipelined rate of small beams set by the host or by the GPU?  BASELINE config 2 (ARES-like lattice, 100 000
particles, float32): time to ENQUEUE n track() calls vs time until they have run."""
import time

import numpy as np

import lynx_amd as lx
from lynx_amd.device import get_runtime

rt = get_runtime()
f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
seg = lx.Segment([lx.BPM(name="B1"), lx.Drift(f(1.0)), lx.BPM(name="B2"), lx.Drift(f(1.0)),
                  lx.VerticalCorrector(f(0.3), angle=f(3.142e-3), name="V"), lx.Drift(f(0.2)),
                  lx.HorizontalCorrector(f(0.3), angle=f(1e-4)), lx.Drift(f(7.0)),
                  lx.HorizontalCorrector(f(0.3), angle=f(-1e-4)), lx.Drift(f(0.05)), lx.BPM(name="B3")])
beam = lx.ParticleBeam.synthetic((1,), 100_000, seed=1)
for _ in range(50):
    out = seg.track(beam)
rt.sync()
for n in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(n):
        out = seg.track(beam)
    t1 = time.perf_counter()
    rt.sync()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {1e6 * (t1 - t0) / n:.1f} us/call, until done {1e6 * (t2 - t0) / n:.1f} us/call")

import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    out = seg.track(beam)
pr.disable()
rt.sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
