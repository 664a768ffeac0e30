"""cProfile of the host path: optimisation-loop pattern (one setting written, track, read sigma_x)."""
import cProfile
import pstats

import numpy as np

import lynx_amd as lx
from lynx_amd.device import get_runtime

rt = get_runtime()
f = lambda v: np.array([v], dtype=np.float32)  # noqa: E731
seg = lx.Segment([lx.BPM(name="B1"), lx.Drift(f(1.0)), lx.BPM(name="B2"), lx.Drift(f(1.0)),
                  lx.VerticalCorrector(f(0.3), angle=f(3.142e-3), name="V"), lx.Drift(f(0.2)),
                  lx.HorizontalCorrector(f(0.3), angle=f(1e-4)), lx.Drift(f(7.0)),
                  lx.HorizontalCorrector(f(0.3), angle=f(-1e-4)), lx.Drift(f(0.05)), lx.BPM(name="B3")])
pb = lx.ParameterBeam.from_parameters()


def loop(n):
    for i in range(n):
        seg.V.angle = f(3e-3 + 1e-9 * i)
        out = seg.track(pb)
        _ = out.sigma_x


loop(200)
pr = cProfile.Profile()
pr.enable()
loop(2000)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
