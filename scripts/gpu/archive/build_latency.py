"""k_build duration vs lattice content (run under rocprofv3 --kernel-trace --stats)."""
import numpy as np

import lynx_amd as lx
from lynx_amd.device import get_runtime

rt = get_runtime()
B = 1024
f = lambda v: np.full(B, v, np.float32)  # noqa: E731
cases = {
    "drift128": [lx.Drift(f(0.5)) for _ in range(128)],
    "fodo128": sum([[lx.Quadrupole(f(0.2), k1=f(4.2)), lx.Drift(f(0.5)), lx.Quadrupole(f(0.2), k1=f(-4.2)), lx.Drift(f(0.5))]
                    for _ in range(32)], []),
    "fodo64": sum([[lx.Quadrupole(f(0.2), k1=f(4.2)), lx.Drift(f(0.5)), lx.Quadrupole(f(0.2), k1=f(-4.2)), lx.Drift(f(0.5))]
                   for _ in range(16)], []),
    "fodo16": sum([[lx.Quadrupole(f(0.2), k1=f(4.2)), lx.Drift(f(0.5)), lx.Quadrupole(f(0.2), k1=f(-4.2)), lx.Drift(f(0.5))]
                   for _ in range(4)], []),
    "quad1": [lx.Quadrupole(f(0.2), k1=f(4.2))],
}
energy = f(1e8)
import time
for name, elements in cases.items():
    seg = lx.Segment(elements)
    for _ in range(3):
        seg.transfer_map(energy)
    rt.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        seg.transfer_map(energy)
    rt.sync()
    print(name, "wall per transfer_map (incl. D2H) us", round((time.perf_counter() - t0) / 20 * 1e6, 1))
