"""Round 3: which output buffers the first calls of BASELINE config 4 get from the pool (addresses, in call order)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402
import lynx_amd as lx  # noqa: E402

rt = lx.device.get_runtime()
B, N = 1024, 100_000
segment = bench.build_segment(lx, "c4", np.arange(B), 64, np.float32, 3)
beam = lx.ParticleBeam.synthetic((B,), N, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3], energy=1e8, seed=2, dtype=np.float32)
rt.sync()
seen = {}
last = None
for k in range(30):
    t0 = time.perf_counter()
    last = segment.track(beam)
    t1 = time.perf_counter()
    ptr = last._particles.device(rt).ptr
    mom = last._moments.device(rt).ptr
    print(k, hex(ptr), "new" if ptr not in seen else "seen@%d" % seen[ptr], hex(mom), "host %.0f us" % (1e6 * (t1 - t0)))
    seen.setdefault(ptr, k)
rt.sync()
