"""Round 3: does a plain copy kernel also start slow after an idle stretch?  40 launches of k_diag_copy (2.87 GB, one 16-byte
vector per thread) from a cold start, each timed by its own events; then config 4's streaming kernel right behind it."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402
import lynx_amd as lx  # noqa: E402

rt = lx.device.get_runtime()
nbytes = 1024 * 100_000 * 7 * 4
a, b = rt.alloc(nbytes), rt.alloc(nbytes)
rt.sync()
time.sleep(0.5)
for shape in (1, 101):
    out = []
    for k in range(40):
        ms = C.c_float()
        rt.check(rt.lib.lynx_diag_copy(rt.ctx, C.c_void_p(b), C.c_void_p(a), nbytes, 1, shape, C.byref(ms)))
        out.append(ms.value)
    print("copy shape %d, ms per launch:" % shape, " ".join("%.4f" % v for v in out))
    time.sleep(0.5)
segment = bench.build_segment(lx, "c4", np.arange(1024), 64, np.float32, 3)
beam = lx.ParticleBeam.synthetic((1024,), 100_000, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3], energy=1e8, seed=2, dtype=np.float32)
o = segment.track(beam)
rt.sync()
for k in range(60):  # keep the GPU busy with copies, then track at once
    ms = C.c_float()
    rt.check(rt.lib.lynx_diag_copy(rt.ctx, C.c_void_p(b), C.c_void_p(a), nbytes, 1, 1, C.byref(ms)))
rt.profile_begin()
for k in range(30):
    o = segment.track(beam)
rt.sync()
import os  # noqa: E402
os.environ["LYNX_PROFILE_DUMP"] = "1"
ms, n = rt.profile_end()
print("tracking behind 60 copies: mean %.4f ms over %d launches (per-launch values on stderr)" % (ms / n, n))
