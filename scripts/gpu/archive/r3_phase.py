"""Round 3: does the streaming kernel's time depend on WHERE its output buffer sits relative to its input?  (From a
cold start consecutive calls alternate between two pool blocks and between 0.93 and 0.96 ms.)  BASELINE config 4 through
the C ABI with the output placed at a sweep of offsets inside one large block; kernel time from the dispatch's stamps."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402
import lynx_amd as lx  # noqa: E402
from lynx_amd import _ffi  # noqa: E402

rt = lx.device.get_runtime()
B, N = 1024, 100_000
segment = bench.build_segment(lx, "c4", np.arange(B), 64, np.float32, 3)
beam = lx.ParticleBeam.synthetic((B,), N, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3], energy=1e8, seed=2, dtype=np.float32)
for _ in range(30):  # steady state first
    out = segment.track(beam)
rt.sync()
lat = segment._lattice_cache._last[3]
p_in = beam._particles.device(rt)
e_in = beam._energy.broadcast_device(rt, (B,))
nbytes = B * N * 7 * 4
big = rt.alloc(nbytes + (64 << 20))
mom = rt.empty((B, _ffi.MOMENT_STRIDE), np.float64)
print("input at", hex(p_in.ptr), "block at", hex(big), "distance mod 2 MiB", (big - p_in.ptr) % (2 << 20))
offsets = [0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 3 << 20, 4 << 20, (4 << 20) + 4096,
           8 << 20, 16 << 20, 32 << 20]
for rep in range(2):
    for off in offsets:
        p_out = C.c_void_p(big + off)
        for _ in range(3):
            rt.check(rt.lib.lynx_track_particles(rt.ctx, lat.handle, N, C.c_void_p(e_in.ptr), C.c_void_p(p_in.ptr), p_out, None,
                                                 C.c_void_p(mom.ptr), _ffi.TRACK_MOMENTS, None))
        rt.sync()
        rt.profile_begin()
        for _ in range(12):
            rt.check(rt.lib.lynx_track_particles(rt.ctx, lat.handle, N, C.c_void_p(e_in.ptr), C.c_void_p(p_in.ptr), p_out, None,
                                                 C.c_void_p(mom.ptr), _ffi.TRACK_MOMENTS, None))
        rt.sync()
        ms, n = rt.profile_end()
        print("rep %d  output offset %9d B  (out - in) mod 64 KiB = %6d  kernel %.4f ms" % (rep, off, (big + off - p_in.ptr) % 65536, ms / n))
rt.free(big)
