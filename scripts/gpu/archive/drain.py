"""
How long does k_reduce_moments take when the kernel in front of it left no dirty lines behind?
Moments of a resident C4-shaped beam (read-only pass + reduction), 20 times; run under rocprofv3 --kernel-trace --stats
and compare k_reduce_moments with its duration behind the storing pass of `bench.py --workload c4`.
"""
import numpy as np

import lynx_amd as lx

beam = lx.ParticleBeam.synthetic((1024,), 100_000, seed=1)
rt = lx.device.get_runtime()
for _ in range(20):
    beam._moments = None
    beam.moment_record()
rt.synchronize() if hasattr(rt, "synchronize") else None
print("ok", float(np.nanmax(beam.moment_record()[..., 35])))
