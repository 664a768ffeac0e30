import sys
import numpy as np
sys.path.insert(0, ".")
import lynx_amd as lx
from oracle import lynx_oracle as o
from tests.helpers import make_lattice, moment_distances
from tests.test_gpu_parity import _bench_workload, _subset
dtype = np.float32
desc, segment, beam, energy = _bench_workload(lx, "c5", dtype)
out = segment.track(beam)
for pick in ([0, 2047, 4095], [3, 71, 175], list(range(0, 128))):
    P = np.asarray(beam.particles)[pick]
    _, specs32 = make_lattice(_subset(desc, pick), dtype)
    _, specs64 = make_lattice(_subset(desc, pick, cast=np.float64), np.float64)
    e = np.full(len(pick), energy, dtype=dtype)
    ref32 = o.segment_track(specs32, o.particle_beam(P, e, dtype), dtype)
    ref64 = o.segment_track(specs64, o.particle_beam(P.astype(np.float64), e.astype(np.float64), np.float64), np.float64)
    m32, m64 = o.beam_moments(ref32, ddof=1), o.beam_moments(ref64, ddof=1)
    got = {key: np.asarray(getattr(out, key))[pick] for key in m32 if hasattr(out, key)}
    print(pick[:5], "mu_p: p-o32 %.2e  p-o64 %.2e  o32-o64 %.2e" % (moment_distances(got, m32, scale=m64)["mu_p"], moment_distances(got, m64, scale=m64)["mu_p"], moment_distances(m32, m64, scale=m64)["mu_p"]))
    print("   mu_p got", got["mu_p"][:3], "o32", np.asarray(m32["mu_p"])[:3], "o64", np.asarray(m64["mu_p"])[:3], "sigma_p", np.asarray(m64["sigma_p"])[:3])
