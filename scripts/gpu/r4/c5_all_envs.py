"""
BASELINE config 5 on the bench beam: product vs float32 oracle vs float64 oracle for ALL 4096 environments (the GPU test
draws eleven per run): per moment the distribution of the distances, and every environment that would fail the test's
assertion  d(product, oracle32) <= max(1e-4, 2 d(oracle32, oracle64)).
    PYTHONPATH=. python scripts/gpu/r4/c5_all_envs.py
"""
import sys

import numpy as np

sys.path.insert(0, ".")
import lynx_amd as lx  # noqa: E402
from oracle import lynx_oracle as o  # noqa: E402
from tests.helpers import MOMENT_KEYS, make_lattice  # noqa: E402
from tests.test_gpu_parity import _bench_workload, _subset  # noqa: E402

dtype = np.float32
desc, segment, beam, energy = _bench_workload(lx, "c5", dtype)
B = beam.batch_shape[0]
out = segment.track(beam)
got_all = {key: np.asarray(getattr(out, key)) for key in MOMENT_KEYS}
P_all = np.asarray(beam.particles)
worst = {key: [] for key in MOMENT_KEYS}
fails = []
for lo in range(0, B, 128):
    pick = list(range(lo, min(lo + 128, B)))
    P = P_all[pick]
    _, specs32 = make_lattice(_subset(desc, pick), dtype)
    _, specs64 = make_lattice(_subset(desc, pick, cast=np.float64), np.float64)
    e = np.full(len(pick), energy, dtype=dtype)
    m32 = o.beam_moments(o.segment_track(specs32, o.particle_beam(P, e, dtype), dtype), ddof=1)
    m64 = o.beam_moments(o.segment_track(specs64, o.particle_beam(P.astype(np.float64), e.astype(np.float64), np.float64), np.float64), ddof=1)

    def scale(key):
        if key.startswith("mu_"):
            return np.abs(m64[key]) + m64["sigma" + key[2:]]
        if key in ("sigma_xxp", "sigma_yyp"):
            a, b = ("sigma_x", "sigma_xp") if key == "sigma_xxp" else ("sigma_y", "sigma_yp")
            return m64[a] * m64[b]
        return m64[key]

    for key in MOMENT_KEYS:
        s = scale(key)
        d_p32 = np.abs(got_all[key][pick] - m32[key]) / s
        d_3264 = np.abs(np.asarray(m32[key], dtype=np.float64) - m64[key]) / s
        worst[key].append(np.stack([d_p32, d_3264], axis=1))
        bad = np.nonzero(d_p32 > np.maximum(1e-4, 2 * d_3264))[0]
        for k in bad:
            fails.append((key, pick[k], float(d_p32[k]), float(d_3264[k])))
    print(f"environments {lo} .. {pick[-1]} done", flush=True)
print(f"{'moment':>10} {'max d(p,o32)':>14} {'99.9 %':>10} {'max d(o32,o64)':>16}")
for key in MOMENT_KEYS:
    w = np.concatenate(worst[key])
    print(f"{key:>10} {w[:, 0].max():14.2e} {np.quantile(w[:, 0], 0.999):10.2e} {w[:, 1].max():16.2e}")
print("environments that fail the test's assertion:", len(fails))
for f in fails[:40]:
    print("  ", f)
