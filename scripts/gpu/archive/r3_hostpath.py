"""Round 3: where the host spends its time per `Segment.track` call (BASELINE config 3 and the 128-sample shard of config 4):
enqueue-only wall time per call against the step time, and cProfile's top entries."""
import cProfile
import io
import pstats
import sys
import time

sys.path.insert(0, ".")
import lynx_amd as lx  # noqa: E402
import numpy as np  # noqa: E402

rt = lx.device.get_runtime()


def run(label, segment, beam, calls=300):
    for _ in range(20):
        out = segment.track(beam)
    rt.sync()
    t0 = time.perf_counter()
    for _ in range(calls):
        out = segment.track(beam)
    t1 = time.perf_counter()
    rt.sync()
    t2 = time.perf_counter()
    print(f"{label}: enqueue {1e6 * (t1 - t0) / calls:.1f} us/call, with the final wait {1e6 * (t2 - t0) / calls:.1f} us/call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(calls):
        out = segment.track(beam)
    pr.disable()
    rt.sync()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
    print("\n".join(s.getvalue().splitlines()[:34]))
    del out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    if which == "c3":
        B, N, dtype = 1, 1_000_000, np.float64
    elif which == "c2":
        B, N, dtype = 1, 100_000, np.float32
    else:
        B, N, dtype = 128, 100_000, np.float32
    f = lambda v: np.full(B, v, dtype=dtype)  # noqa: E731
    rng = np.random.default_rng(3)
    elements = []
    for c in range(64):  # FODO cells: 128 elements
        elements += [lx.Drift(f(0.5), dtype=dtype), lx.Quadrupole(f(0.2), k1=rng.uniform(-3, 3, B).astype(dtype), dtype=dtype)]
    segment = lx.Segment(elements)
    beam = lx.ParticleBeam.from_parameters(num_particles=N, energy=f(1e8), dtype=dtype, seed=1)
    run(which, segment, beam)
