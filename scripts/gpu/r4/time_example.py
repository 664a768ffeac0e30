import time, numpy as np, sys
sys.path.insert(0, '.')
sys.path.insert(0, 'examples')
import gradient_based_tuning as ex
import lynx_amd as lx
f = ex.f
def setup():
    segment = ex.ares_ea()
    segment.AREAMQZM1.k1, segment.AREAMQZM2.k1, segment.AREAMQZM3.k1 = f(5.0), f(-5.0), f(5.0)
    segment.AREAMCVM1.angle, segment.AREAMCHM1.angle = f(1e-3), f(-1e-3)
    beam = lx.ParticleBeam.from_parameters(num_particles=100_000, sigma_x=f(1.75e-4), sigma_xp=f(3.7e-6), sigma_y=f(1.75e-4), sigma_yp=f(3.7e-6), sigma_s=f(8e-6), sigma_p=f(2.3e-3), energy=f(1.07e8), seed=0)
    return segment, beam
segment, beam = setup()
ex.tune(segment, beam, steps=20)
for rep in range(3):
    segment, beam = setup()
    t0 = time.perf_counter()
    h = ex.tune(segment, beam, steps=200)
    dt = time.perf_counter() - t0
    print(f"200 Adam steps: {dt*1e3:.1f} ms = {dt/200*1e6:.1f} us per step (forward 100 000 particles, four moments read, reverse, five gradients read, five settings written); loss {h[0]:.3g} -> {h[-1]:.3g}")
import cProfile, pstats
segment, beam = setup()
pr = cProfile.Profile(); pr.enable(); ex.tune(segment, beam, steps=200); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
