"""Round 4: where the float32 reverse pass's distance from the float64 pass comes from on BASELINE config 5's lattice.
The same 64 environments, 10 000 particles, three bunch lengths: the absolute float32 noise of the forward kick
(cavity.py:150-160: cos(phi + eps) - cos(phi)) does not depend on the bunch length, the phase gradient's signal does."""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402
import lynx_amd as lx  # noqa: E402
import lynx_amd.grad  # noqa: E402,F401
from oracle import lynx_oracle as o  # noqa: E402
from tests.helpers import make_lattice  # noqa: E402

B = 64
ids = np.arange(B)
desc = bench.describe("c5", ids, 8, np.float32, seed=3)
rng = np.random.default_rng(17)
w_mu, w_cov = rng.normal(size=(B, 6)), rng.normal(size=(B, 6, 6)) * 1e3
names = {"drift": ["length"], "quadrupole": ["length", "k1", "misalignment"], "cavity": ["length", "voltage", "phase", "frequency"]}


def gradients(dtype, P, wm=w_mu, wc=w_cov):
    d = [(k, {a: np.asarray(v).astype(dtype) for a, v in kw.items()}) for k, kw in desc]
    elements, _ = make_lattice(d, dtype, lx)
    g = lx.grad.track_vjp(lx.Segment(elements), lx.ParticleBeam(P.astype(dtype), np.full(B, 6e6, dtype), dtype=dtype))(mu_bar=wm, cov_bar=wc)
    out = {(e, n): np.asarray(g[elements[e]][n], dtype=np.float64) for e, (k, _) in enumerate(d) for n in names[k]}
    out["energy"] = np.asarray(g.energy, dtype=np.float64)
    return out


def distance(g, r):
    return float(np.max(np.abs(g - r) / (np.abs(r) + 1e-3 * np.max(np.abs(r)) + 1e-300)))


for which, (wm, wc) in {"random weights": (w_mu, w_cov), "var(x) only (bench --grad)": (np.zeros((B, 6)), np.eye(6)[None, :1, :].repeat(B, 0) * np.eye(6)[None, :, :1].repeat(B, 0).transpose(0, 2, 1) * 0 + np.pad(np.ones((B, 1, 1)), ((0, 0), (0, 5), (0, 5))))}.items():
    for sigma_s in (1e-5, 1e-4, 1e-3):
        sig = list(bench.BEAM_SIGMA)
        sig[4] = sigma_s
        P = o.gaussian_particles((B,), 10_000, seed=2, dtype=np.float32, sigma=sig)
        g32, g64 = gradients(np.float32, P, wm, wc), gradients(np.float64, P, wm, wc)
        worst = {}
        for key, ref in g64.items():
            name = key if isinstance(key, str) else key[1]
            worst[name] = max(worst.get(name, 0.0), distance(g32[key], ref))
        print(f"{which}, sigma_s = {sigma_s:g}:", {k: f"{v:.1e}" for k, v in worst.items()}, flush=True)
