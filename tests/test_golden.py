"""
Committed golden vectors (tests/golden/lynx_golden.npz, made by tests/golden/make_golden.py
from the oracle).  CPU: the oracle still reproduces them (guards against silent edits).
GPU: the HIP path reproduces them through the C ABI.
"""

from pathlib import Path

import numpy as np
import pytest

from oracle import lynx_oracle as o
from tests.golden import make_golden as mg

from .helpers import assert_parameter_beam, map_err, rel_err

G = np.load(Path(__file__).parent / "golden" / "lynx_golden.npz")
DTYPES = [np.float32, np.float64]


def _mixed_desc(tag):
    desc = {}
    for key in G.files:
        parts = key.split("/")
        if parts[0] == "mixed" and parts[1].startswith("elem") and parts[-1] == tag:
            idx, kind = parts[1][4:6], parts[1][7:]
            desc.setdefault((int(idx), kind), {})[parts[2]] = G[key]
    out = []
    n = max(i for i, _ in desc) + 1 if desc else 0
    found = {i: (k, kw) for (i, k), kw in desc.items()}
    # parameter-less elements (marker, bpm) are not in the file: rebuild positions from the generator
    template = mg.mixed_lattice(np.float32, 3, np.random.default_rng(0))
    for i, (k, _) in enumerate(template):
        out.append((k, found[i][1]) if i in found else (k, {}))
    return out


@pytest.mark.parametrize("dtype", DTYPES)
def test_oracle_reproduces_golden(dtype):
    tag = np.dtype(dtype).name
    energy = np.asarray(mg.ENERGY, dtype=dtype)
    for name, kw in mg.ELEMENT_CASES.items():
        kw_t = {k: np.asarray(v, dtype=dtype) for k, v in kw.items()}
        got = o.element_transfer_map(mg.CTOR[name](**kw_t), energy, dtype)
        assert np.array_equal(got, G[f"map/{name}/{tag}"], equal_nan=True), name
    desc = _mixed_desc(tag)
    out = o.segment_track(mg.to_specs(desc), o.particle_beam(G[f"mixed/particles_in/{tag}"], np.full(3, 6e6, dtype), dtype), dtype)
    assert np.array_equal(out["particles"], G[f"mixed/particles_out/{tag}"])
    assert np.array_equal(out["energy"], G[f"mixed/energy_out/{tag}"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_gpu_reproduces_golden(dtype, built_library):
    import lynx_amd as lx

    from .helpers import make_lattice

    tag = np.dtype(dtype).name
    energy = np.asarray(mg.ENERGY, dtype=dtype)
    ctor = {"drift": lx.Drift, "hcor": lx.HorizontalCorrector, "vcor": lx.VerticalCorrector, "quadrupole": lx.Quadrupole,
            "quadrupole_tilt_mis": lx.Quadrupole, "dipole": lx.Dipole, "dipole_thin": lx.Dipole, "rbend": lx.RBend,
            "cavity": lx.Cavity}
    for name, kw in mg.ELEMENT_CASES.items():
        kw_t = {k: np.asarray(v, dtype=dtype) for k, v in kw.items()}
        got = ctor[name](**kw_t, dtype=dtype).transfer_map(energy)
        tol = (5e-4 if dtype == np.float32 else 1e-11)
        assert map_err(got, G[f"map/{name}/{tag}"]) < tol, (name, map_err(got, G[f"map/{name}/{tag}"]))
    # composed maps
    ares, _ = make_lattice([("bpm", {}), ("drift", dict(length=[1.0])), ("bpm", {}), ("drift", dict(length=[1.0])),
                            ("vcor", dict(length=[0.3], angle=[3.142e-3])), ("drift", dict(length=[0.2])),
                            ("hcor", dict(length=[0.3], angle=[1e-4])), ("drift", dict(length=[7.0])),
                            ("hcor", dict(length=[0.3], angle=[-1e-4])), ("drift", dict(length=[0.05])), ("bpm", {})],
                           dtype, lx)
    got = lx.Segment(ares).transfer_map(np.array([1e8], dtype))
    assert map_err(got, G[f"composed/ares/{tag}"]) < (1e-5 if dtype == np.float32 else 1e-12)
    scale = np.linspace(0.5, 1.5, 4).astype(dtype)
    f = lambda v: np.full(4, v, dtype=dtype)  # noqa: E731
    fodo = []
    for _ in range(32):
        fodo += [lx.Quadrupole(f(0.2), k1=dtype(4.2) * scale, dtype=dtype), lx.Drift(f(0.5), dtype=dtype),
                 lx.Quadrupole(f(0.2), k1=-(dtype(4.2) * scale), dtype=dtype), lx.Drift(f(0.5), dtype=dtype)]
    got = lx.Segment(fodo).transfer_map(np.full(4, 1e8, dtype))
    assert map_err(got, G[f"composed/fodo128/{tag}"]) < (5e-4 if dtype == np.float32 else 1e-10)
    # tracked particles + ParameterBeam through the all-kinds lattice
    elements, _ = make_lattice(_mixed_desc(tag), dtype, lx)
    seg = lx.Segment(elements)
    out = seg.track(lx.ParticleBeam(G[f"mixed/particles_in/{tag}"], np.full(3, 6e6, dtype), dtype=dtype))
    got, ref = np.asarray(out.particles), G[f"mixed/particles_out/{tag}"]
    tol = {np.float32: [1e-4] * 4 + [1e-4, 5e-4, 1e-6], np.float64: [1e-9] * 7}[dtype]
    for c in range(7):
        assert rel_err(got[..., c], ref[..., c]) < tol[c], (c, rel_err(got[..., c], ref[..., c]))
    assert rel_err(out.energy, G[f"mixed/energy_out/{tag}"]) < 1e-6
    pb = lx.ParameterBeam.from_parameters(sigma_x=np.full(3, 1e-4, dtype), sigma_xp=np.full(3, 1e-5, dtype),
                                          mu_x=np.asarray([1e-4, -2e-4, 0.0], dtype), energy=np.full(3, 6e6, dtype), dtype=dtype)
    pout = seg.track(pb)
    # mu AND the covariance, entry by entry at north_star's tolerance.  The float32 fixture itself sits 6e-3 from the
    # float64 one in mu_p here (the kick's difference of two float32 cosines, cavity.py:150-160); the product forms that
    # difference without the cancellation (device_cavity_kick) and lands on the float64 fixture: every entry within the
    # tolerance of the float32 fixture or of the float64 one
    assert_parameter_beam(pout, {"mu": G[f"mixed/mu_out/{tag}"], "cov": G[f"mixed/cov_out/{tag}"]},
                          1e-4 if dtype == np.float32 else 1e-9,
                          alt={"mu": G["mixed/mu_out/float64"], "cov": G["mixed/cov_out/float64"]} if dtype == np.float32 else None)
    # C2 moments at N = 100k
    P2 = o.gaussian_particles((1,), 100_000, seed=0, dtype=dtype)
    out2 = lx.Segment(ares).track(lx.ParticleBeam(P2, np.array([1e8], dtype), dtype=dtype))
    gm = G[f"c2/moments/{tag}"]
    names = ("mu_x", "mu_xp", "mu_y", "mu_yp", "mu_s", "mu_p", "sigma_x", "sigma_xp", "sigma_y", "sigma_yp", "sigma_s",
             "sigma_p", "sigma_xxp", "sigma_yyp")
    tol = 1e-4 if dtype == np.float32 else 1e-6
    for i, n in enumerate(names):
        scale_ = abs(gm[i]) if n.startswith("sigma_") and not n.endswith(("xxp", "yyp")) else None
        if n.startswith("mu_"):
            sig = gm[6 + i]
            assert abs(float(getattr(out2, n)[0]) - gm[i]) <= tol * (abs(gm[i]) + sig), n
        elif scale_ is not None:
            assert abs(float(getattr(out2, n)[0]) - gm[i]) <= tol * scale_, n
        else:
            a, b = (gm[6], gm[7]) if n == "sigma_xxp" else (gm[8], gm[9])
            assert abs(float(getattr(out2, n)[0]) - gm[i]) <= tol * a * b, n
