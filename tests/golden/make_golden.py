#!/usr/bin/env python3
"""
Generates tests/golden/*.npz from the oracle (oracle/lynx_oracle.py), seeds recorded below.

The reference itself cannot produce vectors here: `import lynx` raises
ModuleNotFoundError('jax') (an ordinary Python error; jax/equinox are not installed and
there is no network), and its source is not executable JAX.  The
vectors therefore come from the NumPy restatement, which tests/test_oracle_kat.py pins
against the reference's own known answers.  Inputs AND expected outputs are stored, so the
fixtures stay meaningful if the oracle is ever edited.

    python tests/golden/make_golden.py
"""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import lynx_oracle as o  # noqa: E402

OUT = Path(__file__).resolve().parent

ELEMENT_CASES = {
    "drift": dict(length=[0.3, 1.0, 0.0]),
    "hcor": dict(length=[0.3, 0.1, 0.2], angle=[1e-4, 0.0, -2e-3]),
    "vcor": dict(length=[0.3, 0.1, 0.2], angle=[3.142e-3, 0.0, -2e-3]),
    "quadrupole": dict(length=[0.2, 0.1, 0.3], k1=[4.2, -4.2, 0.0]),
    "quadrupole_tilt_mis": dict(length=[1.0, 1.0, 1.0], k1=[1.0, -2.0, 3.0],
                                misalignment=[[0.1, 0.1], [0.0, 0.0], [-1e-3, 2e-3]], tilt=[0.0, 0.3, 0.785398]),
    "dipole": dict(length=[0.5, 0.0, 1.0], angle=[0.1, 0.01, -0.3], e1=[0.05, 0.0, 0.1], e2=[0.02, 0.0, -0.1],
                   tilt=[0.0, 0.2, 1.570796], fringe_integral=[0.3, 0.0, 0.5], gap=[0.02, 0.0, 0.03]),
    "dipole_thin": dict(length=[0.0, 0.0, 0.0], angle=[0.1, 0.01, -0.3]),
    "rbend": dict(length=[0.5, 0.4, 1.0], angle=[0.1, 0.01, -0.3], e1=[0.0, 0.01, 0.0],
                  fringe_integral=[0.3, 0.0, 0.5], fringe_integral_exit=[0.1, 0.2, 0.5], gap=[0.02, 0.0, 0.03]),
    "cavity": dict(length=[1.0377, 3.0441, 1.0], voltage=[0.01815975e9, 48198468.0, 1e6], phase=[0.0, 30.0, -10.0],
                   frequency=[1.3e9, 2.856e9, 1.3e9]),
}
ENERGY = [1e8, 6e6, 1.0732e8]
CTOR = {"drift": o.Drift, "hcor": o.HorizontalCorrector, "vcor": o.VerticalCorrector, "quadrupole": o.Quadrupole,
        "quadrupole_tilt_mis": o.Quadrupole, "dipole": o.Dipole, "dipole_thin": o.Dipole, "rbend": o.RBend,
        "cavity": o.Cavity}


def mixed_lattice(dtype, B, rng):
    f = lambda v: np.full(B, v, dtype=dtype)  # noqa: E731
    r = lambda lo, hi: rng.uniform(lo, hi, B).astype(dtype)  # noqa: E731
    desc = [("drift", dict(length=f(0.6))),
            ("quadrupole", dict(length=f(0.2), k1=r(-5, 5), tilt=r(-1, 1), misalignment=rng.normal(0, 1e-4, (B, 2)).astype(dtype))),
            ("marker", {}),
            ("dipole", dict(length=f(0.5), angle=r(-0.2, 0.2), e1=f(0.05), e2=f(0.02), fringe_integral=f(0.4), gap=f(0.02), tilt=f(0.1))),
            ("hcor", dict(length=f(0.1), angle=r(-1e-4, 1e-4))),
            ("cavity", dict(length=f(1.0377), voltage=r(5e6, 2e7), phase=r(-10, 10), frequency=f(1.3e9))),
            ("rbend", dict(length=f(0.3), angle=f(0.05))),
            ("vcor", dict(length=f(0.1), angle=r(-1e-4, 1e-4))),
            ("bpm", {}),
            ("cavity", dict(length=f(1.0377), voltage=r(5e6, 2e7), phase=f(0.0), frequency=f(1.3e9))),
            ("drift", dict(length=f(0.4)))]
    return desc


def to_specs(desc):
    ctor = {"drift": o.Drift, "quadrupole": o.Quadrupole, "dipole": o.Dipole, "rbend": o.RBend,
            "hcor": o.HorizontalCorrector, "vcor": o.VerticalCorrector, "cavity": o.Cavity, "bpm": o.BPM,
            "marker": o.Marker}
    return [ctor[k](**kw) for k, kw in desc]


def main():
    store = {}
    for dtype in (np.float32, np.float64):
        tag = np.dtype(dtype).name
        energy = np.asarray(ENERGY, dtype=dtype)
        # 1. single-element maps
        for name, kw in ELEMENT_CASES.items():
            kw_t = {k: np.asarray(v, dtype=dtype) for k, v in kw.items()}
            store[f"map/{name}/{tag}"] = o.element_transfer_map(CTOR[name](**kw_t), energy, dtype)
        # 2. composed maps: README segment (C1/C2) and the 128-element FODO (C3/C4) with a k1 scan
        store[f"composed/ares/{tag}"] = o.segment_transfer_map(o.ares_like_segment(dtype, (1,)), np.array([1e8], dtype), dtype)
        scale = np.linspace(0.5, 1.5, 4).astype(dtype)
        store[f"composed/fodo128/{tag}"] = o.segment_transfer_map(
            o.fodo_segment(32, np.dtype(dtype).type, (4,), scale), np.full(4, 1e8, dtype), dtype)
        # 3. tracked particles through a lattice with every element kind and two active cavities
        B, N = 3, 256
        rng = np.random.default_rng(20240607)
        desc = mixed_lattice(dtype, B, rng)
        for i, (k, kw) in enumerate(desc):
            for pk, pv in kw.items():
                store[f"mixed/elem{i:02d}_{k}/{pk}/{tag}"] = pv
        P = o.gaussian_particles((B,), N, seed=11, dtype=dtype, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-4, 1e-3])
        e_in = np.full(B, 6e6, dtype=dtype)
        out = o.segment_track(to_specs(desc), o.particle_beam(P, e_in, dtype), dtype)
        store[f"mixed/particles_in/{tag}"] = P
        store[f"mixed/particles_out/{tag}"] = out["particles"]
        store[f"mixed/energy_out/{tag}"] = out["energy"]
        # ... and the same lattice on a ParameterBeam
        pb = o.parameter_beam_from_parameters(dtype=dtype, sigma_x=np.full(B, 1e-4, dtype), sigma_xp=np.full(B, 1e-5, dtype),
                                              mu_x=np.asarray([1e-4, -2e-4, 0.0], dtype), energy=e_in)
        pout = o.segment_track(to_specs(desc), pb, dtype)
        store[f"mixed/mu_out/{tag}"] = pout["mu"]
        store[f"mixed/cov_out/{tag}"] = pout["cov"]
        # 4. moments of a 100k-particle beam through the README segment (C2), values only
        P2 = o.gaussian_particles((1,), 100_000, seed=0, dtype=dtype)
        out2 = o.segment_track(o.ares_like_segment(dtype, (1,)), o.particle_beam(P2, np.array([1e8], dtype), dtype), dtype)
        m = o.beam_moments(out2, ddof=1)
        store[f"c2/moments/{tag}"] = np.array([m[k][0] for k in (
            "mu_x", "mu_xp", "mu_y", "mu_yp", "mu_s", "mu_p", "sigma_x", "sigma_xp", "sigma_y", "sigma_yp", "sigma_s",
            "sigma_p", "sigma_xxp", "sigma_yyp")], dtype=np.float64)
    np.savez_compressed(OUT / "lynx_golden.npz", **store)
    print("wrote", OUT / "lynx_golden.npz", f"{(OUT / 'lynx_golden.npz').stat().st_size / 1024:.0f} KiB", len(store), "arrays")


if __name__ == "__main__":
    main()
