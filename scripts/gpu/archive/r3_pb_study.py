#!/usr/bin/env python3
"""
Round 3: where the float32 ParameterBeam results behind cavities stand -- product vs float32 oracle, product vs
float64 oracle, float32 oracle vs float64 oracle, per entry of mu and of the 6 x 6 covariance, in units of
(|mu_i| + sigma_i) and sigma_i sigma_j (the scale tests/test_gpu_parity.py asserts in).  Three lattices: the one of
test_parameter_beam_through_mixed_lattice, the one of test_parameter_beam_lanes_path..., the golden `mixed` lattice.
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import lynx_amd as lx  # noqa: E402
from helpers import make_lattice  # noqa: E402
from oracle import lynx_oracle as o  # noqa: E402

np.set_printoptions(linewidth=220, precision=1)


def dist(mu, cov, ref):
    sig = np.sqrt(np.abs(np.einsum("...ii->...i", ref["cov"][..., :6, :6])))
    dmu = np.max((np.abs(mu[..., :6] - ref["mu"][..., :6]) / (np.abs(ref["mu"][..., :6]) + sig + 1e-300)).reshape(-1, 6), axis=0)
    sc = sig[..., :, None] * sig[..., None, :] + 1e-300
    dcov = np.max((np.abs(cov[..., :6, :6] - ref["cov"][..., :6, :6]) / sc).reshape(-1, 6, 6), axis=0)
    return dmu, dcov


def study(name, desc, kw, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    f32 = lambda d: [(k, {a: np.asarray(v, dtype=np.float32) for a, v in kw_.items()}) for k, kw_ in d]  # noqa: E731
    d32 = f32(desc)
    k32 = {a: np.asarray(v, dtype=np.float32) for a, v in kw.items()}
    elements, specs32 = make_lattice(d32, np.float32, lx)
    out = lx.Segment(elements).track(lx.ParameterBeam.from_parameters(**k32, dtype=np.float32))
    mu, cov = np.asarray(out._mu, dtype=np.float64), np.asarray(out._cov, dtype=np.float64)
    r32 = o.segment_track(specs32, o.parameter_beam_from_parameters(dtype=np.float32, **k32), np.float32)
    d64 = [(k, {a: np.asarray(v, dtype=np.float64) for a, v in kw_.items()}) for k, kw_ in d32]
    _, specs64 = make_lattice(d64, np.float64)
    r64 = o.segment_track(specs64, o.parameter_beam_from_parameters(dtype=np.float64, **{a: v.astype(np.float64) for a, v in k32.items()}), np.float64)
    r32d = {"mu": r32["mu"].astype(np.float64), "cov": r32["cov"].astype(np.float64)}
    print(f"==== {name}")
    for label, (a, b) in (("product vs oracle32", ((mu, cov), r32d)), ("product vs oracle64", ((mu, cov), r64)),
                          ("oracle32 vs oracle64", ((r32d["mu"], r32d["cov"]), r64))):
        dmu, dcov = dist(a[0], a[1], b)
        print(f"-- {label}: mu {dmu}  max {dmu.max():.1e};  cov max {dcov.max():.1e}")
        print(dcov)
    for k in (env or {}):
        os.environ.pop(k)


B = 7
rng = np.random.default_rng(9)
f = lambda v: np.full(B, v)  # noqa: E731
desc = [("drift", dict(length=f(0.6))), ("quadrupole", dict(length=f(0.2), k1=rng.uniform(-5, 5, B))),
        ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B), frequency=f(1.3e9))),
        ("drift", dict(length=f(0.4))), ("hcor", dict(length=f(0.1), angle=f(1e-4))),
        ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=f(0.0), frequency=f(1.3e9))),
        ("dipole", dict(length=f(0.5), angle=f(0.1)))]
kw = dict(sigma_x=f(1e-4), sigma_xp=f(1e-5), sigma_y=f(1e-4), sigma_yp=f(1e-5), sigma_s=f(1e-5), sigma_p=f(1e-3),
          mu_x=rng.normal(0, 1e-4, B), energy=f(6e6))
study("mixed lattice, B = 7 (k_track_moments)", desc, kw)

B = 300
rng = np.random.default_rng(12)
desc = [("drift", dict(length=f(0.6))), ("quadrupole", dict(length=f(0.2), k1=rng.uniform(-5, 5, B), tilt=rng.uniform(-1, 1, B))),
        ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=rng.uniform(-10, 10, B), frequency=f(1.3e9))),
        ("drift", dict(length=f(0.4))), ("hcor", dict(length=f(0.1), angle=f(1e-4))),
        ("cavity", dict(length=f(1.0377), voltage=rng.uniform(5e6, 2e7, B), phase=f(0.0), frequency=f(1.3e9))),
        ("dipole", dict(length=f(0.5), angle=f(0.1)))]
for _ in range(3):
    desc += [("quadrupole", dict(length=f(0.2), k1=rng.uniform(-5, 5, B))), ("drift", dict(length=f(0.5))), ("vcor", dict(length=f(0.1), angle=f(1e-4)))]
kw = dict(sigma_x=f(1e-4), sigma_xp=f(1e-5), sigma_y=f(1e-4), sigma_yp=f(1e-5), sigma_s=f(1e-5), sigma_p=f(1e-3),
          mu_x=rng.normal(0, 1e-4, B), energy=f(6e6))
study("lanes-path lattice, B = 300, lanes = samples", desc, kw, {"LYNX_LANES_BUILD_MIN_BATCH": "1"})
study("lanes-path lattice, B = 300, one workgroup per sample", desc, kw, {"LYNX_LANES_BUILD_MIN_BATCH": "1000000"})

sys.path.insert(0, str(ROOT / "tests" / "golden"))
import make_golden as mg  # noqa: E402

gd = mg.mixed_lattice(np.float32, 3, np.random.default_rng(20240607))
kw = dict(sigma_x=np.full(3, 1e-4), sigma_xp=np.full(3, 1e-5), mu_x=np.asarray([1e-4, -2e-4, 0.0]), energy=np.full(3, 6e6))
study("golden mixed lattice (tests/golden/make_golden.py)", gd, kw)
