"""
CPU check of the KERNEL SOURCE for the per-element maps: lynx_amd/csrc/lynx_maps.hpp is
compiled for the host (tests/harness) and compared with the oracle, element kind by
element kind.  Tolerances: the two sides differ only in libm vs NumPy rounding and in FMA
use (fp32 ~1e-6, fp64 ~1e-13 relative to the map's scale).
"""

import numpy as np
import pytest

from lynx_amd import _ffi
from oracle import lynx_oracle as o

from .helpers import harness_map, map_err

TOL = {np.float32: 5e-6, np.float64: 5e-13}
RNG = np.random.default_rng(1234)
# dispersion entries dx = hx/kx2 * (1 - cos) cancel in fp32 (track_methods.py:80): the two
# sides agree to a few ulp of `cos`, i.e. ~1e-4 of a 1e-3-sized entry
DIPOLE_TOL = {np.float32: 3e-4, np.float64: 1e-12}


def _check(h, kind, flags, params, energy, spec, dtype, tol=None):
    got, _ = harness_map(h, kind, flags, params, energy, dtype)
    ref = o.element_transfer_map(spec, np.array([energy], dtype=dtype), dtype)[0]
    err = map_err(got, ref)
    assert err < (tol or TOL[dtype]), (err, got, ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_drift_and_correctors(host_harness, dtype):
    for energy in (1e8, 6e6, 0.0):
        for L in (0.0, 0.3, 7.0):
            a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
            _check(host_harness, _ffi.KIND_DRIFT, 0, [L], energy, o.Drift(a(L)), dtype)
            _check(host_harness, _ffi.KIND_HCOR, 0, [L, 1e-4], energy, o.HorizontalCorrector(a(L), a(1e-4)), dtype)
            _check(host_harness, _ffi.KIND_VCOR, 0, [L, -3e-3], energy, o.VerticalCorrector(a(L), a(-3e-3)), dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quadrupole(host_harness, dtype):
    a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    for _ in range(50):
        L, k1 = RNG.uniform(0.05, 1.0), RNG.choice([0.0, 4.2, -4.2, RNG.uniform(-30, 30)])
        tilt = RNG.choice([0.0, np.pi / 4, RNG.uniform(-3, 3)])
        mis = RNG.choice([0.0, 1.0]) * RNG.normal(0, 1e-3, 2)
        energy = RNG.choice([1e8, 6e6, 1.0732e8])
        flags = (_ffi.FLAG_TILT if tilt != 0 else 0) | (_ffi.FLAG_MISALIGNED if np.any(mis != 0) else 0)
        spec = o.Quadrupole(a(L), k1=a(k1), tilt=a(tilt), misalignment=np.asarray([mis], dtype=dtype))
        _check(host_harness, _ffi.KIND_QUADRUPOLE, flags, [L, k1, tilt, mis[0], mis[1]], energy, spec, dtype,
               tol=2e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_quadrupole_flag_set_by_other_samples(host_harness, dtype):
    """tilt == 0 / misalignment == 0 sample inside a batch where another sample is tilted:
    the rotation by 0 and the zero shift must leave the map unchanged (the whole-batch
    `if any(...)`, track_methods.py:101 / quadrupole.py:75)."""
    got, _ = harness_map(host_harness, _ffi.KIND_QUADRUPOLE, _ffi.FLAG_TILT | _ffi.FLAG_MISALIGNED,
                         [0.3, 4.2, 0.0, 0.0, 0.0], 1e8, dtype)
    plain, _ = harness_map(host_harness, _ffi.KIND_QUADRUPOLE, 0, [0.3, 4.2, 0.0, 0.0, 0.0], 1e8, dtype)
    assert np.array_equal(got, plain)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dipole_and_rbend(host_harness, dtype):
    a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    for _ in range(50):
        L = RNG.choice([0.0, RNG.uniform(0.1, 1.5)])
        angle = RNG.choice([0.0, RNG.uniform(-0.5, 0.5)])
        e1, e2 = RNG.uniform(-0.3, 0.3, 2) * RNG.choice([0, 1])
        tilt = RNG.choice([0.0, np.pi / 2, RNG.uniform(-1, 1)])
        fint, fintx, gap = RNG.uniform(0, 0.5), RNG.uniform(0, 0.5), RNG.uniform(0, 0.05)
        energy = RNG.choice([1e8, 6e6])
        flags = _ffi.FLAG_THICK if L != 0 else 0
        spec = o.Dipole(a(L), angle=a(angle), e1=a(e1), e2=a(e2), tilt=a(tilt), fringe_integral=a(fint),
                        fringe_integral_exit=a(fintx), gap=a(gap))
        _check(host_harness, _ffi.KIND_DIPOLE, flags, [L, angle, e1, e2, tilt, fint, fintx, gap], energy, spec,
               dtype, tol=DIPOLE_TOL[dtype])
        rspec = o.RBend(a(L), angle=a(angle), e1=a(e1), e2=a(e2), tilt=a(tilt), fringe_integral=a(fint),
                        fringe_integral_exit=a(fintx), gap=a(gap))
        h1 = dtype(e1) + dtype(angle) / 2  # rbend.py:79-80, applied on the host at construction
        h2 = dtype(e2) + dtype(angle) / 2
        _check(host_harness, _ffi.KIND_DIPOLE, flags, [L, angle, h1, h2, tilt, fint, fintx, gap], energy, rspec,
               dtype, tol=DIPOLE_TOL[dtype])


def test_dipole_thick_flag_from_other_samples(host_harness):
    """A zero-length sample inside a batch that has a thick sample goes through base_rmatrix
    (dipole.py:119 decides on the whole batch) and loses the thin kick [2,6]."""
    dtype = np.float64
    spec = o.Dipole(np.array([0.0, 0.5]), angle=np.array([0.01, 0.02]))
    ref = o.element_transfer_map(spec, np.array([1e8, 1e8]), dtype)
    got, _ = harness_map(host_harness, _ffi.KIND_DIPOLE, _ffi.FLAG_THICK, [0.0, 0.01, 0, 0, 0, 0, 0, 0], 1e8, dtype)
    assert map_err(got, ref[0]) < 1e-13 and got[2, 6] == 0.0
    thin, _ = harness_map(host_harness, _ffi.KIND_DIPOLE, 0, [0.0, 0.01, 0, 0, 0, 0, 0, 0], 1e8, dtype)
    assert thin[2, 6] == 0.01


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cavity_rmatrix_and_coefficients(host_harness, dtype):
    a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    for _ in range(40):
        L, V = RNG.uniform(0.5, 3.0), RNG.uniform(5e6, 5e7)
        phase, f = RNG.uniform(-30, 30), RNG.choice([1.3e9, 2.856e9])
        energy = RNG.choice([6e6, 1e8])
        spec = o.Cavity(a(L), voltage=a(V), phase=a(phase), frequency=a(f))
        flags = _ffi.FLAG_CAV_BETA | _ffi.FLAG_CAV_GAIN | _ffi.FLAG_CAV_T5XX
        got, coef = harness_map(host_harness, _ffi.KIND_CAVITY, flags, [L, V, phase, f], energy, dtype, True)
        ref = o.cavity_rmatrix(spec, a(energy), dtype)[0]
        tol = 3e-5 if dtype == np.float32 else 1e-11
        assert map_err(got, ref) < tol
        # the non-linear step through the coefficients == oracle `_track_beam` on particles
        P = o.gaussian_particles((1,), 64, seed=3, dtype=dtype, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-4, 1e-3])
        ref_out = o.cavity_track(spec, o.particle_beam(P, a(energy), dtype), dtype)["particles"][0]
        import ctypes as C
        ct = C.c_float if dtype == np.float32 else C.c_double
        kick = host_harness.harness_kick_f32 if dtype == np.float32 else host_harness.harness_kick_f64
        kick.argtypes = [C.c_void_p, ct, ct, C.POINTER(ct), C.POINTER(ct)]
        kick.restype = None
        lin = (P[0].astype(dtype) @ got.T.astype(dtype))
        for n in range(64):
            s_io, d_out = ct(float(lin[n, 4])), ct(0.0)
            kick(coef.ctypes.data, ct(float(P[0, n, 4])), ct(float(P[0, n, 5])), C.byref(s_io), C.byref(d_out))
            # delta carries the fp32 cancellation noise of cos(phi + eps) - cos(phi): absolute bound
            atol_d = (4e-7 if dtype == np.float32 else 1e-14) * abs(coef[1]) + 1e-5 * abs(ref_out[n, 5]) * (dtype == np.float32)
            assert abs(d_out.value - ref_out[n, 5]) <= atol_d + 1e-12 * abs(ref_out[n, 5]), (d_out.value, ref_out[n, 5])
            assert abs(s_io.value - ref_out[n, 4]) <= (2e-5 if dtype == np.float32 else 1e-11) * np.abs(ref_out[:, 4]).max()


def test_cavity_zero_voltage_is_nan_like_the_reference(host_harness):
    """V = 0 gives r12 = inf * 0 = NaN in the reference (cavity.py:269; guard removed at :73-76)."""
    got, _ = harness_map(host_harness, _ffi.KIND_CAVITY, 0, [1.0, 0.0, 0.0, 1.3e9], 1e8, np.float64)
    ref = o.cavity_rmatrix(o.Cavity(np.array([1.0]), voltage=np.array([0.0]), phase=np.array([0.0]),
                                    frequency=np.array([1.3e9])), np.array([1e8]), np.float64)[0]
    assert np.isnan(got[0, 1]) and np.isnan(ref[0, 1])
    assert map_err(got, ref) < 1e-13


def test_custom_and_identity(host_harness):
    tm = RNG.normal(size=(7, 7))
    got, _ = harness_map(host_harness, _ffi.KIND_CUSTOM, 0, tm.reshape(-1), 1e8, np.float64)
    assert np.array_equal(got, tm)
    got, _ = harness_map(host_harness, _ffi.KIND_IDENTITY, 0, [], 1e8, np.float64)
    assert np.array_equal(got, np.eye(7))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_solenoid_and_undulator(host_harness, dtype):
    a = lambda v: np.array([v], dtype=dtype)  # noqa: E731
    for L, k, mis, energy in ((0.5, 2.0, (0.0, 0.0), 1e8), (0.3, 0.0, (0.0, 0.0), 6e6), (0.4, -1.5, (1e-3, -2e-3), 1e8),
                              (0.2, 1.0, (0.0, 0.0), 0.0)):
        flags = _ffi.FLAG_MISALIGNED if any(mis) else 0
        spec = o.Solenoid(a(L), k=a(k), misalignment=np.asarray([mis], dtype=dtype))
        _check(host_harness, _ffi.KIND_SOLENOID, flags, [L, k, mis[0], mis[1]], energy, spec, dtype)
    for L, energy in ((0.25, 1e8), (1.0, 6e6), (0.5, 0.0)):
        _check(host_harness, _ffi.KIND_UNDULATOR, 0, [L], energy, o.Undulator(a(L)), dtype)


ENTRY_U = [0 * 7 + 0, 0 * 7 + 1, 0 * 7 + 6, 1 * 7 + 0, 1 * 7 + 1, 1 * 7 + 6, 2 * 7 + 2, 2 * 7 + 3, 2 * 7 + 6, 3 * 7 + 2, 3 * 7 + 3,
           3 * 7 + 6, 4 * 7 + 4, 4 * 7 + 5, 5 * 7 + 4, 5 * 7 + 5]  # lynx_unit_record.hpp: unit_entry_u


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_the_16_entry_builder_of_structured_maps_equals_the_7x7_builder(host_harness, dtype):
    """
    lynx_maps.hpp: build_entries_u gives the maps of the kinds with class-U structure (drift, correctors, undulator,
    untilted quadrupole with or without misalignment, cavity, identity) as 16 entries in registers; the reverse pass
    differentiates through it (k_build_bwd).  Bit for bit the entries of build_element at the pattern's positions, the
    rest of build_element's 7x7 exactly the identity's; every other kind is refused.
    """
    import ctypes as C

    from lynx_amd import _ffi

    ct = C.c_float if dtype == np.float32 else C.c_double
    fn = host_harness.harness_entries_u_f32 if dtype == np.float32 else host_harness.harness_entries_u_f64
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, ct, C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    cav = _ffi.FLAG_CAV_BETA | _ffi.FLAG_CAV_GAIN | _ffi.FLAG_CAV_T5XX
    cases = [(_ffi.KIND_IDENTITY, 0, [], 1e8), (_ffi.KIND_DRIFT, 0, [0.7], 6e6), (_ffi.KIND_HCOR, 0, [0.3, 1e-3], 1e8),
             (_ffi.KIND_VCOR, 0, [0.3, -2e-3], 6e6), (_ffi.KIND_UNDULATOR, 0, [0.8], 6e6),
             (_ffi.KIND_QUADRUPOLE, 0, [0.2, 4.2, 0.0, 0.0, 0.0], 1e8), (_ffi.KIND_QUADRUPOLE, 0, [0.2, 0.0, 0.0, 0.0, 0.0], 1e8),
             (_ffi.KIND_QUADRUPOLE, _ffi.FLAG_MISALIGNED, [0.1, -3.1, 0.0, 1e-3, -2e-3], 6e6),
             (_ffi.KIND_CAVITY, cav, [1.0377, 1.8e7, 5.0, 1.3e9], 6e6), (_ffi.KIND_CAVITY, cav, [1.0377, 1.2e7, -7.0, 1.3e9], 9e7),
             (_ffi.KIND_CAVITY, _ffi.FLAG_CAV_BETA, [1.0, 0.0, 3.0, 1.3e9], 1e8)]  # V = 0: NaN in r12, as the reference
    for kind, flags, p, energy in cases:
        want_coef = kind == _ffi.KIND_CAVITY
        M, coef = harness_map(host_harness, kind, flags, p, energy, dtype, want_coef)
        pp = np.ascontiguousarray(np.asarray(p if p else [0.0], dtype=dtype))
        m16, c8 = np.zeros(16, dtype), np.zeros(8, dtype)
        assert fn(kind, flags, pp.ctypes.data, ct(float(energy)), m16.ctypes.data, c8.ctypes.data if want_coef else None) == 1
        flat = M.reshape(-1)
        assert np.array_equal(m16, flat[ENTRY_U], equal_nan=True), (kind, flags, m16, flat[ENTRY_U])
        rest = np.delete(flat, ENTRY_U)
        assert np.array_equal(rest, np.delete(np.eye(7, dtype=dtype).reshape(-1), ENTRY_U)), (kind, flags)
        if want_coef:
            assert np.array_equal(c8, coef, equal_nan=True)
    for kind, flags, p in [(_ffi.KIND_QUADRUPOLE, _ffi.FLAG_TILT, [0.2, 1.0, 0.3, 0.0, 0.0]),
                           (_ffi.KIND_DIPOLE, _ffi.FLAG_THICK, [0.5, 0.1, 0, 0, 0, 0, 0, 0]), (_ffi.KIND_SOLENOID, 0, [0.5, 2.0, 0, 0]),
                           (_ffi.KIND_ROTATION, 0, [0.3])]:
        pp = np.ascontiguousarray(np.asarray(p, dtype=dtype))
        m16 = np.zeros(16, dtype)
        assert fn(kind, flags, pp.ctypes.data, ct(1e8), m16.ctypes.data, None) == 0
