"""
CPU check of the dual-number builders (lynx_amd/csrc/lynx_dual.hpp) that the gradient kernels
use for dM/dtheta: derivative of every map entry and cavity coefficient against central
finite differences of the plain builders, for every element kind and parameter.
"""

import ctypes as C

import numpy as np
import pytest

from lynx_amd import _ffi

from .helpers import harness_map

CASES = [
    (_ffi.KIND_DRIFT, 0, [0.7], 1e8),
    (_ffi.KIND_HCOR, 0, [0.3, 1e-3], 6e6),
    (_ffi.KIND_VCOR, 0, [0.3, -2e-3], 1e8),
    (_ffi.KIND_QUADRUPOLE, 0, [0.2, 4.2, 0.0, 0.0, 0.0], 1e8),
    (_ffi.KIND_QUADRUPOLE, _ffi.FLAG_TILT | _ffi.FLAG_MISALIGNED, [0.3, -3.1, 0.4, 1e-3, -2e-3], 6e6),
    (_ffi.KIND_DIPOLE, _ffi.FLAG_THICK, [0.5, 0.12, 0.05, 0.02, 0.3, 0.4, 0.2, 0.03], 1e8),
    (_ffi.KIND_DIPOLE, 0, [0.0, 0.02, 0.0, 0.0, 0.1, 0.0, 0.0, 0.0], 1e8),
    (_ffi.KIND_CAVITY, _ffi.FLAG_CAV_BETA | _ffi.FLAG_CAV_GAIN | _ffi.FLAG_CAV_T5XX, [1.0377, 1.8e7, 5.0, 1.3e9], 6e6),
    (_ffi.KIND_CAVITY, _ffi.FLAG_CAV_BETA | _ffi.FLAG_CAV_GAIN, [1.0, -3e6, 20.0, 2.856e9], 1e8),
    (_ffi.KIND_BASE_RMATRIX, _ffi.FLAG_TILT, [0.4, 2.0, 0.3, 0.2], 1e8),
    (_ffi.KIND_ROTATION, 0, [0.7], 1e8),
    (_ffi.KIND_SOLENOID, _ffi.FLAG_MISALIGNED, [0.5, 2.0, 1e-3, -2e-3], 6e6),
    (_ffi.KIND_UNDULATOR, 0, [0.8], 6e6),
]


def _dual(h, kind, flags, p, energy, seed, want_coef):
    fn = h.harness_build_dual_f64
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    fn.restype = None
    p = np.ascontiguousarray(p, dtype=np.float64)
    M, dM, c, dc = np.zeros(49), np.zeros(49), np.zeros(8), np.zeros(8)
    fn(kind, flags, p.ctypes.data, len(p), energy, seed, M.ctypes.data, dM.ctypes.data, c.ctypes.data,
       dc.ctypes.data, int(want_coef))
    return M, dM, c, dc


@pytest.mark.parametrize("case", range(len(CASES)))
def test_dual_derivatives_match_finite_differences(host_harness, case):
    kind, flags, p, energy = CASES[case]
    want_coef = kind == _ffi.KIND_CAVITY
    p = np.asarray(p, dtype=np.float64)
    M0, c0 = harness_map(host_harness, kind, flags, p, energy, np.float64, want_coef)
    for seed in range(len(p) + 1):
        M, dM, c, dc = _dual(host_harness, kind, flags, p, energy, seed, want_coef)
        # values agree up to the dual's unfused a*b+c in place of fma
        assert np.allclose(M, M0.reshape(-1), rtol=1e-13, atol=1e-15, equal_nan=True) and np.allclose(c, c0, rtol=1e-13, equal_nan=True)
        x0 = p[seed] if seed < len(p) else energy
        if seed < len(p) and p[seed] == 0.0 and kind == _ffi.KIND_QUADRUPOLE and seed == 1:
            continue  # k1 == 0 is replaced by 1e-12 (track_methods.py:67-68): derivative 0 by construction
        h = 1e-6 * max(abs(x0), 1e-3)

        def at(x):
            q, e = p.copy(), energy
            if seed < len(p):
                q[seed] = x
            else:
                e = x
            Mx, cx = harness_map(host_harness, kind, flags, q, e, np.float64, want_coef)
            return Mx.reshape(-1), cx

        (Mp, cp), (Mm, cm) = at(x0 + h), at(x0 - h)
        fdM, fdc = (Mp - Mm) / (2 * h), (cp - cm) / (2 * h)
        for got, fd, what in ((dM, fdM, "map"), (dc, fdc, "coef")):
            scale = np.max(np.abs(fd)) + 1e-300
            assert np.max(np.abs(got - fd)) <= 2e-5 * scale + 1e-12 * np.max(np.abs(got)), (what, seed, got, fd)


def _dual32(h, kind, flags, p, energy, seed):
    fn = h.harness_build_dual_f32
    fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 4 + [C.c_int]
    fn.restype = None
    p = np.ascontiguousarray(p, dtype=np.float32)
    M, dM, c, dc = (np.zeros(n, np.float32) for n in (49, 49, 8, 8))
    fn(kind, flags, p.ctypes.data, len(p), float(energy), seed, M.ctypes.data, dM.ctypes.data, c.ctypes.data, dc.ctypes.data, 1)
    return M, dM, c, dc


@pytest.mark.parametrize("energy", [6e6, 3e7, 8e7, 1.2e8])
def test_float32_map_derivatives_of_the_cavity_do_not_cancel(host_harness, energy):
    """
    cavity.py:296-305: r55_cor carries the bracket g0 g1 (beta0 beta1 - 1) + 1, which cancels twice on an
    ultra-relativistic beam; evaluated in float32 as the reference writes it, d M[4][4] / d(V, phase, f, L, E) was 15-35 %
    off at gamma ~ 150-230 (BASELINE config 5's later cavities) -- and with it every float32 gradient w.r.t. a cavity
    parameter of a loss that sees the (s, delta) plane.  The dual-number instantiation uses an equal form without the
    cancellation (lynx_dual.hpp: cavity_r55_bracket_dual): every derivative of every map entry within 1e-4 of the
    float64 evaluation.  (The forward builders keep the reference's operations.)
    """
    flags = _ffi.FLAG_CAV_BETA | _ffi.FLAG_CAV_GAIN | _ffi.FLAG_CAV_T5XX
    p = np.array([1.0377, 1.5e7, 7.0, 1.3e9], dtype=np.float32)
    e32 = np.float32(energy)
    for seed in range(5):
        _, dM32, _, _ = _dual32(host_harness, _ffi.KIND_CAVITY, flags, p, e32, seed)
        _, dM64, _, _ = _dual(host_harness, _ffi.KIND_CAVITY, flags, p.astype(np.float64), float(e32), seed, True)
        assert np.all(np.abs(dM32 - dM64) <= 1e-4 * np.abs(dM64) + 1e-30), (seed, dM32[32], dM64[32])
