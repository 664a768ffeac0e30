"""
ctypes binding of liblynxhip.so (include/lynx_hip.h).  This is the only place the Python
package touches native code.  There is NO fallback: if the library is missing, or no
gfx950 device is visible, every compute entry point raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "_lib" / "liblynxhip.so"

UNIQUE_ID_BYTES = 128
MOMENT_STRIDE = 36
STEP_STRIDE = 64

F32, F64 = 0, 1

KIND_IDENTITY, KIND_DRIFT, KIND_QUADRUPOLE, KIND_DIPOLE = 0, 1, 2, 3
KIND_HCOR, KIND_VCOR, KIND_CAVITY, KIND_CUSTOM = 4, 5, 6, 7
KIND_BASE_RMATRIX, KIND_ROTATION, KIND_MISALIGNMENT = 8, 9, 10
KIND_SOLENOID, KIND_UNDULATOR = 11, 12
PARAMS_OF_KIND = {0: 0, 1: 1, 2: 5, 3: 8, 4: 2, 5: 2, 6: 4, 7: 49, 8: 4, 9: 1, 10: 3, 11: 4, 12: 1}

FLAG_TILT, FLAG_MISALIGNED, FLAG_THICK = 1, 2, 4
FLAG_CAV_BETA, FLAG_CAV_GAIN, FLAG_CAV_T5XX = 8, 16, 32

STEP_RUN, STEP_CAVITY = 0, 1
STEP_FLAG_RAW = 64
STEP_FLAG_OBSERVE = 128
MAX_OBSERVERS = 8

TRACK_MOMENTS, TRACK_TWO_KERNEL, TRACK_SHARED_INPUT, TRACK_SEQUENTIAL_STEPS, TRACK_COVARIANCE = 1, 2, 4, 8, 16


class LynxError(RuntimeError):
    """A liblynxhip call returned a negative status."""


class Elem(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32), ("param_offset", C.c_int32),
                ("batch_stride", C.c_int32)]


class Step(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first", C.c_int32), ("last", C.c_int32),
                ("flags", C.c_int32)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("lds_bytes_per_cu", C.c_int32), ("hbm_bytes", C.c_int64)]


_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t

# name -> (restype, argtypes); must list every function declared in include/lynx_hip.h
SIGNATURES = {
    "lynx_version": (C.c_char_p, []),
    "lynx_device_count": (_i, [C.POINTER(_i)]),
    "lynx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "lynx_ctx_destroy": (_i, [_vp]),
    "lynx_last_error": (C.c_char_p, [_vp]),
    "lynx_device_info": (_i, [_vp, C.POINTER(DeviceInfo)]),
    "lynx_sync": (_i, [_vp]),
    "lynx_timer_start": (_i, [_vp]),
    "lynx_timer_stop": (_i, [_vp, C.POINTER(C.c_float)]),
    "lynx_profile_begin": (_i, [_vp]),
    "lynx_profile_end": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lynx_profile_launches": (_i, [_vp, C.POINTER(C.c_double), _i64, C.POINTER(C.c_int64)]),
    "lynx_profile_gathers": (_i, [_vp, C.POINTER(C.c_double), _i64, C.POINTER(C.c_int64)]),
    "lynx_ctx_reload_knobs": (_i, [_vp]),
    "lynx_diag_copy": (_i, [_vp, _vp, _vp, _sz, _i, _i, C.POINTER(C.c_float)]),
    "lynx_buf_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "lynx_buf_alloc_result": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "lynx_buf_free": (_i, [_vp, _vp]),
    "lynx_buf_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "lynx_buf_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "lynx_buf_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "lynx_buf_memset": (_i, [_vp, _vp, _i, _sz]),
    "lynx_pool_trim": (_i, [_vp]),
    "lynx_lattice_create": (_i, [_vp, _i, _i64, C.c_int32, C.POINTER(Elem), C.c_int32,
                                 C.POINTER(Step), _vp, _i64, C.POINTER(_vp)]),
    "lynx_lattice_update_params": (_i, [_vp, _i64, _i64, _vp]),
    "lynx_lattice_set_flags": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lynx_lattice_destroy": (_i, [_vp]),
    "lynx_build_compose": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "lynx_track_particles": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "lynx_track_particles_new": (_i, [_vp, _vp, _i64, _vp, _vp, _i, C.c_int32, C.POINTER(_vp)]),
    "lynx_track_moments": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lynx_track_particles_backward": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lynx_track_moments_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lynx_moments": (_i, [_vp, _i, _i64, _i64, _vp, _vp, C.c_int32]),
    "lynx_histogram2d": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "lynx_gaussian_image": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "lynx_diag_phase_trig": (_i, [_vp, _i64, _vp, C.c_int32, _vp, _vp]),
    "lynx_aperture_mask": (_i, [_vp, _i, _i64, _i64, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "lynx_aperture_compact": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "lynx_fill_gaussian": (_i, [_vp, _i, _i64, _i64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.c_uint64, _vp]),
    "lynx_comm_unique_id": (_i, [C.c_char_p]),
    "lynx_comm_init": (_i, [_vp, _i, _i, C.c_char_p]),
    "lynx_comm_destroy": (_i, [_vp]),
    "lynx_comm_info": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lynx_gather_moments": (_i, [_vp, _vp, _vp, _i64]),
}

_lib = None


def library_path() -> Path:
    return Path(os.environ.get("LYNX_HIP_LIBRARY", _LIB_PATH))


def load():
    """Load liblynxhip.so and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise LynxError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or lynx_amd/csrc/build.sh). "
            "lynx_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


ERR_ENERGY = -5


def check(status: int, ctx=None):
    if status != 0:
        lib = load()
        msg = lib.lynx_last_error(ctx)
        if status == ERR_ENERGY:  # the reference's `assert torch.all(Ei > 0)` (cavity.py:260), found on the device
            raise AssertionError(msg.decode() if msg else "Initial energy must be larger than 0")
        raise LynxError(f"liblynxhip status {status}: {msg.decode() if msg else '?'}")
