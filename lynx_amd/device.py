"""
GPU runtime of the package: one `Runtime` (liblynxhip context = one GPU, one HIP stream)
per process, and `DeviceArray`, the HBM-resident array the beams hold.

One process drives one GPU (the multi-GPU model is one process per GPU, see
`lynx_amd.parallel`); the device ordinal is `LYNX_DEVICE`, else `LOCAL_RANK`, else 0.
"""

from __future__ import annotations

import atexit
import ctypes as C
import math
import os

import numpy as np

from . import _ffi

_DTYPES = {np.dtype(np.float32): _ffi.F32, np.dtype(np.float64): _ffi.F64}


def dtype_code(dtype) -> int:
    try:
        return _DTYPES[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"lynx_amd supports float32 and float64, not {dtype}") from None


class Runtime:
    """Owns the liblynxhip context of this process."""

    def __init__(self, device: int | None = None):
        self.lib = _ffi.load()
        if device is None:
            device = int(os.environ.get("LYNX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            count = C.c_int(0)
            self.lib.lynx_device_count(C.byref(count))
            isolated = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
            if 0 < count.value <= device and "LYNX_DEVICE" not in os.environ and isolated and count.value == 1:
                device = 0  # the launcher gave this rank ONE GPU of its own (*_VISIBLE_DEVICES per rank): that one
            elif 0 < count.value <= device and os.environ.get("LYNX_ALLOW_GPU_SHARING") == "1":
                device %= count.value  # rehearsals on a box with fewer GPUs than ranks (no RCCL there)
            elif 0 < count.value <= device:
                raise _ffi.LynxError(
                    f"rank-local device ordinal {device} (LYNX_DEVICE / LOCAL_RANK) but only {count.value} GPU(s) visible "
                    f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                    f"ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')!r}): one process drives one GPU "
                    "(set LYNX_DEVICE, or LYNX_ALLOW_GPU_SHARING=1 for a rehearsal)")
        handle = C.c_void_p()
        _ffi.check(self.lib.lynx_ctx_create(int(device), C.byref(handle)))
        self.ctx = handle
        self.device = int(device)
        self.closed = False
        atexit.register(self.close)

    def close(self):
        """Destroy the context.  Later frees of arrays/lattices that outlive it are no-ops."""
        if not self.closed:
            self.closed = True
            self.lib.lynx_ctx_destroy(self.ctx)

    # -- plumbing -------------------------------------------------------------------------
    def check(self, status: int):
        _ffi.check(status, self.ctx)

    def sync(self):
        self.check(self.lib.lynx_sync(self.ctx))

    def info(self) -> dict:
        di = _ffi.DeviceInfo()
        self.check(self.lib.lynx_device_info(self.ctx, C.byref(di)))
        return {"name": di.name.decode(), "arch": di.arch.decode(), "compute_units": di.compute_units,
                "lds_bytes_per_cu": di.lds_bytes_per_cu, "hbm_bytes": di.hbm_bytes}

    def timer_start(self):
        self.check(self.lib.lynx_timer_start(self.ctx))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.check(self.lib.lynx_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)

    def profile_begin(self):
        self.check(self.lib.lynx_profile_begin(self.ctx))

    def profile_end(self):
        """(total milliseconds, launches) of the streaming kernel since profile_begin()."""
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.lynx_profile_end(self.ctx, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def profile_launches(self) -> list:
        """Milliseconds of every streaming-kernel launch of the profile closed last, in launch order."""
        n = C.c_int64()
        self.check(self.lib.lynx_profile_launches(self.ctx, None, 0, C.byref(n)))
        out = (C.c_double * max(n.value, 1))()
        self.check(self.lib.lynx_profile_launches(self.ctx, out, n.value, C.byref(n)))
        return [float(v) for v in out[: n.value]]

    def profile_gathers(self) -> list:
        """Milliseconds of every RCCL gather of the profile closed last (on the stream it ran on), in issue order."""
        n = C.c_int64()
        self.check(self.lib.lynx_profile_gathers(self.ctx, None, 0, C.byref(n)))
        out = (C.c_double * max(n.value, 1))()
        self.check(self.lib.lynx_profile_gathers(self.ctx, out, n.value, C.byref(n)))
        return [float(v) for v in out[: n.value]]

    def reload_knobs(self):
        """Read the LYNX_* launch-plan switches from the environment again (they are read once, at context creation)."""
        self.check(self.lib.lynx_ctx_reload_knobs(self.ctx))

    def copy_bandwidth(self, nbytes: int, repeats: int = 10, shapes=(1, 4, 0, 101, 104)) -> dict:
        """
        GB/s (read + write bytes) of a plain 16 B/lane device copy of `nbytes` for each launch shape in
        `shapes` (vectors per thread; 0 = grid-stride; 100 + n = n per thread with non-temporal stores):
        the practical HBM ceiling is the best of them.
        """
        nbytes = int(nbytes) // 16 * 16
        a, b = self.alloc(nbytes), self.alloc(nbytes)
        out = {}
        try:
            for vpt in shapes:
                ms = C.c_float()
                self.check(self.lib.lynx_diag_copy(self.ctx, C.c_void_p(b), C.c_void_p(a), nbytes, repeats, int(vpt),
                                                   C.byref(ms)))
                name = "grid_stride" if not vpt else (f"{vpt - 100}_vec_per_thread_nt_store" if vpt >= 100
                                                      else f"{vpt}_vec_per_thread")
                out[name] = 2 * nbytes / (ms.value * 1e-3) / 1e9
        finally:
            self.free(a)
            self.free(b)
        return out

    def alloc(self, nbytes: int) -> int:
        ptr = C.c_void_p()
        self.check(self.lib.lynx_buf_alloc(self.ctx, int(nbytes), C.byref(ptr)))
        return ptr.value

    def free(self, ptr: int):
        if not self.closed:
            self.lib.lynx_buf_free(self.ctx, C.c_void_p(ptr))

    # -- arrays ---------------------------------------------------------------------------
    def empty(self, shape, dtype) -> "DeviceArray":
        return DeviceArray(self, tuple(int(s) for s in shape), np.dtype(dtype))

    def empty_result(self, shape, dtype) -> "DeviceArray":
        """For what a call returns and the host reads next (`lynx_buf_alloc_result`: small blocks are host memory the GPU writes through)."""
        shape, dtype = tuple(int(s) for s in shape), np.dtype(dtype)
        ptr = C.c_void_p()
        self.check(self.lib.lynx_buf_alloc_result(self.ctx, max(math.prod(shape) * dtype.itemsize, 1), C.byref(ptr)))
        return DeviceArray.adopt(self, ptr.value, shape, dtype)

    def to_device_result(self, host) -> "DeviceArray":
        """`to_device` into a block of `empty_result`: a small array the host has just made and the next call reads once (cotangents)."""
        host = np.ascontiguousarray(host)
        arr = self.empty_result(host.shape, host.dtype)
        if host.nbytes:
            self.check(self.lib.lynx_buf_h2d(self.ctx, arr.ptr, host.ctypes.data, host.nbytes))
        return arr

    def to_device(self, host) -> "DeviceArray":
        host = np.ascontiguousarray(host)
        arr = self.empty(host.shape, host.dtype)
        if host.nbytes:
            self.check(self.lib.lynx_buf_h2d(self.ctx, arr.ptr, host.ctypes.data, host.nbytes))
        return arr


_runtime: Runtime | None = None


def get_runtime() -> Runtime:
    """The process-wide runtime (created on first use).  Raises without a GPU."""
    global _runtime
    if _runtime is None:
        _runtime = Runtime()
    return _runtime


class DeviceArray:
    """A C-contiguous array in HBM.  `np.asarray(x)` copies it to the host."""

    # (a plain finaliser instead of `weakref.finalize`: a tracking loop makes and drops three of these per call, and
    # the registry entry of a `finalize` costs more than the allocation; `Runtime.closed` makes a late release a no-op)
    __slots__ = ("rt", "shape", "dtype", "size", "nbytes", "ptr", "_base", "_owned", "__weakref__")

    def __init__(self, rt: Runtime, shape: tuple, dtype: np.dtype):
        self.rt = rt
        self.shape = shape
        self.dtype = dtype
        self.size = math.prod(shape)  # () -> 1; Python ints: no overflow
        self.nbytes = self.size * dtype.itemsize
        self._base = None
        self._owned = False
        self.ptr = rt.alloc(max(self.nbytes, 1))
        self._owned = True

    @classmethod
    def adopt(cls, rt: Runtime, ptr: int, shape: tuple, dtype: np.dtype) -> "DeviceArray":
        """A block the library allocated for the caller (`lynx_track_particles_new`): this object now owns it."""
        self = object.__new__(cls)
        self.rt, self.shape, self.dtype, self.ptr = rt, shape, dtype, ptr
        self.size = math.prod(shape)
        self.nbytes = self.size * dtype.itemsize
        self._base = None
        self._owned = True
        return self

    def __del__(self):
        try:
            if self._owned:
                self._owned = False
                self.rt.free(self.ptr)
        except Exception:  # interpreter teardown: attributes or modules may already be gone
            pass

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            self.rt.check(self.rt.lib.lynx_buf_d2h(self.rt.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def __array__(self, dtype=None, copy=None):
        out = self.numpy()
        return out if dtype is None else out.astype(dtype)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __len__(self):
        return self.shape[0]

    def reshape(self, *shape) -> "DeviceArray":
        """A view with another shape sharing the same HBM block."""
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        view = object.__new__(DeviceArray)
        view.rt, view.dtype, view.ptr = self.rt, self.dtype, self.ptr
        view.shape = tuple(int(s) for s in shape)
        view.size = int(np.prod(view.shape, dtype=np.int64)) if view.shape else 1
        assert view.size == self.size, "reshape must keep the number of elements"
        view.nbytes = self.nbytes
        view._base = self  # keeps the owner alive
        view._owned = False
        return view

    def copy(self) -> "DeviceArray":
        out = self.rt.empty(self.shape, self.dtype)
        self.rt.check(self.rt.lib.lynx_buf_d2d(self.rt.ctx, out.ptr, self.ptr, self.nbytes))
        return out

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.rt.device})"


def frozen(value, copy: bool = True) -> np.ndarray:
    """`value` as a read-only ndarray; a writable input is copied first unless `copy` is False."""
    arr = np.asarray(value)
    if arr.flags.writeable:
        arr = arr.copy() if copy else arr.view()
        arr.setflags(write=False)
    return arr


def as_host(x) -> np.ndarray:
    """NumPy view/copy of a host array, DeviceArray or sequence."""
    if isinstance(x, DeviceArray):
        return x.numpy()
    return np.asarray(x)


class Dual:
    """
    A value that may live on the host, in HBM, or both.  Beams hold these so that a beam
    built from NumPy data is uploaded once, and a beam produced by a kernel is only copied
    back when somebody looks at it.
    """

    def __init__(self, host=None, dev: DeviceArray | None = None, owned: bool = False):
        """
        The host side is kept READ-ONLY and, unless `owned`, as a private copy: the device copy and
        everything derived from it (moment records, packed lattices) is cached, so an in-place
        write -- through the array handed out by a property or through the caller's original --
        would otherwise be silently ignored.  It raises instead; assign a new array.
        """
        assert host is not None or dev is not None
        self._host = None if host is None else frozen(host, copy=not owned)
        self._dev = dev
        self._bcast = {}

    @property
    def shape(self):
        return self._dev.shape if self._dev is not None else self._host.shape

    @property
    def dtype(self):
        return self._dev.dtype if self._dev is not None else self._host.dtype

    @property
    def on_device(self) -> bool:
        return self._dev is not None

    def host(self) -> np.ndarray:
        if self._host is None:
            self._host = frozen(self._dev.numpy(), copy=False)
        return self._host

    def device(self, rt: Runtime | None = None) -> DeviceArray:
        if self._dev is None:
            self._dev = (rt or get_runtime()).to_device(self._host)
        return self._dev

    def broadcast_device(self, rt: Runtime, shape) -> DeviceArray:
        """Device copy broadcast to `shape` (per-sample scalars such as the beam energy)."""
        shape = tuple(shape)
        if tuple(self.shape) == shape:
            return self.device(rt)
        if shape not in self._bcast:
            self._bcast[shape] = rt.to_device(np.ascontiguousarray(np.broadcast_to(self.host(), shape)))
        return self._bcast[shape]
