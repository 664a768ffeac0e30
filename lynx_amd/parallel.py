"""
Multi-GPU execution: one process per GPU, batch samples sharded in contiguous slices,
particles never leave their GPU.  The only exchange is an RCCL all-gather of the
per-sample moment records (36 float64 each) over xGMI; at <= 37 KB per rank it is
latency-bound, so no bucketing or ring tuning applies.

The reference has no counterpart (single process); batch samples are independent in
every function of the path.

Two partitionings:

* **batch-sharded** (C4, C5): `shard_batch` + `assemble_records` -- records are only reordered;
* **particle-sharded** (C3, one sample with many particles): `shard_particles` gives each rank
  a contiguous slice of N, every rank builds the same composed map (cheaper than a broadcast),
  and `merge_records` combines the gathered per-slice records into the record of the whole
  beam with the pairwise mean / co-moment update (Chan et al.), in rank order, in float64 --
  deterministic, and no second collective is needed because the gather already delivered every
  slice's (count, mean, covariance).

Rendezvous: the 128-byte RCCL unique id made by rank 0 has to reach every rank.  The
package does not choose how (`exchange` callable); bench.py sends it over `lynx_amd.rendezvous`
(standard-library sockets: no torch in the process), tests use a gloo broadcast.
"""

from __future__ import annotations

import ctypes as C
from typing import Callable

import numpy as np

from . import _ffi
from .device import DeviceArray, Runtime, get_runtime


def shard_batch(global_batch: int, n_ranks: int, rank: int) -> tuple:
    """Contiguous slice [start, stop) of the batch owned by `rank` (remainder to the first ranks)."""
    assert 0 <= rank < n_ranks and global_batch >= 0
    base, rem = divmod(global_batch, n_ranks)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def assemble_records(gathered: np.ndarray, global_batch: int, n_ranks: int) -> np.ndarray:
    """
    (n_ranks, max_local, 36) all-gather result -> (global_batch, 36) in batch order,
    dropping the padding rows of ranks that own one sample fewer.
    """
    out = np.empty((global_batch, gathered.shape[-1]), dtype=gathered.dtype)
    for r in range(n_ranks):
        a, b = shard_batch(global_batch, n_ranks, r)
        out[a:b] = gathered[r, : b - a]
    return out


def shard_particles(num_particles: int, n_ranks: int, rank: int) -> tuple:
    """Contiguous slice [start, stop) of the particle axis owned by `rank`."""
    return shard_batch(num_particles, n_ranks, rank)


def merge_records(gathered: np.ndarray) -> np.ndarray:
    """
    (n_ranks, ..., 36) records of disjoint particle slices of the same beam(s) -> (..., 36)
    record of the union: n = sum n_r; mean = sum n_r mean_r / n (folded pairwise);
    cov = [n_a cov_a + n_b cov_b + (n_a n_b / n) d d^T] / n with d = mean_b - mean_a
    (biased covariances, layout of include/lynx_hip.h).
    """
    gathered = np.asarray(gathered, dtype=np.float64)
    out = gathered[0].copy()
    iu = [(i, j) for i in range(6) for j in range(i, 6)]  # slots 7..27, row-major upper triangle
    for part in gathered[1:]:
        na, nb = out[..., 35], part[..., 35]
        n = na + nb
        with np.errstate(invalid="ignore", divide="ignore"):
            wa = np.where(n > 0, na / n, 0.0)
            wb = np.where(n > 0, nb / n, 0.0)
        d = part[..., :7] - out[..., :7]
        for slot, (i, j) in enumerate(iu, start=7):
            out[..., slot] = wa * out[..., slot] + wb * part[..., slot] + wa * wb * d[..., i] * d[..., j]
        out[..., :7] = out[..., :7] + wb[..., None] * d
        out[..., 35] = n
    return out


def rccl_unique_id(rt: Runtime | None = None) -> bytes:
    """A fresh 128-byte RCCL unique id (rank 0 makes it, every rank needs it: `lynx_comm_unique_id`)."""
    rt = rt or get_runtime()
    buf = C.create_string_buffer(_ffi.UNIQUE_ID_BYTES)
    _ffi.check(rt.lib.lynx_comm_unique_id(buf))
    return buf.raw


class RcclCommunicator:
    """RCCL communicator of this process' GPU (C ABI: lynx_comm_*, lynx_gather_moments)."""

    def __init__(self, n_ranks: int, rank: int, exchange: Callable[[bytes | None], bytes],
                 rt: Runtime | None = None):
        """
        :param exchange: called on every rank; rank 0 passes its unique id, the others None;
            must return rank 0's id on every rank.
        """
        self.rt = rt or get_runtime()
        self.n_ranks, self.rank = n_ranks, rank
        uid = exchange(rccl_unique_id(self.rt) if rank == 0 else None)
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == _ffi.UNIQUE_ID_BYTES
        self.rt.check(self.rt.lib.lynx_comm_init(self.rt.ctx, n_ranks, rank, bytes(uid)))

    def info(self) -> dict:
        """RCCL version and the communicator's own idea of its size and this rank (`lynx_comm_info`)."""
        v, n, r = C.c_int32(), C.c_int32(), C.c_int32()
        self.rt.check(self.rt.lib.lynx_comm_info(self.rt.ctx, C.byref(v), C.byref(n), C.byref(r)))
        code = v.value  # NCCL_VERSION_CODE: major*10000 + minor*100 + patch since 2.9
        return {"rccl_version": f"{code // 10000}.{code // 100 % 100}.{code % 100}", "rccl_ranks": n.value,
                "rccl_rank": r.value}

    def all_gather(self, local: DeviceArray) -> DeviceArray:
        """
        (rows, 36) float64 per rank -> (n_ranks, rows, 36) on every rank.  Asynchronous: with more than one rank
        it runs on the context's communication stream underneath whatever is tracked next and nothing on the main
        stream waits for it (`LYNX_GATHER_OVERLAP=0`: in line on the main stream); reading the result
        (`np.asarray`, `rt.sync()`) waits for it either way.  Neither array has to be kept alive by the caller:
        the library holds a freed block back until the gather that touches it is done.
        """
        assert local.dtype == np.float64
        out = self.rt.empty((self.n_ranks, *local.shape), np.float64)
        self.rt.check(self.rt.lib.lynx_gather_moments(self.rt.ctx, C.c_void_p(local.ptr), C.c_void_p(out.ptr),
                                                      local.size))
        return out

    def close(self):
        """Destroy the communicator (collective: every rank calls it).  No-op once the runtime is closed."""
        if not self.rt.closed and not getattr(self, "_closed", False):
            self._closed = True
            self.rt.check(self.rt.lib.lynx_comm_destroy(self.rt.ctx))  # waits for the communication stream
