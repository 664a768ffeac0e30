"""
Multi-GPU execution: one process per GPU, batch samples sharded in contiguous slices,
particles never leave their GPU.  The only exchange is an RCCL all-gather of the
per-sample moment records (36 float64 each) over xGMI; at <= 37 KB per rank it is
latency-bound, so no bucketing or ring tuning applies.

The reference has no counterpart (single process); batch samples are independent in
every function of the path.

Rendezvous: the 128-byte RCCL unique id made by rank 0 has to reach every rank.  The
package does not choose how (`exchange` callable); bench.py uses torch.distributed's
store, tests use a gloo broadcast.
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
        uid = None
        if rank == 0:
            buf = C.create_string_buffer(_ffi.UNIQUE_ID_BYTES)
            _ffi.check(self.rt.lib.lynx_comm_unique_id(buf))
            uid = buf.raw
        uid = exchange(uid)
        assert isinstance(uid, (bytes, bytearray)) and len(uid) == _ffi.UNIQUE_ID_BYTES
        self.rt.check(self.rt.lib.lynx_comm_init(self.rt.ctx, n_ranks, rank, bytes(uid)))

    def all_gather(self, local: DeviceArray) -> DeviceArray:
        """(rows, 36) float64 per rank -> (n_ranks, rows, 36) on every rank (async on the stream)."""
        assert local.dtype == np.float64
        out = self.rt.empty((self.n_ranks, *local.shape), np.float64)
        self.rt.check(self.rt.lib.lynx_gather_moments(self.rt.ctx, C.c_void_p(local.ptr), C.c_void_p(out.ptr),
                                                      local.size))
        return out

    def close(self):
        self.rt.lib.lynx_comm_destroy(self.rt.ctx)
