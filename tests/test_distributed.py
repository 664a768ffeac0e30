"""
N > 1 path on the CPU (gloo, world_size 2): batch sharding, unique-id rendezvous plumbing and
the assembly of gathered moment records.  The per-rank tracking result comes from the oracle
here (no GPU in this suite); on the GPU box the same shard plan feeds lynx_amd and the
transport is RCCL (`lynx_amd.parallel.RcclCommunicator`).
"""

import os
import socket

import numpy as np
import pytest

from lynx_amd.parallel import assemble_records, shard_batch
from oracle import lynx_oracle as o

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

GLOBAL_BATCH = 7  # odd on purpose: ranks own 4 and 3 samples
N_PART = 2000


def _records(batch_slice):
    """Moment records (rows of 36) of the tracked beam for the global samples in `batch_slice`."""
    a, b = batch_slice
    scale = (0.5 + np.arange(GLOBAL_BATCH) / (GLOBAL_BATCH - 1))[a:b]
    specs = o.fodo_segment(4, np.float64, (b - a,), scale)
    P = o.gaussian_particles((GLOBAL_BATCH,), N_PART, seed=5, dtype=np.float64,
                             sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3])[a:b]
    out = o.segment_track(specs, o.particle_beam(P, np.full(b - a, 1e8), np.float64), np.float64)
    rec = np.zeros((b - a, 36))
    Q = out["particles"]
    rec[:, :7] = Q.mean(axis=1)
    k = 7
    for i in range(6):
        for j in range(i, 6):
            rec[:, k] = ((Q[..., i] - rec[:, i, None]) * (Q[..., j] - rec[:, j, None])).mean(axis=1)
            k += 1
    rec[:, 35] = N_PART
    return rec


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rendezvous plumbing used by bench.py for the RCCL unique id
        box = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == bytes(range(128))
        a, b = shard_batch(GLOBAL_BATCH, world, rank)
        local = _records((a, b))
        rows = max(shard_batch(GLOBAL_BATCH, world, r)[1] - shard_batch(GLOBAL_BATCH, world, r)[0] for r in range(world))
        padded = np.zeros((rows, 36))
        padded[: b - a] = local
        gathered = [torch.zeros(rows, 36, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(padded))
        full = assemble_records(np.stack([g.numpy() for g in gathered]), GLOBAL_BATCH, world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)  # max-over-ranks as in bench.py
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, full, float(t.item())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_batch_covers_the_batch_once():
    for total in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            slices = [shard_batch(total, world, r) for r in range(world)]
            assert slices[0][0] == 0 and slices[-1][1] == total
            assert all(slices[i][1] == slices[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in slices]
            assert max(sizes) - min(sizes) <= 1
    assert shard_batch(1024, 8, 3) == (384, 512)  # batches [128 g, 128 (g+1)) on GPU g


def test_world_size_2_gather_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reference = _records((0, GLOBAL_BATCH))
    for rank, full, tmax in results:
        assert full.shape == (GLOBAL_BATCH, 36)
        assert np.array_equal(full, reference), rank  # every rank holds the whole batch, in batch order
        assert tmax == 2.0
