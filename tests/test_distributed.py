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

from lynx_amd.parallel import assemble_records, merge_records, shard_batch, shard_particles
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


def _record_of(Q):
    """Moment record(s) of particles Q (..., n, 7), layout of include/lynx_hip.h."""
    rec = np.zeros((*Q.shape[:-2], 36))
    rec[..., :7] = Q.mean(axis=-2)
    k = 7
    for i in range(6):
        for j in range(i, 6):
            rec[..., k] = ((Q[..., i] - rec[..., i, None]) * (Q[..., j] - rec[..., j, None])).mean(axis=-1)
            k += 1
    rec[..., 35] = Q.shape[-2]
    return rec


def _tracked_single_sample():
    """C3-style case: one sample, every rank builds the same map, particles are what is sharded."""
    specs = o.fodo_segment(4, np.float64, (1,))
    P = o.gaussian_particles((1,), 3001, seed=9, dtype=np.float64, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3],
                             mu=[1e-3, 0, -2e-3, 0, 0, 0])
    return specs, P


def _particle_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        specs, P = _tracked_single_sample()
        a, b = shard_particles(P.shape[1], world, rank)
        out = o.segment_track(specs, o.particle_beam(P[:, a:b], np.full(1, 1e8), np.float64), np.float64)
        local = torch.from_numpy(_record_of(out["particles"]))
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        q.put((rank, merge_records(np.stack([g.numpy() for g in gathered]))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


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


def test_merge_records_is_the_record_of_the_union():
    rng = np.random.default_rng(3)
    Q = rng.normal(size=(5, 1000, 7)) * [1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3, 0] + [5e-3, 0, -1e-3, 0, 0, 0, 1]
    whole = _record_of(Q)
    for cuts in ([0, 1000], [0, 1, 1000], [0, 333, 334, 1000], [0, 500, 500, 1000]):  # incl. an empty slice
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = _record_of(Q[:, a:b]) if b > a else np.zeros((5, 36))
            parts.append(part)
        merged = merge_records(np.stack(parts))
        assert np.array_equal(merged[..., 35], whole[..., 35])
        np.testing.assert_allclose(merged[..., :7], whole[..., :7], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(merged[..., 7:28], whole[..., 7:28], rtol=1e-10, atol=1e-24)


def test_world_size_2_particle_sharding_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_particle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    specs, P = _tracked_single_sample()
    out = o.segment_track(specs, o.particle_beam(P, np.full(1, 1e8), np.float64), np.float64)
    reference = _record_of(out["particles"])
    assert np.array_equal(results[0][1], results[1][1])  # same merge order on every rank
    for _, merged in results:
        np.testing.assert_allclose(merged[..., :7], reference[..., :7], rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(merged[..., 7:28], reference[..., 7:28], rtol=1e-9, atol=1e-24)
        assert merged[0, 35] == 3001


# ---------------------------------------------------------------------------------------------
# lynx_amd.rendezvous: the standard-library rendezvous bench.py uses at N > 1 (no torch, no gloo)
# ---------------------------------------------------------------------------------------------


def _rdzv_rank(rank, world, key, queue):
    import struct

    from lynx_amd.rendezvous import Rendezvous

    rz = Rendezvous(rank, world, key=key, timeout_s=60)
    uid = rz.broadcast(bytes(range(128)) if rank == 0 else None)  # the RCCL unique id travels like this
    rz.barrier()
    top = rz.max(10.0 + rank)
    a, b = shard_batch(GLOBAL_BATCH, world, rank)
    rows = _records((a, b))
    pad = np.zeros((shard_batch(GLOBAL_BATCH, world, 0)[1], 36))
    pad[: b - a] = rows
    parts = rz.all_gather(pad.tobytes())  # the --allow-host-gather path
    gathered = np.stack([np.frombuffer(p, dtype=np.float64).reshape(pad.shape) for p in parts])
    ok = rz.all_true(rank != 1)  # one rank reports failure: nobody may see "all true"
    rz.close()
    queue.put((rank, uid, top, assemble_records(gathered, GLOBAL_BATCH, world), ok, struct.calcsize("<d")))


@pytest.mark.parametrize("world", [2, 3])
def test_stdlib_rendezvous_broadcast_barrier_max_gather(world, tmp_path):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    key = f"test-{os.getpid()}-{world}"
    procs = [ctx.Process(target=_rdzv_rank, args=(r, world, key, queue)) for r in range(world)]
    for p in reversed(procs):  # rank 0 last: the others must wait for its port file
        p.start()
    results = sorted(queue.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _records((0, GLOBAL_BATCH))
    for rank, uid, top, records, ok, _ in results:
        assert uid == bytes(range(128))
        assert top == 10.0 + world - 1
        assert np.array_equal(records, want)  # bytes travel unchanged
        assert ok is False
    import tempfile
    from pathlib import Path

    assert not (Path(tempfile.gettempdir()) / f"lynx-rdzv-{key}.port").exists()


def test_a_stale_port_file_of_a_crashed_launch_is_not_mistaken_for_rank_0(tmp_path):
    """Same key, leftover file pointing at a port where somebody else listens: the joiner must end up at the
    real rank 0 (nonce in the file and in the hello reply; rank 0 removes the leftover before it publishes)."""
    import tempfile
    from pathlib import Path

    key = f"test-stale-{os.getpid()}"
    stale = Path(tempfile.gettempdir()) / f"lynx-rdzv-{key}.port"
    with socket.socket() as other:  # a listener that is not a rendezvous
        other.bind(("127.0.0.1", 0))
        other.listen(1)
        stale.write_text(f"{other.getsockname()[1]} {'0' * 32} 1\n")
        ctx = mp.get_context("spawn")
        queue = ctx.Queue()
        procs = [ctx.Process(target=_rdzv_rank, args=(r, 2, key, queue)) for r in range(2)]
        for p in reversed(procs):
            p.start()
        results = sorted(queue.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    assert [r[0] for r in results] == [0, 1] and all(r[1] == bytes(range(128)) for r in results)
    assert not stale.exists()


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N started plainly: it launches its own ranks (no GPU needed for the rehearsal mode)
# ---------------------------------------------------------------------------------------------


def _bench(*args, env=None):
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LYNX_RDZV_KEY")}
    # the launcher and the rehearsal ranks must not need the HIP library (nothing may touch the GPU before the ranks exist)
    e["LYNX_HIP_LIBRARY"] = "/nonexistent/liblynxhip.so"
    e.update(env or {})
    run = subprocess.run([sys.executable, str(root / "bench.py"), *args], env=e, capture_output=True, text=True, timeout=120)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    return run.returncode, [json.loads(ln) for ln in lines], run.stderr


def test_bench_launches_its_own_ranks_and_shards_config_4_strongly():
    rc, lines, err = _bench("--gpus", "8", "--rendezvous-only")
    assert rc == 0, err
    assert len(lines) == 1  # rank 0's line only
    line = lines[0]
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["value"] is None and line["dry_run"] is True
    plan = line["config"]["shard_plan"]
    # batches [128 g, 128 (g + 1)) on GPU g, all 100 000 particles of a sample on its GPU
    assert [(p["first_sample"], p["samples"], p["particles"]) for p in plan] == [(128 * g, 128, 100_000) for g in range(8)]
    assert line["config"]["global_batch"] == 1024 and line["config"]["batch_per_gpu"] == 128
    assert line["config"]["launcher"] == "bench.py"


def test_bench_weak_scaling_and_particle_sharding_are_still_there():
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only", "--weak")
    assert rc == 0, err
    assert lines[0]["scaling"] == "weak" and lines[0]["config"]["global_batch"] == 2048
    rc, lines, err = _bench("--gpus", "3", "--rendezvous-only", "--workload", "c3")
    assert rc == 0, err
    assert [p["particles"] for p in lines[0]["config"]["shard_plan"]] == [333_334, 333_333, 333_333]
    assert lines[0]["config"]["parallelism"] == "particle-sharded x3"


def test_bench_launcher_reports_a_failing_rank():
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only", env={"LYNX_BENCH_TEST_EXIT": "3", "LYNX_BENCH_TEST_EXIT_RANK": "1"})
    assert rc == 3, (rc, err)


def test_bench_under_an_external_launcher_uses_the_ranks_it_is_given():
    # what torchrun does, without torch: two processes with RANK / WORLD_SIZE and a common parent
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    base = {k: v for k, v in os.environ.items() if k != "LYNX_RDZV_KEY"}
    base.update(WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29599", LYNX_HIP_LIBRARY="/nonexistent/liblynxhip.so")
    procs = [subprocess.Popen([sys.executable, str(root / "bench.py"), "--gpus", "2", "--rendezvous-only"],
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in (1, 0)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert outs[0].strip() == "" and '"n_gpus": 2' in outs[1]  # rank 1 prints nothing, rank 0 the line
    # a mismatch between --gpus and the launcher's world size is an error, not a silent single-rank run
    one = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--rendezvous-only"],
                         env=dict(base, RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=60)
    assert one.returncode != 0 and "WORLD_SIZE=2" in one.stderr


def test_bench_launcher_retries_once_without_the_ipc_variable_when_every_rank_fails_rccl():
    """
    The ranks run with HSA_ENABLE_IPC_MODE_LEGACY=0 (this pool's images export it; bench.py: launch_ranks).  Should
    every rank fail its RCCL bring-up with it (exit 3, rank 0's line says "rccl-failed"), the launcher -- which never
    touches the GPU -- starts ONE more, fresh set of ranks with the variable unset; the run's line is the second set's
    and says so.  Rehearsed with ranks that pretend: LYNX_BENCH_TEST_RCCL_FAILS_WITH_IPC_LEGACY.
    """
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only",
                            env={"LYNX_BENCH_TEST_RCCL_FAILS_WITH_IPC_LEGACY": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert rc == 0, (rc, err)
    assert len(lines) == 1, lines  # the first set's line is held back
    assert "HSA_ENABLE_IPC_MODE_LEGACY unset" in lines[0]["config"]["launcher_retry"]
    assert lines[0]["config"]["gather"] == "none" and lines[0]["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] is None
    assert "one more set of ranks" in err
    # no retry for anything else: a rank that simply fails stays a failure
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only", env={"LYNX_BENCH_TEST_EXIT": "3", "LYNX_BENCH_TEST_EXIT_RANK": "0"})
    assert rc == 3 and "one more set of ranks" not in err
    # without the pretence the first set succeeds and carries the variable
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only", env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert rc == 0 and "launcher_retry" not in lines[0]["config"] and lines[0]["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_launcher_runs_the_cpu_baseline_before_the_ranks_and_rank_0_reports_it():
    """N > 1: the CPU-baseline legs run in the launcher parent (they fork; no rank waits for them) and reach the line."""
    rc, lines, err = _bench("--gpus", "2", "--rendezvous-only", "--particles", "2000",
                            env={"LYNX_BENCH_TEST_CPU_BASELINE": "1", "LYNX_BENCH_CPU_BUDGET_S": "0.2"})
    assert rc == 0, err
    base = lines[0]["cpu_baseline"]
    assert base["kind"] == "port" and base["value"] > 0 and base["cores"] >= 1 and base["one_thread"]["value"] > 0
