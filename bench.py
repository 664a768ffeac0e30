#!/usr/bin/env python3
"""
Benchmark of the hot path: `Segment.track()` on a `ParticleBeam`, fused output moments
included, on synthetic lattices/beams of the shapes BASELINE.json names.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (same thing)

A "step" is one `segment.track(beam)` over the whole per-GPU batch: the launch that builds +
composes every sample's element maps, the launch that streams the particles through them
(moments accumulated in its epilogue), the moment finalisation, and (N > 1) the RCCL
all-gather of the per-sample moment records.
Particles, lattice parameters and outputs are resident in HBM when the timed region starts.
Order of a run: CPU baseline (rank 0 of a one-GPU run -- the launcher parent of an N-GPU run --, before the GPU is
opened), W warm-up steps, barrier, K timed steps, barrier (the COLD leg: `ms_per_step_cold`, exactly what the command
line asks for), the plain-copy calibration of this box's HBM ceiling (every rank, about 0.1 s), barrier, the same K
timed steps again, barrier (`value`, `ms_per_step`: the streaming kernel's 20-30 ms start-up transient is behind it by
then).  `hbm_copy_kernel_when` in the line says so; LYNX_BENCH_CALIBRATE_FIRST=0: no cold leg, calibration last.

N > 1: one process per GPU.  Started plainly (`WORLD_SIZE` not in the environment) this process
becomes the LAUNCHER: it starts N fresh children of itself with RANK / LOCAL_RANK / WORLD_SIZE and a
shared rendezvous key BEFORE anything touches the GPU, lets rank 0 print the JSON line and leaves with
the worst child's exit code.  Under torchrun the ranks already exist and are used as they are.
Rendezvous, barriers and the max over ranks go over `lynx_amd.rendezvous` (standard-library
sockets) and the data exchange over RCCL -- the process never imports torch.  If the RCCL
communicator cannot be built the run prints a line with `"value": null` and exits 3
(`--allow-host-gather` turns that into a host gather that says so).

Workload (default `c4`): BASELINE.json config 4, the configuration the metric's target is
quoted on ("1024-batch x 100k-particle x 128-element lattice"): 1024 lattice-parameter
samples (k1 scan) x 128-element FODO x 100 000 particles, fp32 -- 5.73 GB of algorithmic
traffic per step, it fits one GPU.  Scaling is STRONG, as BASELINE config 4 is worded ("1024
lattice-parameter batches ... sharded over 8 x MI355X"): the ONE 1024-sample scan is split in
contiguous slices, GPU g tracks samples [128 g, 128 (g + 1)) at N = 8 (SURVEY.md section 8d/e;
`lynx_amd.parallel.shard_batch`), no particle ever crosses xGMI, the moment records are
all-gathered.  One-sample workloads (c3) split the particles instead.  `--weak` gives every GPU
the whole per-GPU workload (global batch 1024 N).

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (batch per GPU, particles, FODO cells (x4 elements), dtype, description)
    "c4": (1024, 100_000, 32, np.float32,
           "BASELINE config 4: 1024 k1-scan samples x 128-element FODO x 100k particles fp32 (one scan, sharded over the GPUs)"),
    "c4a": (1024, 100_000, 32, np.float32,
            "BASELINE config 4, corrector-angle variant: 1024 angle-scan samples x 128-element FODO with an H and a V "
            "corrector per cell x 100k particles fp32"),
    "c3": (1, 1_000_000, 32, np.float64, "BASELINE config 3: 128-element FODO x 1M particles fp64, batch 1"),
    "c3big": (1, 8_000_000, 32, np.float64, "config 3 at 8M particles (896 MB: defeats the 256 MB Infinity Cache)"),
    "c2": (1, 100_000, 0, np.float32, "BASELINE config 2: 11-element ARES-style segment x 100k particles fp32"),
    "c5": (4096, 10_000, 8, np.float32,
           "BASELINE config 5 (forward): 4096 envs x 32-element [Drift, misaligned Quad, Drift, Cavity]x8 x 10k particles fp32"),
}


def describe(name, sample_ids, cells, dtype, seed):
    """
    The lattice of workload `name` for the GLOBAL samples `sample_ids` (this rank's slice of the scan: a sample's
    parameters depend on its global index only, so the N-rank job is the 1-rank job cut in N pieces), as a list of
    (kind, keyword arguments): the product's elements AND the oracle's specifications are made from it
    (`build_segment`, `oracle_specs`; tests/test_gpu_parity.py checks the very lattices that are timed here).
    """
    batch = len(sample_ids)
    f = lambda v: np.full((batch,), v, dtype=dtype)  # noqa: E731
    if name == "c2":
        return [("bpm", {}), ("drift", dict(length=f(1.0))), ("bpm", {}), ("drift", dict(length=f(1.0))),
                ("vcor", dict(length=f(0.3), angle=f(3.142e-3))), ("drift", dict(length=f(0.2))),
                ("hcor", dict(length=f(0.3), angle=f(1e-4))), ("drift", dict(length=f(7.0))),
                ("hcor", dict(length=f(0.3), angle=f(-1e-4))), ("drift", dict(length=f(0.05))), ("bpm", {})]
    if name == "c5":
        # every environment's parameters are drawn for the whole job and cut to this rank's environments
        rng = np.random.default_rng(seed)
        total = int(sample_ids.max()) + 1 if batch else 0
        pick = lambda a: np.ascontiguousarray(a[sample_ids]).astype(dtype)  # noqa: E731
        desc = []
        for _ in range(cells):
            desc += [("drift", dict(length=f(0.3))),
                     ("quadrupole", dict(length=f(0.1), k1=pick(rng.uniform(-5, 5, total)),
                                         misalignment=pick(rng.normal(0, 1e-4, (total, 2))))),
                     ("drift", dict(length=f(0.3))),
                     ("cavity", dict(length=f(1.0377), voltage=pick(rng.uniform(5e6, 2e7, total)),
                                     phase=pick(rng.uniform(-10, 10, total)), frequency=f(1.3e9)))]
        return desc
    # position of global sample g in the scan: u = (g mod 1024) / 1023   
    u = (sample_ids % 1024) / 1023.0 if batch > 1 else np.full(1, 0.5)
    if name == "c4a":
        # the corrector-angle variant of BASELINE config 4 ("k1/angle scan"): the same FODO, its drifts replaced by a
        # horizontal and a vertical corrector of the drift's length (a corrector IS a drift plus a kick into the affine
        # column: horizontal_corrector.py:52-67, vertical_corrector.py:52-66), k1 fixed, the angles scanned
        k = f(4.2)
        ah, av = (2e-5 * (2 * u - 1)).astype(dtype), (-1e-5 * (2 * u - 1)).astype(dtype)
        desc = []
        for _ in range(cells):
            desc += [("quadrupole", dict(length=f(0.2), k1=k)), ("hcor", dict(length=f(0.5), angle=ah)),
                     ("quadrupole", dict(length=f(0.2), k1=-k)), ("vcor", dict(length=f(0.5), angle=av))]
        return desc
    # k1 scan: k1 = +-4.2 (0.5 + u)
    k = (4.2 * (0.5 + u)).astype(dtype) if batch > 1 else f(4.2)
    desc = []
    for _ in range(cells):
        desc += [("quadrupole", dict(length=f(0.2), k1=k)), ("drift", dict(length=f(0.5))),
                 ("quadrupole", dict(length=f(0.2), k1=-k)), ("drift", dict(length=f(0.5)))]
    return desc


_KINDS = {"drift": "Drift", "quadrupole": "Quadrupole", "hcor": "HorizontalCorrector", "vcor": "VerticalCorrector",
          "cavity": "Cavity", "bpm": "BPM"}


def build_segment(lx, name, sample_ids, cells, dtype, seed):
    """The product's `Segment` of `describe(...)`."""
    elements = []
    for kind, kw in describe(name, sample_ids, cells, dtype, seed):
        ctor = getattr(lx, _KINDS[kind])
        elements.append(ctor(**kw) if kind == "bpm" else ctor(**kw, dtype=dtype))
    return lx.Segment(elements)


def oracle_specs(o, name, sample_ids, cells, dtype, seed):
    """The oracle's specification list of `describe(...)` (cpu_baseline leg and tests only)."""
    return [getattr(o, _KINDS[kind])(**kw) for kind, kw in describe(name, sample_ids, cells, dtype, seed)]


# what every workload's incoming beam looks like (6-D Gaussian, made in HBM: ParticleBeam.synthetic)
BEAM_SIGMA = [1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3]


def beam_energy(name):
    return 6e6 if name == "c5" else 1e8


def shard_of(batch, particles, world, rank, weak):
    """
    -> (global batch, sample ids of this rank, particles of this rank, total particles per sample, parallelism).
    Strong scaling (default): ONE workload of `batch` samples x `particles` particles cut over the ranks --
    contiguous batch slices (primary partitioning), or particle slices when there is only
    one sample (secondary).  Weak: every rank gets the whole per-GPU workload.
    """
    from lynx_amd.parallel import shard_batch, shard_particles

    if weak or world == 1:
        ids = rank * batch + np.arange(batch)
        return batch * world, ids, particles, particles, f"{'particle' if batch == 1 else 'batch'}-sharded x{world}"
    if batch == 1:
        a, b = shard_particles(particles, world, rank)
        return 1, np.zeros(1, dtype=np.int64), b - a, particles, f"particle-sharded x{world}"
    a, b = shard_batch(batch, world, rank)
    return batch, np.arange(a, b), particles, particles, f"batch-sharded x{world}"


def host_cores() -> int:
    """CPUs this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            quota, period = parse(Path(path).read_text())
            if period is None:
                period = Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def _cpu_sample(name, particles, cells, dtype, bs, seed):
    from oracle import lynx_oracle as o

    n = particles if name != "c3big" else 1_000_000
    # `bs` samples spread over the scan (config 5: environments drawn like the job's)
    ids = np.arange(bs) if name == "c5" or bs == 1 else np.linspace(0, 1023, bs).round().astype(np.int64)
    specs = oracle_specs(o, name, ids, cells, dtype, seed)
    P = o.gaussian_particles((bs,), n, seed=seed, dtype=dtype, sigma=BEAM_SIGMA)
    beam = o.particle_beam(P, np.full((bs,), beam_energy(name), dtype=dtype), dtype)

    def run():
        o.beam_moments(o.segment_track(specs, beam, dtype))

    return run, bs * n * len(specs), len(specs), n


def _cpu_worker(job):
    """One process of the all-cores leg: its own samples, one BLAS thread, until the deadline."""
    name, particles, cells, dtype, bs, seed, t_start, t_stop = job
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        pass
    run, steps_per_pass, _, _ = _cpu_sample(name, particles, cells, dtype, bs, seed)
    run()
    while time.time() < t_start:
        time.sleep(0.01)
    reps = 0
    while True:
        run()
        reps += 1
        if time.time() > t_stop:
            return reps * steps_per_pass, time.time()


def cpu_baseline(name, particles, cells, dtype, budget_s=10.0):
    """
    The oracle (NumPy restatement of the reference algorithm: per-sample composition, then one
    `matmul(P, T^T)` and the moment read-out) on a bounded sample of the workload, timed on this box's
    host cores: one thread, and one single-threaded process per usable core (north_star: "the
    reference's own CPU path timed on the host cores of the same box, core count stated").  Must be
    called BEFORE the process touches the GPU (the all-cores leg forks).
    """
    import multiprocessing as mp

    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
    bs = 8 if name in ("c4", "c4a") else (64 if name == "c5" else 1)
    run, steps_per_pass, E, n = _cpu_sample(name, particles, cells, dtype, bs, 2)
    limiter = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        run()
        reps, t0 = 0, time.perf_counter()
        while True:
            run()
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or reps >= 2000:
                break
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    one = steps_per_pass * reps / dt

    cores = min(host_cores(), 64)
    per_worker = 2 if name in ("c4", "c4a") else (16 if name == "c5" else 1)
    t_start = time.time() + 3.0 + 0.05 * cores  # every worker has built its sample by then
    jobs = [(name, particles, cells, dtype, per_worker, 100 + w, t_start, t_start + budget_s) for w in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        done = pool.map(_cpu_worker, jobs, chunksize=1)
    total = sum(d[0] for d in done)
    wall = max(d[1] for d in done) - t_start
    return {"value": total / wall, "unit": "particle-element-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (NumPy restatement of the reference's algorithm) on {cores} processes x {per_worker} "
                      f"samples x {n} particles x {E} elements, one BLAS thread each, {wall:.1f} s "
                      f"({os.cpu_count()} logical CPUs on the host, {host_cores()} usable by this job)",
            "one_thread": {"value": one, "cores": 1,
                           "sample": f"{bs} samples x {n} particles x {E} elements, {reps} passes in {dt:.1f} s"}}


METRIC = "particle-element-steps/sec (whole node) + achieved HBM GB/s, Segment.track ParticleBeam"


def bring_up_rccl(rt, rdzv, rank, world, timeout_s):
    """
    -> (communicator or None, reason or None, stuck).
    RCCL communicator over the ranks of this launch, or the reason there is none.  The unique id
    travels over the rendezvous sockets; the blocking `ncclCommInitRank` runs under a watchdog.
    A bring-up that does not RETURN leaves a thread inside RCCL on this context: nothing may be
    launched on it any more, so the caller must end the process (non-zero) in that case.
    """
    import threading

    from lynx_amd.parallel import RcclCommunicator, rccl_unique_id

    uid = rdzv.broadcast(rccl_unique_id(rt) if rank == 0 else None)
    attempt: dict = {}

    def work():
        try:
            attempt["comm"] = RcclCommunicator(world, rank, lambda _: uid, rt)
        except Exception as exc:  # noqa: BLE001
            attempt["error"] = f"{type(exc).__name__}: {exc}"

    worker = threading.Thread(target=work, daemon=True)
    worker.start()
    worker.join(timeout=timeout_s)
    if worker.is_alive():
        return None, f"ncclCommInitRank did not return within {timeout_s:.0f} s", True
    return attempt.get("comm"), attempt.get("error"), False


def _run_ranks(n_ranks: int, env: dict):
    """One set of N children of this very command -> (exit codes, rank 0's stdout)."""
    import signal
    import subprocess

    children = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is held back: its JSON line is the run's line only if this attempt is the last one
        children.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=e,
                                         stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed_at = None
    grace = float(os.environ.get("LYNX_LAUNCH_GRACE_S", "30"))
    import threading

    captured = []
    reader = threading.Thread(target=lambda: captured.append(children[0].stdout.read()), daemon=True)
    reader.start()
    try:
        while any(c.poll() is None for c in children):
            for c in children:
                rc = c.poll()
                if rc not in (None, 0) and failed_at is None:
                    failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > grace:
                for c in children:
                    if c.poll() is None:
                        c.send_signal(signal.SIGTERM)  # exactly the pids started here
                failed_at = float("inf")
            time.sleep(0.05)
    except KeyboardInterrupt:
        for c in children:
            if c.poll() is None:
                c.send_signal(signal.SIGTERM)
    codes = [c.wait() for c in children]
    reader.join(timeout=10)
    return codes, (captured[0] if captured else b"")


def launch_ranks(n_ranks: int, baseline=None) -> int:
    """
    `python bench.py --gpus N` without a launcher around it: become one.  N children of this very command, one
    per GPU, each with RANK / LOCAL_RANK / WORLD_SIZE and the same rendezvous key, started BEFORE this process has
    touched the GPU (it never does: no re-exec of a process that has opened the device).  Rank 0's stdout is the run's
    JSON line, the other ranks' stdout goes to stderr.  Returns the worst exit code; when a rank fails the others get a
    grace period to notice through the rendezvous, then they are ended by pid.

    `baseline`: the CPU-baseline legs, run by this parent before it starts the children (they fork worker processes
    and take ~25 s: not something a rank should do while its peers wait at the rendezvous); rank 0 puts it in the line.

    ONE retry: the ranks run with HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller's environment says otherwise -- this
    pool's images export it, and its documentation gives the reason: the host driver supports dmabuf IPC only, without
    it RCCL across processes fails in hipIpcGetMemHandle.  The builder has had no multi-GPU box to confirm it on.  If
    EVERY rank comes back with exit code 3 and rank 0's line says the communicator could not be built, the launcher
    -- which has not touched the GPU -- starts a second, fresh set of ranks with the variable UNSET and marks the
    line (`config.launcher_retry`).
    """
    import secrets
    import tempfile

    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_PORT", "0")
    handed = None
    if baseline is not None:
        with tempfile.NamedTemporaryFile("w", suffix=".json", prefix="lynx-cpu-baseline-", delete=False) as fh:
            json.dump(baseline, fh)
            handed = fh.name
        env["LYNX_BENCH_CPU_BASELINE_FILE"] = handed
    try:
        for attempt in (0, 1):
            env["LYNX_RDZV_KEY"] = f"bench-{os.getpid()}-{secrets.token_hex(8)}"
            codes, line = _run_ranks(n_ranks, env)
            rccl_failed = all(c == 3 for c in codes) and b'"rccl-failed"' in line
            if attempt == 0 and rccl_failed and "HSA_ENABLE_IPC_MODE_LEGACY" in env:
                print(f"[launcher] every rank failed to build its RCCL communicator with HSA_ENABLE_IPC_MODE_LEGACY="
                      f"{env['HSA_ENABLE_IPC_MODE_LEGACY']}; one more set of ranks with the variable unset", file=sys.stderr, flush=True)
                env["LYNX_LAUNCHER_RETRY"] = f"second set of ranks, HSA_ENABLE_IPC_MODE_LEGACY unset (first set with ={env.pop('HSA_ENABLE_IPC_MODE_LEGACY')}: RCCL bring-up failed on every rank)"
                continue
            break
    finally:
        if handed:
            try:
                os.unlink(handed)
            except OSError:
                pass
    sys.stdout.buffer.write(line)
    sys.stdout.flush()
    return max([0] + [c if c > 0 else 1 for c in codes if c != 0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("LYNX_BENCH_WORKLOAD", "c4"), choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="override the workload's batch (global; per GPU with --weak)")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: every GPU tracks the whole per-GPU workload (global batch = batch x N) instead of "
                         "a slice of ONE workload")
    ap.add_argument("--no-moments", action="store_true", help="track without the fused moment epilogue")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shared-input", action="store_true",
                    help="variant (SURVEY.md section 8d): ONE incoming beam broadcast lazily over the batch instead of "
                         "one physical beam per sample; reported with its own algorithmic bytes, never the default")
    ap.add_argument("--grad", action="store_true",
                    help="step = forward + reverse pass (gradient of sum of var(x) w.r.t. every element parameter)")
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N > 1 only: if the RCCL communicator cannot be built, gather the moment records through the "
                         "rendezvous sockets instead of failing (config.gather says so); without it the run exits 3")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket the streaming kernel with HIP events (then `roofline.achieved` is null): "
                         "shows what the per-launch time stamps themselves cost a step")
    ap.add_argument("--sync-every-step", action="store_true",
                    help="latency mode: wait for the GPU after every step (no overlap between consecutive calls)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="rehearsal without a GPU: launch, rendezvous, shard plan, barriers and the max over ranks "
                         "only; prints a line with \"value\": null and \"dry_run\": true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing has touched the GPU yet, and this process never will: CPU baseline here, then the ranks
        w = WORKLOADS[args.workload]
        parent_baseline = None
        # (a rehearsal runs it only when a test asks for it: LYNX_BENCH_TEST_CPU_BASELINE=1)
        if not args.no_cpu_baseline and (not args.rendezvous_only or os.environ.get("LYNX_BENCH_TEST_CPU_BASELINE") == "1"):
            parent_baseline = cpu_baseline(args.workload, args.particles or w[1], w[2], np.dtype(w[3]).type,
                                           budget_s=float(os.environ.get("LYNX_BENCH_CPU_BUDGET_S", "10")))
        sys.exit(launch_ranks(args.gpus, parent_baseline))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or without a "
                 "launcher (bench.py starts its own ranks)")

    batch, particles, cells, dtype, descr = WORKLOADS[args.workload]
    batch = args.batch or batch
    particles = args.particles or particles
    dtype = np.dtype(dtype).type
    itemsize = np.dtype(dtype).itemsize

    # CPU baseline first: it forks worker processes, which must happen before this process opens the GPU
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rendezvous_only:
        baseline = cpu_baseline(args.workload, particles, cells, dtype,
                                budget_s=float(os.environ.get("LYNX_BENCH_CPU_BUDGET_S", "10")))
    elif rank == 0 and os.environ.get("LYNX_BENCH_CPU_BASELINE_FILE"):  # the launcher parent ran it (N > 1)
        try:
            baseline = json.loads(Path(os.environ["LYNX_BENCH_CPU_BASELINE_FILE"]).read_text())
        except (OSError, ValueError) as exc:
            print(f"cpu baseline of the launcher not readable: {exc}", file=sys.stderr)

    from lynx_amd.rendezvous import Rendezvous

    # this rank's piece of the job
    global_batch, sample_ids, my_particles, total_particles, parallelism = shard_of(batch, particles, world, rank, args.weak)
    my_batch = len(sample_ids)
    rows = max(len(shard_of(batch, particles, world, r, args.weak)[1]) for r in range(world))  # gathered rows per rank
    strong = world > 1 and not args.weak

    rdzv = Rendezvous(rank, world)  # standard-library sockets; no second communication stack in the process
    config = {"workload": f"{args.workload}: {descr}", "batch_per_gpu": my_batch, "global_batch": global_batch,
              "particles": total_particles, "particles_per_gpu": my_particles, "elements": None,
              "fused_moments": not args.no_moments, "reverse_pass": bool(args.grad),
              "incoming_beam": "one beam shared by the batch (lazy broadcast)" if args.shared_input else "one physical beam per sample",
              "gather": "none", "parallelism": parallelism, "pipelined_calls": not args.sync_every_step,
              "launcher": "none (one process)" if world == 1 and "WORLD_SIZE" not in os.environ
                          else "torchrun/external" if os.environ.get("TORCHELASTIC_RUN_ID") or "LYNX_RDZV_KEY" not in os.environ
                          else "bench.py"}
    if os.environ.get("LYNX_LAUNCHER_RETRY"):
        config["launcher_retry"] = os.environ["LYNX_LAUNCHER_RETRY"]
    if world > 1:
        config["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    scaling = "weak" if (args.weak and world > 1) else "strong"

    if args.rendezvous_only:
        # the N > 1 plumbing without a GPU: every rank reports its slice, rank 0 checks that the slices tile the job
        import struct

        first = int(sample_ids[0]) if my_batch else 0
        parts = rdzv.all_gather(struct.pack("<qqq", first, my_batch, my_particles))
        rdzv.barrier()
        slowest = rdzv.max(float(rank))
        if rank == 0:
            plan = [struct.unpack("<qqq", p) for p in parts]
            if strong and batch > 1:
                assert [p[0] for p in plan] == list(np.cumsum([0] + [p[1] for p in plan[:-1]])) and sum(p[1] for p in plan) == batch
            if strong and batch == 1:
                assert sum(p[2] for p in plan) == particles
            assert slowest == world - 1
            line = {"metric": METRIC, "value": None, "unit": "particle-element-steps/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "scaling": scaling, "dry_run": True,
                    "config": dict(config, shard_plan=[{"first_sample": p[0], "samples": p[1], "particles": p[2]}
                                                       for p in plan])}
            if baseline is not None:
                line["cpu_baseline"] = baseline
            # rehearsal of the launcher's one retry (tests/test_distributed.py): behave like ranks whose RCCL bring-up failed
            if os.environ.get("LYNX_BENCH_TEST_RCCL_FAILS_WITH_IPC_LEGACY") and "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
                line["config"]["gather"] = "rccl-failed"
            print(json.dumps(line))
        rdzv.close()
        if os.environ.get("LYNX_BENCH_TEST_RCCL_FAILS_WITH_IPC_LEGACY") and "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
            sys.exit(3)
        sys.exit(int(os.environ.get("LYNX_BENCH_TEST_EXIT", "0")) if rank == int(os.environ.get("LYNX_BENCH_TEST_EXIT_RANK", "-1")) else 0)

    import lynx_amd as lx
    from lynx_amd.device import get_runtime

    rt = get_runtime()
    lx.config.fused_moments = not args.no_moments

    segment = build_segment(lx, args.workload, sample_ids, cells, dtype, seed=3 + (rank if args.weak else 0))
    n_elements = len(segment.elements)
    config["elements"] = n_elements
    batch = my_batch        # from here on: this rank's samples and particles
    particles = my_particles
    beam = lx.ParticleBeam.synthetic((1,) if args.shared_input else (batch,), particles, sigma=BEAM_SIGMA,
                                     energy=beam_energy(args.workload), seed=2 + rank, dtype=dtype)
    if args.shared_input:
        beam = beam.broadcast((batch,))
        assert beam.is_shared

    def finish(code):
        """Ordinary end of the process; the hard exit is only for a thread stuck inside RCCL."""
        rdzv.close()
        sys.stdout.flush()
        sys.stderr.flush()
        if code == "stuck":
            os._exit(3)
        sys.exit(code)

    comm = None
    force_comm = os.environ.get("LYNX_FORCE_COMM") == "1"  # exercise the RCCL path at world_size 1
    if world > 1 or force_comm:
        comm, why, stuck = bring_up_rccl(rt, rdzv, rank, world, float(os.environ.get("LYNX_COMM_TIMEOUT_S", "300")))
        if comm is None:
            print(f"[rank {rank}] RCCL communicator failed: {why}", file=sys.stderr, flush=True)
        try:
            everyone = (not stuck) and rdzv.all_true(comm is not None)
        except (OSError, TimeoutError, ConnectionError) as exc:  # a peer is gone
            everyone, why = False, why or f"a peer left the rendezvous ({exc})"
        if stuck or not everyone:
            config["gather"] = "rccl-failed"
            if not stuck and args.allow_host_gather:
                if comm is not None:
                    comm.close()
                    comm = None
                config["gather"] = "host-tcp-fallback"
            else:
                if rank == 0:
                    print(json.dumps({"metric": METRIC, "value": None, "unit": "particle-element-steps/s",
                                      "n_gpus": world, "error": f"RCCL communicator could not be built: {why}",
                                      "config": config}))
                finish("stuck" if stuck else 3)
        else:
            config["gather"] = "rccl-allgather"
            config.update(comm.info())
            assert config["rccl_ranks"] == world, config

    grad_cov_bar = None
    if args.grad:
        import lynx_amd.grad as lgrad

        grad_cov_bar = np.zeros((batch, 6, 6))
        grad_cov_bar[:, 0, 0] = 1.0

    def step():
        if args.grad:
            vjp = lgrad.track_vjp(segment, beam)
            grads = vjp(cov_bar=grad_cov_bar)
            return vjp.outgoing, grads
        out = segment.track(beam)
        if args.sync_every_step:
            rt.sync()
        if out._moments is None or (world == 1 and comm is None):
            return out, None
        if comm is not None:
            local = out._moments.device(rt).reshape(batch, 36)
            if batch != rows:  # uneven slices (the batch is not a multiple of N): every rank sends `rows` rows
                padded = rt.empty((rows, 36), np.float64)
                rt.check(rt.lib.lynx_buf_d2d(rt.ctx, padded.ptr, local.ptr, local.nbytes))
                local = padded
            return out, comm.all_gather(local)
        mine = np.zeros((rows, 36))
        mine[:batch] = out.moment_record().reshape(batch, 36)
        parts = rdzv.all_gather(mine.tobytes())
        return out, np.stack([np.frombuffer(p, dtype=np.float64).reshape(rows, 36) for p in parts])

    last = None
    for _ in range(args.warmup):
        last = step()
    rt.sync()
    # The practical ceiling on this box -- a plain 16 B/lane device copy of one pass' bytes, five launch shapes -- is
    # measured HERE, between the warm-up and the timed region, on every rank, for about 0.1 s.  The streaming kernel
    # has a start-up transient of 20-30 ms (BASELINE config 4: 1.00, 1.00, 1.09, 1.17, 1.16, ... 0.95 ms per launch
    # after 20 launches, the same curve with nothing running next to it and also right behind other kernels; plain
    # copies show none: profiles/r03_first_launches.txt), and five warm-up steps end inside it.
    # Behind the calibration's copies the kernel's second start is much milder (1.00 -> 0.94 over the 20 timed steps),
    # which is closer to what a long job sees.  LYNX_BENCH_CALIBRATE_FIRST=0 puts the calibration behind the timed
    # region again (rank 0 of a one-GPU run only).
    copy_gbs = None
    calibrate_first = os.environ.get("LYNX_BENCH_CALIBRATE_FIRST", "1") != "0"

    def calibrate():
        nbytes = min(batch * particles * 7 * np.dtype(dtype).itemsize, 4 << 30)
        est_s = 2 * nbytes / 5e12  # one copy launch
        repeats = int(min(200, max(10, np.ceil(0.1 / (5 * est_s))))) if calibrate_first else 10
        try:
            return rt.copy_bandwidth(nbytes, repeats=repeats)
        except Exception as exc:  # pragma: no cover
            print(f"copy calibration failed: {exc}", file=sys.stderr)
            return None

    def timed(steps):
        """K steps between two barriers -> (slowest rank's seconds, this rank's seconds, kernel ms, launches, per-launch ms, gather ms)."""
        nonlocal last
        rdzv.barrier()
        rt.sync()
        if not args.no_kernel_timing:
            rt.profile_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        rt.sync()
        mine = time.perf_counter() - t0  # this rank's K steps, GPU drained
        rdzv.barrier()
        kern_ms, launches = rt.profile_end() if not args.no_kernel_timing else (0.0, 0)
        per_launch = rt.profile_launches() if not args.no_kernel_timing else []
        gathers = rt.profile_gathers() if not args.no_kernel_timing else []
        return rdzv.max(mine), mine, kern_ms, launches, per_launch, gathers

    # The same K steps twice.  COLD: right behind the W warm-up steps, as the command line says -- this is what a short
    # job sees (`ms_per_step_cold`, `launch_ms_cold`).  Then the calibration copies, then the K steps `value` is
    # computed from (LYNX_BENCH_CALIBRATE_FIRST=0: no cold leg, calibration behind the timed region).
    cold = None
    if calibrate_first:
        cold = timed(args.steps)
        copy_gbs = calibrate()
    elapsed, my_elapsed, kern_ms, launches, per_launch, gather_ms = timed(args.steps)
    import struct

    per_rank_s = [struct.unpack("<d", b)[0] for b in rdzv.all_gather(struct.pack("<d", my_elapsed))]

    # sanity on the last result (outside the timed region): finite moments, right count
    out, gathered = last
    if out._moments is not None:
        rec = out.moment_record()
        have = ~np.isnan(rec)  # a property-set record marks the covariance entries it does not carry with NaN
        assert np.all(have[..., :7]) and np.all(have[..., [7, 8, 13, 18, 19, 22, 25, 27]]), "bench: moments missing"
        assert np.all(np.isfinite(rec[have])) and np.all(rec[..., 35] == particles), "bench: bad moment records"
    if gathered is not None and not args.grad:
        g = np.asarray(gathered)
        assert g.shape == (world, rows, 36) and np.allclose(g[rank, :batch], rec.reshape(batch, 36), equal_nan=True)
        if global_batch == 1 and world > 1:
            # one sample: the ranks hold disjoint particle slices of one beam (SURVEY.md section 8e,
            # secondary partitioning); its record is the merge of the gathered slice records
            from lynx_amd.parallel import merge_records

            whole = merge_records(g)
            want = total_particles if strong else world * particles
            assert whole[0, 35] == want and np.all(np.isfinite(whole[~np.isnan(whole)])), "bench: bad merged record"
        elif strong:
            # the scan's records in batch order, as a single-GPU run would hold them
            from lynx_amd.parallel import assemble_records

            whole = assemble_records(g, global_batch, world)
            assert whole.shape == (global_batch, 36) and np.all(whole[:, 35] == total_particles), "bench: bad gathered records"
            assert np.all(np.isfinite(whole[~np.isnan(whole)])), "bench: bad gathered records"

    if not calibrate_first and rank == 0 and world == 1:
        copy_gbs = calibrate()

    # HBM traffic of the streaming kernel as measured with rocprofv3 PMC counters (separate
    # FETCH_SIZE / WRITE_SIZE passes of this same command; profiles/*_pmc_traffic.json).  It
    # cannot be collected from inside the process, so the committed measurement is quoted and
    # its provenance named; null for workloads that were not profiled.
    traffic, traffic_src = None, None
    pmc = sorted((ROOT / "profiles").glob(f"*_{args.workload}_pmc_traffic.json"))
    if pmc and batch == WORKLOADS[args.workload][0] and particles == WORKLOADS[args.workload][1]:
        try:
            rec = json.loads(pmc[-1].read_text())
            # (the streaming kernel of the workload: k_track_direct, or k_track_unit_pairs / k_track_units for config 5)
            stream = "k_track_unit" if args.workload == "c5" else "k_track_direct"
            fetch = next(v["mean_KB"] for k, v in rec.items() if stream in k and k.endswith("FETCH_SIZE"))
            write = next(v["mean_KB"] for k, v in rec.items() if stream in k and k.endswith("WRITE_SIZE"))
            # gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM);
            # calibrated on k_diag_copy in the same profile run.  Units: KiB.
            traffic = (2.0 * fetch + write) * 1024.0
            traffic_src = f"profiles/{pmc[-1].name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE x2 on gfx950)"
        except Exception as exc:  # pragma: no cover
            print(f"could not read {pmc[-1]}: {exc}", file=sys.stderr)

    if rank == 0:
        # particle passes of the whole job per step: one workload cut in N (strong), or N workloads (weak)
        passes = global_batch * total_particles if strong else batch * world * particles
        steps_per_pass = passes * n_elements
        alg_bytes = 2 * batch * particles * 7 * itemsize  # per launch of the streaming kernel, per GPU
        if args.shared_input:  # the incoming beam is read once, every sample's outgoing beam is written
            alg_bytes = (1 + batch) * particles * 7 * itemsize
        kern_s = kern_ms / 1e3 / max(launches, 1)
        achieved = alg_bytes / kern_s / 1e9 if launches else None
        result = {
            # BASELINE.json's metric, verbatim; `value` is its first half (whole-node steps/s), the
            # achieved HBM GB/s is `roofline.achieved` (kernel) and `hbm_gbs_whole_step` (wall clock)
            "metric": METRIC,
            "value": steps_per_pass * args.steps / elapsed,
            "unit": "particle-element-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the same K steps right behind the W warm-up steps, before the calibration copies (the streaming kernel's
            # start-up transient, DESIGN.md section 5, is in this figure and mostly out of `ms_per_step`)
            "ms_per_step_cold": (cold[0] / args.steps * 1e3) if cold else None,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if itemsize == 4 else "f64",
            "data": "synthetic",
            "config": config,
            # config 5's step loop is the structured one (lynx_units.hpp) unless LYNX_TRACK_UNITS=0 asks for the dense loop
            # (its cells are merged [run, cavity] pairs of class U: k_track_unit_pairs, unless LYNX_UNIT_PAIRS=0 keeps the general kernel)
            "roofline": {"bound": "hbm", "kernel": ("k_track_direct" if (args.workload != "c5" or os.environ.get("LYNX_TRACK_UNITS", "1") == "0")
                                                    else "k_track_units" if os.environ.get("LYNX_UNIT_PAIRS", "1") == "0" else "k_track_unit_pairs"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": kern_s * 1e3, "launches": launches,
                         "avg_launch_ms_cold": (cold[2] / max(cold[3], 1)) if cold else None,
                         # every launch of the timed steps, in order (at most the first 64 of each leg)
                         "launch_ms": [round(v, 4) for v in per_launch[:64]],
                         "launch_ms_cold": [round(v, 4) for v in cold[4][:64]] if cold else None},
            "hbm_gbs_whole_step": (alg_bytes if args.shared_input else 2 * passes * 7 * itemsize) * args.steps / elapsed / 1e9,
            # the practical ceiling: best plain-copy shape measured in this process (MI355X_MICROARCH.md: ~6.3 TB/s)
            "hbm_copy_kernel_gbs": max(copy_gbs.values()) if copy_gbs else None,
            "hbm_copy_kernel_shapes": copy_gbs,
            "hbm_copy_kernel_when": "between the cold leg (W warm-up steps + K timed steps: ms_per_step_cold) and the timed region of `value`"
                                    if calibrate_first else "after the timed region",
            "device": rt.info(),
        }
        if world > 1 or comm is not None:
            # diagnosis of a scaling result from ONE run: every rank's own clock, and what the gathers took on the stream
            # they ran on (with more than one rank that includes the wait for the slowest rank of the step)
            result["ms_per_step_per_rank"] = [round(t / args.steps * 1e3, 5) for t in per_rank_s]
            result["gather_ms"] = {"n": len(gather_ms), "mean": float(np.mean(gather_ms)) if gather_ms else None,
                                   "max": float(np.max(gather_ms)) if gather_ms else None, "rank": 0}
        if baseline is not None:
            result["cpu_baseline"] = baseline
        print(json.dumps(result))
    if comm is not None:
        comm.close()
    finish(0)


if __name__ == "__main__":
    main()
