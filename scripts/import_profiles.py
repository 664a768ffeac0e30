#!/usr/bin/env python3
"""
Copy the summaries of one profile series (scripts/gpu/prof_r2.sh, SERIES=<letter>, merged back by gpurun into
gpurun_out/prof_r2_<letter>/) into profiles/ under the names profiles/README.md describes.

    python scripts/import_profiles.py a            # gpurun_out/prof_r3_a -> profiles/r03_a_*
    python scripts/import_profiles.py d r02        # gpurun_out/prof_r2_d -> profiles/r02_d_*
"""
import glob
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
series = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = ROOT / "gpurun_out" / f"prof_r{int(rnd[1:])}_{series}"
dst = ROOT / "profiles"
copied = []


def put(path, name):
    if Path(path).is_file() and Path(path).stat().st_size > 0:
        shutil.copyfile(path, dst / f"{rnd}_{series}_{name}")
        copied.append(name)


for w in ("c4", "c4a", "c3", "c3big", "c5", "c2", "c5grad"):
    stats = sorted(glob.glob(str(src / f"trace_{w}" / "*" / "*kernel_stats.csv")))
    if stats:
        put(stats[-1], f"{w}_kernel_stats.csv")
    put(src / f"trace_{w}.json", f"{w}_bench_under_rocprof.json")
for w in ("c4", "c4a", "c3big", "c5"):
    put(src / f"{w}_pmc_traffic.json", f"{w}_pmc_traffic.json")
put(src / "default.json", "c4_bench_default.json")
put(src / "rccl_world1.json", "c4_bench_rccl_world1.json")
for w in ("c3", "c2"):
    put(src / f"steady_{w}.json", f"{w}_bench_steady.json")
    put(src / f"latency_{w}.json", f"{w}_bench_waited_for.json")
put(src / "latency_table.json", "latency_table.json")
# round 3: the strong-scaling shards of config 4, config 5 A/B and default line, SQ counters of config 5's kernels
put(src / "strong_scaling_shards.json", "c4_strong_scaling_shards.json")
for b in (1024, 512, 256, 128):
    put(src / f"shard_b{b}_1.json", f"c4_shard_b{b}_bench.json")
put(src / "c5_units1.json", "c5_bench_units.json")
put(src / "c5_units0.json", "c5_bench_dense_step_loop.json")
put(src / "c5_general.json", "c5_bench_general_kernel.json")
put(src / "c5_default.json", "c5_bench_default.json")
put(src / "c5grad.json", "c5grad_bench.json")
put(src / "steady_c3big.json", "c3big_bench_steady.json")
put(src / "pmc_c5" / "c5_pmc_sq.json", "c5_pmc_sq.json")
# round 4: the corrector-angle variant of config 4, the reverse pass's counters, the dense reverse pass, kernel timelines
put(src / "c4a_default.json", "c4a_bench_default.json")
put(src / "pmc_c5" / "c5grad_pmc_sq.json", "c5grad_pmc_sq.json")
put(src / "c5grad_dense.json", "c5grad_bench_dense_reverse.json")
put(src / "timeline.txt", "small_configs_kernel_timeline.txt")
print(f"{len(copied)} files -> profiles/{rnd}_{series}_*:", ", ".join(copied))
