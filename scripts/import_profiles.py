#!/usr/bin/env python3
"""
Copy the summaries of one profile series (scripts/gpu/prof_r2.sh, SERIES=<letter>, merged back by gpurun into
gpurun_out/prof_r2_<letter>/) into profiles/ under the names profiles/README.md describes.

    python scripts/import_profiles.py d            # -> profiles/r02_d_*
"""
import glob
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
series = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = ROOT / "gpurun_out" / f"prof_r2_{series}"
dst = ROOT / "profiles"
copied = []


def put(path, name):
    if Path(path).is_file() and Path(path).stat().st_size > 0:
        shutil.copyfile(path, dst / f"{rnd}_{series}_{name}")
        copied.append(name)


for w in ("c4", "c3", "c3big", "c5", "c2", "c5grad"):
    stats = sorted(glob.glob(str(src / f"trace_{w}" / "*" / "*kernel_stats.csv")))
    if stats:
        put(stats[-1], f"{w}_kernel_stats.csv")
    put(src / f"trace_{w}.json", f"{w}_bench_under_rocprof.json")
for w in ("c4", "c3big"):
    put(src / f"{w}_pmc_traffic.json", f"{w}_pmc_traffic.json")
put(src / "default.json", "c4_bench_default.json")
put(src / "rccl_world1.json", "c4_bench_rccl_world1.json")
for w in ("c3", "c2"):
    put(src / f"steady_{w}.json", f"{w}_bench_steady.json")
    put(src / f"latency_{w}.json", f"{w}_bench_waited_for.json")
put(src / "latency_table.json", "latency_table.json")
print(f"{len(copied)} files -> profiles/{rnd}_{series}_*:", ", ".join(copied))
