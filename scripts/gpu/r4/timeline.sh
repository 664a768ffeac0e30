#!/bin/bash
# Round 4: kernel timeline (rocprofv3 --kernel-trace) of the small configurations, pipelined and waited for.
# Usage: bash scripts/gpu/r4/timeline.sh <tag> [extra env assignments ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r4tl}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do export "$e"; done
for w in c3 c2; do
  for mode in pipelined waited; do
    extra=""; [ $mode == waited ] && extra="--sync-every-step"
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/${w}_$mode -- python3 bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing $extra > $OUT/${w}_$mode.json 2> $OUT/${w}_$mode.err
    python3 - $OUT/${w}_$mode <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "diag_copy" not in r["Kernel_Name"] and "fill_gaussian" not in r["Kernel_Name"]]
tail = rows[-16:]
t0 = int(tail[0]["Start_Timestamp"])
print(sys.argv[1])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f}"
    print(f"  {r['Kernel_Name'][:60]:60s} q{r.get('Queue_Id', '?'):>3} start {s / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f} us  {gap}")
    prev_end = e
PY
  done
done
