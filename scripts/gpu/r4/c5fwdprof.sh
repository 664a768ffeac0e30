#!/bin/bash
# Round 4: per-kernel times of BASELINE config 5, forward only (rocprofv3 --kernel-trace --stats).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r4c5fwd}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do export "$e"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "diag_copy" in r["Name"] or "fill_gaussian" in r["Name"]: continue
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
