#!/bin/bash
# Round 4: kernel trace of the gradient-based tuning example (examples/gradient_based_tuning.py): where a step's time goes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r4example}; mkdir -p $OUT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 examples/gradient_based_tuning.py > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']:>6}%")
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"]); prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev is None else f"gap {(s - prev) / 1e3:7.1f}"
    print(f"  {r['Kernel_Name'][:70]:70s} start {s / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f} us  {gap}")
    prev = e
PY
