#!/bin/bash
# Round 4 experiment: k_build_bwd cut short behind its phases 1 (element maps), 2 (chains), 3 (dual tasks): kernel time each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 4 5 6; do
  OUT=gpurun_out/r4bwdphase_$n; mkdir -p $OUT
  LYNX_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/liblynxhip_stop$n.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload c5 --grad --steps 6 --warmup 2 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python3 - $OUT $n <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "k_build_bwd" in r["Name"]:
        print(f"stop after phase {sys.argv[2]}: k_build_bwd avg {float(r['AverageNs'])/1e3:.1f} us")
PY
done
