#!/bin/bash
# Round 4 experiment: the build stream confined to n CUs (LYNX_BUILD_CUS) so that the next call's build does not take
# wave slots from the streaming kernel everywhere.  Usage: cumask.sh <tag>
TAG=${1:-r4cumask}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[2]:30s} us/step {1e3 * d['ms_per_step']:8.2f}  cold {1e3 * (d.get('ms_per_step_cold') or 0):8.2f}  kernel {1e3 * (r['avg_launch_ms'] or 0):7.2f}")
except Exception as exc:
    print(sys.argv[1], "unreadable", exc)
PY
}
for rep in 1 2; do
for n in 0 16 32 64; do
  export LYNX_BUILD_CUS=$n
  run c5_cus${n}_$rep --workload c5 --steps 40 --warmup 5
  run c4_cus${n}_$rep --gpus 1 --steps 20 --warmup 5
  LYNX_FORCE_COMM=1 run shard128_cus${n}_$rep --batch 128 --steps 60 --warmup 5
done
done
