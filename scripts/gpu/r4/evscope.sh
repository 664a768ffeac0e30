#!/bin/bash
# Round 4: the scope of the release behind the library's inter-stream events (LYNX_EVENT_SCOPE: 0 = HIP's default, a
# system-scope fence; 1 = hipEventReleaseToDevice; 2 = hipEventDisableSystemFence) over the workloads.  Usage: evscope.sh <tag>
TAG=${1:-r4evscope}; OUT=gpurun_out/$TAG; mkdir -p $OUT
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
for scope in 0 1 2; do
  export LYNX_EVENT_SCOPE=$scope
  run c4_scope${scope}_$rep --gpus 1 --steps 20 --warmup 5
  LYNX_FORCE_COMM=1 run shard128_scope${scope}_$rep --batch 128 --steps 60 --warmup 5
  run c3big_scope${scope}_$rep --workload c3big --steps 100 --warmup 5
  run c3_scope${scope}_$rep --workload c3 --steps 300 --warmup 20
  run c5_scope${scope}_$rep --workload c5 --steps 40 --warmup 5
done
done
