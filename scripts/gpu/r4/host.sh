#!/bin/bash
# Round 4: the small configurations (host path): bench lines pipelined / waited for, and the latency table.
# Usage: bash scripts/gpu/r4/host.sh <tag> [ENV=value ...]
TAG=${1:-r4h}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do export "$e"; done
run() { local name=$1; shift
  timeout -k 10 200 python bench.py "$@" --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
  python - $OUT/$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(f"{sys.argv[1]:45s} us/step {1e3 * d['ms_per_step']:7.2f}  cold {1e3 * (d.get('ms_per_step_cold') or 0):7.2f}  kernel {1e3 * (r['avg_launch_ms'] or 0):6.2f}")
except Exception as exc:
    print(sys.argv[1], "unreadable", exc)
PY
}
for rep in 1 2; do
run c3_$rep --workload c3 --steps 300 --warmup 20
run c3_waited_$rep --workload c3 --steps 300 --warmup 20 --sync-every-step
run c2_$rep --workload c2 --steps 1000 --warmup 50
run c2_waited_$rep --workload c2 --steps 1000 --warmup 50 --sync-every-step
done
run c3big --workload c3big --steps 100 --warmup 5
PYTHONPATH=. timeout -k 10 300 python scripts/gpu/latency.py > $OUT/latency_table.json 2> $OUT/latency_table.err; cat $OUT/latency_table.json
