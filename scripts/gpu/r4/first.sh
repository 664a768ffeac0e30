#!/bin/bash
# Round 4, first GPU call: the whole GPU suite (new full-size parity tests included), then one bench line per workload.
# Usage (from the repo root, on the GPU box): bash scripts/gpu/r4/first.sh <tag>
set -o pipefail
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 -s > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed ($?)"
  python - "$OUT/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], "ms/step", round(d["ms_per_step"], 5), "cold", d.get("ms_per_step_cold") and round(d["ms_per_step_cold"], 5),
          "kernel ms", r["avg_launch_ms"] and round(r["avg_launch_ms"], 5), "frac", r["frac"] and round(r["frac"], 4))
except Exception as exc:
    print(sys.argv[1], "unreadable:", exc)
PY
}
run c4_default --gpus 1 --steps 20 --warmup 5
run c4a --workload c4a --steps 20 --warmup 5 --no-cpu-baseline
run c3 --workload c3 --steps 200 --warmup 10 --no-cpu-baseline
run c3_waited --workload c3 --steps 200 --warmup 10 --no-cpu-baseline --sync-every-step
run c3big --workload c3big --steps 60 --warmup 5 --no-cpu-baseline
run c2 --workload c2 --steps 500 --warmup 20 --no-cpu-baseline
run c5 --workload c5 --steps 40 --warmup 5 --no-cpu-baseline
run c5grad --workload c5 --grad --steps 20 --warmup 3 --no-cpu-baseline
