#!/bin/bash
# Round 4: BASELINE config 5 -- gradient tests, then forward and forward + reverse bench lines (two runs each).
# Usage: bash scripts/gpu/r4/c5.sh <tag> [ENV=value ...]
TAG=${1:-r4c5}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do export "$e"; done
timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -q -x -s > $OUT/pytest_grad.log 2>&1; echo "grad tests exit $?"
grep -E "passed|failed|Error" $OUT/pytest_grad.log | tail -5
grep -A8 "structured-dense" $OUT/pytest_grad.log | head -10
for rep in 1 2; do
  for mode in fwd grad; do
    extra=""; [ $mode == grad ] && extra="--grad"
    timeout -k 10 200 python bench.py --workload c5 $extra --steps 30 --warmup 5 --no-cpu-baseline > $OUT/c5_${mode}_$rep.json 2> $OUT/c5_${mode}_$rep.err || echo "c5 $mode failed"
    python - $OUT/c5_${mode}_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} ms/step {d['ms_per_step']:.4f}  cold {d.get('ms_per_step_cold') or 0:.4f}  fwd kernel {d['roofline']['avg_launch_ms']:.4f}")
PY
  done
done
