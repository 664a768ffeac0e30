#!/bin/bash
# Round 4: every other long call walks the batch from its end (LYNX_ALTERNATE_ORDER, default 1), so that what the previous
# pass over the SAME incoming beam left in the 256 MB Infinity Cache is what the next pass reads first.
TAG=${1:-r4alt}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f"{sys.argv[2]:30s} us/step {1e3 * d['ms_per_step']:8.2f}  cold {1e3 * (d.get('ms_per_step_cold') or 0):8.2f}  kernel {1e3 * (r['avg_launch_ms'] or 0):7.2f}")
PY
}
for rep in 1 2 3; do for alt in 0 1; do
  export LYNX_ALTERNATE_ORDER=$alt
  run c4_alt${alt}_$rep --gpus 1 --steps 20 --warmup 5
  run c3big_alt${alt}_$rep --workload c3big --steps 100 --warmup 5
done; done
