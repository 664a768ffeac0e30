#!/bin/bash
# Round 4: BASELINE config 5 forward (and forward + reverse) with the kernel written for lattices of merged [run, cavity]
# pairs of class U (LYNX_UNIT_PAIRS, default 1) and with the general one.
TAG=${1:-r4pairform}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in 1 0 1 0; do
  LYNX_UNIT_PAIRS=$v timeout -k 10 200 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5_pairs$v.json 2>> $OUT/err.log || exit 1
  python3 - $OUT/c5_pairs$v.json $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"LYNX_UNIT_PAIRS={sys.argv[2]}  ms_per_step {r['ms_per_step']:.4f}  cold {r['ms_per_step_cold']:.4f}  kernel {r['roofline']['avg_launch_ms']:.4f}  frac {r['roofline']['frac']:.3f}")
PY
done
