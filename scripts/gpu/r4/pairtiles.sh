#!/bin/bash
# Round 4: tiles per workgroup (LYNX_MIN_TILES_PER_WG) for BASELINE config 5 with k_track_unit_pairs.
TAG=${1:-r4pairtiles}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in 16 7 5 4 2 16 7 5 4 2; do
  LYNX_MIN_TILES_PER_WG=$v timeout -k 10 200 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5_tiles$v.json 2>> $OUT/err.log || exit 1
  python3 - $OUT/c5_tiles$v.json $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"LYNX_MIN_TILES_PER_WG={sys.argv[2]}  ms_per_step {r['ms_per_step']:.4f}  cold {r['ms_per_step_cold']:.4f}  kernel {r['roofline']['avg_launch_ms']:.4f}  frac {r['roofline']['frac']:.3f}")
PY
done
