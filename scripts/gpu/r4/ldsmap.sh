#!/bin/bash
# Round 4 experiment (VERDICT next #4): the headline kernel with its map read row by row from LDS instead of 49 scalars
# spilled to vector lanes -- the driver's command, three runs each, alternating; first-launch curves in the lines.
OUT=gpurun_out/r4ldsmap; mkdir -p $OUT
for rep in 1 2 3; do
  for v in base ldsmap; do
    lib=lynx_amd/_lib/liblynxhip.so; [ $v == ldsmap ] && lib=build/liblynxhip_ldsmap.so
    LYNX_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
    python - $OUT/${v}_$rep.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f"{sys.argv[2]:7s} ms/step {d['ms_per_step']:.4f} cold {d['ms_per_step_cold']:.4f} kernel {r['avg_launch_ms']:.4f} cold kernel {r['avg_launch_ms_cold']:.4f}  cold launches {[round(x,3) for x in r['launch_ms_cold'][:12]]}")
PY
  done
done
