#!/bin/bash
# Round 4: BASELINE config 3 (1 M particles, float64, one sample) under launch-plan knobs: pipelined and waited for.
OUT=gpurun_out/r4c3knobs; mkdir -p $OUT
run() { local name=$1; shift
  for mode in p w; do
    extra=""; [ $mode == w ] && extra="--sync-every-step"
    env "$@" timeout -k 10 200 python bench.py --workload c3 --steps 300 --warmup 20 --no-cpu-baseline $extra > $OUT/${name}_$mode.json 2> $OUT/${name}_$mode.err
    python - $OUT/${name}_$mode.json $name $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {'pipelined' if sys.argv[3]=='p' else 'waited   '} us/step {1e3*d['ms_per_step']:7.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:6.2f}")
PY
  done
}
run default LYNX_NOOP=1
run tiles2 LYNX_MIN_TILES_PER_WG=2
run tiles4 LYNX_MIN_TILES_PER_WG=4
run tiles8 LYNX_MIN_TILES_PER_WG=8
run tiles4_twolaunch LYNX_MIN_TILES_PER_WG=4 LYNX_REDUCE_TICKET=0
run wide_tiles2 LYNX_MIN_TILES_PER_WG=2 LYNX_REDUCE_WIDE=1
