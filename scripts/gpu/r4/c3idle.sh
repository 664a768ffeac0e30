#!/bin/bash
# Round 4: config 3 pipelined / waited for with the short-call policy's forms forced, then the kernel timeline of the default.
# Usage: bash scripts/gpu/r4/c3idle.sh <tag>
TAG=${1:-r4c3idle}; OUT=gpurun_out/$TAG; mkdir -p $OUT
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
run inline LYNX_SMALL_INLINE=1
run async LYNX_SMALL_INLINE=0
run default_again LYNX_NOOP=1
bash scripts/gpu/r4/timeline.sh $TAG/tl > $OUT/timeline_default.txt 2>&1
bash scripts/gpu/r4/timeline.sh $TAG/tl_inline LYNX_SMALL_INLINE=1 > $OUT/timeline_inline.txt 2>&1
grep -A16 "c3_pipelined" $OUT/timeline_default.txt
grep -A16 "c3_pipelined" $OUT/timeline_inline.txt
