#!/bin/bash
# Round 4: config 2 (and the latency table) under the forms its three kernels can take.  Usage: c2forms.sh <tag>
TAG=${1:-r4c2forms}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { local name=$1; shift
  for mode in p w; do
    extra=""; [ $mode == w ] && extra="--sync-every-step"
    env "$@" timeout -k 10 200 python bench.py --workload c2 --steps 1000 --warmup 50 --no-cpu-baseline --no-kernel-timing $extra > $OUT/${name}_$mode.json 2> $OUT/${name}_$mode.err
    python - $OUT/${name}_$mode.json $name $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {'pipelined' if sys.argv[3]=='p' else 'waited   '} us/step {1e3*d['ms_per_step']:7.2f}")
PY
  done
  env "$@" PYTHONPATH=. timeout -k 10 300 python scripts/gpu/latency.py > $OUT/${name}_latency.json 2> $OUT/${name}_latency.err
  python - $OUT/${name}_latency.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   ", {k: round(v, 1) for k, v in d.items() if isinstance(v, (int, float))})
PY
}
run default LYNX_NOOP=1
run wide LYNX_REDUCE_WIDE=1
run fused LYNX_FUSE_MAX_CHUNKS=100000
run fused_wide LYNX_FUSE_MAX_CHUNKS=100000 LYNX_REDUCE_WIDE=1
run tiles4 LYNX_MIN_TILES_PER_WG=4
run tiles4_wide LYNX_MIN_TILES_PER_WG=4 LYNX_REDUCE_WIDE=1
