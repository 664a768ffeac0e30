#!/bin/bash
# Round 4: small lattices' programs in the kernel arguments (LYNX_INLINE_POOL): GPU suite, then config 2 / config 3 and
# the latency table with it on and off.  Usage: inline.sh <tag>
TAG=${1:-r4inline}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { local name=$1; shift
  for w in c2 c3; do for mode in p w; do
    extra=""; [ $mode == w ] && extra="--sync-every-step"
    env "$@" timeout -k 10 200 python bench.py --workload $w --steps 500 --warmup 50 --no-cpu-baseline --no-kernel-timing $extra > $OUT/${name}_${w}_$mode.json 2> $OUT/${name}_${w}_$mode.err
    python - $OUT/${name}_${w}_$mode.json $name $w $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:12s} {sys.argv[3]} {'pipelined' if sys.argv[4]=='p' else 'waited   '} us/step {1e3*d['ms_per_step']:7.2f}")
PY
  done; done
  env "$@" PYTHONPATH=. timeout -k 10 300 python scripts/gpu/latency.py > $OUT/${name}_latency.json 2> $OUT/${name}_latency.err
  python - $OUT/${name}_latency.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   ", {k: round(v, 1) for k, v in d.items() if isinstance(v, (int, float))})
PY
}
run inline LYNX_INLINE_POOL=1
run memory LYNX_INLINE_POOL=0
run inline2 LYNX_INLINE_POOL=1
run memory2 LYNX_INLINE_POOL=0
