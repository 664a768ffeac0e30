#!/bin/bash
# Round 4: the moment fold in the streaming kernel of small beams: GPU suite, then config 2 and the latency table with
# the fold on / off and with the build fused into the kernel too.  Usage: fold.sh <tag>
TAG=${1:-r4fold}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
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
run nofold LYNX_FOLD_REDUCE=0
run fused LYNX_FUSE_MAX_CHUNKS=100000
run default_again LYNX_NOOP=1
bash scripts/gpu/r4/timeline.sh $TAG/tl > $OUT/timeline_default.txt 2>&1
grep -A16 "c2_pipelined" $OUT/timeline_default.txt
