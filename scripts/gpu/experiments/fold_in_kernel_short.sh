#!/bin/bash
# Round 4: fold tests + config 2 timing, short form.  Usage: fold2.sh <tag>
TAG=${1:-r4fold2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fold or forms or variant or c2_ares" > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
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
}
run default LYNX_NOOP=1
run nofold LYNX_FOLD_REDUCE=0
run fused LYNX_FUSE_MAX_CHUNKS=100000
bash scripts/gpu/r4/timeline.sh $TAG/tl > $OUT/timeline_default.txt 2>&1
grep -A8 "c2_pipelined" $OUT/timeline_default.txt
