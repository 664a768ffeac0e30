#!/bin/bash
# Round 3: the pair tree of the lanes build in one launch (LYNX_PAIR_LEVELS_FUSED) against one launch per level, and
# the lanes build for the 128-sample shard (default there: the workgroup build); shards of BASELINE config 4
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3levels}; rm -rf $OUT; mkdir -p $OUT
run() { # name batch env...
  local name=$1 b=$2; shift 2
  env "$@" LYNX_FORCE_COMM=1 timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
}
for rep in 1 2 3; do
  for b in 1024 256; do
    run b${b}_fused_$rep $b LYNX_PAIR_LEVELS_FUSED=1
    run b${b}_levels_$rep $b LYNX_PAIR_LEVELS_FUSED=0
  done
  run b128_wg_$rep 128 LYNX_PAIR_LEVELS_FUSED=1
  run b128_lanes_$rep 128 LYNX_LANES_BUILD_MIN_BATCH=128
  run b128_lanes_hw0_$rep 128 LYNX_LANES_BUILD_MIN_BATCH=128 LYNX_BUILD_HOST_WAIT=0
done
python3 - <<PY
import json,glob,os
out='$OUT'
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(30), 'ms/step %.4f kern %.4f'%(d['ms_per_step'], r['avg_launch_ms']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
