#!/bin/bash
# small configurations (C2, C3): do the side stream / host-side build wait help or hurt there?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3small; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
for w in c3 c2; do
for v in "LYNX_BUILD_HOST_WAIT=1 LYNX_SIDE_REDUCE=1" "LYNX_BUILD_HOST_WAIT=0 LYNX_SIDE_REDUCE=1" "LYNX_BUILD_HOST_WAIT=0 LYNX_SIDE_REDUCE=0" "LYNX_BUILD_HOST_WAIT=1 LYNX_SIDE_REDUCE=0" "X=default"; do
  n=$(echo $v | tr -c 'A-Za-z0-9=\n' '_')
  env $v timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 10 --no-cpu-baseline > $OUT/${w}_${n}_$rep.json 2> $OUT/${w}_${n}_$rep.err
done; done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(56), 'us/step %.1f kern %.1f'%(d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
