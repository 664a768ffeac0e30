#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3hostwait; rm -rf $OUT; mkdir -p $OUT
export LYNX_FORCE_COMM=1
for rep in 1 2 3; do
for hw in 0 1; do
for b in 128 1024; do
LYNX_BUILD_HOST_WAIT=$hw timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 > $OUT/b${b}_hw${hw}_$rep.json 2> $OUT/b${b}_hw${hw}_$rep.err
done
LYNX_FORCE_COMM=0 LYNX_BUILD_HOST_WAIT=$hw timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 > $OUT/c3big_hw${hw}_$rep.json 2> $OUT/c3big_hw${hw}_$rep.err
LYNX_FORCE_COMM=0 LYNX_BUILD_HOST_WAIT=$hw timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --steps 40 --warmup 5 > $OUT/c5_hw${hw}_$rep.json 2> $OUT/c5_hw${hw}_$rep.err
done
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
