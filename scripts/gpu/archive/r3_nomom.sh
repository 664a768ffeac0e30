#!/bin/bash
# what the step of the 128-sample shard costs without the moment reduction on the main stream (upper bound of what
# moving it to the side stream can return), and with the build in line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3nomom; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
for b in 128 1024; do
timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 > $OUT/mom_b${b}_$rep.json 2> $OUT/mom_b${b}_$rep.err
timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 --no-moments > $OUT/nomom_b${b}_$rep.json 2> $OUT/nomom_b${b}_$rep.err
LYNX_ASYNC_BUILD=0 timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 --no-moments > $OUT/nomom_sync_b${b}_$rep.json 2> $OUT/nomom_sync_b${b}_$rep.err
done
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 > $OUT/c3big_mom_$rep.json 2> $OUT/c3big_mom_$rep.err
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 --no-moments > $OUT/c3big_nomom_$rep.json 2> $OUT/c3big_nomom_$rep.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
