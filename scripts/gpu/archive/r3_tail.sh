#!/bin/bash
# build at the head of the previous streaming kernel (LYNX_BUILD_IN_TAIL=0) vs in its tail (1): C5, C4, the 128-sample shard, c3big
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3tail; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do for t in 1 0; do
LYNX_BUILD_IN_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --steps 40 --warmup 5 > $OUT/c5_tail${t}_$rep.json 2> $OUT/c5_tail${t}_$rep.err
LYNX_BUILD_IN_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > $OUT/c4_tail${t}_$rep.json 2> $OUT/c4_tail${t}_$rep.err
LYNX_FORCE_COMM=1 LYNX_BUILD_IN_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --batch 128 --steps 60 --warmup 5 > $OUT/b128_tail${t}_$rep.json 2> $OUT/b128_tail${t}_$rep.err
LYNX_BUILD_IN_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 > $OUT/c3big_tail${t}_$rep.json 2> $OUT/c3big_tail${t}_$rep.err
LYNX_BUILD_IN_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --grad --steps 10 --warmup 2 > $OUT/c5grad_tail${t}_$rep.json 2> $OUT/c5grad_tail${t}_$rep.err
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
