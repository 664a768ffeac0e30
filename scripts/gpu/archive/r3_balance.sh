#!/bin/bash
# whole rounds of workgroups (LYNX_BALANCE=1) vs the plain partition (0)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3balance; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3; do for t in 1 0; do
LYNX_FORCE_COMM=1 LYNX_BALANCE=$t timeout -k 10 200 python bench.py --no-cpu-baseline --batch 128 --steps 60 --warmup 5 > $OUT/b128_bal${t}_$rep.json 2> $OUT/b128_bal${t}_$rep.err
LYNX_FORCE_COMM=1 LYNX_BALANCE=$t timeout -k 10 200 python bench.py --no-cpu-baseline --batch 256 --steps 60 --warmup 5 > $OUT/b256_bal${t}_$rep.json 2> $OUT/b256_bal${t}_$rep.err
LYNX_BALANCE=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 5 > $OUT/c4_bal${t}_$rep.json 2> $OUT/c4_bal${t}_$rep.err
LYNX_BALANCE=$t timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 > $OUT/c3big_bal${t}_$rep.json 2> $OUT/c3big_bal${t}_$rep.err
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
