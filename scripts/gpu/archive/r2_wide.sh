#!/bin/bash
# Moment reduction for beams of few samples: one 1024-thread workgroup per sample (LYNX_REDUCE_WIDE=1, default) vs a level of groups + final.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2wide; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for m in 1 0; do for w in c2 c3; do
  LYNX_REDUCE_WIDE=$m timeout -k 10 180 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline > $OUT/${w}_wide${m}_$i.json 2> $OUT/${w}_wide${m}_$i.err
done; done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2wide/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(24), 'us/step %.1f kern %.1f'%(d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))
    except Exception as e: print(f, 'ERR', e)
PY
