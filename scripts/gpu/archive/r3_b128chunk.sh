#!/bin/bash
# Round 3: k_build's chunk (elements built and composed per round) for the 128-sample shard of BASELINE config 4
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3b128chunk}; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3 4; do
  for c in 32 64 48; do
    LYNX_BUILD_CHUNK=$c LYNX_FORCE_COMM=1 timeout -k 10 200 python bench.py --no-cpu-baseline --batch 128 --steps 60 --warmup 5 > $OUT/b128_chunk${c}_$rep.json 2> $OUT/b128_chunk${c}_$rep.err || echo "$c failed"
  done
done
python3 - <<PY
import json,glob,os
out='$OUT'
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(26), 'ms/step %.4f kern %.4f'%(d['ms_per_step'], r['avg_launch_ms']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
