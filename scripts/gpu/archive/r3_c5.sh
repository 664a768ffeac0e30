#!/bin/bash
# Round 3: BASELINE config 5 forward with the structured step loop (lynx_units.hpp) vs the dense one, same box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3c5}; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
for un in 1 0; do
LYNX_TRACK_UNITS=$un timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --steps 40 --warmup 5 > $OUT/c5_units${un}_$rep.json 2> $OUT/c5_units${un}_$rep.err
done
done
for v in "$@"; do
env $v timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --steps 40 --warmup 5 > $OUT/c5_${v//[^A-Za-z0-9=]/_}.json 2> $OUT/c5_${v//[^A-Za-z0-9=]/_}.err
done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(34), 'ms/step %.4f kern %.4f  step-kern %.1f us  frac %.3f'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3, r['frac']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
