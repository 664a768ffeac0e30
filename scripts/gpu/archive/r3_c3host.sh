#!/bin/bash
# Round 3: BASELINE config 3 (1 sample, 1 M particles, float64) is bound by the host's enqueue rate: which of the
# per-call stream operations pay for themselves there
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3c3host}; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1 w=$2; shift 2
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps 300 --warmup 20 > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
for rep in 1 2; do for w in c3 c2; do
  run ${w}_default_$rep $w LYNX_X=0
  run ${w}_notail_$rep $w LYNX_BUILD_IN_TAIL=0
  run ${w}_noside_$rep $w LYNX_SIDE_REDUCE=0
  run ${w}_notail_noside_$rep $w LYNX_BUILD_IN_TAIL=0 LYNX_SIDE_REDUCE=0
  run ${w}_inline_$rep $w LYNX_ASYNC_BUILD=0 LYNX_SIDE_REDUCE=0
  run ${w}_async_$rep $w LYNX_ASYNC_BUILD=1
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(30), 'us/step %.1f kern %.1f'%(d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
