#!/bin/bash
# Small beams: map build as a launch of its own vs in the streaming kernel's prologue (LYNX_FUSE_MAX_CHUNKS).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2fuse; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for f in 0 1000; do for w in c2 c3; do
  LYNX_FUSE_MAX_CHUNKS=$f timeout -k 10 180 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline > $OUT/${w}_fuse${f}_$i.json 2> $OUT/${w}_fuse${f}_$i.err
  LYNX_FUSE_MAX_CHUNKS=$f timeout -k 10 180 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --sync-every-step > $OUT/${w}_fuse${f}_sync_$i.json 2> $OUT/${w}_fuse${f}_sync_$i.err
done; done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2fuse/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(28), 'us/step %.1f kern %.1f'%(d['ms_per_step']*1e3, r['avg_launch_ms']*1e3))
    except Exception as e: print(f, 'ERR', e)
PY
