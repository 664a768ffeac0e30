#!/bin/bash
# What separates the C4 streaming kernel from a plain copy?  Moments off / on, tiles per workgroup, particles per lane.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2dec; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
EXTRA=""
run a_default X=0
for u in 1 2 4; do for t in 1 2 4; do run m_u${u}_t$t LYNX_UNROLL=$u LYNX_MIN_TILES_PER_WG=$t; done; done
EXTRA="--no-moments"
for u in 1 2 4; do for t in 1 2; do run n_u${u}_t$t LYNX_UNROLL=$u LYNX_MIN_TILES_PER_WG=$t; done; done
EXTRA=""
run z_default X=0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2dec/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f kern %.4f GB/s %.0f copy %s'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r.get('hbm_copy_kernel_gbs')))
    except Exception as e: print(f, 'ERR', e)
PY
