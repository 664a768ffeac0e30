#!/bin/bash
# C5 forward: particles per lane / tiles per workgroup / workgroups per CU.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2c5s; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout -k 10 180 python bench.py --workload c5 --steps 30 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run a_default X=0
run u4 LYNX_UNROLL=4
run u4_t8 LYNX_UNROLL=4 LYNX_MIN_TILES_PER_WG=8
run u4_t4 LYNX_UNROLL=4 LYNX_MIN_TILES_PER_WG=4
run u2_t32 LYNX_MIN_TILES_PER_WG=32
run u2_t8 LYNX_MIN_TILES_PER_WG=8
run u1 LYNX_UNROLL=1
run nomerge LYNX_MERGE_STEPS=0
run z_default X=0
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c5s/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
    except Exception as e: print(f, 'ERR', e)
PY
