#!/bin/bash
# Launch-plan sweeps on one box (tiles per workgroup, particles per lane, access form, moment mode); edit the list.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2c4; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; w=$2; shift; shift; env "$@" timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run c4_base1 c4
for t in 1 2 3 4; do run c4_x1_t$t c4 LYNX_XPOSE=1 LYNX_MIN_TILES_PER_WG=$t; done
run c4_base2 c4
run c4_x1_t2_mom2 c4 LYNX_XPOSE=1 LYNX_MIN_TILES_PER_WG=2 LYNX_MOM=2
run c4_base3 c4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c4/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(20), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
PY
