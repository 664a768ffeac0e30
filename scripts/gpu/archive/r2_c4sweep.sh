#!/bin/bash
# Launch-plan sweeps on one box (tiles per workgroup, moment mode, interleave, workgroups per CU); edit the list.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2c4; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; w=$2; shift; shift; env "$@" timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run c4_base1 c4
run c4_mom2 c4 LYNX_MOM=2
run c4_mom3 c4 LYNX_MOM=3
run c4_inter c4 LYNX_INTERLEAVE=1
run c4_wgs256 c4 LYNX_WGS_PER_CU=256
run c4_wgs128 c4 LYNX_WGS_PER_CU=128
run c4_base2 c4
run c4_nomom c4 X=0
run c3big_base c3big
run c3big_inter c3big LYNX_INTERLEAVE=1
run c3big_t3 c3big LYNX_MIN_TILES_PER_WG=3
run c4_base3 c4
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c4/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(20), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
PY
