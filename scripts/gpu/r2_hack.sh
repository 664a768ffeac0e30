#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2hack; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; w=$2; shift; shift
  env "$@" timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed: $(tail -3 $OUT/$name.err)"; }
H=LYNX_HIP_LIBRARY=$GRAFT_REPO_ROOT/lynx_amd/_lib/liblynxhip_hack.so
for w in c3big c4 c3 c5; do
  run ${w}_full_x0 $w LYNX_XPOSE=0
  run ${w}_full_x1 $w LYNX_XPOSE=1
  run ${w}_hack_x0 $w LYNX_XPOSE=0 $H
  run ${w}_hack_x1 $w LYNX_XPOSE=1 $H
done
run c3big_hack_x1_t2 c3big LYNX_XPOSE=1 LYNX_MIN_TILES_PER_WG=2 $H
run c3big_hack_x1_t8 c3big LYNX_XPOSE=1 LYNX_MIN_TILES_PER_WG=8 $H
run c3big_hack_x0_t2 c3big LYNX_XPOSE=0 LYNX_MIN_TILES_PER_WG=2 $H
run c3big_hack_x0_t8 c3big LYNX_XPOSE=0 LYNX_MIN_TILES_PER_WG=8 $H
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2hack/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(24), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
PY
