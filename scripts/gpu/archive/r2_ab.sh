#!/bin/bash
# Round-2 A/B: wave-tile (XPOSE) vs per-particle accesses, build on the second stream or in line,
# on every bench workload; GPU test suite first.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2ab; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { # name workload env...
  name=$1; w=$2; shift; shift
  env "$@" timeout -k 10 180 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed: $(tail -3 $OUT/$name.err)"
}
for w in c4 c3big c3 c5 c2; do
  STEPS=20; [ $w = c3 -o $w = c2 ] && STEPS=100
  run ${w}_x1_a1 $w LYNX_XPOSE=1 LYNX_ASYNC_BUILD=1
  run ${w}_x0_a1 $w LYNX_XPOSE=0 LYNX_ASYNC_BUILD=1
  run ${w}_x1_a0 $w LYNX_XPOSE=1 LYNX_ASYNC_BUILD=0
done
EXTRA=--no-moments run c4_x1_nomom c4 LYNX_XPOSE=1
EXTRA=--no-moments run c4_x0_nomom c4 LYNX_XPOSE=0
EXTRA=--no-moments run c3big_x1_nomom c3big LYNX_XPOSE=1
EXTRA=--sync-every-step run c3_x1_sync c3 LYNX_XPOSE=1
EXTRA=--sync-every-step run c2_x1_sync c2 LYNX_XPOSE=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'], 'copy %.0f'%(d.get('hbm_copy_kernel_gbs') or 0))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
PY
