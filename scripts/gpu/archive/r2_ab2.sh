#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2ab2; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { name=$1; w=$2; shift; shift
  env "$@" timeout -k 10 180 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed: $(tail -3 $OUT/$name.err)"; }
for w in c4 c3big c3 c5 c2; do
  STEPS=20; [ $w = c3 -o $w = c2 ] && STEPS=100
  run ${w}_default $w
  run ${w}_x0 $w LYNX_XPOSE=0
  run ${w}_x1 $w LYNX_XPOSE=1
  (cd _old && timeout -k 10 180 python bench.py --workload $w --steps $STEPS --warmup 3 --no-cpu-baseline) > $OUT/${w}_old.json 2> $OUT/${w}_old.err
done
EXTRA=--sync-every-step STEPS=100 run c3_sync c3
EXTRA=--sync-every-step STEPS=100 run c2_sync c2
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2ab2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
PY
