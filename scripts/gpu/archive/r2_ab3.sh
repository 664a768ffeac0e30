#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2ab3; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { name=$1; w=$2; shift; shift
  env "$@" timeout -k 10 180 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline $EXTRA > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed: $(tail -3 $OUT/$name.err)"; }
for w in c4 c5; do
  run ${w}_lanes $w
  run ${w}_lanes_p4 $w LYNX_PIECE=4
  run ${w}_lanes_sync $w LYNX_ASYNC_BUILD=0
  run ${w}_wg $w LYNX_LANES_BUILD_MIN_BATCH=100000000
  (cd _old && timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline) > $OUT/${w}_old.json 2> $OUT/${w}_old.err
done
for w in c4 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/st_$w.json 2> $OUT/st_$w.err
done
python3 - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r2ab3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
for w in ('c4','c5'):
    print('==',w)
    for f in glob.glob('gpurun_out/r2ab3/st_%s/*/*kernel_stats.csv'%w):
        for r in csv.DictReader(open(f)):
            n=r['Name']
            if 'diag_copy' in n or 'fill_gaussian' in n or 'rocclr' in n: continue
            print('  %-46s calls %3s avg %10.1f us  min %9.1f  max %9.1f'%(n.replace('void lynx::','').replace('lynx::','')[:46], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
