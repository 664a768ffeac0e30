#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2nt; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; w=$2; shift; shift; env "$@" timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
for i in 1 2; do
run c4_pp_$i c4 LYNX_XPOSE=0
run c4_default_$i c4 X=0
run c2_$i c2 X=0
run c2_pp_$i c2 LYNX_XPOSE=0
run c3_$i c3 X=0



run c5_$i c5 X=0

run c3big_t2_$i c3big X=0

done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2nt/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
    except Exception as e: print(f, 'ERR', e)
PY
