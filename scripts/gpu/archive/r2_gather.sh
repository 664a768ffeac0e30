#!/bin/bash
# RCCL all-gather of the moment records at world size 1 (LYNX_FORCE_COMM=1): in line on the main stream (default) vs on the
# communication stream underneath the next streaming kernel (LYNX_GATHER_OVERLAP=1).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2gather; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q -k "rccl or comm or gather" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 > $OUT/nocomm_$i.json 2> $OUT/nocomm_$i.err
LYNX_FORCE_COMM=1 LYNX_GATHER_OVERLAP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 > $OUT/overlap_$i.json 2> $OUT/overlap_$i.err
LYNX_FORCE_COMM=1 LYNX_GATHER_OVERLAP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 > $OUT/inline_$i.json 2> $OUT/inline_$i.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2gather/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(26), 'ms/step %.4f kern %.4f'%(d['ms_per_step'], r['avg_launch_ms']), d['config'].get('gather'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
