#!/bin/bash
# The main stream's wait for the build of the next call: a device-side event wait (default) vs the host waiting for the build
# before it enqueues the streaming kernel (LYNX_HOST_WAIT_BUILD=1: no barrier packet in front of the kernel, host one call ahead at most).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2hw; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do for m in 0 1; do
  for w in c4 c5; do LYNX_HOST_WAIT_BUILD=$m timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${w}_hw${m}_$i.json 2> $OUT/${w}_hw${m}_$i.err; done
  for w in c3 c3big; do LYNX_HOST_WAIT_BUILD=$m timeout -k 10 180 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline > $OUT/${w}_hw${m}_$i.json 2> $OUT/${w}_hw${m}_$i.err; done
done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2hw/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f kern %.4f gap us %.1f'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f, 'ERR', e)
PY
