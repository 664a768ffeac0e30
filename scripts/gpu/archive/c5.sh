#!/bin/bash
# the whole GPU suite, then every workload (and the unroll variants of the cavity workload)
mkdir -p gpurun_out/c5; rm -f gpurun_out/c5/*
timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for r in 1 2; do
for wl in c4 c5 c3 c3big c2; do
  timeout -k 10 120 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c5/r${r}_$wl.json 2> gpurun_out/c5/r${r}_$wl.err
done
for u in 1 2 4; do
LYNX_UNROLL=$u timeout -k 10 120 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c5/r${r}_c5_unroll$u.json 2> gpurun_out/c5/r${r}_c5_unroll$u.err
done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c5/r*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(28), 'ms/step %.3f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'steps/s %.3e'%d['value'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-500:])
PY
