#!/bin/bash
mkdir -p gpurun_out/ab
for r in 1 2 3; do
for wl in c4 c5 c3; do
  timeout -k 10 120 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/r${r}_${wl}_slp.json 2> gpurun_out/ab/r${r}_${wl}_slp.err
  LYNX_HIP_LIBRARY=$GRAFT_REPO_ROOT/lynx_amd/_lib/liblynxhip_noslp.so timeout -k 10 120 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/r${r}_${wl}_noslp.json 2> gpurun_out/ab/r${r}_${wl}_noslp.err
done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/r*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(28), 'ms/step %.3f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
