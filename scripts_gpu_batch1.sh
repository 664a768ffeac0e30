#!/bin/bash
# one-off measurement batch (round 1): bench variants + rocprofv3 kernel trace
mkdir -p gpurun_out/b1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in c4 c3 c3big c2; do
  timeout -k 10 120 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b1/$w.json 2> gpurun_out/b1/$w.err
done
timeout -k 10 120 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline --no-moments > gpurun_out/b1/c4_nomom.json 2> gpurun_out/b1/c4_nomom.err
LYNX_TWO_KERNEL=1 timeout -k 10 120 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b1/c4_two.json 2> gpurun_out/b1/c4_two.err
LYNX_TWO_KERNEL=1 timeout -k 10 120 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b1/c3_two.json 2> gpurun_out/b1/c3_two.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1/prof_c4 -- python3 bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b1/prof_c4.json 2> gpurun_out/b1/prof_c4.err
ls -R gpurun_out/b1/prof_c4 | head -30
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b1/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.3f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'], 'steps/s %.3e'%d['value'])
    except Exception as e:
        print(f, 'ERR', e)
PY
