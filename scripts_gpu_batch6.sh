#!/bin/bash
mkdir -p gpurun_out/b6
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload ${WL:-c4} --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/b6/$name.json 2> gpurun_out/b6/$name.err
}
for r in 1 2; do
  for w in 64 128 256 512 1024; do for u in 2 4; do
    run r${r}_w${w}_u${u} LYNX_WGS_PER_CU=$w LYNX_UNROLL=$u
  done; done
  WL=c3 run r${r}_c3 A=1
  WL=c3 run r${r}_c3_w32 LYNX_WGS_PER_CU=32
  WL=c3big run r${r}_c3big A=1
  WL=c2 run r${r}_c2 A=1
done
python3 - <<'PY'
import json,glob,collections,statistics
res=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/b6/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        res[f.split('/')[-1].split('_',1)[1][:-5]].append((r['achieved'], d['ms_per_step'], r['avg_launch_ms']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
for k,v in sorted(res.items(), key=lambda kv:-statistics.median([a for a,_,_ in kv[1]])):
    print(k.ljust(16), 'kern GB/s', ' '.join('%.0f'%x for x,_,_ in v), ' ms/step', ' '.join('%.3f'%y for _,y,_ in v), ' kern ms', ' '.join('%.4f'%z for _,_,z in v))
PY
